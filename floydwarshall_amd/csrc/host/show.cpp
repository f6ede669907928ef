// show.cpp -- GHC's `show` for the value types the reference prints.
//
//   show :: Double   (GHC.Float showFloat / formatRealFloat FFGeneric): shortest digits that
//                    round-trip; fixed notation for 0.1 <= |x| < 10^7, else d.ddde<N>
//                    -- e.g. 1000.0, 9.0e-4, 1.0, 1001.0 in /root/reference/README.md:188-246
//   show :: UTCTime  "2017-11-01 09:42:23 UTC"   (README.md:189, ProcessRequestsTest.hs:77-78)
//   show :: String   quoted with Haskell escapes, inside parseTimeM's message
//                    (ParserTest.hs:46-49, ProcessRequestsTest.hs:66)
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "host_types.hpp"

namespace fwxh {

std::string show_double(double x)
{
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x < 0 ? "-Infinity" : "Infinity";
    std::string out;
    if (std::signbit(x)) {
        out = "-";
        x = -x;
    }
    if (x == 0.0) return out + "0.0";
    // shortest round-trip digits: "d.ddddde[+-]XX"
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::scientific);
    *r.ptr = 0;
    std::string digits;
    int exp10 = 0;
    {
        const char *e = strchr(buf, 'e');
        for (const char *p = buf; p < e; ++p)
            if (*p != '.') digits.push_back(*p);
        exp10 = atoi(e + 1);
    }
    // floatToDigits: x = 0.d1d2... * 10^e  with e = exp10 + 1
    const int e = exp10 + 1;
    if (0 < e && e <= 7) {
        // fixed: integer part = first e digits (zero padded), fraction = rest or "0"
        std::string ip, fp;
        for (int i = 0; i < e; ++i) ip.push_back(i < (int)digits.size() ? digits[i] : '0');
        if ((int)digits.size() > e) fp = digits.substr(e);
        if (fp.empty()) fp = "0";
        return out + ip + "." + fp;
    }
    if (e == 0) {
        // 0.1 <= x < 1: "0." ++ ds
        return out + "0." + digits;
    }
    // exponent format: d.ddd e (e-1); a single digit gets ".0"
    std::string m(1, digits[0]);
    m += ".";
    m += digits.size() > 1 ? digits.substr(1) : "0";
    return out + m + "e" + std::to_string(e - 1);
}

static int64_t days_from_civil(int64_t y, unsigned m, unsigned d)
{
    y -= m <= 2;
    const int64_t era = (y >= 0 ? y : y - 399) / 400;
    const unsigned yoe = (unsigned)(y - era * 400);
    const unsigned doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
    const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return era * 146097 + (int64_t)doe - 719468;
}

static void civil_from_days(int64_t z, int64_t &y, unsigned &m, unsigned &d)
{
    z += 719468;
    const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
    const unsigned doe = (unsigned)(z - era * 146097);
    const unsigned yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
    y = (int64_t)yoe + era * 400;
    const unsigned doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
    const unsigned mp = (5 * doy + 2) / 153;
    d = doy - (153 * mp + 2) / 5 + 1;
    m = mp < 10 ? mp + 3 : mp - 9;
    y += m <= 2;
}

int64_t posix_from_civil(int64_t y, unsigned mo, unsigned d, unsigned h, unsigned mi, unsigned s)
{
    return days_from_civil(y, mo, d) * 86400 + (int64_t)h * 3600 + (int64_t)mi * 60 + s;
}

std::string show_utctime(int64_t t)
{
    int64_t days = t >= 0 ? t / 86400 : -((-t + 86399) / 86400);
    int64_t sod = t - days * 86400;
    int64_t y;
    unsigned m, d;
    civil_from_days(days, y, m, d);
    char buf[64];
    snprintf(buf, sizeof(buf), "%04lld-%02u-%02u %02d:%02d:%02d UTC", (long long)y, m, d,
             (int)(sod / 3600), (int)(sod % 3600 / 60), (int)(sod % 60));
    return buf;
}

std::string show_string(const std::string &s)
{
    std::string o = "\"";
    for (size_t i = 0; i < s.size();) {
        unsigned char c = (unsigned char)s[i];
        if (c < 0x80) {
            ++i;
            switch (c) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\n': o += "\\n"; break;
            case '\t': o += "\\t"; break;
            case '\r': o += "\\r"; break;
            case '\a': o += "\\a"; break;
            case '\b': o += "\\b"; break;
            case '\f': o += "\\f"; break;
            case '\v': o += "\\v"; break;
            case 0x7f: o += "\\DEL"; break;
            default:
                if (c < 0x20) {
                    o += "\\" + std::to_string((int)c);
                    // "\SOH" style names are what GHC prints; numeric form kept for simplicity
                } else {
                    o.push_back((char)c);
                }
            }
            continue;
        }
        // UTF-8 -> code point -> \DDDD (GHC escapes everything above 0x7f numerically)
        unsigned cp = 0;
        int extra = 0;
        if ((c & 0xE0) == 0xC0) { cp = c & 0x1F; extra = 1; }
        else if ((c & 0xF0) == 0xE0) { cp = c & 0x0F; extra = 2; }
        else if ((c & 0xF8) == 0xF0) { cp = c & 0x07; extra = 3; }
        else { cp = c; }
        ++i;
        for (int k = 0; k < extra && i < s.size(); ++k, ++i) cp = (cp << 6) | ((unsigned char)s[i] & 0x3F);
        o += "\\" + std::to_string(cp);
        // GHC inserts "\&" when a digit follows a numeric escape
        if (i < s.size() && s[i] >= '0' && s[i] <= '9') o += "\\&";
    }
    o += "\"";
    return o;
}

}  // namespace fwxh
