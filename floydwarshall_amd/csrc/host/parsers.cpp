// parsers.cpp -- host mirror of the reference's attoparsec grammars, with attoparsec's own
// failure texts (they are part of what the reference prints).
//
//   exchRatesParser  /root/reference/src/lib/Parsers.hs:25-40
//   exchPairParser   /root/reference/src/lib/Parsers.hs:46-55
//   alphabets        /root/reference/src/lib/Parsers.hs:57-58   (many1 letter)
//   simpleParse      /root/reference/src/lib/Parsers.hs:70-73   (parseOnly: trailing input ignored)
//
// attoparsec conventions reproduced: `fail s` -> "Failed reading: s"; a failing `letter`
// (= satisfy isAlpha <?> "letter") -> "letter: Failed reading: satisfy"; running out of input
// under parseOnly -> "not enough input" (prefixed by the context, e.g. "letter: ").
// Expected texts pinned by src/test/ParserTest.hs:46-79 and README.md:170-207.
#include <cctype>
#include <cstdlib>
#include <cstring>

#include "host_types.hpp"

namespace fwxh {

int64_t posix_from_civil(int64_t y, unsigned mo, unsigned d, unsigned h, unsigned mi, unsigned s);

namespace {

struct Cursor {
    const std::string &s;
    size_t pos = 0;
    explicit Cursor(const std::string &str) : s(str) {}
    bool eof() const { return pos >= s.size(); }
    unsigned char peek() const { return (unsigned char)s[pos]; }
};

bool is_space(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
// `letter` is Data.Char.isAlpha; bytes >= 0x80 (UTF-8 multi-byte letters) are let through.
bool is_letter(unsigned char c) { return std::isalpha(c) || c >= 0x80; }

void skip_space(Cursor &c)
{
    while (!c.eof() && is_space(c.peek())) ++c.pos;
}

// skipSpace >> many1 letter
bool alphabets(Cursor &c, std::string &out, std::string &err)
{
    skip_space(c);
    if (c.eof()) { err = "letter: not enough input"; return false; }
    if (!is_letter(c.peek())) { err = "letter: Failed reading: satisfy"; return false; }
    out.clear();
    while (!c.eof() && is_letter(c.peek())) out.push_back(c.s[c.pos++]);
    return true;
}

// skipSpace >> double   (Data.Attoparsec.Text.double = scientifically ...)
bool parse_double(Cursor &c, double &out, std::string &err)
{
    skip_space(c);
    if (c.eof()) { err = "not enough input"; return false; }            // peekChar'
    size_t p = c.pos;
    if (c.s[p] == '+' || c.s[p] == '-') ++p;
    const size_t digits0 = p;
    while (p < c.s.size() && std::isdigit((unsigned char)c.s[p])) ++p;
    if (p == digits0) {                                                  // decimal = takeWhile1
        err = p >= c.s.size() ? "not enough input" : "Failed reading: takeWhile1";
        return false;
    }
    if (p < c.s.size() && c.s[p] == '.') {                               // '.' then takeWhile isDigit
        ++p;
        while (p < c.s.size() && std::isdigit((unsigned char)c.s[p])) ++p;
    }
    if (p < c.s.size() && (c.s[p] == 'e' || c.s[p] == 'E')) {            // optional exponent,
        size_t q = p + 1;                                                // backtracks if malformed
        if (q < c.s.size() && (c.s[q] == '+' || c.s[q] == '-')) ++q;
        const size_t e0 = q;
        while (q < c.s.size() && std::isdigit((unsigned char)c.s[q])) ++q;
        if (q > e0) p = q;
    }
    std::string tok = c.s.substr(c.pos, p - c.pos);
    if (!tok.empty() && tok.back() == '.') tok.push_back('0');
    out = strtod(tok.c_str(), nullptr);                                  // correctly rounded
    c.pos = p;
    return true;
}

bool two_digits(const std::string &s, size_t p, unsigned &v)
{
    if (p + 2 > s.size() || !std::isdigit((unsigned char)s[p]) || !std::isdigit((unsigned char)s[p + 1]))
        return false;
    v = (unsigned)(s[p] - '0') * 10 + (unsigned)(s[p + 1] - '0');
    return true;
}

// parseTimeM True defaultTimeLocale "%Y-%m-%dT%H:%M:%S%z"   (Parsers.hs:39)
bool parse_timestamp(const std::string &t, int64_t &out)
{
    size_t b = 0, e = t.size();
    while (b < e && is_space((unsigned char)t[b])) ++b;                  // acceptWS = True
    while (e > b && is_space((unsigned char)t[e - 1])) --e;
    const std::string s = t.substr(b, e - b);
    size_t p = 0;
    if (s.size() < 4) return false;
    int64_t year = 0;
    for (int i = 0; i < 4; ++i, ++p) {
        if (!std::isdigit((unsigned char)s[p])) return false;
        year = year * 10 + (s[p] - '0');
    }
    unsigned mo, d, h, mi, sec;
    if (p >= s.size() || s[p++] != '-' || !two_digits(s, p, mo)) return false;
    p += 2;
    if (p >= s.size() || s[p++] != '-' || !two_digits(s, p, d)) return false;
    p += 2;
    if (p >= s.size() || s[p++] != 'T' || !two_digits(s, p, h)) return false;
    p += 2;
    if (p >= s.size() || s[p++] != ':' || !two_digits(s, p, mi)) return false;
    p += 2;
    if (p >= s.size() || s[p++] != ':' || !two_digits(s, p, sec)) return false;
    p += 2;
    if (p >= s.size() || (s[p] != '+' && s[p] != '-')) return false;    // %z: +HHMM or +HH:MM
    const int sign = s[p++] == '-' ? -1 : 1;
    unsigned zh, zm;
    if (!two_digits(s, p, zh)) return false;
    p += 2;
    if (p < s.size() && s[p] == ':') ++p;
    if (!two_digits(s, p, zm)) return false;
    p += 2;
    if (p != s.size()) return false;
    static const unsigned mdays[] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    const bool leap = (year % 4 == 0 && year % 100 != 0) || year % 400 == 0;
    if (mo < 1 || mo > 12) return false;
    const unsigned dim = mdays[mo - 1] + (mo == 2 && leap ? 1 : 0);
    if (d < 1 || d > dim || h > 23 || mi > 59 || sec > 60 || zm > 59) return false;
    out = posix_from_civil(year, mo, d, h, mi, sec) - sign * (int64_t)(zh * 3600 + zm * 60);
    return true;
}

std::string upper(std::string s)
{
    for (auto &ch : s) ch = (char)std::toupper((unsigned char)ch);
    return s;
}

}  // namespace

bool parse_rates(const std::string &line, ParsedRates &out, std::string &err)
{
    Cursor c(line);
    // :27  tS <- skipSpace >> many1 (satisfy (/= ' '))
    skip_space(c);
    if (c.eof()) { err = "not enough input"; return false; }
    std::string ts;
    while (!c.eof() && c.peek() != ' ') ts.push_back(c.s[c.pos++]);
    if (ts.empty()) { err = "Failed reading: satisfy"; return false; }
    // :28  time <- parseTimestamp tS
    if (!parse_timestamp(ts, out.time)) {
        err = "Failed reading: parseTimeM: no parse of " + show_string(ts);
        return false;
    }
    std::string exch, src, dest;
    if (!alphabets(c, exch, err) || !alphabets(c, src, err) || !alphabets(c, dest, err))   // :29-31
        return false;
    auto positive = [&](double r) {                                                         // :40
        if (r <= 0) { err = "Failed reading: Rate must be > 0"; return false; }
        return true;
    };
    if (!parse_double(c, out.fwd, err) || !positive(out.fwd)) return false;                 // :32
    if (!parse_double(c, out.bkd, err) || !positive(out.bkd)) return false;                 // :33
    if (out.fwd * out.bkd > 1.0) {                                                          // :34
        err = "Failed reading: Product of " + show_double(out.fwd) + " and " +
              show_double(out.bkd) + " must be <= 1.0";
        return false;
    }
    exch = upper(exch); src = upper(src); dest = upper(dest);                               // :35
    if (src == dest) { err = "Failed reading: The currencies must be different"; return false; }  // :36
    out.src = Vertex{exch, src};
    out.dest = Vertex{exch, dest};
    return true;
}

bool parse_exch_pair(const std::string &line, Vertex &src, Vertex &dest, std::string &err)
{
    Cursor c(line);
    std::string a, b, d, e;
    if (!alphabets(c, a, err) || !alphabets(c, b, err) || !alphabets(c, d, err) ||
        !alphabets(c, e, err))                                                              // :48-51
        return false;
    src = Vertex{upper(a), upper(b)};                                                       // :52-53
    dest = Vertex{upper(d), upper(e)};
    if (src == dest) {                                                                      // :54
        err = "Failed reading: source must be different from destination";
        return false;
    }
    return true;
}

}  // namespace fwxh
