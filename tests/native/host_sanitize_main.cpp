// ASan/UBSan driver for the host mirror's pure-CPU pieces (parsers, show, buildMatrix, optimum).
// Built and run by tests/test_native_sanitizers.py; GPU AddressSanitizer is not available on this
// pool, so sanitizers run on the CPU build only.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "host_types.hpp"

using namespace fwxh;

int main()
{
    // show
    const double vals[] = {0.0, -0.0, 1000.0, 0.0009, 1e7, 9999999.0, 5e-324, 1.7976931348623157e308,
                           0.1, 123456.789, 1e-7, 1.0 / 3.0};
    for (double v : vals)
        if (show_double(v).empty()) return 1;
    if (show_utctime(1509529343) != "2017-11-01 09:42:23 UTC") return 2;
    if (show_utctime(-1) != "1969-12-31 23:59:59 UTC") return 3;
    if (show_string("a\"b\\c\n\xc3\xa9" "1") != "\"a\\\"b\\\\c\\n\\233\\&1\"") return 4;

    // parsers on structured and random input
    const char *lines[] = {
        "2017-11-01T09:42:23+00:00 KRAKEN BTC USD 1000.0 0.0009", "", " ", "KRAKEN BTC GDAX USD",
        "2017-11-01T09:42:23+00:00 KRAKEN BTC USD 1e400 0.0", "2017-02-30T00:00:00+00:00 A B C 1 1",
        "2017-11-01T09:42:23+00:00 K A B 1.", "x", "2017-11-01T09:42:23-23:59 K A B 0.5 0.5 trailing",
        "2017-11-01T09:42:23+00:00 K A B .5 1", "9999-12-31T23:59:60+00:00 K A B 1 1"};
    for (const char *l : lines) {
        ParsedRates pr;
        Vertex a, b;
        std::string err;
        (void)parse_rates(l, pr, err);
        (void)parse_exch_pair(l, a, b, err);
    }
    std::mt19937 rng(7);
    const std::string alphabet = "0123456789-+:.TZeE abcXYZ\t\"\\\xc3\xa9";
    for (int it = 0; it < 20000; ++it) {
        std::string s;
        const int len = (int)(rng() % 60);
        for (int i = 0; i < len; ++i) s.push_back(alphabet[rng() % alphabet.size()]);
        ParsedRates pr;
        Vertex a, b;
        std::string err;
        (void)parse_rates(s, pr, err);
        (void)parse_exch_pair(s, a, b, err);
    }

    // buildMatrix + optimum on a small market
    ExchRateTimes rates;
    const char *ex[] = {"GDAX", "KRAKEN", "BITTREX"};
    const char *cc[] = {"BTC", "USD", "ETH"};
    for (int e = 0; e < 3; ++e)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                if (i != j) rates[VertexPair(Vertex{ex[e], cc[i]}, Vertex{ex[e], cc[j]})] =
                    std::make_pair(0.5 + 0.1 * (i + j + e), (int64_t)1000 + e);
    DenseMatrix m = build_matrix(rates);
    if (m.n() != 9) return 5;
    for (int i = 0; i < m.n(); ++i)
        for (int j = 0; j < m.n(); ++j) {
            OptimumResult r = optimum_dense(m.vertices, m.n(), m.rate.data(), m.next.data(),
                                            m.vertices[i], m.vertices[j]);
            if (r.ok != (m.next[(size_t)i * m.n() + j] >= 0)) return 6;   // unsolved: direct edges only
        }
    OptimumResult miss = optimum_dense(m.vertices, m.n(), m.rate.data(), m.next.data(),
                                       Vertex{"NOPE", "BTC"}, m.vertices[0]);
    if (miss.ok || miss.error != "(NOPE, BTC) is not entered before") return 7;
    OptimumResult empty_rows = optimum_dense(m.vertices, 0, nullptr, nullptr, m.vertices[0], m.vertices[1]);
    if (empty_rows.error != "The matrix is empty") return 8;
    puts("host sanitize ok");
    return 0;
}
