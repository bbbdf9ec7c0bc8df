// Host-side plan builder (xmhw_amd/csrc/plan.cpp) under AddressSanitizer + UBSan on the CPU:
// calendars with leap days, partial years, tstep axes, pathological label sequences, every
// (subs, yps) table and chunking the kernels ask for.  Built and run by tests/test_plan_sanitize.py.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../xmhw_amd/csrc/plan.cpp"

static bool leap(int y) { return (y % 4 == 0 && y % 100 != 0) || y % 400 == 0; }

static std::vector<int32_t> daily(int y0, int y1, int skip_first = 0, int skip_last = 0) {
    std::vector<int32_t> d;
    for (int y = y0; y <= y1; ++y) {
        const int n = leap(y) ? 366 : 365;
        for (int k = 1; k <= n; ++k) d.push_back((!leap(y) && k >= 60) ? k + 1 : k);
    }
    d.erase(d.begin(), d.begin() + skip_first);
    d.resize(d.size() - skip_last);
    return d;
}

static int exercise(const std::vector<int32_t>& doy, int w) {
    xmhw::Plan p;
    if (!p.build(doy.data(), static_cast<int64_t>(doy.size()), w)) return 0;   // refused: fine
    long checksum = p.D + p.ntracks;
    for (int subs : {8, 16})
        for (int yps = 1; yps <= 6; ++yps) {
            if (static_cast<long>(subs) * yps < p.ntracks) continue;
            const auto t = p.ring_table(subs, yps);
            if (t.size() != static_cast<size_t>(p.nsteps) * subs * yps) { std::fprintf(stderr, "table size\n"); std::exit(1); }
            for (uint32_t e : t) checksum += e & 1u;
        }
    for (int nc : {1, 2, 3, 7, 16, 366, 5000}) {
        const auto ch = p.make_chunks(nc);
        int32_t covered = 0;
        for (const auto& c : ch) {
            if (c.begin < c.warm_start - 0 && c.warm_start > c.begin) { std::fprintf(stderr, "warm_start after begin\n"); std::exit(1); }
            covered += c.end - c.begin;
        }
        if (covered != p.D) { std::fprintf(stderr, "chunks cover %d of %d rows\n", covered, p.D); std::exit(1); }
        checksum += static_cast<long>(ch.size());
    }
    return static_cast<int>(checksum & 0x7fffffff);
}

int main() {
    long acc = 0;
    for (int w : {0, 1, 2, 5, 15}) {
        acc += exercise(daily(1982, 2021), w);
        acc += exercise(daily(2001, 2003, 100, 200), w);          // partial first and last years
        acc += exercise(daily(2001, 2001), w);                     // one non-leap year: D = 365
        acc += exercise(daily(2004, 2004, 58, 300), w);            // a few days around Feb 29
    }
    {   // tstep axis: 73 steps per year, 9 years; and one with a broken last cycle
        std::vector<int32_t> d;
        for (int y = 0; y < 9; ++y) for (int k = 1; k <= 73; ++k) d.push_back(k);
        acc += exercise(d, 2);
        d.resize(d.size() - 10);
        acc += exercise(d, 2);
    }
    {   // pathological labels: constant, strictly decreasing, random, single step, huge labels
        std::vector<int32_t> c(500, 7), dec, rnd, one(1, 366), big;
        for (int k = 400; k >= 1; --k) dec.push_back(k);
        unsigned s = 1;
        for (int k = 0; k < 3000; ++k) { s = s * 1664525u + 1013904223u; rnd.push_back(1 + static_cast<int32_t>((s >> 8) % 366)); }
        for (int k = 0; k < 50; ++k) big.push_back(2000000000 - 50 + k);
        for (const auto* v : {&c, &dec, &rnd, &one, &big}) acc += exercise(*v, 5);
        std::vector<int32_t> neg = {3, -1, 5};
        acc += exercise(neg, 5);
    }
    std::printf("plan sanitize ok %ld\n", acc);
    return 0;
}
