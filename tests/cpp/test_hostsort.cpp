// Unit test of line3d_amd/csrc/l3d_hostsort.hpp (host-only header): parallel_stable_order must equal std::stable_sort on
// (major, minor) for every thread count, including heavily skewed and tied keys.  Built and run by tests/test_host_units.py.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#include "../../line3d_amd/csrc/l3d_hostsort.hpp"

static int check(size_t n, size_t n_major, size_t n_minor, int skew, unsigned nt, unsigned seed)
{
    std::mt19937 rng(seed);
    std::vector<uint32_t> major(n), minor(n);
    for (size_t i = 0; i < n; ++i) {
        uint32_t a = rng() % n_major, b = rng() % n_minor;
        if (skew == 1 && rng() % 4) a = (uint32_t)(n_major / 2);            // 75 % of the records in one bucket
        if (skew == 2) b = b % 3;                                         // few distinct minor keys: long runs of ties
        if (skew == 3) { a = (uint32_t)(n_major - 1); b = (uint32_t)(n_minor - 1 - (i % 7)); }   // one bucket, top minor values
        major[i] = a; minor[i] = b;
    }
    std::vector<uint32_t> ref(n), got, start;
    std::iota(ref.begin(), ref.end(), 0u);
    std::stable_sort(ref.begin(), ref.end(), [&](uint32_t x, uint32_t y) { return major[x] < major[y] || (major[x] == major[y] && minor[x] < minor[y]); });
    l3d::parallel_stable_order(n, n_major, n_minor, [&](size_t i) { return major[i]; }, [&](size_t i) { return minor[i]; }, nt, got, &start);
    if (got != ref) { fprintf(stderr, "order differs: n %zu majors %zu minors %zu skew %d threads %u\n", n, n_major, n_minor, skew, nt); return 1; }
    if (start.size() != n_major + 1 || start[0] != 0 || start[n_major] != n) { fprintf(stderr, "bucket starts wrong\n"); return 1; }
    for (size_t b = 0; b < n_major; ++b)
        for (uint32_t k = start[b]; k < start[b + 1]; ++k) if (major[got[k]] != b) { fprintf(stderr, "bucket %zu holds a foreign record\n", b); return 1; }
    return 0;
}

int main()
{
    int bad = 0;
    const size_t sizes[] = { 0, 1, 17, 5000, 60000, 300000 };
    for (size_t n : sizes)
        for (unsigned nt : { 1u, 2u, 5u, 16u })
            for (int skew = 0; skew < 4; ++skew) {
                bad += check(n, 65536, 65536, skew, nt, 11u + (unsigned)n + nt);
                bad += check(n, 1000, 100000, skew, nt, 23u + (unsigned)n + nt);
                bad += check(n, 7, 3, skew, nt, 5u + (unsigned)n);
            }
    // float keys: the monotone map
    const float vals[] = { -3.5f, -0.0f, 0.0f, 1e-30f, 0.25f, 0.250001f, 1.0f, 7e9f };
    for (size_t i = 0; i + 1 < sizeof(vals) / sizeof(vals[0]); ++i) {
        const uint32_t a = l3d::float_order_key(vals[i]), b = l3d::float_order_key(vals[i + 1]);
        if (vals[i] == vals[i + 1] ? a != b : !(a < b)) { fprintf(stderr, "float key order broken at %g, %g\n", vals[i], vals[i + 1]); ++bad; }
    }
    printf(bad ? "FAILED %d\n" : "ok\n", bad);
    return bad ? 1 : 0;
}
