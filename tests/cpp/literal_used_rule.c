/* tests/cpp/literal_used_rule.c -- TEST HELPER (not product).  The candidate enumeration of Line3D::clusterSegments2D
 * (line3D.cc:968-1221) with the reference's `used` bookkeeping applied LITERALLY, on one thread, over the flat tables the device
 * fill works on (dense segment ids, CSR potential correspondences / collinearities, one hypothesis per segment or none):
 *   used[src][x]  <=>  x was met earlier in src's own iteration, or src was met while x was the source (an earlier iteration).
 * For every source hypothesis in order: its potential correspondences t (family 0) and the segments collinear with an accepted t
 * (family 1), then the segments collinear with the source itself (family 2); a segment that is `used` is skipped, otherwise
 * it is marked and, if it has a hypothesis, yields a candidate (source hypothesis, its hypothesis, family, collinearity weight).
 * The product's device fill (l3d_affinity.hip) replaces the maps by one "expanded" bit per (source, target); the full-size test
 * compares its edge list with what this enumeration gives.  Built by the test with gcc. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int32_t a, b, kind; float cw; } item_t;

static int has(const int32_t* p, int32_t n, int32_t d)
{
    int32_t lo = 0, hi = n;
    while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (p[mid] < d) lo = mid + 1; else hi = mid; }
    return lo < n && p[lo] == d;
}
static int cmp_i32(const void* a, const void* b) { const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b; return (x > y) - (x < y); }

/* returns the number of candidates; items (capacity cap) receives the first min(count, cap) of them, in the reference's order */
int64_t literal_used_rule(int32_t n_dense, int32_t n_hyp, const int32_t* hyp_dense, const int32_t* best,
                          const int64_t* pot_start, const int32_t* pot_tgt, const int64_t* coll_start, const int32_t* coll_other, const float* coll_w,
                          item_t* items, int64_t cap)
{
    int32_t* stamp = (int32_t*)calloc((size_t)n_dense + 1, 4);
    int64_t* met_off = (int64_t*)calloc((size_t)n_hyp + 1, 8);
    int32_t* met_len = (int32_t*)calloc((size_t)n_hyp + 1, 4);
    size_t met_cap = 1u << 20, met_n = 0;
    int32_t* met = (int32_t*)malloc(met_cap * 4);
    int64_t n_items = 0;
    for (int32_t si = 0; si < n_hyp; ++si) {
        const int32_t d = hyp_dense[si], st = si + 1;
        const size_t m0 = met_n;
#define RESERVE() do { if (met_n + 1 > met_cap) { met_cap *= 2; met = (int32_t*)realloc(met, met_cap * 4); } } while (0)
#define USED(x) (stamp[x] == st || (best[x] >= 0 && best[x] < si && has(met + met_off[best[x]], met_len[best[x]], d)))
#define MARK(x) do { stamp[x] = st; RESERVE(); met[met_n++] = (x); } while (0)
#define EMIT(hb, k, w) do { if (n_items < cap) { items[n_items].a = si; items[n_items].b = (hb); items[n_items].kind = (k); items[n_items].cw = (w); } ++n_items; } while (0)
        for (int64_t e = pot_start[d]; e < pot_start[d + 1]; ++e) {                    /* line3D.cc:996-1138 */
            const int32_t t = pot_tgt[e];
            if (USED(t)) continue;
            MARK(t);
            if (best[t] < 0) continue;
            EMIT(best[t], 0, 0.0f);
            for (int64_t c = coll_start[t]; c < coll_start[t + 1]; ++c) {              /* :1065-1136 */
                const int32_t x = coll_other[c];
                if (USED(x)) continue;
                MARK(x);
                if (best[x] >= 0) EMIT(best[x], 1, 0.0f);
            }
        }
        for (int64_t c = coll_start[d]; c < coll_start[d + 1]; ++c) {                  /* :1141-1214 */
            const int32_t x = coll_other[c];
            if (USED(x)) continue;
            MARK(x);
            if (best[x] >= 0) EMIT(best[x], 2, coll_w[c]);
        }
        qsort(met + m0, met_n - m0, 4, cmp_i32);
        met_off[si] = (int64_t)m0; met_len[si] = (int32_t)(met_n - m0);
    }
    free(stamp); free(met_off); free(met_len); free(met);
    return n_items;
}
