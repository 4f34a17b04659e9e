// l3d_affinity.hip -- the affinity fill of Line3D::clusterSegments2D (line3D.cc:968-1221) on the device: candidate
// enumeration with the reference's `used` bookkeeping, similarity_coll3D, thresholds, first-touch node numbering and the
// symmetric edge list, from flat per-segment tables (l3d_affinity_input).
//
// What the reference does.  For every source segment s with a 3-D hypothesis, in (view, segment) order: walk its
// potential correspondences t (other views, ascending key); skip t if it is `used` -- met earlier in this iteration, or
// s was met while t was the source (an earlier iteration); otherwise mark it, and if t has a hypothesis: one candidate
// edge (family 0), then the same for every segment c collinear with t (family 1).  Finally the segments collinear with s
// itself (family 2).  `used` is a map of maps updated as the loops go.
//
// The same result without the sequential maps.  Within one source, targets of different views never interact (a mark is a
// segment of the target's view), so the walk decomposes into GROUPS = the targets of one view, ascending.  In a group
//   * target t_k is marked as a target  <=>  no EXPANDED earlier target t_i (i < k) lists it as collinear, and it is
//     not m-used;  t_k is expanded  <=>  marked as a target and it has a hypothesis;
//   * a collinear entry c of an expanded t_k is marked  <=>  c is none of t_j (j < k), no expanded t_i (i < k) lists c,
//     and c is not m-used;
//   * x is m-used  <=>  x has a hypothesis xb EARLIER than the source's and xb's iteration marked the source segment d
//     <=>  d is one of xb's targets, or an expanded target of xb (in d's view) lists d as collinear
// (an earlier source can never find a later one m-used, so nothing else can have stopped it from marking d).  Hence the
// only state that flows from source to source is one bit per (source, target): "expanded".  It is computed view by view
// (k_aff_groups: a target's bit only needs the bits of EARLIER views), with the few targets of a group resolved in
// lane-parallel rounds; everything else is data parallel over all sources of all views at once:
//   k_aff_words   how many 64-entry words the flattened (expanded target, collinear entry) sequence of a source takes
//   k_aff_items<0> decides every entry (one ballot = one word), counts the candidates of the source
//   k_aff_items<1> replays the words: similarity, weight, threshold -> candidate arrays in the reference's order
//   k_aff_first / k_aff_nodes / k_aff_edges   first-touch numbering (line3D.cc:1020-1048 etc.) and the edge list
// The family-2 entries need no state at all: x collinear with s is skipped  <=>  x's hypothesis is earlier and x lists s.
//
// Potential correspondences are recorded in both directions (line3D.cc:861-865), so "d is one of t's targets" holds for
// nearly every target t of d: k_aff_sym looks that up for all sources at once (bit 2), and a target with that bit is m-used
// as soon as its hypothesis is earlier -- no bit of another source is read.  Only views holding an entry WITHOUT the
// reverse record (early-return views, cudawrapper.cu:877-878) have to wait for the bits of the views before them: the
// launches of k_aff_groups are cut there; everywhere else consecutive views share one launch.
#include "l3d_sort.hpp"

#include "l3d_ctx.hpp"
#include "l3d_similarity.hpp"

using namespace l3d;

namespace l3d {

struct AffIn {
    int n_views, n_hyp, n_dense, chunk;             // chunk: targets handled per wave pass (64; smaller only in tests)
    const int* seg_base;                            // n_views + 1
    const int* dview;                               // n_dense: view index of a dense segment id
    const Hypothesis* hyp;
    const float* score;
    const int* hyp_dense;                           // n_hyp: dense id of the hypothesis' segment
    const int* best;                                // n_dense: hypothesis index or -1
    const long long* pot_start;                     // n_dense + 1
    const int* pot_tgt;                             // dense ids, ascending per segment
    const long long* coll_start;                    // n_dense + 1
    const int* coll_other;                          // dense ids (same view), ascending per segment
    const float* coll_w;
    unsigned char* flags;                           // per potential correspondence: bit 0 marked as a target, bit 1 expanded
    int pot_trusted;                                // the potential correspondences were built by the library's own products builder (resident tables): k_aff_validate
                                                    // does not walk the table again (600 M entries at 64 x 4000 x 24: 8 ms of every fill)
    int coll_sym;                                   // the collinearity table holds every entry both ways (the reference's always does: segments.h:94-95) -- found by
                                                    // k_aff_validate; then "does an earlier target list c" is asked from c's own short list
};

// d in C(t)?
__device__ __forceinline__ bool coll_has(const AffIn& a, int t, int d)
{
    long long lo = a.coll_start[t], hi = a.coll_start[t + 1];
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        const int x = a.coll_other[mid];
        if (x == d) return true;
        if (x < d) lo = mid + 1; else hi = mid;
    }
    return false;
}

// first potential correspondence of segment x (entries [pb, pe)) whose target is >= id
__device__ __forceinline__ long long pot_lower(const AffIn& a, long long pb, long long pe, int id)
{
    while (pb < pe) {
        const long long mid = (pb + pe) >> 1;
        if (a.pot_tgt[mid] < id) pb = mid + 1; else pe = mid;
    }
    return pb;
}

__device__ __forceinline__ unsigned char load_flag(const AffIn& a, long long e)
{
    return __hip_atomic_load(a.flags + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// is d one of the targets of segment x?
__device__ __forceinline__ bool pot_has(const AffIn& a, int x, int d)
{
    const long long pe = a.pot_start[x + 1];
    const long long p = pot_lower(a, a.pot_start[x], pe, d);
    return p < pe && a.pot_tgt[p] == d;
}

// did the iteration of the EARLIER hypothesis xb mark segment d (of view vd)?
__device__ __forceinline__ bool met_by(const AffIn& a, int xb, int d, int vd)
{
    const int x = a.hyp_dense[xb];
    const long long pe = a.pot_start[x + 1];
    const int hi_id = a.seg_base[vd + 1];
    for (long long e = pot_lower(a, a.pot_start[x], pe, a.seg_base[vd]); e < pe; ++e) {
        const int t = a.pot_tgt[e];
        if (t >= hi_id) break;
        if (t == d) return true;
        if ((load_flag(a, e) & 2) && coll_has(a, t, d)) return true;
    }
    return false;
}

__global__ void k_aff_dview(const int* __restrict__ seg_base, int n_views, int* __restrict__ dview)
{
    const int v = blockIdx.y;
    const int i = seg_base[v] + blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n_views && i < seg_base[v + 1]) dview[i] = v;
}

// The tables come from the caller: every id in range, lists strictly ascending, targets in OTHER views, collinear segments in
// the SAME view, hypotheses numbered in dense order and consistent with `best` -- checked once, in parallel, so that a bad
// table is an error message and not a fault inside the later kernels.
__global__ void k_aff_validate(AffIn a, long long n_pot, long long n_coll, int* __restrict__ bad)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n_dense) {
        const long long pb = a.pot_start[i], pe = a.pot_start[i + 1], cb = a.coll_start[i], ce = a.coll_start[i + 1];
        if (pb < 0 || pe < pb || pe > n_pot || cb < 0 || ce < cb || ce > n_coll) { *bad = 1; return; }
        const int v = a.dview[i];
        if (!a.pot_trusted)
            for (long long e = pb; e < pe; ++e) {
                const int t = a.pot_tgt[e];
                if (t < 0 || t >= a.n_dense || a.dview[t] == v || (e > pb && a.pot_tgt[e - 1] >= t)) { *bad = 2; return; }
            }
        for (long long q = cb; q < ce; ++q) {
            const int x = a.coll_other[q];
            if (x < 0 || x >= a.n_dense || a.dview[x] != v || x == (int)i || (q > cb && a.coll_other[q - 1] >= x)) { *bad = 3; return; }
        }
        // (second word: the table is NOT symmetric -- no error, the kernels then take the general path.  Ranges of x are checked by x's own thread;
        // a bad range there is an error anyway)
        for (long long q = cb; q < ce; ++q) {
            const int x = a.coll_other[q];
            const long long xb = a.coll_start[x], xe = a.coll_start[x + 1];
            if (xb < 0 || xe < xb || xe > n_coll) break;
            if (!coll_has(a, x, (int)i)) { bad[1] = 1; break; }
        }
        const int hb = a.best[i];
        if (hb < -1 || hb >= a.n_hyp || (hb >= 0 && a.hyp_dense[hb] != (int)i)) *bad = 4;
    }
    if (i < a.n_hyp) {
        const int d = a.hyp_dense[i];
        if (d < 0 || d >= a.n_dense || a.best[d] != (int)i || (i > 0 && a.hyp_dense[i - 1] >= d)) *bad = 5;
    }
}

// bit 2 of a source's entry: the target has an earlier hypothesis and records the source among its own targets (=> m-used).
// needs_prev[view] is raised when a target with an earlier hypothesis lacks the reverse record: its m-used test reads bits of
// earlier views.
__global__ __launch_bounds__(256) void k_aff_sym(AffIn a, int* __restrict__ needs_prev, int assume_symmetric)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int si = blockIdx.x * 4 + wave;
    if (si >= a.n_hyp) return;
    const int d = a.hyp_dense[si];
    const long long pe = a.pot_start[d + 1];
    bool need = false;
    for (long long e = a.pot_start[d] + lane; e < pe; e += 64) {
        const int t = a.pot_tgt[e];
        const int hb = a.best[t];
        if (hb >= 0 && hb < si) {
            if (assume_symmetric || pot_has(a, t, d)) a.flags[e] = 4; else need = true;
        }
    }
    if (__ballot(need) && lane == 0) needs_prev[a.dview[d]] = 1;
}

// The "expanded" bits of the sources [h0, h1) (one view, or several that need no bits of each other).  One wave per source, one lane per target (passes of a.chunk
// targets); a target waits for the earlier targets of its group: round r resolves the r-th target of every group.
__global__ __launch_bounds__(256) void k_aff_groups(AffIn a, int h0, int h1)
{
    __shared__ int s_t[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int si = h0 + blockIdx.x * 4 + wave;
    if (si >= h1) return;
    const int d = a.hyp_dense[si];
    const int vi = a.dview[d];
    const long long pb = a.pot_start[d], pe = a.pot_start[d + 1];
    int carry_view = -1;                                                       // view and start of the group the previous pass ended in
    long long carry_gs = pb;
    for (long long c0 = pb; c0 < pe; c0 += a.chunk) {
        const long long e = c0 + lane;
        const bool valid = lane < a.chunk && e < pe;
        int t = -1, hb = -1, tv = -2;
        bool mused = false;
        if (valid) {
            t = a.pot_tgt[e];
            const unsigned char f0 = a.flags[e];                               // (bit 2 only: written by k_aff_sym, an earlier launch)
            hb = a.best[t];
            tv = a.dview[t];
            mused = hb >= 0 && hb < si && ((f0 & 4) || met_by(a, hb, d, vi));
        }
        // first target of the lane's group: the targets ascend, a group is a run of one view
        const int tv_prev = __shfl_up(tv, 1);
        const unsigned long long starts = __ballot(valid && (lane == 0 ? tv != carry_view : tv != tv_prev));
        const unsigned long long below = starts & ((lane == 63 ? 0ull : (1ull << (lane + 1))) - 1ull);
        long long gs = below ? c0 + (63 - __clzll((long long)below)) : carry_gs;
        {   // hand the last group over to the next pass
            const int last = (int)(pe - c0 < (long long)a.chunk ? pe - c0 : (long long)a.chunk) - 1;
            carry_view = __shfl(tv, last);
            carry_gs = __shfl(gs, last);
        }
        s_t[wave][lane] = t;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                 // (the other lanes' targets are read below)
        __builtin_amdgcn_wave_barrier();
        const long long g0 = gs > c0 ? gs : c0;                                // the group's part inside this pass starts here
        // Resolution.  A target is decided once every earlier target of its group is: the first undecided target of a group
        // (its head) has by then been tested against every expanded target before it.  When a head turns out expanded, ALL
        // undecided targets behind it test its collinear list at once -- the usual group (fragments of one line) is done
        // after two iterations however long it is.
        bool marked = false, expanded = false;
        bool resolved = !valid;
        const unsigned long long group_below = ((1ull << lane) - 1ull) & ~((1ull << (int)(g0 - c0)) - 1ull);   // earlier lanes of my group
        if (a.coll_sym) {
            // Symmetric table: "an expanded earlier target lists t" <=> "an element of C(t) is an expanded earlier target of the group" -- asked from
            // t's own short list instead of every earlier target's.  The earlier targets of the group that are collinear with t: in an earlier pass of
            // this wave (their bits are final: read back) or lanes of this pass (predmask).  A lane is decided as soon as all lanes of its predmask
            // are: rounds = the longest chain of collinear targets, not the number of expanded targets of the group.
            unsigned long long predmask = 0ull;
            if (valid && !mused) {
                const int lo_lane = (int)(g0 - c0);
                const int t_first = s_t[wave][lo_lane];                        // first target of my group inside this pass
                const long long qe = a.coll_start[t + 1];
                for (long long q = a.coll_start[t]; q < qe; ++q) {
                    const int x = a.coll_other[q];
                    if (x >= t) break;                                         // (ascending: only earlier targets matter)
                    if (x < t_first) {
                        if (gs < c0 && !resolved) {                            // the group began in an earlier pass
                            const long long p = pot_lower(a, gs, c0, x);
                            if (p < c0 && a.pot_tgt[p] == x && (load_flag(a, p) & 2)) resolved = true;   // listed by an expanded target: not marked
                        }
                    } else {
                        int l0 = lo_lane, l1 = lane;                           // x among the lanes [lo_lane, lane) of my group?
                        while (l0 < l1) { const int mid = (l0 + l1) >> 1; if (s_t[wave][mid] < x) l0 = mid + 1; else l1 = mid; }
                        if (l0 < lane && s_t[wave][l0] == x) predmask |= 1ull << l0;
                    }
                }
            }
            unsigned long long exp_mask = 0ull;
            for (;;) {
                const unsigned long long undecided = __ballot(!resolved);
                if (!undecided) break;
                if (!resolved && (predmask & undecided) == 0ull) {             // every collinear earlier target of the group is decided
                    marked = !mused && (predmask & exp_mask) == 0ull;
                    expanded = marked && hb >= 0;
                    resolved = true;
                }
                exp_mask |= __ballot(expanded);
            }
        } else {
        if (valid) {
            bool hit = false;
            for (long long e2 = gs; e2 < c0 && !hit; ++e2)                     // the group began in an earlier pass of this wave
                if ((load_flag(a, e2) & 2) && coll_has(a, a.pot_tgt[e2], t)) hit = true;
            resolved = hit;                                                    // listed by an expanded target: not marked
        }
        for (;;) {
            const unsigned long long undecided = __ballot(!resolved);
            if (!undecided) break;
            const bool head = !resolved && (undecided & group_below) == 0ull;
            if (head) { marked = !mused; expanded = marked && hb >= 0; resolved = true; }
            const unsigned long long fresh = __ballot(head && expanded) & group_below;   // (at most one bit: my group's head)
            if (!resolved && fresh) {
                if (coll_has(a, s_t[wave][__ffsll((long long)fresh) - 1], t)) resolved = true;
            }
        }
        }
        if (valid) __hip_atomic_store(a.flags + e, (unsigned char)((marked ? 1 : 0) | (expanded ? 2 : 0) | (mused ? 4 : 0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                 // (the next pass of this wave may read them back)
    }
}

// Flattened entries of one pass of targets: every EXPANDED target contributes itself (family 0) and its collinear list
// (family 1).  pre = exclusive prefix of the contributions over the lanes; returns the total.
__device__ __forceinline__ int pass_layout(const AffIn& a, long long e, bool valid, int& t, long long& cs, int& len, int& pre)
{
    const int lane = threadIdx.x & 63;
    t = -1; cs = 0; len = 0;
    if (valid && (load_flag(a, e) & 2)) {
        t = a.pot_tgt[e];
        cs = a.coll_start[t];
        len = 1 + (int)(a.coll_start[t + 1] - cs);
    }
    int incl = len;
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    pre = incl - len;
    return __shfl(incl, 63);
}

__global__ __launch_bounds__(256) void k_aff_words(AffIn a, int h0, int h1, int* __restrict__ nwords)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int si = h0 + blockIdx.x * 4 + wave;
    if (si >= h1) return;
    const int d = a.hyp_dense[si];
    const long long pb = a.pot_start[d], pe = a.pot_start[d + 1];
    int words = 0;
    for (long long c0 = pb; c0 < pe; c0 += a.chunk) {
        int t, len, pre; long long cs;
        const int T = pass_layout(a, c0 + lane, lane < a.chunk && c0 + lane < pe, t, cs, len, pre);
        words += (T + 63) >> 6;
    }
    words += (int)((a.coll_start[d + 1] - a.coll_start[d] + 63) >> 6);
    if (lane == 0) nwords[si] = words;
}

// kEmit = false: decide every flattened entry of the sources [h0, h1); the ballots are stored (words), the candidates counted.
// kEmit = true:  replay the words: candidate k of the source gets its pair, its weight (similarity * mean score
//                [* collinearity weight]) or -1 when the weight does not pass the family's threshold.
// word_off / cnt / item_off are indexed by the source; the words and the candidates of a launch are numbered from the offsets' own zero (a
// block of sources at a time: affinity_fill_core).
template <bool kEmit>
__global__ __launch_bounds__(256) void k_aff_items(AffIn a, int h0, int h1, const int* __restrict__ word_off, unsigned long long* __restrict__ words,
                                                   int* __restrict__ cnt, const int* __restrict__ item_off, int2* __restrict__ pairs,
                                                   float* __restrict__ wgt, float sigma_a, float two_log)
{
    __shared__ int s_pre[4][64];
    __shared__ int s_t[4][64];
    __shared__ long long s_cs[4][64];
    __shared__ long long s_e[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int si = h0 + blockIdx.x * 4 + wave;
    if (si >= h1) return;
    const int d = a.hyp_dense[si];
    const int vi = a.dview[d];
    const long long pb = a.pot_start[d], pe = a.pot_start[d + 1];
    long long w = word_off[si];
    int written = 0;
    const int out0 = kEmit ? item_off[si] : 0;
    const float s_src = kEmit ? a.score[si] : 0.0f;

    auto emit = [&](bool on, int b, int family, float cw) {
        // (all lanes call; `on` lanes hold a candidate (si, b))
        const unsigned long long bal = __ballot(on);
        if (on) {
            const int k = out0 + written + __popcll(bal & ((1ull << lane) - 1ull));
            const float sim = similarity_coll3D(a.hyp[si], a.hyp[b], sigma_a, two_log);
            float wv = 0.5f * (s_src + a.score[b]) * sim;                      // line3D.cc:1014, :1085
            if (family == 2) wv = cw * 0.5f * (s_src + a.score[b]) * sim;      // :1163
            const float thr = family == 0 ? 0.25f : 0.01f;                     // L3D_MIN_AFFINITY / 0.01f
            pairs[k] = make_int2(si, b);
            wgt[k] = wv > thr ? wv : -1.0f;
        }
        written += __popcll(bal);
    };

    for (long long c0 = pb; c0 < pe; c0 += a.chunk) {
        const long long e = c0 + lane;
        int t, len, pre; long long cs;
        const int T = pass_layout(a, e, lane < a.chunk && e < pe, t, cs, len, pre);
        s_pre[wave][lane] = pre;
        s_t[wave][lane] = t; s_cs[wave][lane] = cs; s_e[wave][lane] = e;
        for (int q0 = 0; q0 < T; q0 += 64, ++w) {
            const int q = q0 + lane;
            bool on = false;
            int b = -1, family = 0;
            // the target this entry belongs to: the last contributing lane with pre <= q
            int k = -1;
            const unsigned long long contrib = __ballot(len != 0);
            if (q < T) {
                unsigned long long m = contrib;
                while (m) {                                                    // (few expanded targets per pass)
                    const int j = __ffsll((long long)m) - 1;
                    m &= m - 1ull;
                    if (s_pre[wave][j] <= q) k = j; else break;
                }
            }
            if (!kEmit) {
                if (k >= 0) {
                    const int pos = q - s_pre[wave][k] - 1;
                    const int tk = s_t[wave][k];
                    if (pos < 0) {                                             // the expanded target itself
                        on = true;
                    } else {
                        const int c = a.coll_other[s_cs[wave][k] + pos];
                        const long long ek = s_e[wave][k];
                        const long long gs = pot_lower(a, pb, ek, a.seg_base[a.dview[tk]]);
                        bool skip = false;
                        {   // c is an earlier target of the group?
                            const long long p = pot_lower(a, gs, ek, c);
                            skip = p < ek && a.pot_tgt[p] == c;
                        }
                        if (a.coll_sym) {                                       // an earlier expanded target lists c?  symmetric table: ask c's own list
                            const int t_lo = gs < ek ? a.pot_tgt[gs] : 0x7fffffff;
                            const long long qe = a.coll_start[c + 1];
                            for (long long q2 = a.coll_start[c]; q2 < qe && !skip; ++q2) {
                                const int x = a.coll_other[q2];
                                if (x < t_lo) continue;
                                if (x >= tk) break;                             // (the group's earlier targets are smaller than tk)
                                const long long p = pot_lower(a, gs, ek, x);
                                if (p < ek && a.pot_tgt[p] == x && (load_flag(a, p) & 2)) skip = true;
                            }
                        } else
                        for (long long e2 = gs; e2 < ek && !skip; ++e2)         // an earlier expanded target lists c?
                            if ((load_flag(a, e2) & 2) && coll_has(a, a.pot_tgt[e2], c)) skip = true;
                        if (!skip) {
                            const int hb = a.best[c];
                            if (hb >= 0 && !(hb < si && met_by(a, hb, d, vi))) on = true;   // marked; a candidate when it has a hypothesis
                        }
                    }
                }
                const unsigned long long bal = __ballot(on);
                if (lane == 0) words[w] = bal;
                written += __popcll(bal);
            } else {
                const unsigned long long bal = words[w];
                on = (bal >> lane) & 1ull;
                if (on) {
                    const int pos = q - s_pre[wave][k] - 1;
                    if (pos < 0) { b = a.best[s_t[wave][k]]; family = 0; }
                    else { b = a.best[a.coll_other[s_cs[wave][k] + pos]]; family = 1; }
                }
                emit(on, b, family, 0.0f);
            }
        }
    }
    // family 2: the segments collinear with the source itself (line3D.cc:1141-1214)
    const long long cb = a.coll_start[d], ce = a.coll_start[d + 1];
    for (long long q0 = cb; q0 < ce; q0 += 64, ++w) {
        const long long q = q0 + lane;
        if (!kEmit) {
            bool on = false;
            if (q < ce) {
                const int x = a.coll_other[q];
                const int hb = a.best[x];
                on = hb >= 0 && !(hb < si && coll_has(a, x, d));
            }
            const unsigned long long bal = __ballot(on);
            if (lane == 0) words[w] = bal;
            written += __popcll(bal);
        } else {
            const unsigned long long bal = words[w];
            const bool on = (bal >> lane) & 1ull;
            emit(on, on ? a.best[a.coll_other[q]] : -1, 2, on ? a.coll_w[q] : 0.0f);
        }
    }
    if (!kEmit && lane == 0) cnt[si] = written;
}

// first-touch numbering: a node is created the first time a hypothesis appears in a candidate that passed its threshold,
// source before target (line3D.cc:1020-1048 and the two other families).  Candidates are produced a block of sources at a time; the
// position of a candidate in the whole enumeration is 64 bits wide (pos0 = the candidates in front of this block), so `first` holds
// 2 * position + side as an unsigned 64-bit minimum and nothing here depends on the total fitting 31 bits.
constexpr unsigned long long kFirstNone = ~0ull;
__global__ void k_aff_first(const int2* __restrict__ pairs, const float* __restrict__ wgt, int n, unsigned long long pos0, unsigned long long* __restrict__ first, int* __restrict__ kept)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const bool on = wgt[k] > 0.0f;
    kept[k] = on ? 1 : 0;
    if (on) { const unsigned long long p = 2ull * (pos0 + (unsigned long long)k); atomicMin(&first[pairs[k].x], p); atomicMin(&first[pairs[k].y], p + 1ull); }
}
// the candidates of a block that passed, appended to the list of all passed candidates (in enumeration order: erank = exclusive scan of kept)
__global__ void k_aff_compact(const int2* __restrict__ pairs, const float* __restrict__ wgt, const int* __restrict__ erank, int n, const int* __restrict__ loc2glob,
                              int2* __restrict__ out_pairs, float* __restrict__ out_w)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n || !(wgt[k] > 0.0f)) return;
    const int r = erank[k];
    out_pairs[r] = loc2glob ? make_int2(loc2glob[pairs[k].x], loc2glob[pairs[k].y]) : pairs[k];
    out_w[r] = wgt[k];
}
// a rank's first-touch minima (local hypothesis numbers) into the whole-scene array
__global__ void k_aff_first_scatter(const unsigned long long* __restrict__ first, const int* __restrict__ loc2glob, int n, unsigned long long* __restrict__ out)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < n) out[loc2glob[h]] = first[h];
}
// node index = rank of a hypothesis' first position among all first positions: (first, hypothesis) sorted by first
__global__ void k_aff_node_keys(const unsigned long long* __restrict__ first, int n_hyp, unsigned long long none_key, unsigned long long* __restrict__ keys, unsigned* __restrict__ vals)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_hyp) return;
    const unsigned long long f = first[h];
    keys[h] = f == kFirstNone ? none_key : f;
    vals[h] = (unsigned)h;
}
__global__ void k_aff_nodes(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals, int n_hyp, unsigned long long none_key,
                            int* __restrict__ node, int* __restrict__ node_hyp, int* __restrict__ n_nodes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_hyp) return;
    const bool touched = keys[i] != none_key;
    const int h = (int)vals[i];
    node[h] = touched ? i : -1;
    if (touched) { node_hyp[i] = h; if (i + 1 == n_hyp || keys[i + 1] == none_key) *n_nodes = i + 1; }
}
__global__ void k_aff_edges(const int2* __restrict__ pairs, const float* __restrict__ wgt, const int* __restrict__ node, long long n, l3d_edge* __restrict__ A)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int na = node[pairs[r].x], nb = node[pairs[r].y];
    A[2 * (size_t)r] = { na, nb, wgt[r] };
    A[2 * (size_t)r + 1] = { nb, na, wgt[r] };
}
// 64-bit sum of n ints (out zeroed by the caller): whether a whole range of sources fits ONE block is a scalar question -- the per-source counts
// are only downloaded when it does not
__global__ __launch_bounds__(256) void k_aff_sum64(const int* __restrict__ in, int n, unsigned long long* __restrict__ out)
{
    unsigned long long t = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) t += (unsigned long long)(unsigned)in[i];
    for (int o = 32; o > 0; o >>= 1) { const unsigned lo = __shfl_down((unsigned)t, o), hi = __shfl_down((unsigned)(t >> 32), o); t += ((unsigned long long)hi << 32) | lo; }
    if ((threadIdx.x & 63) == 0 && t) atomicAdd(out, t);
}
__global__ void k_aff_fill64(unsigned long long* p, int n, unsigned long long v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
}  // namespace l3d

namespace {

// exclusive sum of n ints into out[0..n] (out[n] = total); tmp: the context's scratch
int scan_excl(l3d_ctx* c, const int* in, int* out, int n, hipStream_t st)
{
    size_t bytes = 0;
    HIPCHK(c, exclusive_sum_int(nullptr, bytes, in, out, n + 1, st));
    HIPCHK(c, c->g7.reserve(bytes + 256));
    HIPCHK(c, exclusive_sum_int(c->g7.p, bytes, in, out, n + 1, st));
    return L3D_OK;
}

}  // namespace

namespace {

// a device buffer grown WITHOUT losing its first `used` bytes (DevBuf::reserve drops the content)
int grow_keep(l3d_ctx* c, DevBuf& b, size_t bytes, size_t used, hipStream_t st)
{
    if (bytes <= b.cap) return L3D_OK;
    const size_t want = bytes + bytes / 2 + 4096;
    void* np = nullptr;
    if (hipMalloc(&np, want) != hipSuccess) { (void)hipGetLastError(); return fail(c, L3D_ERR_NOMEM, "affinity fill: growing the list of passed candidates to " + std::to_string(want >> 20) + " MB"); }
    if (used && b.p) { HIPCHK(c, hipMemcpyAsync(np, b.p, used, hipMemcpyDeviceToDevice, st)); HIPCHK(c, hipStreamSynchronize(st)); }
    if (b.p) (void)hipFree(b.p);
    b.p = np; b.cap = want;
    return L3D_OK;
}

// First-touch node numbering (line3D.cc:1020-1048 and the two other families) and the symmetric edge list from the candidates that passed,
// in enumeration order, with the first-touch minima of all hypotheses: node index = rank of a hypothesis' first position among all first
// positions (one sort of n_hyp 64-bit keys: nothing depends on the number of candidates), edges (a,b,w), (b,a,w) per passed candidate.
int affinity_number_edges(l3d_ctx* c, const unsigned long long* first, int nh, const int2* pass_pairs, const float* pass_w, long long n_passed,
                          l3d_edge** edges_out, int* n_edges_out, int32_t** node_hyp_out, int* n_nodes_out)
{
    hipStream_t st = c->stream;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    if (n_passed == 0) return L3D_OK;
    // the edge list and everything behind it (diffusion, edge order, merge loop) count entries with 31 bits: reported, not wrapped
    if (2 * n_passed > 0x7ffffff0ll) return fail(c, L3D_ERR_UNSUPPORTED, "affinity fill: " + std::to_string(2 * n_passed) + " affinity entries -- the clustering stages take at most 2^31");
    const size_t kb = al((size_t)nh * 8), vb = al((size_t)nh * 4);
    size_t tb = 0;
    HIPCHK(c, sort_pairs_u64_u32(nullptr, tb, nullptr, nullptr, nullptr, nullptr, nh, 0, 64, st));
    HIPCHK(c, c->g5.reserve(2 * kb + 3 * vb + al(tb) + 1024));
    unsigned char* sc = c->g5.as<unsigned char>();
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(sc);
    unsigned long long* keys2 = reinterpret_cast<unsigned long long*>(sc + kb);
    unsigned* vals = reinterpret_cast<unsigned*>(sc + 2 * kb);
    unsigned* vals2 = reinterpret_cast<unsigned*>(sc + 2 * kb + vb);
    int* node = reinterpret_cast<int*>(sc + 2 * kb + 2 * vb);
    void* tmp = sc + 2 * kb + 3 * vb;
    int* d_nn = reinterpret_cast<int*>(sc + 2 * kb + 3 * vb + al(tb));
    // (every real key is below 2^63: positions carry at most a rank in bits 44.. and twice a 43-bit count; the "never touched" key sorts last)
    const unsigned long long none_key = 1ull << 63;
    HIPCHK(c, hipMemsetAsync(d_nn, 0, 4, st));
    hipLaunchKernelGGL(k_aff_node_keys, dim3((nh + 255) / 256), dim3(256), 0, st, first, nh, none_key, keys, vals);
    HIPCHK(c, sort_pairs_u64_u32(tmp, tb, keys, keys2, vals, vals2, nh, 0, 64, st));
    // node_hyp sits behind the edges in g6 (l3d_fit_labelled_clusters reads it there); at most nh nodes
    HIPCHK(c, c->g6.reserve(al((size_t)n_passed * 2 * sizeof(l3d_edge)) + (size_t)nh * 4 + 512));
    l3d_edge* dA = c->g6.as<l3d_edge>();
    int* d_node_hyp = reinterpret_cast<int*>(c->g6.as<char>() + al((size_t)n_passed * 2 * sizeof(l3d_edge)));
    hipLaunchKernelGGL(k_aff_nodes, dim3((nh + 255) / 256), dim3(256), 0, st, keys2, vals2, nh, none_key, node, d_node_hyp, d_nn);
    hipLaunchKernelGGL(k_aff_edges, dim3((unsigned)((n_passed + 255) / 256)), dim3(256), 0, st, pass_pairs, pass_w, node, n_passed, dA);
    int n_nodes = 0;
    HIPCHK(c, hipMemcpyAsync(&n_nodes, d_nn, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    // (edges_out == nullptr: the list stays on the device only -- l3d_perform_clustering_device walks it there, l3d_resident_edges_get copies it)
    l3d_edge* A = edges_out ? static_cast<l3d_edge*>(malloc((size_t)n_passed * 2 * sizeof(l3d_edge))) : nullptr;
    int32_t* nh_out = static_cast<int32_t*>(malloc((size_t)n_nodes * 4 + 4));
    if ((edges_out && !A) || !nh_out) { free(A); free(nh_out); return fail(c, L3D_ERR_NOMEM, "affinity fill: host allocation failed"); }
    hipError_t e1 = A ? hipMemcpyAsync(A, dA, (size_t)n_passed * 2 * sizeof(l3d_edge), hipMemcpyDeviceToHost, st) : hipSuccess;
    hipError_t e2 = hipMemcpyAsync(nh_out, d_node_hyp, (size_t)n_nodes * 4, hipMemcpyDeviceToHost, st);
    hipError_t e3 = hipStreamSynchronize(st);
    hipError_t e4 = hipGetLastError();
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
        free(A); free(nh_out);
        return fail(c, L3D_ERR_HIP, std::string("affinity fill: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2 != hipSuccess ? e2 : e3 != hipSuccess ? e3 : e4));
    }
    if (c->opt.timing) fprintf(stderr, "[l3d affinity] numbering + edges: %d nodes, %lld entries%s\n", n_nodes, 2 * n_passed, A ? " (downloaded)" : "");
    if (edges_out) *edges_out = A;
    *n_edges_out = (int)(2 * n_passed); *node_hyp_out = nh_out; *n_nodes_out = n_nodes;
    c->resident_edges = (int)(2 * n_passed);                                   // (the list stays in g6 for l3d_clustering_edges)
    c->kept_edges = 0;
    c->resident_nodes = n_nodes; c->resident_nodes_p = d_node_hyp;
    c->resident_hyp = nh;
    return L3D_OK;
}

// The fill proper, on tables that are already on the device (`a`: everything but dview / flags, which live in the context's scratch).
// seg_base_h / vhb_h: host copies of the view tables (launch geometry).  part != nullptr: a rank's part of a fill sharded by source key
// (SURVEY 8e): candidates of the sources [h0, h1) only, left as the passed list + first-touch minima in the context (no numbering).
int affinity_fill_core(l3d_ctx* c, AffIn a, const int32_t* seg_base_h, const int32_t* vhb_h, long long n_pot, long long n_coll, float sigma_a,
                       l3d_edge** edges_out, int* n_edges_out, int32_t** node_hyp_out, int* n_nodes_out, int* n_candidates_out, const l3d::FillPart* part)
{
    const int V = a.n_views, nh = a.n_hyp, nd = a.n_dense;
    hipStream_t st = c->stream;
    const bool timing = c->opt.timing != 0;
    double tl = now_s();
    auto lap = [&](const char* what) { if (timing) { (void)hipStreamSynchronize(st); const double t = now_s(); fprintf(stderr, "[l3d affinity] %-34s %8.2f ms\n", what, (t - tl) * 1e3); tl = t; } };
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    {   // dview + per-entry flags
        const size_t o_fl = al((size_t)nd * 4);
        HIPCHK(c, c->products.aux.reserve(o_fl + al((size_t)n_pot + 4)));
        a.dview = c->products.aux.as<int>();
        a.flags = reinterpret_cast<unsigned char*>(c->products.aux.as<char>() + o_fl);
        HIPCHK(c, hipMemsetAsync(a.flags, 0, (size_t)n_pot + 4, st));
    }
    a.chunk = 64;
    if (c->opt.aff_chunk > 0) a.chunk = std::min(64, c->opt.aff_chunk);       // tests: forces multi-pass groups on small scenes
    char* const base = nullptr; (void)base;
    int maxS = 1;
    for (int v = 0; v < V; ++v) maxS = std::max(maxS, seg_base_h[v + 1] - seg_base_h[v]);
    hipLaunchKernelGGL(k_aff_dview, dim3((maxS + 255) / 256, V), dim3(256), 0, st, a.seg_base, V, const_cast<int*>(a.dview));
    {
        HIPCHK(c, c->g1.reserve(((size_t)nh + 2) * 4 * 4 + (size_t)V * 4 + 1024));
        int* bad = c->g1.as<int>();
        HIPCHK(c, hipMemsetAsync(bad, 0, 8, st));
        const int nmax = std::max(nd, nh);
        a.coll_sym = 0;
        hipLaunchKernelGGL(k_aff_validate, dim3((nmax + 255) / 256), dim3(256), 0, st, a, n_pot, n_coll, bad);
        int h_bad2[2] = { 0, 0 };
        HIPCHK(c, hipMemcpyAsync(h_bad2, bad, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        const int h_bad = h_bad2[0];
        a.coll_sym = (h_bad2[1] == 0 && c->opt.aff_sym != 0) ? 1 : 0;
        static const char* what[] = { "", "a CSR range is out of bounds", "a potential correspondence is out of range, in the source's own view, or out of order",
                                      "a collinearity entry is out of range, in another view, a self entry, or out of order", "best[] and hyp_dense[] disagree",
                                      "hypotheses are not numbered in dense order" };
        if (h_bad) return fail(c, L3D_ERR_INVALID, std::string("affinity fill: ") + what[std::min(h_bad, 5)]);
    }
    lap("upload + table check");

    // ---- expanded bits: consecutive views share a launch unless one of them reads bits of the views before it
    const dim3 gsrc((nh + 3) / 4);
    HIPCHK(c, c->g1.reserve(((size_t)nh + 2) * 4 * 4 + (size_t)V * 4 + 1024));
    int* needs_prev = c->g1.as<int>() + ((size_t)nh + 2) * 4;
    HIPCHK(c, hipMemsetAsync(needs_prev, 0, (size_t)V * 4, st));
    { ProfScope p(c, "aff_sym", st); hipLaunchKernelGGL(k_aff_sym, gsrc, dim3(256), 0, st, a, needs_prev, part ? part->assume_symmetric : 0); }
    std::vector<int> cut((size_t)V, 0);
    HIPCHK(c, hipMemcpyAsync(cut.data(), needs_prev, (size_t)V * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const bool per_view = c->opt.aff_per_view != 0;   // tests: the general schedule everywhere
    int n_launches = 0;
    for (int v0 = 0; v0 < V;) {
        int v1 = v0 + 1;
        while (v1 < V && !cut[(size_t)v1] && !per_view) ++v1;
        const int h0 = vhb_h[v0], h1 = vhb_h[v1];
        if (h1 > h0) { ProfScope p(c, "aff_groups", st); hipLaunchKernelGGL(k_aff_groups, dim3((h1 - h0 + 3) / 4), dim3(256), 0, st, a, h0, h1); ++n_launches; }
        v0 = v1;
    }
    if (timing) fprintf(stderr, "[l3d affinity] %d launch(es) of k_aff_groups for %d views\n", n_launches, V);
    lap("reverse records + groups");

    // ---- words, decisions, candidates: a BLOCK OF SOURCES at a time.  The reference enumerates its candidates source by source (line3D.cc:996-1221);
    // nothing in that order needs all of them at once, so the transient arrays -- decision words, candidate pairs and weights, their flags and
    // ranks: 20 bytes per candidate -- are bounded by the block, and no count of the whole fill has to fit 31 bits (round 4 stopped at 2^30
    // candidates: 256 x 4000 x 24).  Outer blocks are cut by decision WORDS (known before the decisions: 2^26 = 512 MB), inner blocks by the
    // candidates the decisions counted (2^27 = 2.7 GB of transients; options aff_block / aff_word_block force small blocks in tests).
    // What survives a block: the candidates that passed their threshold, appended in enumeration order (12 bytes each), and the 64-bit
    // first-touch minimum per hypothesis.
    const int h0 = part ? part->h0 : 0, h1 = part ? part->h1 : nh;
    const unsigned long long pos_base = part ? part->pos_base : 0ull;
    const int* loc2glob = part ? part->loc2glob : nullptr;
    int* nwords = c->g1.as<int>();
    int* word_off = nwords + (nh + 2);
    int* cnt = word_off + (nh + 2);
    int* item_off = cnt + (nh + 2);
    HIPCHK(c, hipMemsetAsync(nwords, 0, ((size_t)nh + 2) * 4 * 4, st));
    HIPCHK(c, c->aff_first.reserve((size_t)nh * 8 + 64));              // (+ a 64-bit scratch word behind the minima)
    unsigned long long* first = c->aff_first.as<unsigned long long>();
    hipLaunchKernelGGL(k_aff_fill64, dim3((nh + 255) / 256), dim3(256), 0, st, first, nh, kFirstNone);
    c->fill_items = 0; c->fill_passed = 0;
    const float two_log = 2.0f * logf(0.01f);      // view.cc:376
    long long n_items_total = 0, n_passed = 0;
    if (h1 > h0) {
        const dim3 gpart((h1 - h0 + 3) / 4);
        { ProfScope p(c, "aff_words", st); hipLaunchKernelGGL(k_aff_words, gpart, dim3(256), 0, st, a, h0, h1, nwords); }
        std::vector<int> h_nwords, h_cnt;
        const long long word_budget = c->opt.aff_word_block > 0 ? c->opt.aff_word_block : (1ll << 26);
        const long long cand_budget = c->opt.aff_block > 0 ? c->opt.aff_block : (1ll << 27);
        // (the usual scene is ONE block: asked with a 64-bit sum on the device -- a scalar comes back, not a table)
        unsigned long long* d_sum = reinterpret_cast<unsigned long long*>(c->aff_first.as<unsigned char>() + (size_t)nh * 8 + 8);
        auto device_sum = [&](const int* in, int n, unsigned long long& out) -> int {
            HIPCHK(c, hipMemsetAsync(d_sum, 0, 8, st));
            hipLaunchKernelGGL(k_aff_sum64, dim3(std::min(1024, (n + 255) / 256)), dim3(256), 0, st, in, n, d_sum);
            HIPCHK(c, hipMemcpyAsync(&out, d_sum, 8, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            return L3D_OK;
        };
        unsigned long long words_all = 0;
        if (int rc = device_sum(nwords + h0, h1 - h0, words_all)) return rc;
        lap("  words");
        const bool one_wblock = (long long)words_all <= word_budget;
        if (!one_wblock) {
            h_nwords.resize((size_t)(h1 - h0));
            HIPCHK(c, hipMemcpyAsync(h_nwords.data(), nwords + h0, (size_t)(h1 - h0) * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
        }
        int n_wblocks = 0, n_cblocks = 0;
        for (int w0 = h0; w0 < h1;) {
            long long nw = 0;
            int w1 = w0;
            if (one_wblock) { nw = (long long)words_all; w1 = h1; }
            else while (w1 < h1 && (w1 == w0 || nw + h_nwords[(size_t)(w1 - h0)] <= word_budget)) { nw += h_nwords[(size_t)(w1 - h0)]; ++w1; }
            if (nw > 0x7ffffff0ll) return fail(c, L3D_ERR_UNSUPPORTED, "affinity fill: one source segment's candidate walk takes more than 2^31 decision words");
            ++n_wblocks;
            if (int rc = scan_excl(c, nwords + w0, word_off + w0, w1 - w0, st)) return rc;
            HIPCHK(c, c->g2.reserve(((size_t)nw + 1) * 8));
            { ProfScope p(c, "aff_decide", st);
              hipLaunchKernelGGL(k_aff_items<false>, dim3((w1 - w0 + 3) / 4), dim3(256), 0, st, a, w0, w1, word_off, c->g2.as<unsigned long long>(), cnt, (const int*)nullptr, (int2*)nullptr, (float*)nullptr, sigma_a, two_log); }
            unsigned long long cand_all = 0;
            if (int rc = device_sum(cnt + w0, w1 - w0, cand_all)) return rc;
            lap("  decisions");
            const bool one_cblock = (long long)cand_all <= cand_budget;
            if (!one_cblock) {
                h_cnt.resize((size_t)(w1 - w0));
                HIPCHK(c, hipMemcpyAsync(h_cnt.data(), cnt + w0, (size_t)(w1 - w0) * 4, hipMemcpyDeviceToHost, st));
                HIPCHK(c, hipStreamSynchronize(st));
            }
            for (int g0 = w0; g0 < w1;) {
                long long ni = 0;
                int g1 = g0;
                if (one_cblock) { ni = (long long)cand_all; g1 = w1; }
                else while (g1 < w1 && (g1 == g0 || ni + h_cnt[(size_t)(g1 - w0)] <= cand_budget)) { ni += h_cnt[(size_t)(g1 - w0)]; ++g1; }
                if (ni > 0x3ffffff0ll) return fail(c, L3D_ERR_UNSUPPORTED, "affinity fill: one source segment has more than 2^30 candidate pairs");
                if (ni > 0) {
                    ++n_cblocks;
                    const int n_items = (int)ni;
                    if (int rc = scan_excl(c, cnt + g0, item_off + g0, g1 - g0, st)) return rc;
                    HIPCHK(c, c->g3.reserve((size_t)n_items * 8 + 256));
                    HIPCHK(c, c->g4.reserve((size_t)n_items * 4 + 256));
                    int2* pairs = c->g3.as<int2>();
                    float* wgt = c->g4.as<float>();
                    { ProfScope p(c, "aff_emit", st);
                      hipLaunchKernelGGL(k_aff_items<true>, dim3((g1 - g0 + 3) / 4), dim3(256), 0, st, a, g0, g1, word_off, c->g2.as<unsigned long long>(), (int*)nullptr, item_off, pairs, wgt, sigma_a, two_log); }
                    // kept[] n_items + 1 | erank[] n_items + 1
                    HIPCHK(c, c->g5.reserve(((size_t)n_items + 1) * 2 * 4 + 1024));
                    int* kept = c->g5.as<int>();
                    int* erank = kept + ((size_t)n_items + 1);
                    HIPCHK(c, hipMemsetAsync(kept + n_items, 0, 4, st));
                    hipLaunchKernelGGL(k_aff_first, dim3((n_items + 255) / 256), dim3(256), 0, st, pairs, wgt, n_items, pos_base + (unsigned long long)n_items_total, first, kept);
                    if (int rc = scan_excl(c, kept, erank, n_items, st)) return rc;
                    int n_kept_blk = 0;
                    HIPCHK(c, hipMemcpyAsync(&n_kept_blk, erank + n_items, 4, hipMemcpyDeviceToHost, st));
                    HIPCHK(c, hipStreamSynchronize(st));
                    lap("  candidates (similarity, first touch, ranks)");
                    if (n_kept_blk > 0) {
                        if (int rc = grow_keep(c, c->aff_pass_pairs, (size_t)(n_passed + n_kept_blk) * 8 + 256, (size_t)n_passed * 8, st)) return rc;
                        if (int rc = grow_keep(c, c->aff_pass_w, (size_t)(n_passed + n_kept_blk) * 4 + 256, (size_t)n_passed * 4, st)) return rc;
                        hipLaunchKernelGGL(k_aff_compact, dim3((n_items + 255) / 256), dim3(256), 0, st, pairs, wgt, erank, n_items, loc2glob,
                                           c->aff_pass_pairs.as<int2>() + n_passed, c->aff_pass_w.as<float>() + n_passed);
                    }
                    n_items_total += n_items; n_passed += n_kept_blk;
                }
                g0 = g1;
            }
            w0 = w1;
        }
        if (timing) fprintf(stderr, "[l3d affinity] sources %d..%d in %d word block(s), %d candidate block(s): %lld candidates, %lld passed\n", h0, h1, n_wblocks, n_cblocks, n_items_total, n_passed);
    }
    HIPCHK(c, hipStreamSynchronize(st));
    { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(c, L3D_ERR_HIP, std::string("affinity fill: ") + hipGetErrorString(e_)); }
    lap("words + decisions + candidates");
    c->fill_items = n_items_total; c->fill_passed = n_passed;
    if (n_candidates_out) *n_candidates_out = (int)std::min<long long>(n_items_total, 0x7fffffffll);      // (l3d_last_fill_counts has the 64-bit figures)
    if (part) return L3D_OK;            // (a rank's part of a sharded fill: the caller puts the parts together, then affinity_number_edges)
    return affinity_number_edges(c, first, nh, c->aff_pass_pairs.as<int2>(), c->aff_pass_w.as<float>(), n_passed, edges_out, n_edges_out, node_hyp_out, n_nodes_out);
}

}  // namespace

extern "C" {

int l3d_affinity_fill(l3d_ctx* c, const l3d_affinity_input* in, l3d_edge** edges_out, int* n_edges_out, int32_t** node_hyp_out, int* n_nodes_out,
                      int* n_candidates_out)
{
    if (!c) return L3D_ERR_INVALID;
    if (!in || !edges_out || !n_edges_out || !node_hyp_out || !n_nodes_out) return fail(c, L3D_ERR_INVALID, "bad argument");
    *edges_out = nullptr; *n_edges_out = 0; *node_hyp_out = nullptr; *n_nodes_out = 0;
    c->resident_edges = 0; c->kept_edges = 0; c->resident_nodes = 0; c->resident_labels = 0;
    if (n_candidates_out) *n_candidates_out = 0;
    const int V = in->n_views, nh = in->n_hyp;
    if (V < 0 || nh < 0 || (V > 0 && (!in->seg_base || !in->view_hyp_begin))) return fail(c, L3D_ERR_INVALID, "bad argument");
    if (nh == 0) return L3D_OK;
    const int nd = in->seg_base[V];
    if (nd <= 0 || !in->hyp || !in->score || !in->hyp_dense || !in->best || !in->pot_start || !in->coll_start) return fail(c, L3D_ERR_INVALID, "bad argument");
    // the view tables decide where k_aff_dview writes and how the arena is laid out: checked on the host before anything is launched
    // (ascending from 0 up to seg_base[V] = nd, so no view's range leaves the dense ids)
    if (in->pot_start[0] != 0 || in->coll_start[0] != 0) return fail(c, L3D_ERR_INVALID, "affinity fill: CSR tables must start at 0");
    for (int v = 0; v < V; ++v)
        if (in->seg_base[v + 1] < in->seg_base[v] || in->view_hyp_begin[v + 1] < in->view_hyp_begin[v]) return fail(c, L3D_ERR_INVALID, "affinity fill: view tables must ascend");
    if (in->seg_base[0] != 0 || in->view_hyp_begin[0] != 0 || in->view_hyp_begin[V] != nh) return fail(c, L3D_ERR_INVALID, "affinity fill: view tables do not cover the hypotheses");
    const long long n_pot = in->pot_start[nd], n_coll = in->coll_start[nd];
    if (n_pot < 0 || n_coll < 0 || n_coll > 0x7fffffffll || (n_pot > 0 && !in->pot_tgt) || (n_coll > 0 && (!in->coll_other || !in->coll_w)))
        return fail(c, L3D_ERR_INVALID, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;

    // ---- inputs to the device (one arena, 256-byte aligned slices)
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_base = 0;
    const size_t o_score = o_base + al((size_t)(V + 1) * 4);
    const size_t o_hd = o_score + al((size_t)nh * 4);
    const size_t o_best = o_hd + al((size_t)nh * 4);
    const size_t o_ps = o_best + al((size_t)nd * 4);
    const size_t o_pt = o_ps + al((size_t)(nd + 1) * 8);
    const size_t o_cs = o_pt + al((size_t)n_pot * 4 + 4);
    const size_t o_co = o_cs + al((size_t)(nd + 1) * 8);
    const size_t o_cw = o_co + al((size_t)n_coll * 4 + 4);
    const size_t total = o_cw + al((size_t)n_coll * 4 + 4);
    HIPCHK(c, c->g0.reserve(total));
    char* base = c->g0.as<char>();
    auto up = [&](size_t off, const void* src, size_t bytes) { return bytes ? hipMemcpyAsync(base + off, src, bytes, hipMemcpyHostToDevice, st) : hipSuccess; };
    HIPCHK(c, up(o_base, in->seg_base, (size_t)(V + 1) * 4));
    // (the hypothesis table gets a buffer of its own: the line fit of the same finish reads it again, l3d_fit_clusters)
    c->resident_hyp = 0;
    HIPCHK(c, c->aff_hyp.reserve((size_t)nh * sizeof(Hypothesis) + 64));
    HIPCHK(c, hipMemcpyAsync(c->aff_hyp.p, in->hyp, (size_t)nh * sizeof(Hypothesis), hipMemcpyHostToDevice, st));
    HIPCHK(c, up(o_score, in->score, (size_t)nh * 4));
    HIPCHK(c, up(o_hd, in->hyp_dense, (size_t)nh * 4));
    HIPCHK(c, up(o_best, in->best, (size_t)nd * 4));
    HIPCHK(c, up(o_ps, in->pot_start, (size_t)(nd + 1) * 8));
    HIPCHK(c, up(o_pt, in->pot_tgt, (size_t)n_pot * 4));
    HIPCHK(c, up(o_cs, in->coll_start, (size_t)(nd + 1) * 8));
    HIPCHK(c, up(o_co, in->coll_other, (size_t)n_coll * 4));
    HIPCHK(c, up(o_cw, in->coll_w, (size_t)n_coll * 4));
    AffIn a;
    a.n_views = V; a.n_hyp = nh; a.n_dense = nd;
    a.seg_base = reinterpret_cast<const int*>(base + o_base);
    a.hyp = c->aff_hyp.as<Hypothesis>();
    a.score = reinterpret_cast<const float*>(base + o_score);
    a.hyp_dense = reinterpret_cast<const int*>(base + o_hd);
    a.best = reinterpret_cast<const int*>(base + o_best);
    a.pot_start = reinterpret_cast<const long long*>(base + o_ps);
    a.pot_tgt = reinterpret_cast<const int*>(base + o_pt);
    a.coll_start = reinterpret_cast<const long long*>(base + o_cs);
    a.coll_other = reinterpret_cast<const int*>(base + o_co);
    a.coll_w = reinterpret_cast<const float*>(base + o_cw);
    a.pot_trusted = 0; a.coll_sym = 0;
    return affinity_fill_core(c, a, in->seg_base, in->view_hyp_begin, n_pot, n_coll, in->sigma_a, edges_out, n_edges_out, node_hyp_out, n_nodes_out, n_candidates_out, nullptr);
}

int l3d_last_fill_counts(l3d_ctx* c, int64_t* n_candidates, int64_t* n_passed)
{
    if (!c) return L3D_ERR_INVALID;
    if (n_candidates) *n_candidates = c->fill_items;
    if (n_passed) *n_passed = c->fill_passed;
    return L3D_OK;
}

}  // extern "C"

namespace {

// the tables of a fill on the resident products (hypotheses of l3d_products_hypotheses, potential correspondences and best matches of the
// chain): only the collinearity CSR comes from the host -- uploaded when it changed, otherwise the copy of the previous call is used
int resident_tables(l3d_ctx* c, const int64_t* coll_start, const int32_t* coll_other, const float* coll_w, int coll_changed, AffIn& a, long long& n_coll_out)
{
    Products& P = c->products;
    const int nd = P.n_dense, nh = P.n_hyp, V = P.n_views_all;
    hipStream_t st = c->stream;
    const long long n_coll = coll_start[nd];
    if (coll_start[0] != 0 || n_coll < 0 || n_coll > 0x7fffffffll || (n_coll > 0 && (!coll_other || !coll_w))) return fail(c, L3D_ERR_INVALID, "affinity fill: bad collinearity table");
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_cs = 0, o_co = al((size_t)(nd + 1) * 8), o_cw = o_co + al((size_t)n_coll * 4 + 4), o_sb = o_cw + al((size_t)n_coll * 4 + 4);
    // the resident copy is reused only for the table it was uploaded from: the caller's flag, and -- because that flag is kept by host code
    // that also feeds the non-resident fill -- sizes and a checksum of the row starts and of the dense map
    unsigned long long sig = 1469598103934665603ull;
    auto mix = [&](unsigned long long v) { sig = (sig ^ v) * 1099511628211ull; };
    mix((unsigned long long)nd); mix((unsigned long long)V); mix((unsigned long long)n_coll);
    for (int k = 0; k <= nd; ++k) mix((unsigned long long)coll_start[k]);
    for (int v = 0; v <= V; ++v) mix((unsigned long long)P.seg_base[(size_t)v]);
    if (coll_changed || P.coll_n != n_coll || P.coll_sig != sig || P.coll.cap < o_sb + al((size_t)(V + 1) * 4)) {
        HIPCHK(c, P.coll.reserve(o_sb + al((size_t)(V + 1) * 4)));
        char* cb = P.coll.as<char>();
        HIPCHK(c, hipMemcpyAsync(cb + o_cs, coll_start, (size_t)(nd + 1) * 8, hipMemcpyHostToDevice, st));
        if (n_coll) {
            HIPCHK(c, hipMemcpyAsync(cb + o_co, coll_other, (size_t)n_coll * 4, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(cb + o_cw, coll_w, (size_t)n_coll * 4, hipMemcpyHostToDevice, st));
        }
        HIPCHK(c, hipMemcpyAsync(cb + o_sb, P.seg_base.data(), (size_t)(V + 1) * 4, hipMemcpyHostToDevice, st));
        P.coll_n = n_coll; P.coll_sig = sig;
    }
    char* cb = P.coll.as<char>();
    a.n_views = V; a.n_hyp = nh; a.n_dense = nd; a.chunk = 64;
    a.seg_base = reinterpret_cast<const int*>(cb + o_sb);
    a.dview = nullptr; a.flags = nullptr;
    a.hyp = c->aff_hyp.as<Hypothesis>();
    a.score = P.score.as<float>();
    a.hyp_dense = P.hyp_dense.as<int>();
    a.best = P.best_hyp.as<int>();
    a.pot_start = P.pot_start.as<long long>();
    a.pot_tgt = P.pot_tgt.as<int>();
    a.coll_start = reinterpret_cast<const long long*>(cb + o_cs);
    a.coll_other = reinterpret_cast<const int*>(cb + o_co);
    a.coll_w = reinterpret_cast<const float*>(cb + o_cw);
    a.pot_trusted = c->opt.check_pot ? 0 : 1;      // (the products builder's own table; tests walk it all the same)
    a.coll_sym = 0;
    n_coll_out = n_coll;
    if ((int)P.view_hyp_begin.size() != V + 1) return fail(c, L3D_ERR_INVALID, "l3d_affinity_fill_resident: hypothesis ranges missing");
    return L3D_OK;
}

}  // namespace

namespace l3d {

// first-touch minima of all ranks (each a whole-scene array, "never touched" outside the rank's hypotheses) -> their minimum
__global__ __launch_bounds__(256) void k_aff_first_min(const unsigned long long* __restrict__ all, size_t stride_words, int world, int nh, unsigned long long* __restrict__ out)
{
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= nh) return;
    unsigned long long m = kFirstNone;
    for (int r = 0; r < world; ++r) { const unsigned long long v = all[(size_t)r * stride_words + h]; m = v < m ? v : m; }
    out[h] = m;
}

}  // namespace l3d

extern "C" {

// The same fill on the resident tables: hypotheses (l3d_products_hypotheses), potential correspondences and best matches
// (l3d_match_chain_resident) never left the device; the collinearity CSR is uploaded when it changed.
int l3d_affinity_fill_resident(l3d_ctx* c, const int64_t* coll_start, const int32_t* coll_other, const float* coll_w, int coll_changed, float sigma_a,
                               l3d_edge** edges_out, int* n_edges_out, int32_t** node_hyp_out, int* n_nodes_out, int* n_candidates_out)
{
    if (!c) return L3D_ERR_INVALID;
    if (!n_edges_out || !node_hyp_out || !n_nodes_out || !coll_start) return fail(c, L3D_ERR_INVALID, "bad argument");
    if (edges_out) *edges_out = nullptr;
    *n_edges_out = 0; *node_hyp_out = nullptr; *n_nodes_out = 0;
    c->resident_edges = 0; c->kept_edges = 0; c->resident_nodes = 0; c->resident_labels = 0;
    if (n_candidates_out) *n_candidates_out = 0;
    Products& P = c->products;
    if (!P.valid || !P.hyp_valid) return fail(c, L3D_ERR_INVALID, "l3d_affinity_fill_resident: no resident products / hypotheses");
    if (P.part.active && P.part.world > 1) return fail(c, L3D_ERR_INVALID, "l3d_affinity_fill_resident: the products are partitioned over the ranks (l3d_affinity_fill_sharded)");
    if (P.n_hyp == 0) return L3D_OK;
    HIPCHK(c, hipSetDevice(c->device));
    AffIn a;
    long long n_coll = 0;
    if (int rc = resident_tables(c, coll_start, coll_other, coll_w, coll_changed, a, n_coll)) return rc;
    return affinity_fill_core(c, a, P.seg_base.data(), P.view_hyp_begin.data(), P.n_pot, n_coll, sigma_a, edges_out, n_edges_out, node_hyp_out, n_nodes_out, n_candidates_out, nullptr);
}

// The fill SHARDED BY SOURCE KEY over the ranks of a partitioned job (l3d_match_chain_partition left every rank the rows, best matches
// and hypotheses its block's sources can reach).  Every rank enumerates the candidates of its own block's sources (in blocks of sources, as the
// one-GPU fill does); what has to be global is small and is all-gathered: how many hypotheses every view has (-> global hypothesis numbers), the
// first-touch minima (a hypothesis is first touched by a candidate of its own rank or of a neighbour's), the candidates that PASSED (12 bytes each,
// concatenated in rank order = source order: the reference's enumeration order) and the hypotheses of every block (for the line fit).  Node
// numbering and the edge list are then formed by every rank from the same data (replicas, like the clustering stages behind them).
int l3d_affinity_fill_sharded(l3d_ctx* c, const int64_t* coll_start, const int32_t* coll_other, const float* coll_w, int coll_changed, float sigma_a,
                              l3d_exchange_fn exchange, void* exchange_user, int* n_edges_out, int32_t** node_hyp_out, int* n_nodes_out,
                              int64_t* n_candidates_all, int32_t* view_hyp_begin_global, int32_t** hyp_dense_global, int* n_hyp_global)
{
    if (!c) return L3D_ERR_INVALID;
    if (!n_edges_out || !node_hyp_out || !n_nodes_out || !coll_start || !exchange || !view_hyp_begin_global || !hyp_dense_global || !n_hyp_global) return fail(c, L3D_ERR_INVALID, "bad argument");
    *n_edges_out = 0; *node_hyp_out = nullptr; *n_nodes_out = 0; *hyp_dense_global = nullptr; *n_hyp_global = 0;
    if (n_candidates_all) *n_candidates_all = 0;
    c->resident_edges = 0; c->kept_edges = 0; c->resident_nodes = 0; c->resident_labels = 0;
    Products& P = c->products;
    if (!P.valid || !P.hyp_valid || !P.part.active) return fail(c, L3D_ERR_INVALID, "l3d_affinity_fill_sharded: no partitioned products / hypotheses (l3d_match_chain_partition, l3d_products_hypotheses)");
    const ProductsPart part = P.part;
    const int world = part.world, rank = part.rank, V = P.n_views_all, nh = P.n_hyp;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    // A rank that fails on its own between two collectives still enters the next one, with a mark: nobody is left waiting (as in l3d_match_chain_blocks)
    int local_rc = L3D_OK;
    std::string local_err;
    auto note = [&](int rc) { if (rc && !local_rc) { local_rc = rc; std::lock_guard<std::mutex> lk(c->err_mu); local_err = c->err; } };
#define L3D_SOFT(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) note(fail(c, L3D_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_))); } while (0)
    if (c->ch_hdr.reserve(512 * (size_t)(world + 2) + 256) != hipSuccess) return fail(c, L3D_ERR_NOMEM, "l3d_affinity_fill_sharded: status words");     // (before the first collective)
    long long* st_own = c->ch_hdr.as<long long>();
    long long* st_all = reinterpret_cast<long long*>(c->ch_hdr.as<unsigned char>() + 256);
    std::vector<long long> words((size_t)world * 4, 0);
    // four words per rank; word 0 negative = the rank failed
    auto all_gather_words = [&](long long w0, long long w1, long long w2, long long w3, const char* what) -> int {
        long long mine[4] = { local_rc ? -(long long)local_rc : w0, w1, w2, w3 };
        hipError_t e = hipMemcpyAsync(st_own, mine, 32, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { note(fail(c, L3D_ERR_HIP, std::string("l3d_affinity_fill_sharded: status words: ") + hipGetErrorString(e))); (void)hipMemsetAsync(st_own, 0xff, 32, st); }
        if (exchange(exchange_user, -3, st_own, st_all, 256, world, (void*)st)) return fail(c, L3D_ERR_HIP, std::string("l3d_affinity_fill_sharded: the exchange of the status words failed (") + what + ")");
        e = hipSuccess;
        for (int r = 0; r < world && e == hipSuccess; ++r) e = hipMemcpyAsync(&words[(size_t)r * 4], reinterpret_cast<const unsigned char*>(st_all) + (size_t)r * 256, 32, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return fail(c, L3D_ERR_HIP, std::string("l3d_affinity_fill_sharded: status words: ") + hipGetErrorString(e));
        if (local_rc) return fail(c, local_rc, local_err);
        for (int r = 0; r < world; ++r)
            if (words[(size_t)r * 4] < 0) return fail(c, L3D_ERR_HIP, "l3d_affinity_fill_sharded: rank " + std::to_string(r) + " failed (code " + std::to_string(-words[(size_t)r * 4]) + ") while " + what);
        return L3D_OK;
    };
    auto gather = [&](int tag, size_t slot, const char* what) -> int {
        if (exchange(exchange_user, tag, c->ch_send.p, c->ch_gathered.p, slot, world, (void*)st)) return fail(c, L3D_ERR_HIP, std::string("l3d_affinity_fill_sharded: the exchange of ") + what + " failed");
        return L3D_OK;
    };
    auto reserve_slots = [&](size_t slot) { hipError_t e = c->ch_send.reserve(slot + 256); if (e == hipSuccess) e = c->ch_gathered.reserve(slot * (size_t)world + 256); if (e != hipSuccess) note(fail(c, L3D_ERR_NOMEM, "l3d_affinity_fill_sharded: exchange slots")); return e == hipSuccess; };

    // ---- 1. hypotheses per view: a view's count comes from the rank that owns it; every holder of the view must agree
    const size_t vslot = al((size_t)(V + 2) * 4);
    // (a view this rank has hypotheses of without holding it: an early-return view -- every rank finds its best matches from the all-gathered
    // records that point at it -- or a view an early return's local camera numbers name, whose best matches its owner sent: l3d_match_chain_partition)
    std::vector<int> cnt_loc((size_t)V + 2, -1);           // -1: nothing is known about the view here
    for (int v = 0; v < V; ++v) {
        const int n = P.view_hyp_begin[(size_t)v + 1] - P.view_hyp_begin[(size_t)v];
        if ((v >= part.held_dv0 && v < part.held_dv1) || n > 0) cnt_loc[(size_t)v] = n;
    }
    cnt_loc[(size_t)V] = part.own_dv0; cnt_loc[(size_t)V + 1] = part.own_dv1;
    if (reserve_slots(vslot)) { L3D_SOFT(hipMemcpyAsync(c->ch_send.p, cnt_loc.data(), (size_t)(V + 2) * 4, hipMemcpyHostToDevice, st)); L3D_SOFT(hipStreamSynchronize(st)); }
    if (int rc = all_gather_words(0, 0, 0, 0, "publishing its hypothesis counts")) return rc;
    if (int rc = gather(-7, vslot, "the hypothesis counts")) return rc;
    std::vector<int> cnt_all((size_t)world * (size_t)(V + 2), 0);
    for (int r = 0; r < world; ++r) L3D_SOFT(hipMemcpyAsync(&cnt_all[(size_t)r * (size_t)(V + 2)], c->ch_gathered.as<unsigned char>() + (size_t)r * vslot, (size_t)(V + 2) * 4, hipMemcpyDeviceToHost, st));
    L3D_SOFT(hipStreamSynchronize(st));
    std::vector<int> vhb((size_t)V + 1, 0), blk0((size_t)world), blk1((size_t)world);
    if (!local_rc) {
        for (int r = 0; r < world; ++r) { blk0[(size_t)r] = cnt_all[(size_t)r * (size_t)(V + 2) + (size_t)V]; blk1[(size_t)r] = cnt_all[(size_t)r * (size_t)(V + 2) + (size_t)V + 1]; }
        for (int v = 0; v < V && !local_rc; ++v) {
            int own = -1;
            for (int r = 0; r < world; ++r) if (v >= blk0[(size_t)r] && v < blk1[(size_t)r]) own = cnt_all[(size_t)r * (size_t)(V + 2) + (size_t)v];
            // (one rank's share of a larger job exercised alone -- options part_vrank / part_vworld at world 1: the other ranks' blocks are nobody's here; their
            // views count with what this rank happens to know of them, the share's hypothesis numbers are then its own)
            if (own < 0 && world == 1 && c->opt.part_vworld > 0) own = std::max(0, cnt_all[(size_t)v]);
            if (own < 0) { note(fail(c, L3D_ERR_INVALID, "l3d_affinity_fill_sharded: a view belongs to no rank's block")); break; }
            for (int r = 0; r < world; ++r) {
                const int x = cnt_all[(size_t)r * (size_t)(V + 2) + (size_t)v];
                if (x >= 0 && x != own) { note(fail(c, L3D_ERR_INVALID, "l3d_affinity_fill_sharded: rank " + std::to_string(r) + " holds " + std::to_string(x) + " hypotheses of dense view " + std::to_string(v) + ", its owner " + std::to_string(own) + " (the ranks' kept lists differ)")); break; }
            }
            vhb[(size_t)v + 1] = vhb[(size_t)v] + own;
        }
    }
    const int nh_all = vhb[(size_t)V];
    // local -> global hypothesis numbers (both ascend with the dense segment id: every comparison of two local numbers is the global one)
    if (!local_rc && nh > 0) {
        std::vector<int> l2g((size_t)nh);
        for (int v = 0; v < V; ++v)
            for (int h = P.view_hyp_begin[(size_t)v]; h < P.view_hyp_begin[(size_t)v + 1]; ++h) l2g[(size_t)h] = vhb[(size_t)v] + (h - P.view_hyp_begin[(size_t)v]);
        L3D_SOFT(c->aff_l2g.reserve((size_t)nh * 4 + 64));
        if (!local_rc) { L3D_SOFT(hipMemcpyAsync(c->aff_l2g.p, l2g.data(), (size_t)nh * 4, hipMemcpyHostToDevice, st)); L3D_SOFT(hipStreamSynchronize(st)); }
    }

    // ---- 2. this rank's candidates: the sources of its block
    long long n_coll = 0;
    if (!local_rc && nh > 0) {
        AffIn a;
        int rc = resident_tables(c, coll_start, coll_other, coll_w, coll_changed, a, n_coll);
        if (!rc) {
            FillPart fp;
            fp.h0 = P.view_hyp_begin[(size_t)part.own_dv0]; fp.h1 = P.view_hyp_begin[(size_t)part.own_dv1];
            fp.pos_base = (unsigned long long)rank << 44; fp.loc2glob = c->aff_l2g.as<int>(); fp.assume_symmetric = 1;
            int n_e = 0, n_n = 0, n_c = 0; int32_t* nhp = nullptr;
            rc = affinity_fill_core(c, a, P.seg_base.data(), P.view_hyp_begin.data(), P.n_pot, n_coll, sigma_a, nullptr, &n_e, &nhp, &n_n, &n_c, &fp);
        }
        note(rc);
    } else { c->fill_items = 0; c->fill_passed = 0; }
    const long long my_items = local_rc ? 0 : c->fill_items, my_passed = local_rc ? 0 : c->fill_passed;
    const int own_h0 = nh > 0 ? P.view_hyp_begin[(size_t)part.own_dv0] : 0, own_h1 = nh > 0 ? P.view_hyp_begin[(size_t)part.own_dv1] : 0;
    if (int rc = all_gather_words(0, my_items, my_passed, own_h1 - own_h0, "enumerating its candidates")) return rc;
    long long items_all = 0, passed_all = 0, max_passed = 0, max_blk = 0;
    std::vector<long long> passed_of((size_t)world), blk_of((size_t)world);
    for (int r = 0; r < world; ++r) {
        items_all += words[(size_t)r * 4 + 1]; passed_of[(size_t)r] = words[(size_t)r * 4 + 2]; blk_of[(size_t)r] = words[(size_t)r * 4 + 3];
        passed_all += passed_of[(size_t)r]; max_passed = std::max(max_passed, passed_of[(size_t)r]); max_blk = std::max(max_blk, blk_of[(size_t)r]);
    }
    if (n_candidates_all) *n_candidates_all = items_all;

    // ---- 3. first-touch minima: every rank's, over the global hypothesis numbers -> the minimum
    const size_t fslot = al((size_t)std::max(nh_all, 1) * 8);
    if (reserve_slots(fslot)) {
        hipLaunchKernelGGL(k_aff_fill64, dim3((std::max(nh_all, 1) + 255) / 256), dim3(256), 0, st, c->ch_send.as<unsigned long long>(), std::max(nh_all, 1), kFirstNone);
        if (nh > 0 && !local_rc) hipLaunchKernelGGL(k_aff_first_scatter, dim3((nh + 255) / 256), dim3(256), 0, st, c->aff_first.as<unsigned long long>(), c->aff_l2g.as<int>(), nh, c->ch_send.as<unsigned long long>());
    }
    if (int rc = all_gather_words(0, 0, 0, 0, "staging its first-touch positions")) return rc;
    if (int rc = gather(-8, fslot, "the first-touch positions")) return rc;
    L3D_SOFT(c->aff_first.reserve((size_t)std::max(nh_all, 1) * 8 + 64));
    if (!local_rc && nh_all > 0) hipLaunchKernelGGL(k_aff_first_min, dim3((nh_all + 255) / 256), dim3(256), 0, st, c->ch_gathered.as<unsigned long long>(), fslot / 8, world, nh_all, c->aff_first.as<unsigned long long>());
    L3D_SOFT(hipStreamSynchronize(st));            // (the gathered buffer is reused below)

    // ---- 4. the candidates that passed, concatenated in rank order
    const size_t o_w = al((size_t)max_passed * 8), pslot = o_w + al((size_t)max_passed * 4 + 4);
    if (reserve_slots(pslot) && my_passed > 0) {
        L3D_SOFT(hipMemcpyAsync(c->ch_send.p, c->aff_pass_pairs.p, (size_t)my_passed * 8, hipMemcpyDeviceToDevice, st));
        L3D_SOFT(hipMemcpyAsync(c->ch_send.as<unsigned char>() + o_w, c->aff_pass_w.p, (size_t)my_passed * 4, hipMemcpyDeviceToDevice, st));
    }
    if (int rc = all_gather_words(0, 0, 0, 0, "staging its passed candidates")) return rc;
    if (int rc = gather(-9, pslot, "the passed candidates")) return rc;
    L3D_SOFT(c->aff_pass_pairs.reserve((size_t)passed_all * 8 + 256));
    L3D_SOFT(c->aff_pass_w.reserve((size_t)passed_all * 4 + 256));
    if (!local_rc) {
        long long at = 0;
        for (int r = 0; r < world; ++r) {
            if (passed_of[(size_t)r] > 0) {
                L3D_SOFT(hipMemcpyAsync(c->aff_pass_pairs.as<int2>() + at, c->ch_gathered.as<unsigned char>() + (size_t)r * pslot, (size_t)passed_of[(size_t)r] * 8, hipMemcpyDeviceToDevice, st));
                L3D_SOFT(hipMemcpyAsync(c->aff_pass_w.as<float>() + at, c->ch_gathered.as<unsigned char>() + (size_t)r * pslot + o_w, (size_t)passed_of[(size_t)r] * 4, hipMemcpyDeviceToDevice, st));
            }
            at += passed_of[(size_t)r];
        }
        L3D_SOFT(hipStreamSynchronize(st));
    }

    // (one rank's share exercised alone -- part_vrank / part_vworld at world 1: the share of the fill ends here; numbering, clustering and the fits need the
    // other ranks' blocks.  l3d_last_fill_counts has the share's candidates and passed pairs; an empty edge list goes back)
    if (world == 1 && c->opt.part_vworld > 0) {
        HIPCHK(c, hipStreamSynchronize(st));
        if (local_rc) return local_rc;
        int32_t* hd0 = static_cast<int32_t*>(malloc(8));
        int32_t* nh0 = static_cast<int32_t*>(malloc(8));
        if (!hd0 || !nh0) { free(hd0); free(nh0); return fail(c, L3D_ERR_NOMEM, "malloc"); }
        for (int v = 0; v <= V; ++v) view_hyp_begin_global[v] = 0;
        *hyp_dense_global = hd0; *node_hyp_out = nh0;
        return L3D_OK;
    }
    // ---- 5. the hypotheses of every block (the line fit reads them by global number) and the segments they belong to
    const size_t o_hd = al((size_t)max_blk * sizeof(Hypothesis)), hslot = o_hd + al((size_t)max_blk * 4 + 4);
    if (reserve_slots(hslot) && own_h1 > own_h0 && !local_rc) {
        L3D_SOFT(hipMemcpyAsync(c->ch_send.p, c->aff_hyp.as<Hypothesis>() + own_h0, (size_t)(own_h1 - own_h0) * sizeof(Hypothesis), hipMemcpyDeviceToDevice, st));
        L3D_SOFT(hipMemcpyAsync(c->ch_send.as<unsigned char>() + o_hd, P.hyp_dense.as<int>() + own_h0, (size_t)(own_h1 - own_h0) * 4, hipMemcpyDeviceToDevice, st));
    }
    if (int rc = all_gather_words(0, 0, 0, 0, "staging its hypotheses")) return rc;
    if (int rc = gather(-10, hslot, "the hypotheses")) return rc;
    // (past the last collective: a failure from here on is this rank's alone)
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, c->aff_hyp.reserve((size_t)std::max(nh_all, 1) * sizeof(Hypothesis) + 64));
    int32_t* hd = static_cast<int32_t*>(malloc((size_t)nh_all * 4 + 4));
    if (!hd) return fail(c, L3D_ERR_NOMEM, "malloc");
    {
        long long at = 0;
        for (int r = 0; r < world; ++r) {
            const long long n = blk_of[(size_t)r];
            if (at != vhb[(size_t)blk0[(size_t)r]]) { free(hd); return fail(c, L3D_ERR_INVALID, "l3d_affinity_fill_sharded: the blocks' hypotheses do not line up with the global numbering"); }
            if (n > 0) {
                hipError_t e = hipMemcpyAsync(c->aff_hyp.as<Hypothesis>() + at, c->ch_gathered.as<unsigned char>() + (size_t)r * hslot, (size_t)n * sizeof(Hypothesis), hipMemcpyDeviceToDevice, st);
                if (e == hipSuccess) e = hipMemcpyAsync(hd + at, c->ch_gathered.as<unsigned char>() + (size_t)r * hslot + o_hd, (size_t)n * 4, hipMemcpyDeviceToHost, st);
                if (e != hipSuccess) { free(hd); return fail(c, L3D_ERR_HIP, std::string("l3d_affinity_fill_sharded: ") + hipGetErrorString(e)); }
            }
            at += n;
        }
        if (at != nh_all) { free(hd); return fail(c, L3D_ERR_INVALID, "l3d_affinity_fill_sharded: the blocks' hypotheses do not add up"); }
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) { free(hd); return fail(c, L3D_ERR_HIP, std::string("l3d_affinity_fill_sharded: ") + hipGetErrorString(e)); }
    }
    memcpy(view_hyp_begin_global, vhb.data(), (size_t)(V + 1) * 4);
    *hyp_dense_global = hd; *n_hyp_global = nh_all;
    c->fill_items = items_all; c->fill_passed = passed_all;
    P.n_hyp = nh_all;                                 // (l3d_products_hypotheses_get now returns the whole table; the local index tables of the fill are spent)
    // ---- 6. numbering and edges: every rank from the same data
    int rc = affinity_number_edges(c, c->aff_first.as<unsigned long long>(), nh_all, c->aff_pass_pairs.as<int2>(), c->aff_pass_w.as<float>(), passed_all, nullptr, n_edges_out, node_hyp_out, n_nodes_out);
    if (rc) { free(hd); *hyp_dense_global = nullptr; *n_hyp_global = 0; }
    c->resident_hyp = nh_all;
    return rc;
#undef L3D_SOFT
}

}  // extern "C"

void l3d::warm_affinity() { touch_kernel(reinterpret_cast<const void*>(&k_aff_fill64)); }
