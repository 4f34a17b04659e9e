// l3d_products.hip -- what Line3D::performMatching leaves behind for the rest of compute3Dmodel, built ON THE DEVICE from the kept
// arena of the resident chain (which never leaves HBM):
//
//   potential_correspondences_   line3D.cc:861-865: for every kept match (view, segment) <-> (camera, target) both directions, in a
//                                map of maps: set semantics, ascending iteration.  Here: one 64-bit key per direction over DENSE
//                                segment ids, ONE stable radix sort (hipCUB), duplicates dropped, CSR -- the reference's host loop
//                                over std::map inserts (and rounds 1-2's finaliser threads on the host) disappear.
//   the match files              line3D.cc:884, view.cc:165-183: a view's own list overwrites whatever earlier views pushed, reduced
//                                to the best match per segment (first of the highest confidence): found by the kept writer while
//                                the confidences are in registers (l3d_kept.hpp); here only turned into arena references.
//   median depths                cudawrapper.cu:1058-1076: std::sort + middle element on the host there; a radix select per view here.
//   greedy selection             line3D.cc:899-965 + L3DView::unprojectSegment view.cc:302-342: one 3-D hypothesis per segment with a
//                                best match, numbered in dense order (prefix sum), unprojected in double with the host's operations.
//
// Views with nothing left to match return early (cudawrapper.cu:877-878): their list is the localized existing list -- what the
// earlier views pushed, in push order, LOCAL camera ids, confidence 0 -- and the reference then files its entries under those
// local numbers read as view ids (line3D.cc:838-866).  Reproduced: such a view's entries are formed from its sources' records.
#include "l3d_sort.hpp"

#include <algorithm>
#include <vector>

#include "l3d_products.hpp"
#include "l3d_linalg.hpp"
#include "l3d_unproject.hpp"
#include "l3d_runtable.hpp"

using namespace l3d;

namespace l3d {

struct ProdView {                   // per chain index
    long long kept_base;
    int n_kept, R;
    int dense_base, S;
    int early, pad;
    const int* bestpos;
    const float2* best;
};
// The keys are formed, sorted and reduced to CSR rows in BLOCKS of consecutive dense views (ascending: the blocks' row ranges
// concatenate to the table), so that the transient memory -- two 64-bit key arrays, flags, positions: 24 bytes per key slot -- is
// bounded by the block size and not by the scene (the reference spills every view's matches to disk, view.cc:150-224).
// A block holds the keys whose SOURCE segment lies in its dense range [d0, d1): the forward key of a record of one of its views, the
// backward key of a record that points into it.  Every chain view / early-return pair that can contribute gets a region of
// 2 x (its records) slots in the block's key array (out_off; -1: it cannot contribute); a key outside the range is written invalid.
struct ProdBlock {
    int d0, d1;                     // dense range
    long long slots;                // key slots of the block (without the sentinel)
    size_t off_view, off_src;       // this block's out_off tables inside the uploaded table buffer (long long per chain view / per ProdSrc)
};
struct ProdSrc {                    // one (early-return view, source) pair
    int view, src;                  // chain indices
    int alias_base, alias_S;        // dense range of the view the source's LOCAL camera number names (-1: no such view)
    long long out_off;              // first key slot of this pair
    int rank, pad;                  // position of the source in the view's list (push order)
};

constexpr unsigned long long kInvalidKey = ~0ull;

__device__ __forceinline__ int find_view(const unsigned* __restrict__ ids, int n, unsigned id)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (ids[mid] < id) lo = mid + 1; else hi = mid; }
    return lo < n && ids[lo] == id ? lo : -1;
}

// two keys per kept record of the verified views: (dense source << nb | dense target) and the reverse (line3D.cc:864-865)
__global__ __launch_bounds__(256) void k_prod_keys(const Match* __restrict__ arena, const ProdView* __restrict__ pv, const long long* __restrict__ out_off,
                                                   const unsigned* __restrict__ ids, const int* __restrict__ seg_base, int n_all, int nb, int d0, int d1,
                                                   unsigned long long* __restrict__ keys)
{
    const ProdView v = pv[blockIdx.y];
    const long long off = out_off[blockIdx.y];
    if (v.early || off < 0) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < v.n_kept; i += gridDim.x * 256) {
        const Match r = arena[v.kept_base + i];
        const int t = find_view(ids, n_all, r.camID2);
        unsigned long long f = kInvalidKey, b = kInvalidKey;
        if (t >= 0 && (int)r.segID1 < v.S) {
            const int tb = seg_base[t], tS = seg_base[t + 1] - tb;
            if ((int)r.segID2 < tS) {
                const int ai = v.dense_base + (int)r.segID1, di = tb + (int)r.segID2;
                const unsigned long long a = (unsigned long long)ai, d = (unsigned long long)di;
                if (ai >= d0 && ai < d1) f = (a << nb) | d;
                if (di >= d0 && di < d1) b = (d << nb) | a;
            }
        }
        keys[off + 2 * (long long)i] = f;
        keys[off + 2 * (long long)i + 1] = b;
    }
}

// early-return views: the records of source `src` that point at the view, read reversed; the camera of such an entry is the
// source's LOCAL number in the view's neighbour list (alias = the view that number happens to name).  Also the view's best
// match per segment: the first entry of the list in push order (all confidences are 0): atomicMin of (source rank, record).
// out_off == nullptr: the pass that only finds the best matches and the list lengths (once per build); otherwise the keys of one block.
__global__ __launch_bounds__(256) void k_prod_keys_early(const Match* __restrict__ arena, const ProdView* __restrict__ pv, const ProdSrc* __restrict__ ps,
                                                         const long long* __restrict__ out_off, const unsigned* __restrict__ chain_view_id, int nb, int d0, int d1,
                                                         unsigned long long* __restrict__ keys, unsigned long long* __restrict__ best_ref, int* __restrict__ list_len)
{
    const ProdSrc e = ps[blockIdx.y];
    const long long off = out_off ? out_off[blockIdx.y] : 0;
    if (off < 0) return;
    const ProdView v = pv[e.view], s = pv[e.src];
    const unsigned vid = chain_view_id[e.view];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < s.n_kept; i += gridDim.x * 256) {
        const Match r = arena[s.kept_base + i];
        unsigned long long f = kInvalidKey, b = kInvalidKey;
        if (r.camID2 == vid) {
            if (!out_off) atomicAdd(&list_len[e.view], 1);                      // (length of the view's list: statistics)
            // the list entry: segID1 = r.segID2 (the view's segment), camID2 = local number, segID2 = r.segID1
            if ((int)r.segID2 < v.S) {
                if (!out_off) { if (!e.pad) atomicMin(&best_ref[v.dense_base + (int)r.segID2], ((unsigned long long)e.rank << 40) | (unsigned long long)(s.kept_base + i)); }
                else if (e.alias_base >= 0 && (int)r.segID1 < e.alias_S) {
                    const int ai = v.dense_base + (int)r.segID2, di = e.alias_base + (int)r.segID1;
                    const unsigned long long a = (unsigned long long)ai, d = (unsigned long long)di;
                    if (ai >= d0 && ai < d1) f = (a << nb) | d;
                    if (di >= d0 && di < d1) b = (d << nb) | a;
                }
            }
        }
        if (out_off) { keys[off + 2 * (long long)i] = f; keys[off + 2 * (long long)i + 1] = b; }
    }
}

// best match per segment as a reference into the arena: verified views from the kept writer's positions, early-return views
// from the atomicMin keys (-> record index, "read reversed")
__global__ __launch_bounds__(256) void k_prod_best(const ProdView* __restrict__ pv, long long* __restrict__ best_ref)
{
    const ProdView v = pv[blockIdx.y];
    for (int s = blockIdx.x * 256 + threadIdx.x; s < v.S; s += gridDim.x * 256) {
        long long* o = best_ref + v.dense_base + s;
        if (v.early) { const long long k = *o; if (k != -1) *o = (k & kBestIndexMask) | kBestReversed; }
        else if (v.bestpos) { const int p = v.bestpos[s]; *o = p < 0 || v.n_kept == 0 ? -1 : v.kept_base + p; }
    }
}

// the median of the depths of every segment's best hypothesis (cudawrapper.cu:1058-1076: both depths of every segment that has
// one, sorted, element size/2): radix select over order-preserving keys, one workgroup per view
__global__ __launch_bounds__(256) void k_prod_median(const ProdView* __restrict__ pv, float* __restrict__ median)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned s_sel[3];       // prefix, rank, count
    const ProdView v = pv[blockIdx.x];
    const int tid = threadIdx.x;
    if (v.early || !v.best || v.R == 0) { if (tid == 0) median[blockIdx.x] = 1.0f; return; }     // untouched (line3D.cc:811, cudawrapper.cu:955-956)
    auto key_of = [](float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    if (tid == 0) s_sel[2] = 0;
    __syncthreads();
    unsigned cnt = 0;
    for (int s = tid; s < v.S; s += 256) cnt += v.best[s].x != -1.0f ? 2u : 0u;
    atomicAdd(&s_sel[2], cnt);
    __syncthreads();
    const unsigned n2 = s_sel[2];
    if (n2 == 0) { if (tid == 0) median[blockIdx.x] = -1.0f; return; }                           // cudawrapper.cu:1066
    if (tid == 0) { s_sel[0] = 0; s_sel[1] = n2 / 2; }
    for (int pass = 3; pass >= 0; --pass) {
        hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = s_sel[0];
        const int shift = pass * 8;
        for (int s = tid; s < v.S; s += 256) {
            const float2 b = v.best[s];
            if (b.x == -1.0f) continue;
            const unsigned k0 = key_of(b.x), k1 = key_of(b.y);
            if (pass == 3 || (k0 >> (shift + 8)) == prefix) atomicAdd(&hist[(k0 >> shift) & 255u], 1u);
            if (pass == 3 || (k1 >> (shift + 8)) == prefix) atomicAdd(&hist[(k1 >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned rank = s_sel[1], cum = 0;
            int bkt = 0;
            for (; bkt < 256; ++bkt) { if (cum + hist[bkt] > rank) break; cum += hist[bkt]; }
            s_sel[0] = (prefix << 8) | (unsigned)bkt;
            s_sel[1] = rank - cum;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const unsigned k = s_sel[0];
        median[blockIdx.x] = __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
    }
}

// sorted keys -> first-occurrence flags (the sentinel behind the last key is invalid by construction)
__global__ __launch_bounds__(256) void k_prod_flags(const unsigned long long* __restrict__ keys, long long n, int* __restrict__ flag)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    flag[i] = (k != kInvalidKey && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
}

// unique keys -> CSR.  The thread of a first occurrence writes its target; where the source changes it also writes the row starts
// of every source in between (segments without entries), and the first invalid key closes the table.
// One block of sources [d0, d1): its rows start at `base` (the entries of the blocks in front of it); the closing (first invalid) key
// writes the start of row d1 = the end of this block's rows = the base of the next block (or of the table when d1 = n_dense).
__global__ __launch_bounds__(256) void k_prod_csr(const unsigned long long* __restrict__ keys, const int* __restrict__ flag, const int* __restrict__ pos,
                                                  long long n, int nb, int d0, int d1, long long base, long long* __restrict__ pot_start, int* __restrict__ pot_tgt)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    const unsigned long long mask = (1ull << nb) - 1ull;
    const bool invalid = k == kInvalidKey;
    if (invalid && i > 0 && keys[i - 1] == kInvalidKey) return;
    const long long p = base + pos[i];
    const long long src = invalid ? (long long)d1 : (long long)(k >> nb);
    const long long prev = i == 0 ? (long long)d0 - 1 : (long long)(keys[i - 1] >> nb);
    if (!invalid && flag[i]) pot_tgt[p] = (int)(k & mask);
    if (i == 0 || prev != src) for (long long d = prev + 1; d <= src; ++d) pot_start[d] = p;
}

// partitioned products: the rows this rank does not hold are empty -- in front of its rows they start at 0, behind them at its last entry
__global__ __launch_bounds__(256) void k_prod_fill_rows(long long* __restrict__ pot_start, long long lo, long long hi, long long value)
{
    const long long i = lo + (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < hi) pot_start[i] = value;
}

// a rank's piece of the table (l3d_match_chain_blocks): its row starts, numbered from 0, shifted to where its entries sit in the whole table
__global__ __launch_bounds__(256) void k_prod_shift_rows(const long long* __restrict__ piece, long long n_rows, long long base, long long* __restrict__ pot_start)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n_rows) pot_start[i] = piece[i] + base;
}


// =================================================================================================================================
// Round 6: the table as a TRANSPOSE of run tables instead of a sort of keys.  A row of the table -- the potential correspondences of one dense
// segment (t, u) -- is the union of
//   (F) the segment's own kept records, read forward: one contiguous run of its view's list (run table: rt[0][u] .. rt[N][u]), and
//   (B) the records of other views that point at it, read backward.  The records of view v towards its local camera q (= view t) are the runs
//       rt_v[q][s] .. rt_v[q + 1][s] over all s: a sparse S_v x S_t matrix in row order; B needs it in column order.  ONE workgroup per (v, q) pair
//       transposes it in LDS (histogram over u, scan, scatter): no global atomics, nothing scanned that is not used.
// Every list is read through its 4-byte side array qt = (local camera << 16 | target) written by the kept writer beside the records (l3d_kept.hpp;
// rebuilt here, block by block, for lists that did not come out of the single-GPU chain): the 32-byte records are not read at all.
//   k_prodv_pair_counts     records of every pair that targets the block (sums of run lengths)        -> where its transposed entries go
//   k_prodv_pair_transpose  per pair: column starts boff[u] and the source segments grouped by u (E)
//   k_prodv_rows<false>     one wave per row, view by view of the views the row's view touches (ascending: dense order): the F run of that camera and the B column
//   exclusive_sum_int       of that pair set bits in a bitmap of the view's segments -- sorted and unique by construction; popcount = the row's length; the scan
//   k_prodv_rows<true>      gives the CSR's row starts; the bitmaps again, expanded into the row's place.  Rows of at most 64 entries: ranked by shuffles instead.
// Transients: 4 bytes per backward entry (+ 4 for the staging of the two-level scatter; rounds 3-5: 48 bytes per record in key arrays); the table's entries are
// counted before they are written, so the table is reserved at its size.  The hipCUB radix sort is off matchViews' path (kept as A/B and as the way out for lists
// that are not ordered).
struct ProdNbQ { int t_base, t_S; };                                  // chain view k, LOCAL camera q: the target view's dense range (-1: not in the map)
struct ProdViewQ { const unsigned* qt; const int* rt; int nb_off, N; }; // a chain view's list through its side array and run table (null: no records here)
struct ProdPair { int k, q, off_off, pad; };                            // a (chain view, local camera) pair that targets a view of the block; where its column starts live in boff
struct ProdTouch { int y_base, y_S, fq, pair; };                        // a view y the rows of view x can name: its dense range; the local camera number y has in x's own chain view
                                                                       // (F runs; -1: none); the pair (y's chain view, its camera number of x) whose columns are x's B (-1: none)
struct ProdRowView { int dense_base, S, own_k, t0, t1, pad; };          // own_k: the chain view whose list holds the rows' F runs (-1: none); [t0, t1): its ProdTouch entries, ascending y

__global__ __launch_bounds__(256) void k_prodv_pair_counts(const ProdPair* __restrict__ pairs, const ProdViewQ* __restrict__ vq, const ProdView* __restrict__ pv, int* __restrict__ pcnt)
{
    __shared__ int s_w[4];
    const ProdPair pr = pairs[blockIdx.x];
    const ProdViewQ v = vq[pr.k];
    const int S = pv[pr.k].S;
    int t = 0;
    if (v.rt) for (int s = threadIdx.x; s < S; s += 256) t += v.rt[(size_t)(pr.q + 1) * S + s] - v.rt[(size_t)pr.q * S + s];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) pcnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// One workgroup of 512 threads per pair, ONE RUN PER THREAD with four of its words in flight: g lanes sharing a run left every wave with one dependent
// load outstanding at a time (two levels, ~2 us each under load: 5 ms per build at 40 x 4000 x 24); a thread per run keeps 4 x 512 independent loads
// in flight per workgroup, and a run's words share cache lines across the thread's iterations.  LDS: S_t + 1 counters.
constexpr int kPairThreads = 512;
template <class F> __device__ __forceinline__ void for_run_words(const unsigned* __restrict__ qt, int a, int b, F&& f)
{
    for (int i = a; i < b; i += 4) {
        const unsigned w0 = qt[i], w1 = i + 1 < b ? qt[i + 1] : 0u, w2 = i + 2 < b ? qt[i + 2] : 0u, w3 = i + 3 < b ? qt[i + 3] : 0u;
        f(i, w0);
        if (i + 1 < b) f(i + 1, w1);
        if (i + 2 < b) f(i + 2, w2);
        if (i + 3 < b) f(i + 3, w3);
    }
}
// T (optional; round 6, late): the scatter in two levels.  The direct scatter below sends every 4-byte store of a pair's 650 KB region of E its own way:
// 792 pairs' regions are sixteen times the L2s, every store ends as a partial-line write-back -- WRITE_SIZE 4.03 GB per launch for 0.62 GB of entries
// (profiles/r6_cfg5_traffic.json).  With T the records first go, packed (u within its bucket << 14 | s), to one of NB buckets of consecutive u in a staging
// region of the pair's size (NB sequential write streams per workgroup: their open lines fit the L2 and leave it full), then bucket by bucket into an LDS image of
// the bucket's piece of E at their exact places (LDS cursors), and the image goes out in whole lines.  NB = the smallest power of two for which the largest bucket
// fits `cap` entries of LDS (a pair whose single busiest u does not fit -- more than `cap` sources for one target segment -- scatters directly).
// one pair's transpose by one workgroup of kPairThreads: the runs r0[s] .. r1[s] (s < S) of the side array qt, targets below St; column starts to bo[0 .. St], the
// source segments grouped by target to e[0 ..); t: the staging region of the two-level scatter (null: direct).  s_h: St + 2 + 2 * cap ints of LDS.
__device__ __forceinline__ void pair_transpose_wg(const unsigned* __restrict__ qt, const int* __restrict__ r0, const int* __restrict__ r1, int S, int St, int g,
                                                  int* __restrict__ bo, unsigned* __restrict__ e, unsigned* __restrict__ t, int cap, int* s_h, int* s_w, int* s_kbp)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int& s_kb = *s_kbp;
    for (int u = tid; u <= St; u += kPairThreads) s_h[u] = 0;
    __syncthreads();
    const int grp = g ? tid / g : 0, gl = g ? tid - grp * g : 0, ngrp = g ? kPairThreads / g : 1;
    if (g == 0) {
        for (int s = tid; s < S; s += kPairThreads)
            for_run_words(qt, r0[s], r1[s], [&](int, unsigned w) { const int u = (int)(w & 0xffffu); if (u < St) atomicAdd(&s_h[u], 1); });
    } else {
        for (int s = grp; s < S; s += ngrp) {
            const int a = r0[s], b = r1[s];
            for (int i = a + gl; i < b; i += g) { const int u = (int)(qt[i] & 0xffffu); if (u < St) atomicAdd(&s_h[u], 1); }
        }
    }
    __syncthreads();
    // exclusive scan of the St counters: a contiguous piece per thread
    const int per = (St + kPairThreads - 1) / kPairThreads, u0 = tid * per, u1 = min(St, u0 + per);
    int sum = 0;
    for (int u = u0; u < u1; ++u) sum += s_h[u];
    int incl = sum;
    for (int d = 1; d < 64; d <<= 1) { const int x = __shfl_up(incl, d); if (lane >= d) incl += x; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int run = incl - sum, all = 0;
    for (int w = 0; w < kPairThreads / 64; ++w) { if (w < wave) run += s_w[w]; all += s_w[w]; }
    for (int u = u0; u < u1; ++u) { const int c = s_h[u]; s_h[u] = run; bo[u] = run; run += c; }
    if (tid == 0) { bo[St] = all; s_h[St] = all; }
    __syncthreads();
    if (t && cap > 0 && all > 32768) {                          // (a small region -- config 2: 3 k entries per pair -- is written in place: it stays in L2)
        // bucket width: the smallest shift kb (from "32 buckets" down to one u per bucket) for which every bucket's piece of E fits the LDS image
        if (tid == 0) s_kb = -1;
        __syncthreads();
        int kb0 = 0;
        while ((St >> kb0) > 32) ++kb0;
        for (int kb = kb0; kb >= 0; --kb) {
            const int nbk = (St + (1 << kb) - 1) >> kb;
            int bad = nbk > cap ? 1 : 0;                              // (level 1 keeps a cursor per bucket in the image)
            for (int b = tid; b < nbk; b += kPairThreads) bad |= (s_h[min(St, (b + 1) << kb)] - s_h[b << kb]) > cap ? 1 : 0;
            if (__syncthreads_or(bad) == 0) { if (tid == 0) s_kb = kb; break; }
        }
        __syncthreads();
        const int kb = s_kb;
        if (kb >= 0) {
            int* img = s_h + St + 2;                                   // the bucket's piece of E (cap ints); in front of it, level 1's write cursors of the buckets (reused)
            const int nbk = (St + (1 << kb) - 1) >> kb;
            int* bcur = img;                                          // (nbk <= cap: asserted by the host's choice of cap >= 512)
            for (int b = tid; b < nbk; b += kPairThreads) bcur[b] = s_h[b << kb];
            __syncthreads();
            const unsigned umask = (1u << kb) - 1u;
            auto lvl1 = [&](int s, unsigned w) { const unsigned u = w & 0xffffu; if ((int)u < St) t[atomicAdd(&bcur[u >> kb], 1)] = ((u & umask) << 14) | (unsigned)s; };
            if (g == 0) { for (int s = tid; s < S; s += kPairThreads) for_run_words(qt, r0[s], r1[s], [&](int, unsigned w) { lvl1(s, w); }); }
            else for (int s = grp; s < S; s += ngrp) { const int a = r0[s], b = r1[s]; for (int i = a + gl; i < b; i += g) lvl1(s, qt[i]); }
            __threadfence_block();
            __syncthreads();
            for (int b = 0; b < nbk; ++b) {
                const int ub = b << kb, p0 = s_h[ub], p1 = s_h[min(St, (b + 1) << kb)];     // (s_h[u] is still the START of column u: level 1 used its own cursors)
                __syncthreads();                                       // (the image of the bucket before is out)
                // the bucket's records, in any order, to their places: a cursor per u of the bucket (the counts are gone: cursors run from the column starts,
                // kept apart in the image's tail so that s_h stays the column starts)
                int* cur = img + cap;                                  // 1 << kb ints
                for (int j = tid; j < (1 << kb) && ub + j <= St; j += kPairThreads) cur[j] = s_h[min(St, ub + j)] - p0;
                __syncthreads();
                for (int i = p0 + tid; i < p1; i += kPairThreads) { const unsigned x = t[i]; img[atomicAdd(&cur[x >> 14], 1)] = (int)(x & 0x3fffu); }
                __syncthreads();
                for (int i = tid; i < p1 - p0; i += kPairThreads) e[p0 + i] = (unsigned)img[i];
            }
            return;
        }
    }
    if (g == 0) {
        for (int s = tid; s < S; s += kPairThreads)
            for_run_words(qt, r0[s], r1[s], [&](int, unsigned w) { const int u = (int)(w & 0xffffu); if (u < St) e[atomicAdd(&s_h[u], 1)] = (unsigned)s; });
    } else {
        for (int s = grp; s < S; s += ngrp) {
            const int a = r0[s], b = r1[s];
            for (int i = a + gl; i < b; i += g) { const int u = (int)(qt[i] & 0xffffu); if (u < St) e[atomicAdd(&s_h[u], 1)] = (unsigned)s; }
        }
    }
}

__global__ __launch_bounds__(kPairThreads) void k_prodv_pair_transpose(const ProdPair* __restrict__ pairs, const ProdViewQ* __restrict__ vq, const ProdView* __restrict__ pv,
                                                                       const ProdNbQ* __restrict__ nbq, const int* __restrict__ poff, int g, int* __restrict__ boff, unsigned* __restrict__ E,
                                                                       unsigned* __restrict__ T, int cap)
{
    extern __shared__ int s_h[];
    __shared__ int s_w[kPairThreads / 64];
    __shared__ int s_kb;
    const ProdPair pr = pairs[blockIdx.x];
    const ProdViewQ v = vq[pr.k];
    const int S = pv[pr.k].S, St = nbq[v.nb_off + pr.q].t_S;
    pair_transpose_wg(v.qt, v.rt + (size_t)pr.q * S, v.rt + (size_t)(pr.q + 1) * S, S, St, g, boff + pr.off_off, E + poff[blockIdx.x], T ? T + poff[blockIdx.x] : nullptr, cap, s_h, s_w, &s_kb);
}

// ---- the transposes of ONE view's pairs, launched by the chain on a side stream right behind the view's kept writer (round 6, late): a pair depends on its view's
// list alone, so only the rows are left for the end of matchViews.  Canonical places, independent of the tables the end builds: counts and E offsets per (chain view,
// camera) at [k * maxN + q]; the pair's entries inside its view's piece of an arena-aligned E (the pairs of a view hold exactly its records), camera by camera.
__global__ __launch_bounds__(256) void k_prode_counts(const EarlyView* __restrict__ views, int k0, int maxN, const ChainResult* __restrict__ res, int* __restrict__ pcnt_kq)
{
    __shared__ int s_w[4];
    const EarlyView v = views[k0 + blockIdx.y];
    const int q = blockIdx.x;
    if (!v.rt || q >= v.N) return;
    const ChainResult r = res[v.k];
    int t = 0;
    if (!r.overflow && r.n_kept > 0 && v.St[q] > 0)
        for (int s = threadIdx.x; s < v.S; s += 256) t += v.rt[(size_t)(q + 1) * v.S + s] - v.rt[(size_t)q * v.S + s];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) pcnt_kq[(size_t)v.k * maxN + q] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(kPairThreads) void k_prode_transpose(const EarlyView* __restrict__ views, int k0, int maxN, const ChainResult* __restrict__ res, const unsigned* __restrict__ qt_arena,
                                                                  const int* __restrict__ pcnt_kq, unsigned* __restrict__ poff_kq, int g, int* __restrict__ boff, unsigned* __restrict__ E,
                                                                  unsigned* __restrict__ T, int cap)
{
    extern __shared__ int s_h[];
    __shared__ int s_w[kPairThreads / 64];
    __shared__ int s_kb;
    const EarlyView v = views[k0 + blockIdx.y];
    const int q = blockIdx.x;
    if (!v.rt || q >= v.N) return;
    const ChainResult r = res[v.k];
    const int St = v.St[q];
    unsigned off = r.kept_base;
    for (int j = 0; j < q; ++j) off += (unsigned)pcnt_kq[(size_t)v.k * maxN + j];
    if (threadIdx.x == 0) poff_kq[(size_t)v.k * maxN + q] = off;
    if (r.overflow || r.n_kept == 0 || St <= 0) return;                 // (an overflowed view is run again, and transposed again behind that run)
    pair_transpose_wg(qt_arena + r.kept_base, v.rt + (size_t)q * v.S, v.rt + (size_t)(q + 1) * v.S, v.S, St, g, boff + v.boff_off[q], E + off, T ? T + off : nullptr, cap, s_h, s_w, &s_kb);
}

// The early-return quirk through run tables (the chain's side arrays; with rebuilt ones the scans of k_prod_keys_early / k_prodt_early below stay):
// the records of source `src` that point at the early-return view are the runs (s, slot) of its list.  MODE 0: the view's best match per segment
// and its list length (k_prod_keys_early with out_off == nullptr); 1 / 2: count / scatter of the extra entries of both rows (k_prodt_early).
template <int MODE>
__global__ __launch_bounds__(256) void k_prod_early_rt(const ProdView* __restrict__ pv, const ProdViewQ* __restrict__ vq, const ProdSrc* __restrict__ ps, const int* __restrict__ ps_slot,
                                                       const int* __restrict__ src_list, int d0, int d1, int* __restrict__ cnt, const int* __restrict__ bstart,
                                                       unsigned* __restrict__ ent, unsigned long long* __restrict__ best_ref, int* __restrict__ list_len)
{
    const int q = src_list ? src_list[blockIdx.y] : (int)blockIdx.y;
    const ProdSrc e = ps[q];
    const int slot = ps_slot[q];
    if (slot < 0 || (MODE != 0 && e.alias_base < 0)) return;
    const ProdView v = pv[e.view], s = pv[e.src];
    const ProdViewQ sq = vq[e.src];
    if (!sq.rt || s.n_kept == 0) return;
    const int* r0 = sq.rt + (size_t)slot * s.S;
    const int* r1 = r0 + s.S;
    // sixteen lanes per run (coalesced words; a run per thread touched 64 lines per wave instruction), the list length by a reduction over the wave -- one
    // atomic per record on ONE counter was 1.5 ms per build at 40 x 4000 x 24
    constexpr int g = 16;
    const int grp = threadIdx.x / g, gl = threadIdx.x % g, ngrp = 256 / g;
    int n_mine = 0;
    for (int sg = blockIdx.x * ngrp + grp; sg < s.S; sg += gridDim.x * ngrp) {
        const int ra = r0[sg], rb = r1[sg];
        for (int i = ra + gl; i < rb; i += g) {
            const unsigned w = sq.qt[i];
            const int u = (int)(w & 0xffffu);                   // the view's segment (the record's segID2); sg = the record's segID1
            if (MODE == 0) {
                ++n_mine;
                if (u < v.S && !e.pad) atomicMin(&best_ref[v.dense_base + u], ((unsigned long long)e.rank << 40) | (unsigned long long)(s.kept_base + i));
            } else if (u < v.S && sg < e.alias_S) {
                const int ai = v.dense_base + u, di = e.alias_base + sg;
                if (ai >= d0 && ai < d1) {
                    if (MODE == 1) atomicAdd(&cnt[ai - d0], 1);
                    else ent[bstart[ai - d0] + atomicSub(&cnt[ai - d0], 1) - 1] = (unsigned)di;
                }
                if (di >= d0 && di < d1) {
                    if (MODE == 1) atomicAdd(&cnt[di - d0], 1);
                    else ent[bstart[di - d0] + atomicSub(&cnt[di - d0], 1) - 1] = (unsigned)ai;
                }
            }
        }
    }
    if (MODE == 0) {
        for (int d = 32; d > 0; d >>= 1) n_mine += __shfl_down(n_mine, d);
        if ((threadIdx.x & 63) == 0 && n_mine) atomicAdd(&list_len[e.view], n_mine);
    }
}

// the entries of the early-return quirk (k_prod_keys_early's keys), both directions, through a counted scatter with global atomics (a handful of
// views): per row the DENSE ids of its extra targets
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_prodt_early(const Match* __restrict__ arena, const ProdView* __restrict__ pv, const ProdSrc* __restrict__ ps,
                                                     const int* __restrict__ src_list, const unsigned* __restrict__ chain_view_id, int d0, int d1, int* __restrict__ cnt,
                                                     const int* __restrict__ bstart, unsigned* __restrict__ ent)
{
    const int q = src_list[blockIdx.y];
    const ProdSrc e = ps[q];
    if (e.alias_base < 0) return;
    const ProdView v = pv[e.view], s = pv[e.src];
    const unsigned vid = chain_view_id[e.view];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < s.n_kept; i += gridDim.x * 256) {
        const Match r = arena[s.kept_base + i];
        if (r.camID2 != vid || (int)r.segID2 >= v.S || (int)r.segID1 >= e.alias_S) continue;
        const int ai = v.dense_base + (int)r.segID2, di = e.alias_base + (int)r.segID1;
        if (ai >= d0 && ai < d1) {
            if (!SCATTER) atomicAdd(&cnt[ai - d0], 1);
            else ent[bstart[ai - d0] + atomicSub(&cnt[ai - d0], 1) - 1] = (unsigned)di;
        }
        if (di >= d0 && di < d1) {
            if (!SCATTER) atomicAdd(&cnt[di - d0], 1);
            else ent[bstart[di - d0] + atomicSub(&cnt[di - d0], 1) - 1] = (unsigned)ai;
        }
    }
}

// One wave per row (x, u).  The row is put together from the views x touches, ascending, a GROUP of consecutive ones at a time (as many as fit 512
// words of bitmap: 8 views of 2000 segments, 4 of 4000): for view y the run (u, camera of y) of x's own list (F) and column u of the pair (y, camera of
// x) (B) set bits in y's words of the bitmap -- sorted and unique by construction; the lanes look all views' run and column bounds up at once, 64 views at
// a time.  2 KB of LDS per wave: every wave slot of the CU is used (the first version's bitmap over all touched views, 12.5 KB at 25 x 4000, left 12).
template <bool WRITE>
__global__ __launch_bounds__(256) void k_prodv_rows(const ProdRowView* __restrict__ rv, int x0, int d0, const ProdViewQ* __restrict__ vq, const ProdTouch* __restrict__ tl,
                                                    const ProdPair* __restrict__ pairs, const int* __restrict__ poff, const int* __restrict__ boff, const unsigned* __restrict__ E,
                                                    const int* __restrict__ bstart, const unsigned* __restrict__ ent, int* __restrict__ ucnt, const int* __restrict__ ustart,
                                                    long long base, int* __restrict__ pot_tgt, int group_words, int* __restrict__ stage,
                                                    const unsigned* __restrict__ poff_kq, int maxN)
{
    constexpr int kWords = 512;
    __shared__ unsigned s_bm[4][kWords];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const ProdRowView v = rv[x0 + blockIdx.y];
    const int u = blockIdx.x * 4 + wave;
    if (u >= v.S) return;                                       // (whole waves leave; no workgroup barrier below)
    unsigned* bm = s_bm[wave];
    const int row = v.dense_base + u - d0;
    const unsigned* fq = nullptr;
    const int* frt = nullptr;
    if (v.own_k >= 0) { const ProdViewQ o = vq[v.own_k]; if (o.rt) { fq = o.qt; frt = o.rt; } }
    const int e0 = bstart ? bstart[row] : 0, ne = bstart ? bstart[row + 1] - e0 : 0;
    int total = 0;
    long long o = WRITE ? base + ustart[row] : 0;
    if (!WRITE && stage && lane == 63) stage[(size_t)row * 64 + 63] = 0;         // (no staged row yet)
    if (WRITE && stage && stage[(size_t)row * 64 + 63] == -2) {
        // the counting pass left the row here, sorted and unique (stage[.. + 63] = -2 marks it)
        const int n = ustart[row + 1] - ustart[row];
        if (lane < n) pot_tgt[o + lane] = stage[(size_t)row * 64 + lane];
        return;
    }
    for (int t0 = v.t0; t0 < v.t1; t0 += 64) {
        const int j = t0 + lane;
        ProdTouch e;
        e.y_base = 0; e.y_S = 0; e.fq = -1; e.pair = -1;
        int f_a = 0, f_n = 0, b_a = 0, b_n = 0;
        if (j < v.t1) {
            e = tl[j];
            if (e.fq >= 0 && frt) { f_a = frt[(size_t)e.fq * v.S + u]; f_n = frt[(size_t)(e.fq + 1) * v.S + u] - f_a; }
            if (e.pair >= 0) {
                const ProdPair pr = pairs[e.pair];
                const int* bo = boff + pr.off_off;
                const int a = bo[u];
                b_n = bo[u + 1] - a;
                b_a = (poff_kq ? (int)poff_kq[(size_t)pr.k * maxN + pr.q] : poff[e.pair]) + a;         // (the chain's transposes: E is aligned with the arena; < 2^31 records asserted by the host)
            }
        }
        const int cnt = min(64, v.t1 - t0);
        const int words = j < v.t1 ? (e.y_S + 31) >> 5 : 0;
        // SHORT ROWS (config 2: 18 + 18 entries over 12 views): at most 64 entries in all -- one per lane, as dense ids; ranked against each other by
        // shuffles, put in order, neighbours compared: sorted and unique without a bitmap, a fence or a sweep per touched view
        if (v.t1 - v.t0 <= 64) {
            const int mine = f_n + b_n;
            int incl = mine;
            for (int d = 1; d < 64; d <<= 1) { const int x = __shfl_up(incl, d); if (lane >= d) incl += x; }
            const int excl = incl - mine, tot = __shfl(incl, 63);
            if (tot + ne <= 64) {
                int id = 0x7fffffff;
                int l = 0;                                       // owner of entry `lane`: the last lane whose exclusive count is <= lane
                for (int step = 32; step > 0; step >>= 1) { const int cand = l + step; const int ex = __shfl(excl, cand & 63); if (cand < 64 && ex <= lane) l = cand; }
                const int off = lane - __shfl(excl, l), ofn = __shfl(f_n, l), ofa = __shfl(f_a, l), oba = __shfl(b_a, l), oyS = __shfl(e.y_S, l), oyb = __shfl(e.y_base, l);
                if (lane < tot) {
                    const unsigned t = off < ofn ? (fq[ofa + off] & 0xffffu) : E[oba + off - ofn];
                    if ((int)t < oyS) id = oyb + (int)t;
                } else if (lane < tot + ne) id = (int)ent[e0 + lane - tot];
                int rank = 0;
                for (int k = 0; k < 64; ++k) { const int other = __shfl(id, k); rank += (other < id || (other == id && k < lane)) ? 1 : 0; }
                const int sorted = __builtin_amdgcn_ds_permute(rank << 2, id);          // lane `rank` receives this lane's id
                const int before = __shfl_up(sorted, 1);
                const bool keep = sorted != 0x7fffffff && (lane == 0 || before != sorted);
                const unsigned long long km = __ballot(keep);
                if (!WRITE) {
                    const int n = __popcll(km);
                    if (lane == 0) ucnt[row] = n;
                    if (stage && n < 64) {
                        if (keep) stage[(size_t)row * 64 + __popcll(km & ((1ull << lane) - 1ull))] = sorted;
                        if (lane == 63) stage[(size_t)row * 64 + 63] = -2;
                    }
                }
                else if (keep) pot_tgt[o + __popcll(km & ((1ull << lane) - 1ull))] = sorted;
                return;
            }
        }
        int jj = 0;
        while (jj < cnt) {
            // the group [jj, j1): views whose words fit the bitmap together (a single view always fits: <= 16384 segments)
            int j1 = jj, gw = 0, any = ne;
            while (j1 < cnt) { const int w = __shfl(words, j1); if (j1 > jj && gw + w > group_words) break; gw += w; any += __shfl(f_n, j1) + __shfl(b_n, j1); ++j1; }
            if (any == 0) { jj = j1; continue; }
            for (int w = lane; w < gw; w += 64) bm[w] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            int wo = 0;
            for (int k = jj; k < j1; ++k) {
                const int fn = __shfl(f_n, k), bn = __shfl(b_n, k), fa = __shfl(f_a, k), ba = __shfl(b_a, k), yS = __shfl(e.y_S, k), yb = __shfl(e.y_base, k);
                unsigned* bw = bm + wo;
                for (int i = lane; i < fn; i += 64) { const unsigned t = fq[fa + i] & 0xffffu; if ((int)t < yS) atomicOr(&bw[t >> 5], 1u << (t & 31u)); }
                for (int i = lane; i < bn; i += 64) { const unsigned t = E[ba + i]; if ((int)t < yS) atomicOr(&bw[t >> 5], 1u << (t & 31u)); }
                for (int i = lane; i < ne; i += 64) { const unsigned t = ent[e0 + i] - (unsigned)yb; if (t < (unsigned)yS) atomicOr(&bw[t >> 5], 1u << (t & 31u)); }
                wo += (yS + 31) >> 5;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (!WRITE) {
                for (int w = lane; w < gw; w += 64) total += __popc(bm[w]);
            } else {
                for (int wb = 0; wb < gw; wb += 64) {
                    const int w = wb + lane;
                    unsigned word = w < gw ? bm[w] : 0u;
                    if (__ballot(word != 0u) == 0ull) continue;
                    // the view a word belongs to: the last one of the group whose first word is <= w
                    int dbase = 0, acc = 0;
                    for (int k = jj; k < j1; ++k) { const int yb = __shfl(e.y_base, k), yw = __shfl(words, k); if (w >= acc) dbase = yb + ((w - acc) << 5); acc += yw; }
                    const int pc = __popc(word);
                    int incl = pc;
                    for (int d = 1; d < 64; d <<= 1) { const int x = __shfl_up(incl, d); if (lane >= d) incl += x; }
                    long long p = o + (incl - pc);
                    while (word) { const int bit = __ffs(word) - 1; pot_tgt[p++] = dbase + bit; word &= word - 1u; }
                    o += __shfl(incl, 63);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            jj = j1;
        }
    }
    if (!WRITE) {
        for (int d = 32; d > 0; d >>= 1) total += __shfl_down(total, d);
        if (lane == 0) ucnt[row] = total;
    }
}

// row starts of one block: pot_start[d0 + r] = base + ustart[r], r = 0 .. rows (the last one = where the next block's entries start)
__global__ __launch_bounds__(256) void k_prodt_row_starts(const int* __restrict__ ustart, int rows, long long base, long long* __restrict__ pot_start_at)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r <= rows) pot_start_at[r] = base + ustart[r];
}

}  // namespace l3d

namespace {

int bits_for(int n) { int b = 1; while ((1ll << b) <= (long long)n) ++b; return b; }

}  // namespace

void l3d::launch_prod_shift_rows(const long long* piece, long long n_rows, long long base, long long* pot_start_at, hipStream_t st)
{
    if (n_rows > 0) hipLaunchKernelGGL(k_prod_shift_rows, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, piece, n_rows, base, pot_start_at);
}

void l3d::launch_early_transposes(l3d_ctx* c, const EarlyView* views_dev, int k0, int nb, int maxN, int maxSt, const ChainResult* res,
                                  const unsigned* qt_arena, int* pcnt_kq, unsigned* poff_kq, int* boff, unsigned* E, unsigned* T, double avg_run, hipStream_t st)
{
    if (maxN <= 0 || nb <= 0) return;
    int g = 1;
    while (g < 64 && g < avg_run / 4.0) g <<= 1;
    if (c->opt.prod_pair_g >= 0) g = c->opt.prod_pair_g;
    int cap = 0;
    if (c->opt.prod_pair_stage != 0 && T) {
        cap = 12288;
        while (cap > 1024 && ((size_t)maxSt + 2 + 2 * (size_t)cap) * 4 > 64 * 1024) cap -= 1024;
        if (((size_t)maxSt + 2 + 2 * (size_t)cap) * 4 > 64 * 1024) cap = 0;
    }
    ProfScope p(c, "prod_keys", st);
    hipLaunchKernelGGL(k_prode_counts, dim3((unsigned)maxN, (unsigned)nb), dim3(256), 0, st, views_dev, k0, maxN, res, pcnt_kq);
    hipLaunchKernelGGL(k_prode_transpose, dim3((unsigned)maxN, (unsigned)nb), dim3(kPairThreads), ((size_t)maxSt + 2 + 2 * (size_t)cap) * 4, st, views_dev, k0, maxN, res, qt_arena, (const int*)pcnt_kq, poff_kq, g,
                       boff, E, cap ? T : (unsigned*)nullptr, cap);
}

int l3d::build_products(l3d_ctx* c, const l3d_chain_view* views, int n_views, const ProdChainView* pvh, const ChainResult* hres,
                        const l3d_dense_map* map, l3d_chain_summary* summary, int64_t* n_pot_out, int dv0, int dv1, const char* held, const unsigned* qt_arena, const ProdEarly* early)
{
    Products& P = c->products;
    P.valid = false; P.hyp_valid = false;
    hipStream_t st = c->stream;
    const int nv = map->n_views;
    if (nv <= 0) return fail(c, L3D_ERR_INVALID, "products: empty dense map");
    for (int i = 0; i < nv; ++i) {
        if (map->seg_base[i + 1] < map->seg_base[i] || (i && map->view_ids[i] <= map->view_ids[i - 1]) || map->seg_base[0] != 0)
            return fail(c, L3D_ERR_INVALID, "products: the dense map must ascend");
    }
    const int nd = map->seg_base[nv];
    if (nd <= 0) return fail(c, L3D_ERR_INVALID, "products: no segments");
    // [dv0, dv1): the dense views whose rows of the table this call builds (dv1 < 0: all) -- a rank of l3d_match_chain_blocks builds the rows of
    // its own block only, with offsets that start at 0; the pieces are all-gathered and put together by the caller
    const bool partial = dv1 >= 0;
    if (!partial) { dv0 = 0; dv1 = nv; }
    if (dv0 < 0 || dv1 > nv || dv0 > dv1) return fail(c, L3D_ERR_INVALID, "products: bad view range");
    P.seg_base.assign(map->seg_base, map->seg_base + nv + 1);
    P.view_ids.assign(map->view_ids, map->view_ids + nv);
    P.res.assign(hres, hres + n_views);
    P.n_dense = nd; P.n_views_all = nv; P.n_chain = n_views;
    P.chain_view.assign((size_t)n_views, -1);
    P.chain_verified.assign((size_t)n_views, 0);
    for (int k = 0; k < n_views; ++k) P.chain_verified[(size_t)k] = pvh[k].verified ? 1 : 0;
    P.chain_view_id.resize((size_t)n_views);
    P.early_src_index.assign((size_t)n_views, {}); P.early_src_cam.assign((size_t)n_views, {});
    auto view_of = [&](unsigned id) { auto it = std::lower_bound(P.view_ids.begin(), P.view_ids.end(), id); return it != P.view_ids.end() && *it == id ? (int)(it - P.view_ids.begin()) : -1; };

    // ---- host tables: one ProdView per chain view, one ProdSrc per (early-return view, source)
    std::vector<ProdView> pv((size_t)n_views);
    std::vector<ProdSrc> ps;
    std::vector<int> ps_alias_view;                         // dense view a ProdSrc's alias names (-1: none)
    long long total_kept = 0, early_slots = 0;
    for (int k = 0; k < n_views; ++k) if (pvh[k].verified) total_kept = std::max(total_kept, (long long)hres[k].kept_base + hres[k].n_kept);
    int max_kept = 0, maxS = 1;
    for (int k = 0; k < n_views; ++k) {
        const l3d_chain_view& v = views[k];
        const int vi = view_of(v.view_id);
        if (vi < 0 || P.seg_base[(size_t)vi + 1] - P.seg_base[(size_t)vi] != v.S_src) return fail(c, L3D_ERR_INVALID, "products: a chain view is missing from the dense map");
        P.chain_view[(size_t)k] = vi; P.chain_view_id[(size_t)k] = v.view_id;
        ProdView& o = pv[(size_t)k];
        o.kept_base = pvh[k].verified ? hres[k].kept_base : 0; o.n_kept = pvh[k].verified ? hres[k].n_kept : 0; o.R = pvh[k].verified ? hres[k].R : 0;
        o.dense_base = P.seg_base[(size_t)vi]; o.S = v.S_src; o.early = pvh[k].verified ? 0 : 1; o.pad = 0;
        o.bestpos = pvh[k].bestpos; o.best = pvh[k].best;
        max_kept = std::max(max_kept, o.n_kept); maxS = std::max(maxS, o.S);
        if (!pvh[k].verified) {
            P.early_src_index[(size_t)k].assign(v.source_index, v.source_index + v.n_sources);
            P.early_src_cam[(size_t)k].assign(v.source_cam, v.source_cam + v.n_sources);
            for (int q = 0; q < v.n_sources; ++q) {
                const int si = v.source_index[q];
                if (si < 0 || si >= k || !pvh[si].verified) continue;       // (a source is a verified earlier view)
                ProdSrc e;
                e.view = k; e.src = si; e.rank = q; e.pad = 0;     // (partitioned: an early-return view's best matches are found on EVERY rank -- the records that point at it are
                                                                    // all-gathered, in list order -- because what is filed under its local camera numbers makes it a target of far-away rows)
                const int av = view_of((unsigned)v.source_cam[q]);           // the LOCAL camera number read as a view id (line3D.cc:861-865)
                e.alias_base = av >= 0 ? P.seg_base[(size_t)av] : -1; e.alias_S = av >= 0 ? P.seg_base[(size_t)av + 1] - P.seg_base[(size_t)av] : 0;
                e.out_off = 0;
                early_slots += 2 * (long long)hres[si].n_kept;
                ps.push_back(e);
                ps_alias_view.push_back(av);
            }
        }
    }
    const long long slots_all = 2 * total_kept + early_slots;      // every key there can be: the bound of the table's entries
    const int nb = bits_for(nd);
    P.total_kept = total_kept;

    // ---- blocks of consecutive dense views.  touching[x] = the chain views / early pairs that can have a key with a source in dense
    // view x: a verified chain view touches its own view (forward keys) and its neighbours' (backward keys); an early pair its view and
    // the view its alias names.  A block is grown view by view while its key slots stay within the budget.
    std::vector<std::vector<int>> touch_view((size_t)nv), touch_src((size_t)nv);
    for (int k = 0; k < n_views; ++k) {
        if (!pvh[k].verified || hres[k].n_kept == 0) continue;
        touch_view[(size_t)P.chain_view[(size_t)k]].push_back(k);
        for (int q = 0; q < views[k].N; ++q) {
            const int t = view_of(views[k].local2global[q]);
            if (t >= 0 && t != P.chain_view[(size_t)k]) touch_view[(size_t)t].push_back(k);
        }
    }
    for (size_t q = 0; q < ps.size(); ++q) {
        if (hres[ps[q].src].n_kept == 0) continue;
        const int a = P.chain_view[(size_t)ps[q].view], b = ps_alias_view[q];
        touch_src[(size_t)a].push_back((int)q);
        if (b >= 0 && b != a) touch_src[(size_t)b].push_back((int)q);
    }
    // key slots per block: 2^28 (6.4 GB of transients at 24 B per slot) -- up to 2^30 when a quarter of the free HBM allows it: every block
    // re-reads the records of all views that touch it (its views and their neighbours), so a dense scene wants few, big blocks
    long long budget = 1ll << 28;
    if (c->opt.prod_block_keys > 0) budget = c->opt.prod_block_keys;
    else if (slots_all > budget) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) budget = std::max(budget, std::min<long long>((long long)(fr / 4 / 24), (1ll << 30) - 64));
    }
    std::vector<ProdBlock> blocks;
    std::vector<long long> off_tab;                         // per block: out_off of every chain view, then of every early pair
    {
        std::vector<int> mark_v((size_t)n_views, -1), mark_s(ps.size(), -1);
        int x = dv0;
        while (x < dv1) {
            const int bi = (int)blocks.size();
            ProdBlock B;
            B.d0 = P.seg_base[(size_t)x]; B.slots = 0;
            B.off_view = off_tab.size();
            off_tab.resize(off_tab.size() + (size_t)n_views + ps.size(), -1);
            B.off_src = B.off_view + (size_t)n_views;
            int x1 = x;
            for (; x1 < dv1; ++x1) {
                long long add = 0;
                for (int k : touch_view[(size_t)x1]) if (mark_v[(size_t)k] != bi) add += 2 * (long long)hres[k].n_kept;
                for (int q : touch_src[(size_t)x1]) if (mark_s[(size_t)q] != bi) add += 2 * (long long)hres[ps[(size_t)q].src].n_kept;
                if (x1 > x && B.slots + add > budget) break;
                for (int k : touch_view[(size_t)x1]) if (mark_v[(size_t)k] != bi) { mark_v[(size_t)k] = bi; off_tab[B.off_view + (size_t)k] = B.slots; B.slots += 2 * (long long)hres[k].n_kept; }
                for (int q : touch_src[(size_t)x1]) if (mark_s[(size_t)q] != bi) { mark_s[(size_t)q] = bi; off_tab[B.off_src + (size_t)q] = B.slots; B.slots += 2 * (long long)hres[ps[(size_t)q].src].n_kept; }
            }
            B.d1 = P.seg_base[(size_t)x1];
            if (B.slots > 0x7ffffff0ll) return fail(c, L3D_ERR_UNSUPPORTED, "products: one view and its neighbours hold more than 2^30 kept matches");
            blocks.push_back(B);
            x = x1;
        }
    }
    long long max_slots = 0;
    for (const ProdBlock& B : blocks) max_slots = std::max(max_slots, B.slots);
    const long long n_keys_max = max_slots + 1;            // (+ the sentinel)

    // ---- upload tables: [ProdView n_views][ProdSrc][ids nv][seg_base nv+1][chain view ids][out_off tables of all blocks]
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_pv = 0, o_ps = o_pv + al(pv.size() * sizeof(ProdView)), o_ids = o_ps + al(ps.size() * sizeof(ProdSrc) + 16),
                 o_sb = o_ids + al((size_t)nv * 4), o_cv = o_sb + al((size_t)(nv + 1) * 4), o_off = o_cv + al((size_t)n_views * 4), o_evq = o_off + al(off_tab.size() * 8 + 16),
                 o_slot = o_evq + al((size_t)n_views * sizeof(ProdViewQ) + 16), tab_total = o_slot + al(ps.size() * 4 + 16);
    HIPCHK(c, P.tables.reserve(tab_total));
    // the chain's side arrays and run tables (round 6), when every list with records has them: the early-return quirk reads runs instead of scanning lists
    bool native_rt = qt_arena != nullptr && c->opt.prod_transpose != 0;
    for (int k = 0; k < n_views; ++k) if (pvh[k].verified && hres[k].n_kept > 0 && !pvh[k].rt) native_rt = false;
    std::vector<ProdViewQ> evq((size_t)n_views);
    for (int k = 0; k < n_views; ++k) {
        const bool has = native_rt && pvh[k].verified && hres[k].n_kept > 0;
        evq[(size_t)k] = ProdViewQ{ has ? qt_arena + hres[k].kept_base : nullptr, has ? pvh[k].rt : nullptr, 0, views[k].N };
    }
    std::vector<int> ps_slot(ps.size() + 1, -1);                // the LOCAL camera number of the early-return view in its source's neighbour list
    for (size_t q = 0; q < ps.size(); ++q) {
        const l3d_chain_view& sv = views[ps[q].src];
        for (int j = 0; j < sv.N; ++j) if (sv.local2global[j] == P.chain_view_id[(size_t)ps[q].view]) { ps_slot[q] = j; break; }
    }
    char* tb = P.tables.as<char>();
    std::vector<char> tblob(tab_total);
    {
        auto put = [&](size_t off, const void* src, size_t bytes) { if (bytes) memcpy(tblob.data() + off, src, bytes); };
        put(o_pv, pv.data(), pv.size() * sizeof(ProdView)); put(o_ps, ps.data(), ps.size() * sizeof(ProdSrc)); put(o_ids, P.view_ids.data(), (size_t)nv * 4);
        put(o_sb, P.seg_base.data(), (size_t)(nv + 1) * 4); put(o_cv, P.chain_view_id.data(), (size_t)n_views * 4); put(o_off, off_tab.data(), off_tab.size() * 8);
        put(o_evq, evq.data(), evq.size() * sizeof(ProdViewQ)); put(o_slot, ps_slot.data(), ps_slot.size() * 4);
    }
    HIPCHK(c, hipMemcpyAsync(tb, tblob.data(), tblob.size(), hipMemcpyHostToDevice, st));
    const ProdView* dpv = reinterpret_cast<const ProdView*>(tb + o_pv);
    const ProdSrc* dps = reinterpret_cast<const ProdSrc*>(tb + o_ps);
    const unsigned* dids = reinterpret_cast<const unsigned*>(tb + o_ids);
    const int* dsb = reinterpret_cast<const int*>(tb + o_sb);
    const unsigned* dcv = reinterpret_cast<const unsigned*>(tb + o_cv);
    const long long* doff = reinterpret_cast<const long long*>(tb + o_off);
    const ProdViewQ* devq = reinterpret_cast<const ProdViewQ*>(tb + o_evq);
    const int* dslot = reinterpret_cast<const int*>(tb + o_slot);

    // ---- best references, medians (once); per block: keys, sort, unique, CSR rows
    const double t_res0 = now_s();
    HIPCHK(c, P.pot_start.reserve(((size_t)nd + 2) * 8));
    HIPCHK(c, P.best_ref.reserve((size_t)nd * 8 + 64));
    HIPCHK(c, P.median.reserve((size_t)n_views * 8 + 64));          // medians | list lengths of the early-return views
    if (c->opt.timing) fprintf(stderr, "[l3d products] %zu block(s) of up to %lld keys, table of up to %lld entries: buffers reserved in %.2f ms\n", blocks.size(), n_keys_max, slots_all, (now_s() - t_res0) * 1e3);
    int* d_list_len = reinterpret_cast<int*>(P.median.as<float>() + n_views);
    HIPCHK(c, hipMemsetAsync(d_list_len, 0, (size_t)n_views * 4, st));
    const Match* arena = c->ch_kept.as<Match>();
    HIPCHK(c, hipMemsetAsync(P.best_ref.p, 0xff, (size_t)nd * 8, st));
    const unsigned gx = (unsigned)std::max(1, std::min(512, (max_kept + 1023) / 1024));
    {
        ProfScope p(c, "prod_keys", st);
        if (!ps.empty() && native_rt)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_prod_early_rt<0>), dim3((unsigned)std::max(1, std::min(256, (maxS + 15) / 16)), (unsigned)ps.size()), dim3(256), 0, st, dpv, devq, dps, dslot, (const int*)nullptr, 0, 0,
                               (int*)nullptr, (const int*)nullptr, (unsigned*)nullptr, P.best_ref.as<unsigned long long>(), d_list_len);
        else if (!ps.empty()) hipLaunchKernelGGL(k_prod_keys_early, dim3(gx, (unsigned)ps.size()), dim3(256), 0, st, arena, dpv, dps, (const long long*)nullptr, dcv, nb, 0, 0,
                                                 (unsigned long long*)nullptr, P.best_ref.as<unsigned long long>(), d_list_len);
        hipLaunchKernelGGL(k_prod_best, dim3((maxS + 255) / 256, n_views), dim3(256), 0, st, dpv, P.best_ref.as<long long>());
        hipLaunchKernelGGL(k_prod_median, dim3(n_views), dim3(256), 0, st, dpv, P.median.as<float>());
    }
    // ---- the transposed build (round 6; the kernels above).  kSortInstead: the records are not what it assumes -- the caller sorts.
    constexpr int kSortInstead = -12345;
    auto transposed = [&](long long& n_pot_out) -> int {
        const double t_t0 = now_s();
        auto seg_count = [&](int x) { return P.seg_base[(size_t)x + 1] - P.seg_base[(size_t)x]; };
        for (int x = dv0; x < dv1; ++x) if (seg_count(x) > 16000) return kSortInstead;                                  // (the pair transposes keep a view's segments in 64 KB of LDS)
        const bool rebuild = !native_rt;                        // (a list without a run table: all are rebuilt, block by block)
        const bool use_early = early && !rebuild && early->E && early->boff && early->poff_kq;      // the chain transposed every pair behind its view's kept writer
        // T(x): the dense views a row of view x can name -- its chain view's neighbours, the views it is a neighbour of, the early-return aliases
        // (the dense view of every (chain view, local camera), looked up ONCE: the table loops below asked three times per pair and 144 times per touched view --
        // 0.12 ms of host time per config-2 pass with the GPU idle behind it)
        std::vector<int> nbv_off((size_t)n_views + 1, 0), nbv;
        for (int k = 0; k < n_views; ++k) {
            nbv_off[(size_t)k] = (int)nbv.size();
            if (pvh[k].verified) for (int q = 0; q < views[k].N; ++q) nbv.push_back(view_of(views[k].local2global[q]));
        }
        nbv_off[(size_t)n_views] = (int)nbv.size();
        std::vector<std::vector<int>> T((size_t)nv);
        for (int x = 0; x < nv; ++x) T[(size_t)x].reserve(32);
        for (int k = 0; k < n_views; ++k) {
            if (!pvh[k].verified) continue;
            const int vi = P.chain_view[(size_t)k];
            for (int q = 0; q < views[k].N; ++q) {
                const int t = nbv[(size_t)nbv_off[(size_t)k] + q];
                if (t >= 0) { T[(size_t)vi].push_back(t); T[(size_t)t].push_back(vi); }
            }
        }
        for (size_t q = 0; q < ps.size(); ++q) {
            const int a = P.chain_view[(size_t)ps[q].view], b = ps_alias_view[q];
            if (b >= 0) { T[(size_t)a].push_back(b); T[(size_t)b].push_back(a); }
        }
        for (int x = 0; x < nv; ++x) { std::vector<int>& t = T[(size_t)x]; std::sort(t.begin(), t.end()); t.erase(std::unique(t.begin(), t.end()), t.end()); }
        // per chain view and LOCAL camera: the target view (ProdNbQ); the pairs grouped by target view; the sorted ids for lists whose side arrays are rebuilt
        std::vector<int> nb_off((size_t)n_views, 0);
        std::vector<ProdNbQ> nbt;
        std::vector<unsigned> ids_sorted;
        std::vector<int> qs_sorted;
        std::vector<std::vector<ProdPair>> pairs_of((size_t)nv);
        std::vector<int> own_k((size_t)nv, -1);
        std::vector<std::pair<unsigned, int>> byid;
        nbt.reserve(nbv.size()); ids_sorted.reserve(nbv.size()); qs_sorted.reserve(nbv.size());
        for (int k = 0; k < n_views; ++k) {
            nb_off[(size_t)k] = (int)nbt.size();
            if (!pvh[k].verified) continue;
            const int vi = P.chain_view[(size_t)k], N = views[k].N;
            if (hres[k].n_kept > 0) own_k[(size_t)vi] = k;
            byid.clear();
            for (int q = 0; q < N; ++q) {
                const int t = nbv[(size_t)nbv_off[(size_t)k] + q];
                nbt.push_back(ProdNbQ{ t >= 0 ? P.seg_base[(size_t)t] : -1, t >= 0 ? seg_count(t) : 0 });
                byid.push_back({ views[k].local2global[q], q });
                if (t >= 0 && hres[k].n_kept > 0) pairs_of[(size_t)t].push_back(ProdPair{ k, q, 0, 0 });
            }
            std::sort(byid.begin(), byid.end());
            for (auto& e : byid) { ids_sorted.push_back(e.first); qs_sorted.push_back(e.second); }
        }
        std::vector<ProdRowView> rv((size_t)nv);
        std::vector<ProdPair> pairs;
        std::vector<ProdTouch> tl;
        pairs.reserve(nbv.size()); tl.reserve(2 * nbv.size() + 64);
        int max_touch = 0;
        for (int x = 0; x < nv; ++x) {
            ProdRowView& r = rv[(size_t)x];
            r.dense_base = P.seg_base[(size_t)x]; r.S = seg_count(x); r.own_k = own_k[(size_t)x]; r.t0 = r.t1 = 0; r.pad = 0;
            if (x < dv0 || x >= dv1) continue;
            const int pair0 = (int)pairs.size();
            for (const ProdPair& pr : pairs_of[(size_t)x]) pairs.push_back(pr);
            r.t0 = (int)tl.size();
            for (int y : T[(size_t)x]) {
                ProdTouch e;
                e.y_base = P.seg_base[(size_t)y]; e.y_S = seg_count(y); e.fq = -1; e.pair = -1;
                if (r.own_k >= 0) for (int q = 0; q < views[r.own_k].N; ++q) if (nbv[(size_t)nbv_off[(size_t)r.own_k] + q] == y) { e.fq = q; break; }
                for (int p = pair0; p < (int)pairs.size(); ++p) if (P.chain_view[(size_t)pairs[(size_t)p].k] == y) { e.pair = p; break; }
                tl.push_back(e);
            }
            r.t1 = (int)tl.size();
            max_touch = std::max(max_touch, r.t1 - r.t0);
        }
        // where the pairs of view x start in `pairs` (the blocks below cut the list at view boundaries)
        std::vector<int> pair_begin((size_t)nv + 1, 0);
        {
            int p = 0;
            for (int x = 0; x < nv; ++x) { pair_begin[(size_t)x] = p; if (x >= dv0 && x < dv1) p += (int)pairs_of[(size_t)x].size(); }
            pair_begin[(size_t)nv] = p;
        }
        // blocks of consecutive dense views: the records of the chain views that touch a block (its B entries; with rebuilt side arrays also their qt) within the budget
        struct TBlock { int x0, x1; size_t item0, n_items, src0, n_srcs; long long rec_n, rt_n, boff_n, early_n; int maxS, pair0, pair1; };
        std::vector<TBlock> tblocks;
        std::vector<int> items;                                 // chain views touching the block
        std::vector<int> srcs;
        const long long t_budget = c->opt.prod_block_keys > 0 ? c->opt.prod_block_keys : (1ll << 30) - 64;
        {
            std::vector<int> mark_v((size_t)n_views, -1), mark_s(ps.size(), -1);
            int x = dv0;
            while (x < dv1) {
                const int bi = (int)tblocks.size();
                TBlock B;
                B.x0 = x; B.item0 = items.size(); B.src0 = srcs.size(); B.rec_n = 0; B.rt_n = 0; B.boff_n = 0; B.early_n = 0; B.maxS = 1; B.pair0 = pair_begin[(size_t)x];
                int x1 = x;
                for (; x1 < dv1; ++x1) {
                    long long add = 0;
                    for (int k : touch_view[(size_t)x1]) if (mark_v[(size_t)k] != bi) add += hres[k].n_kept;
                    for (int q : touch_src[(size_t)x1]) if (mark_s[(size_t)q] != bi) add += 2 * (long long)hres[ps[(size_t)q].src].n_kept;
                    if (x1 > x && B.rec_n + B.early_n + add > t_budget) break;
                    for (int k : touch_view[(size_t)x1]) if (mark_v[(size_t)k] != bi) {
                        mark_v[(size_t)k] = bi; items.push_back(k); B.rec_n += hres[k].n_kept; B.rt_n += ((long long)views[k].N + 1) * views[k].S_src;
                    }
                    for (int q : touch_src[(size_t)x1]) if (mark_s[(size_t)q] != bi) { mark_s[(size_t)q] = bi; srcs.push_back(q); B.early_n += 2 * (long long)hres[ps[(size_t)q].src].n_kept; }
                    B.maxS = std::max(B.maxS, seg_count(x1));
                    for (int p = pair_begin[(size_t)x1]; p < pair_begin[(size_t)x1 + 1]; ++p) {
                        if (use_early) { pairs[(size_t)p].off_off = early->boff_off_host[(size_t)pairs[(size_t)p].k * early->maxN + pairs[(size_t)p].q]; continue; }   // (the chain's transposes: canonical places)
                        pairs[(size_t)p].off_off = (int)B.boff_n; B.boff_n += seg_count(x1) + 1;
                    }
                }
                B.x1 = x1; B.n_items = items.size() - B.item0; B.n_srcs = srcs.size() - B.src0; B.pair1 = pair_begin[(size_t)x1];
                if (B.rec_n > 0x7ffffff0ll || B.early_n > 0x7ffffff0ll || B.boff_n > 0x7ffffff0ll || B.rt_n > 0x7ffffff0ll)
                    return fail(c, L3D_ERR_UNSUPPORTED, "products: one view and its neighbours hold more than 2^31 kept matches");
                tblocks.push_back(B);
                x = x1;
            }
        }
        long long max_rec = 0, max_rt = 0, max_boff = 0, max_early = 0;
        int max_rows = 0, max_pairs = 0;
        for (const TBlock& B : tblocks) {
            max_rec = std::max(max_rec, B.rec_n); max_rt = std::max(max_rt, B.rt_n); max_boff = std::max(max_boff, B.boff_n); max_early = std::max(max_early, B.early_n);
            max_rows = std::max(max_rows, P.seg_base[(size_t)B.x1] - P.seg_base[(size_t)B.x0]); max_pairs = std::max(max_pairs, B.pair1 - B.pair0);
        }
        // ---- tables: [ProdViewQ x n_views (rewritten per block when rebuilt)][ProdNbQ][ids][qs][ProdRowView][ProdTouch][pairs][srcs][RtJobs]
        std::vector<ProdViewQ> vqh((size_t)n_views);
        for (int k = 0; k < n_views; ++k) {
            ProdViewQ& o = vqh[(size_t)k];
            o.nb_off = nb_off[(size_t)k]; o.N = views[k].N;
            const bool has = pvh[k].verified && hres[k].n_kept > 0 && !rebuild;
            o.qt = has ? qt_arena + hres[k].kept_base : nullptr; o.rt = has ? pvh[k].rt : nullptr;
        }
        const size_t q_vq = 0, q_nb = q_vq + al(vqh.size() * sizeof(ProdViewQ) + 16), q_ids = q_nb + al(nbt.size() * sizeof(ProdNbQ) + 16), q_qs = q_ids + al(ids_sorted.size() * 4 + 16),
                     q_rv = q_qs + al(qs_sorted.size() * 4 + 16), q_tl = q_rv + al(rv.size() * sizeof(ProdRowView)), q_pr = q_tl + al(tl.size() * sizeof(ProdTouch) + 16),
                     q_sr = q_pr + al(pairs.size() * sizeof(ProdPair) + 16), q_job = q_sr + al(srcs.size() * 4 + 16), q_total = q_job + al((size_t)n_views * sizeof(RtJob) + 16);
        HIPCHK(c, P.ttab.reserve(q_total));
        char* tt = P.ttab.as<char>();
        std::vector<char> blob(q_job);                          // (one copy instead of eight: a config-2 pass pays ~8 us per small copy)
        auto put = [&](size_t off, const void* src, size_t bytes) { if (bytes) memcpy(blob.data() + off, src, bytes); };
        put(q_vq, vqh.data(), vqh.size() * sizeof(ProdViewQ)); put(q_nb, nbt.data(), nbt.size() * sizeof(ProdNbQ)); put(q_ids, ids_sorted.data(), ids_sorted.size() * 4);
        put(q_qs, qs_sorted.data(), qs_sorted.size() * 4); put(q_rv, rv.data(), rv.size() * sizeof(ProdRowView)); put(q_tl, tl.data(), tl.size() * sizeof(ProdTouch));
        put(q_pr, pairs.data(), pairs.size() * sizeof(ProdPair)); put(q_sr, srcs.data(), srcs.size() * 4);
        HIPCHK(c, hipMemcpyAsync(tt, blob.data(), blob.size(), hipMemcpyHostToDevice, st));
        const ProdViewQ* dvq = reinterpret_cast<const ProdViewQ*>(tt + q_vq);
        const ProdNbQ* dnb = reinterpret_cast<const ProdNbQ*>(tt + q_nb);
        const unsigned* dids = reinterpret_cast<const unsigned*>(tt + q_ids);
        const int* dqs = reinterpret_cast<const int*>(tt + q_qs);
        const ProdRowView* drv = reinterpret_cast<const ProdRowView*>(tt + q_rv);
        const ProdTouch* dtl = reinterpret_cast<const ProdTouch*>(tt + q_tl);
        const ProdPair* dpr = reinterpret_cast<const ProdPair*>(tt + q_pr);
        const int* dsr = reinterpret_cast<const int*>(tt + q_sr);
        RtJob* djob = reinterpret_cast<RtJob*>(tt + q_job);
        // ---- transients: E (transposed entries; early entries behind them), column starts, pair counts / offsets, four row arrays (+ the error counter);
        // rebuilt side arrays: qt and rt of the block's chain views
        const size_t ra = ((size_t)max_rows + 18 + 63) & ~(size_t)63;         // ints per row array (rows + 1 entries, the error counter in the last eight of the last one)
        const size_t pa = ((size_t)max_pairs + 2 + 63) & ~(size_t)63;
        const double t_res1 = now_s();
        HIPCHK(c, P.keys2.reserve(((use_early ? 0 : (size_t)max_rec) + (size_t)max_early) * 4 + 256));
        HIPCHK(c, P.flag.reserve((5 * ra + 2 * pa + (size_t)max_boff) * 4 + 256));
        if (rebuild) { HIPCHK(c, P.keys.reserve((size_t)max_rec * 4 + 256)); HIPCHK(c, P.pos.reserve((size_t)max_rt * 4 + 256)); HIPCHK(c, P.tstage.reserve((size_t)max_rec * 4 + 256)); }
        size_t tbs = 0;
        HIPCHK(c, exclusive_sum_int(nullptr, tbs, nullptr, nullptr, std::max(max_rows, max_pairs) + 1, st));
        HIPCHK(c, P.tmp.reserve(tbs + 256));
        if (tblocks.size() > 1) HIPCHK(c, P.pot_tgt.reserve(((size_t)slots_all + 2) * 4));        // (several blocks: the bound; one block: its count, below)
        if (c->opt.timing) fprintf(stderr, "[l3d products] transposed (%s side arrays): %zu block(s), <= %lld records, %d pairs, %d rows, a row touches <= %d views: tables in %.2f ms, buffers in %.2f ms\n",
                                   rebuild ? "rebuilt" : "the chain's", tblocks.size(), max_rec, max_pairs, max_rows, max_touch, (t_res1 - t_t0) * 1e3, (now_s() - t_res1) * 1e3);
        unsigned* E = use_early ? const_cast<unsigned*>(early->E) : P.keys2.as<unsigned>();
        int* cnt = P.flag.as<int>();
        int *bstart = cnt + ra, *ucnt = cnt + 2 * ra, *ustart = cnt + 3 * ra, *spare = cnt + 4 * ra, *pcnt = cnt + 5 * ra, *poff = pcnt + pa, *boff = use_early ? const_cast<int*>(early->boff) : poff + pa;
        const unsigned* poff_kq = use_early ? early->poff_kq : nullptr;
        const int e_maxN = use_early ? early->maxN : 0;
        int* err = spare + ra - 8;
        HIPCHK(c, hipMemsetAsync(err, 0, 4, st));
        long long base = 0;
        std::vector<RtJob> jobs;
        for (const TBlock& B : tblocks) {
            const int d0 = P.seg_base[(size_t)B.x0], d1 = P.seg_base[(size_t)B.x1], rows = d1 - d0, n_pairs = B.pair1 - B.pair0;
            int scal[2] = { 0, 0 };                                            // {entries of the block, error count}
            if (rows > 0) {
                unsigned* ent = use_early ? P.keys2.as<unsigned>() : E + B.rec_n;      // (early-return entries behind the transposed ones; the chain's transposes: a buffer of their own)
                {
                    ProfScope p(c, "prod_keys", st);
                    if (rebuild) {
                        // the side arrays of the block's chain views, from their records
                        jobs.clear();
                        long long qo = 0, ro = 0;
                        int max_n = 0, max_cells = 0;
                        for (size_t i = 0; i < B.n_items; ++i) {
                            const int k = items[B.item0 + i];
                            RtJob j;
                            j.recs = arena + hres[k].kept_base; j.qt = P.keys.as<unsigned>() + qo; j.rt = P.pos.as<int>() + ro;
                            j.ids = dids + nb_off[(size_t)k]; j.qs = dqs + nb_off[(size_t)k]; j.n = hres[k].n_kept; j.S = views[k].S_src; j.N = views[k].N; j.pad = 0;
                            j.skey = P.tstage.as<unsigned>() + qo;            // (the staging region of the pair transposes, not in use yet)
                            vqh[(size_t)k].qt = j.qt; vqh[(size_t)k].rt = j.rt;
                            qo += j.n; ro += ((long long)j.N + 1) * j.S;
                            max_n = std::max(max_n, j.n); max_cells = std::max(max_cells, (j.N + 1) * j.S);
                            jobs.push_back(j);
                        }
                        HIPCHK(c, hipMemcpyAsync(djob, jobs.data(), jobs.size() * sizeof(RtJob), hipMemcpyHostToDevice, st));
                        HIPCHK(c, hipMemcpyAsync(tt + q_vq, vqh.data(), vqh.size() * sizeof(ProdViewQ), hipMemcpyHostToDevice, st));
                        launch_qt_from_records(djob, (int)jobs.size(), max_n, err, st);
                        launch_rt_from_qt(djob, (int)jobs.size(), max_cells, st);
                        HIPCHK(c, hipStreamSynchronize(st));                   // (the host vectors above are re-used by the next block)
                    }
                    if (n_pairs > 0 && !use_early) {
                        hipLaunchKernelGGL(k_prodv_pair_counts, dim3((unsigned)n_pairs), dim3(256), 0, st, dpr + B.pair0, dvq, dpv, pcnt);
                        HIPCHK(c, hipMemsetAsync(pcnt + n_pairs, 0, 4, st));
                        size_t t1 = tbs;
                        HIPCHK(c, exclusive_sum_int(P.tmp.p, t1, pcnt, poff, n_pairs + 1, st));
                        // lanes per run (prod_pair_g: -1 = a quarter of the average run of the block's lists rounded up to a power of two, 0 = a run per thread)
                        double recs = 0, cells = 0;
                        for (size_t i = 0; i < B.n_items; ++i) { const int k = items[B.item0 + i]; recs += hres[k].n_kept; cells += (double)views[k].N * views[k].S_src; }
                        int g = 1;
                        while (g < 64 && g < recs / std::max(1.0, cells) / 4.0) g <<= 1;
                        if (c->opt.prod_pair_g >= 0) g = c->opt.prod_pair_g;
                        // two-level scatter (prod_pair_stage: 1 on, 0 the direct scatter): an LDS image of `cap` entries of E per workgroup + a cursor per u of a bucket
                        int cap = 0;
                        unsigned* T = nullptr;
                        if (c->opt.prod_pair_stage != 0) {
                            cap = 12288;
                            while (cap > 1024 && ((size_t)B.maxS + 2 + 2 * (size_t)cap) * 4 > 64 * 1024) cap -= 1024;
                            if (((size_t)B.maxS + 2 + 2 * (size_t)cap) * 4 <= 64 * 1024) { HIPCHK(c, P.tstage.reserve((size_t)max_rec * 4 + 256)); T = P.tstage.as<unsigned>(); } else cap = 0;
                        }
                        hipLaunchKernelGGL(k_prodv_pair_transpose, dim3((unsigned)n_pairs), dim3(kPairThreads), ((size_t)B.maxS + 2 + 2 * (size_t)cap) * 4, st, dpr + B.pair0, dvq, dpv, dnb,
                                           (const int*)poff, g, boff, E, T, cap);
                    }
                    if (B.n_srcs) {
                        HIPCHK(c, hipMemsetAsync(cnt, 0, ((size_t)rows + 1) * 4, st));
                        const dim3 eg((unsigned)std::max(1, std::min(256, (maxS + 15) / 16)), (unsigned)B.n_srcs);
                        if (!rebuild) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_prod_early_rt<1>), eg, dim3(256), 0, st, dpv, dvq, dps, dslot, dsr + B.src0, d0, d1, cnt, (const int*)bstart, ent,
                                                         (unsigned long long*)nullptr, (int*)nullptr);
                        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_prodt_early<false>), dim3(gx, (unsigned)B.n_srcs), dim3(256), 0, st, arena, dpv, dps, dsr + B.src0, dcv, d0, d1, cnt, (const int*)bstart, ent);
                        size_t t1 = tbs;
                        HIPCHK(c, exclusive_sum_int(P.tmp.p, t1, cnt, bstart, rows + 1, st));
                        if (!rebuild) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_prod_early_rt<2>), eg, dim3(256), 0, st, dpv, dvq, dps, dslot, dsr + B.src0, d0, d1, cnt, (const int*)bstart, ent,
                                                         (unsigned long long*)nullptr, (int*)nullptr);
                        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_prodt_early<true>), dim3(gx, (unsigned)B.n_srcs), dim3(256), 0, st, arena, dpv, dps, dsr + B.src0, dcv, d0, d1, cnt, (const int*)bstart, ent);
                    }
                }
                const dim3 rg((unsigned)((B.maxS + 3) / 4), (unsigned)(B.x1 - B.x0));
                const int* bs = B.n_srcs ? bstart : nullptr;
                // short rows (on average at most 48 entries before the duplicates go) are left sorted in a staging row of 64 ints by the counting pass; the writing pass copies
                int* stage = nullptr;
                if (2.0 * (double)B.rec_n <= 48.0 * rows && (size_t)rows * 256 <= ((size_t)1 << 30)) { HIPCHK(c, P.rowstage.reserve((size_t)rows * 256 + 256)); stage = P.rowstage.as<int>(); }
                {
                    ProfScope p(c, "prod_rows", st);
                    HIPCHK(c, hipMemsetAsync(ucnt + rows, 0, 4, st));
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_prodv_rows<false>), rg, dim3(256), 0, st, drv, B.x0, d0, dvq, dtl, dpr, (const int*)poff - B.pair0, (const int*)boff,
                                       (const unsigned*)E, bs, (const unsigned*)ent, ucnt, (const int*)ustart, base, (int*)nullptr, c->opt.prod_row_group, stage, poff_kq, e_maxN);
                    size_t t1 = tbs;
                    HIPCHK(c, exclusive_sum_int(P.tmp.p, t1, ucnt, ustart, rows + 1, st));
                }
                HIPCHK(c, hipMemcpyAsync(&scal[0], ustart + rows, 4, hipMemcpyDeviceToHost, st));
                HIPCHK(c, hipMemcpyAsync(&scal[1], err, 4, hipMemcpyDeviceToHost, st));
                HIPCHK(c, hipStreamSynchronize(st));
                if (scal[1]) return kSortInstead;
                if (tblocks.size() == 1) HIPCHK(c, P.pot_tgt.reserve(((size_t)scal[0] + 2) * 4));
                {
                    ProfScope p(c, "prod_rows", st);
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_prodv_rows<true>), rg, dim3(256), 0, st, drv, B.x0, d0, dvq, dtl, dpr, (const int*)poff - B.pair0, (const int*)boff,
                                       (const unsigned*)E, bs, (const unsigned*)ent, ucnt, (const int*)ustart, base, P.pot_tgt.as<int>(), c->opt.prod_row_group, stage, poff_kq, e_maxN);
                    hipLaunchKernelGGL(k_prodt_row_starts, dim3((unsigned)((rows + 256) / 256)), dim3(256), 0, st, (const int*)ustart, rows, base, P.pot_start.as<long long>() + d0);
                }
            } else {
                hipLaunchKernelGGL(k_prod_fill_rows, dim3(1), dim3(256), 0, st, P.pot_start.as<long long>(), (long long)d0, (long long)d0 + 1, base);
            }
            base += scal[0];
        }
        if (tblocks.empty()) HIPCHK(c, P.pot_tgt.reserve(64));
        n_pot_out = base;
        return L3D_OK;
    };
    long long n_pot = 0;
    bool rows_built = false;
    if (c->opt.prod_transpose != 0) {
        const int rc = transposed(n_pot);
        if (rc == L3D_OK) rows_built = true;
        else if (rc != kSortInstead) return rc;
        else if (c->opt.timing) fprintf(stderr, "[l3d products] transposed build refused (a slice is not ordered by segment, or a target outside the bit space): sorting instead\n");
    }
    if (!rows_built) {
        // ---- the sorted build (rounds 3-5; A/B: L3D_PROD_TRANSPOSE=0, and the way out for slices the transposed build refuses)
        HIPCHK(c, P.keys.reserve((size_t)n_keys_max * 8 + 64));
        HIPCHK(c, P.keys2.reserve((size_t)n_keys_max * 8 + 64));
        HIPCHK(c, P.flag.reserve(((size_t)n_keys_max + 2) * 4));
        HIPCHK(c, P.pos.reserve(((size_t)n_keys_max + 2) * 4));
        HIPCHK(c, P.pot_tgt.reserve(((size_t)slots_all + 2) * 4));
        unsigned long long* keys = P.keys.as<unsigned long long>();
        unsigned long long* keys2 = P.keys2.as<unsigned long long>();
        size_t tb1 = 0, tb2 = 0;
        HIPCHK(c, sort_keys_u64(nullptr, tb1, keys, keys2, (int)n_keys_max, 0, std::min(64, 2 * nb), st));
        HIPCHK(c, exclusive_sum_int(nullptr, tb2, P.flag.as<int>(), P.pos.as<int>(), (int)n_keys_max + 1, st));
        HIPCHK(c, P.tmp.reserve(std::max(tb1, tb2) + 256));
        long long base = 0;
        for (size_t bi = 0; bi < blocks.size(); ++bi) {
            const ProdBlock& B = blocks[bi];
            const long long n_keys = B.slots + 1;
            {
                ProfScope p(c, "prod_keys", st);
                HIPCHK(c, hipMemsetAsync(keys + B.slots, 0xff, 8, st));          // the sentinel
                if (max_kept > 0) hipLaunchKernelGGL(k_prod_keys, dim3(gx, n_views), dim3(256), 0, st, arena, dpv, doff + B.off_view, dids, dsb, nv, nb, B.d0, B.d1, keys);
                if (!ps.empty()) hipLaunchKernelGGL(k_prod_keys_early, dim3(gx, (unsigned)ps.size()), dim3(256), 0, st, arena, dpv, dps, doff + B.off_src, dcv, nb, B.d0, B.d1,
                                                    keys, P.best_ref.as<unsigned long long>(), d_list_len);
            }
            {
                ProfScope p(c, "prod_sort", st);
                size_t t1 = tb1, t2 = tb2;
                HIPCHK(c, sort_keys_u64(P.tmp.p, t1, keys, keys2, (int)n_keys, 0, std::min(64, 2 * nb), st));
                const unsigned nblk = (unsigned)((n_keys + 255) / 256);
                HIPCHK(c, hipMemsetAsync(P.flag.as<int>() + n_keys, 0, 4, st));
                hipLaunchKernelGGL(k_prod_flags, dim3(nblk), dim3(256), 0, st, keys2, n_keys, P.flag.as<int>());
                HIPCHK(c, exclusive_sum_int(P.tmp.p, t2, P.flag.as<int>(), P.pos.as<int>(), (int)n_keys + 1, st));
                hipLaunchKernelGGL(k_prod_csr, dim3(nblk), dim3(256), 0, st, keys2, P.flag.as<int>(), P.pos.as<int>(), n_keys, nb, B.d0, B.d1, base, P.pot_start.as<long long>(), P.pot_tgt.as<int>());
            }
            if (bi + 1 < blocks.size()) {                      // the next block's rows start behind this block's entries
                int n_unique = 0;
                HIPCHK(c, hipMemcpyAsync(&n_unique, P.pos.as<int>() + n_keys, 4, hipMemcpyDeviceToHost, st));
                HIPCHK(c, hipStreamSynchronize(st));
                base += n_unique;
            }
        }
        const long long last_keys = blocks.empty() ? 0 : blocks.back().slots + 1;
        int n_last = 0;
        if (!blocks.empty()) HIPCHK(c, hipMemcpyAsync(&n_last, P.pos.as<int>() + last_keys, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        n_pot = base + n_last;
    }
    // ---- the scalars the host needs
    std::vector<float> med((size_t)2 * n_views, 1.0f);
    HIPCHK(c, hipMemcpyAsync(med.data(), P.median.p, (size_t)n_views * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(c, L3D_ERR_HIP, std::string("products: ") + hipGetErrorString(e_)); }
    P.n_pot = n_pot;
    if (held) {
        const long long r0 = dv0 >= dv1 ? (long long)nd + 1 : (long long)P.seg_base[(size_t)dv0], r1 = dv0 >= dv1 ? (long long)nd + 1 : (long long)P.seg_base[(size_t)dv1] + 1;
        if (r0 > 0) hipLaunchKernelGGL(k_prod_fill_rows, dim3((unsigned)((std::min<long long>(r0, nd + 1) + 255) / 256)), dim3(256), 0, st, P.pot_start.as<long long>(), 0ll, std::min<long long>(r0, nd + 1), 0ll);
        if (r1 <= nd) hipLaunchKernelGGL(k_prod_fill_rows, dim3((unsigned)(((long long)nd + 1 - r1 + 255) / 256)), dim3(256), 0, st, P.pot_start.as<long long>(), r1, (long long)nd + 1, n_pot);
        HIPCHK(c, hipStreamSynchronize(st));
        if (c->opt.part_release != 0) { P.keys.release(); P.keys2.release(); P.flag.release(); P.pos.release(); P.tmp.release(); }     // (the key blocks: transients)
    }
    for (int k = 0; k < n_views; ++k) {
        l3d_chain_summary& s = summary[k];
        s.verified = pvh[k].verified; s.n_kept = pvh[k].verified ? hres[k].n_kept : reinterpret_cast<const int*>(med.data() + n_views)[k];
        s.n_candidates = pvh[k].verified ? hres[k].R : 0;
        s.median_depth = med[(size_t)k]; s.pad = 0;
    }
    if (n_pot_out) *n_pot_out = n_pot;
    P.valid = !partial;
    return L3D_OK;
}

// =================================================================================================================================
// greedy selection on the resident products
namespace l3d {

struct ViewGeo {
    double RtKinv[9], C[3];
    float k_lower, k_upper, median_depth;
    int pad;
    const float4* segs;
};

__global__ __launch_bounds__(256) void k_hyp_flag(const long long* __restrict__ best_ref, int nd, int* __restrict__ flag)
{
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d < nd) flag[d] = best_ref[d] >= 0 ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_hyp_build(const Match* __restrict__ arena, const long long* __restrict__ best_ref, const int* __restrict__ hyp_of,
                                                   const int* __restrict__ seg_base, int nv, const ViewGeo* __restrict__ geo, int nd,
                                                   Hypothesis* __restrict__ hyp, float* __restrict__ score, int* __restrict__ hyp_dense, int* __restrict__ best_hyp,
                                                   int* __restrict__ view_hyp_begin)
{
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d <= nv) view_hyp_begin[d] = hyp_of[seg_base[d]];          // (d doubles as a view index for the first nv + 1 threads)
    if (d >= nd) return;
    const long long ref = best_ref[d];
    if (ref < 0) { best_hyp[d] = -1; return; }
    int lo = 0, hi = nv;                                          // view of the dense id: last view whose base is <= d
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_base[mid] <= d) lo = mid; else hi = mid; }
    const ViewGeo g = geo[lo];
    const Match r = arena[ref & kBestIndexMask];
    const bool rev = (ref & kBestReversed) != 0;
    const float d1 = rev ? r.depths[2] : r.depths[0], d2 = rev ? r.depths[3] : r.depths[1];
    const float conf = rev ? 0.0f : r.confidence;
    const float4 sg = g.segs[d - seg_base[lo]];
    la::M3 M;
    for (int i = 0; i < 9; ++i) M.m[i] = g.RtKinv[i];
    la::V3 P1, P2, dir;
    unproject_segment_f64(M, la::V3{ g.C[0], g.C[1], g.C[2] }, sg.x, sg.y, sg.z, sg.w, d1, d2, P1, P2, dir);
    const int h = hyp_of[d];
    Hypothesis o;
    o.P1[0] = P1.x; o.P1[1] = P1.y; o.P1[2] = P1.z; o.P2[0] = P2.x; o.P2[1] = P2.y; o.P2[2] = P2.z; o.dir[0] = dir.x; o.dir[1] = dir.y; o.dir[2] = dir.z;
    o.depth_p1 = d1; o.depth_p2 = d2; o.k_lower = g.k_lower; o.k_upper = g.k_upper; o.median_depth = g.median_depth; o.pad = 0;
    hyp[h] = o;
    score[h] = fminf(conf, 1.0f);                                 // line3D.cc:927
    hyp_dense[h] = d;
    best_hyp[d] = h;
}

// a view's best matches as records (inspection): the arena record, reversed where the reference read it reversed
__global__ __launch_bounds__(256) void k_prod_best_records(const Match* __restrict__ arena, const long long* __restrict__ best_ref, const int* __restrict__ seg_base, int nv,
                                                           const unsigned* __restrict__ early_cam, int nd, Match* __restrict__ out)
{
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= nd) return;
    const long long ref = best_ref[d];
    Match o;
    o.segID1 = 0xffffffffu; o.camID2 = 0; o.segID2 = 0; o.depths[0] = o.depths[1] = o.depths[2] = o.depths[3] = 0.0f; o.confidence = 0.0f;
    if (ref >= 0) {
        int lo = 0, hi = nv;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_base[mid] <= d) lo = mid; else hi = mid; }
        const Match r = arena[ref & kBestIndexMask];
        if (ref & kBestReversed) {
            o.segID1 = r.segID2; o.segID2 = r.segID1; o.camID2 = early_cam ? early_cam[d] : 0;
            o.depths[0] = r.depths[2]; o.depths[1] = r.depths[3]; o.depths[2] = r.depths[0]; o.depths[3] = r.depths[1];
        } else o = r;
    }
    out[d] = o;
}

}  // namespace l3d

extern "C" {

int l3d_products_hypotheses(l3d_ctx* c, const l3d_view_geometry* geometry, int n_views, int32_t* view_hyp_begin, int32_t** hyp_dense_out, int* n_hyp_out)
{
    if (!c) return L3D_ERR_INVALID;
    Products& P = c->products;
    if (!geometry || !view_hyp_begin || !hyp_dense_out || !n_hyp_out) return fail(c, L3D_ERR_INVALID, "l3d_products_hypotheses: bad argument");
    *hyp_dense_out = nullptr; *n_hyp_out = 0;
    if (!P.valid) return fail(c, L3D_ERR_INVALID, "l3d_products_hypotheses: no resident products (run l3d_match_chain_resident first)");
    if (n_views != P.n_views_all) return fail(c, L3D_ERR_INVALID, "l3d_products_hypotheses: geometry does not match the dense map");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int nd = P.n_dense, nv = n_views;
    std::vector<ViewGeo> g((size_t)nv);
    for (int i = 0; i < nv; ++i) {
        const l3d_view_geometry& s = geometry[i];
        const int S = P.seg_base[(size_t)i + 1] - P.seg_base[(size_t)i];
        if (s.n_segments != S) return fail(c, L3D_ERR_INVALID, "l3d_products_hypotheses: segment count differs from the dense map");
        ViewGeo& o = g[(size_t)i];
        memcpy(o.RtKinv, s.RtKinv, 72); memcpy(o.C, s.C, 24);
        o.k_lower = s.k_lower; o.k_upper = s.k_upper; o.median_depth = s.median_depth; o.pad = 0;
        o.segs = reinterpret_cast<const float4*>(resident_ptr(c, s.segments, (size_t)S * 16));
        if (!o.segs && S > 0) return fail(c, L3D_ERR_INVALID, "l3d_products_hypotheses: a view's segments are not registered (l3d_register_segments)");
    }
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_geo = 0, o_vhb = al((size_t)nv * sizeof(ViewGeo)), o_sb = o_vhb + al((size_t)(nv + 2) * 4);
    HIPCHK(c, P.geo.reserve(o_sb + al((size_t)(nv + 1) * 4)));
    char* gb = P.geo.as<char>();
    HIPCHK(c, hipMemcpyAsync(gb + o_geo, g.data(), (size_t)nv * sizeof(ViewGeo), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(gb + o_sb, P.seg_base.data(), (size_t)(nv + 1) * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, P.hyp_of.reserve(((size_t)nd + 2) * 8));                 // flags | exclusive sums
    int* flag = P.hyp_of.as<int>();
    int* hyp_of = flag + (nd + 2);
    HIPCHK(c, hipMemsetAsync(flag + nd, 0, 4, st));
    ProfScope p(c, "hypotheses", st);
    hipLaunchKernelGGL(k_hyp_flag, dim3((nd + 255) / 256), dim3(256), 0, st, P.best_ref.as<long long>(), nd, flag);
    size_t tb = 0;
    HIPCHK(c, exclusive_sum_int(nullptr, tb, flag, hyp_of, nd + 1, st));
    HIPCHK(c, P.tmp.reserve(tb + 256));
    HIPCHK(c, exclusive_sum_int(P.tmp.p, tb, flag, hyp_of, nd + 1, st));
    // (the number of hypotheses is at most the number of segments: the tables are sized for that, no round trip before the build)
    HIPCHK(c, c->aff_hyp.reserve((size_t)nd * sizeof(Hypothesis) + 64));
    HIPCHK(c, P.score.reserve((size_t)nd * 4 + 64));
    HIPCHK(c, P.hyp_dense.reserve((size_t)nd * 4 + 64));
    HIPCHK(c, P.best_hyp.reserve((size_t)nd * 4 + 64));
    hipLaunchKernelGGL(k_hyp_build, dim3((std::max(nd, nv + 1) + 255) / 256), dim3(256), 0, st, c->ch_kept.as<Match>(), P.best_ref.as<long long>(), hyp_of,
                       reinterpret_cast<const int*>(gb + o_sb), nv, reinterpret_cast<const ViewGeo*>(gb + o_geo), nd, c->aff_hyp.as<Hypothesis>(), P.score.as<float>(),
                       P.hyp_dense.as<int>(), P.best_hyp.as<int>(), reinterpret_cast<int*>(gb + o_vhb));
    HIPCHK(c, hipMemcpyAsync(view_hyp_begin, gb + o_vhb, (size_t)(nv + 1) * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const int nh = view_hyp_begin[nv];
    int32_t* hd = static_cast<int32_t*>(malloc((size_t)nh * 4 + 4));
    if (!hd) return fail(c, L3D_ERR_NOMEM, "malloc");
    if (nh > 0) {
        hipError_t e = hipMemcpyAsync(hd, P.hyp_dense.p, (size_t)nh * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { free(hd); return fail(c, L3D_ERR_HIP, std::string("l3d_products_hypotheses: ") + hipGetErrorString(e)); }
    }
    { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { free(hd); return fail(c, L3D_ERR_HIP, std::string("l3d_products_hypotheses: ") + hipGetErrorString(e_)); } }
    P.view_hyp_begin.assign(view_hyp_begin, view_hyp_begin + nv + 1);
    P.n_hyp = nh; P.hyp_valid = true;
    c->resident_hyp = nh;
    *hyp_dense_out = hd; *n_hyp_out = nh;
    return L3D_OK;
}

int l3d_products_hypotheses_get(l3d_ctx* c, l3d_hypothesis* hyp, float* score)
{
    if (!c) return L3D_ERR_INVALID;
    Products& P = c->products;
    if (!P.valid || !P.hyp_valid) return fail(c, L3D_ERR_INVALID, "no resident hypotheses");
    HIPCHK(c, hipSetDevice(c->device));
    if (hyp && P.n_hyp) HIPCHK(c, hipMemcpyAsync(hyp, c->aff_hyp.p, (size_t)P.n_hyp * sizeof(Hypothesis), hipMemcpyDeviceToHost, c->stream));
    if (score && P.n_hyp) HIPCHK(c, hipMemcpyAsync(score, P.score.p, (size_t)P.n_hyp * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return L3D_OK;
}

int l3d_chain_kept_list(l3d_ctx* c, int index, l3d_match** out, int* n)
{
    if (!c) return L3D_ERR_INVALID;
    Products& P = c->products;
    if (!out || !n) return fail(c, L3D_ERR_INVALID, "bad argument");
    *out = nullptr; *n = 0;
    if (!P.valid || index < 0 || index >= P.n_chain) return fail(c, L3D_ERR_INVALID, "l3d_chain_kept_list: no such view in the resident products");
    HIPCHK(c, hipSetDevice(c->device));
    auto fetch = [&](int k, std::vector<l3d_match>& v) -> int {
        const ChainResult& r = P.res[(size_t)k];
        v.resize((size_t)r.n_kept);
        if (r.n_kept) HIPCHK(c, hipMemcpy(v.data(), c->ch_kept.as<Match>() + r.kept_base, (size_t)r.n_kept * sizeof(Match), hipMemcpyDeviceToHost));
        return L3D_OK;
    };
    std::vector<l3d_match> lst;
    if (P.chain_verified[(size_t)index]) { if (int rc = fetch(index, lst)) return rc; }
    if (!P.chain_verified[(size_t)index]) {
        // cudawrapper.cu:877-878: the localized existing list -- what the earlier views pushed (line3D.cc:838-872), in push order
        lst.clear();
        std::vector<l3d_match> src;
        const unsigned vid = P.chain_view_id[(size_t)index];
        for (size_t q = 0; q < P.early_src_index[(size_t)index].size(); ++q) {
            if (int rc = fetch(P.early_src_index[(size_t)index][q], src)) return rc;
            for (const l3d_match& mp : src) {
                if (mp.camID2 != vid) continue;
                l3d_match r;
                r.segID1 = mp.segID2; r.segID2 = mp.segID1; r.confidence = 0.0f; r.camID2 = (uint32_t)P.early_src_cam[(size_t)index][q];
                r.depths[0] = mp.depths[2]; r.depths[1] = mp.depths[3]; r.depths[2] = mp.depths[0]; r.depths[3] = mp.depths[1];
                lst.push_back(r);
            }
        }
    }
    l3d_match* o = static_cast<l3d_match*>(malloc(lst.size() * sizeof(l3d_match) + 32));
    if (!o) return fail(c, L3D_ERR_NOMEM, "malloc");
    if (!lst.empty()) memcpy(o, lst.data(), lst.size() * sizeof(l3d_match));
    *out = o; *n = (int)lst.size();
    return L3D_OK;
}

int l3d_chain_products_get(l3d_ctx* c, int64_t* pot_start, int32_t* pot_tgt, l3d_match* best_match)
{
    if (!c) return L3D_ERR_INVALID;
    Products& P = c->products;
    if (!P.valid) return fail(c, L3D_ERR_INVALID, "no resident products");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    if (pot_start) HIPCHK(c, hipMemcpyAsync(pot_start, P.pot_start.p, ((size_t)P.n_dense + 1) * 8, hipMemcpyDeviceToHost, st));
    if (pot_tgt && P.n_pot) HIPCHK(c, hipMemcpyAsync(pot_tgt, P.pot_tgt.p, (size_t)P.n_pot * 4, hipMemcpyDeviceToHost, st));
    if (best_match) {
        // the LOCAL camera number of an early-return view's entry: the rank of its source, per dense id
        std::vector<unsigned> cam((size_t)P.n_dense, 0u);
        bool any_early = false;
        std::vector<long long> ref((size_t)P.n_dense);
        HIPCHK(c, hipMemcpyAsync(ref.data(), P.best_ref.p, (size_t)P.n_dense * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        for (int k = 0; k < P.n_chain; ++k) {
            if (P.chain_verified[(size_t)k]) continue;
            any_early = true;
            const int vi = P.chain_view[(size_t)k];
            // which source a reversed reference points into: the one whose slice holds the record
            for (int d = P.seg_base[(size_t)vi]; d < P.seg_base[(size_t)vi + 1]; ++d) {
                if (ref[(size_t)d] < 0) continue;
                const long long idx = ref[(size_t)d] & kBestIndexMask;
                for (size_t q = 0; q < P.early_src_index[(size_t)k].size(); ++q) {
                    const ChainResult& r = P.res[(size_t)P.early_src_index[(size_t)k][q]];
                    if (idx >= r.kept_base && idx < (long long)r.kept_base + r.n_kept) { cam[(size_t)d] = (unsigned)P.early_src_cam[(size_t)k][q]; break; }
                }
            }
        }
        HIPCHK(c, P.keys.reserve((size_t)P.n_dense * (sizeof(Match) + 4) + 256));       // (scratch: the keys are consumed)
        Match* dout = P.keys.as<Match>();
        unsigned* dcam = reinterpret_cast<unsigned*>(P.keys.as<char>() + (((size_t)P.n_dense * sizeof(Match) + 255) & ~(size_t)255));
        if (any_early) HIPCHK(c, hipMemcpyAsync(dcam, cam.data(), (size_t)P.n_dense * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, P.flag.reserve(((size_t)P.n_views_all + 2) * 4 + 256));      // (scratch: the flags are consumed)
        int* dsb = P.flag.as<int>();
        HIPCHK(c, hipMemcpyAsync(dsb, P.seg_base.data(), ((size_t)P.n_views_all + 1) * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_prod_best_records, dim3((P.n_dense + 255) / 256), dim3(256), 0, st, c->ch_kept.as<Match>(), P.best_ref.as<long long>(), dsb, P.n_views_all,
                           any_early ? dcam : nullptr, P.n_dense, dout);
        HIPCHK(c, hipMemcpyAsync(best_match, dout, (size_t)P.n_dense * sizeof(Match), hipMemcpyDeviceToHost, st));
    }
    HIPCHK(c, hipStreamSynchronize(st));
    return L3D_OK;
}

}  // extern "C"

void l3d::warm_products() { touch_kernel(reinterpret_cast<const void*>(&k_prod_flags)); }
