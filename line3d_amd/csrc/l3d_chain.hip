// l3d_chain.hip -- Line3D::matchViews (line3D.cc:620-648) as ONE device-resident chain.
//
// The reference processes views strictly one after the other and crosses the host<->device boundary several
// times per view (uploads, a dense download per neighbour, host sort, download of confidences, and the
// verified matches of a view travel through the host -- a file! -- to become candidates of later views,
// line3D.cc:838-872, view.cc:162-224).  Here the whole schedule is static: which neighbours a view still has to
// match (toBeMatched) and which earlier views feed it with reverse matches depends only on the neighbour graph
// and the processing order, not on data.  So the host enqueues everything without ever waiting:
//
//   phase 1  stage 1 (pair test -> bit rows -> row counts) of ALL views: independent, back to back
//   phase 2  per view, in order: reverse matches are pulled on the device out of the kept lists of the earlier
//            views (they never leave HBM), prefix sums, depth records, verification, per-segment best/filter,
//            ordered compaction of the kept matches into one arena
//
// and only trails behind the GPU to hand each view's kept list to the caller's bookkeeping (a callback), which
// overlaps with the GPU working on later views.  Results are identical to the per-view entry point
// (l3d_compute_pairwise_matches): same kernels, same candidate order.
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "l3d_ctx.hpp"
#include "l3d_scan.hpp"
#include "l3d_kept.hpp"
#include "l3d_products.hpp"
#include "l3d_chain_common.hpp"
#include "l3d_runtable.hpp"

#ifndef L3D_AHEAD
#define L3D_AHEAD 4
#define L3D_S1AHEAD 8
#endif

using namespace l3d;

namespace l3d {

// reverse matches for view `view_id` out of the kept lists of earlier views (blockIdx.y = source): count per
// (segment, camera) row.  (seg, tgt) swap roles and the depth pairs swap, line3D.cc:847-856.
constexpr int kCamScanMin = 262144;     // records of a source's list from which the side array is scanned first
// cams (round 5): the target camera of every record of the arena, written beside it by the kept writer -- a later view reads 4 bytes per record
// of its sources' lists and the 32-byte record only where it points at that view (one in N): at 4000 segments x 24 neighbours the two scans of a
// view's sources were 2.9 GB of reads per view
__global__ void k_exist_count(const Match* __restrict__ arena, const unsigned* __restrict__ cams, const ChainResult* __restrict__ res, const int* __restrict__ src_index,
                              const int* __restrict__ src_cam, unsigned view_id, int N, int S, int* __restrict__ rowcnt)
{
    const ChainResult* src = res + src_index[blockIdx.y];
    const int cam = src_cam[blockIdx.y];
    const int n = src->n_kept;
    const Match* kept = arena + src->kept_base;
    // (short lists -- config 2 keeps 36 k matches per view, 1.2 MB -- are scanned record by record: one pass over cache-resident data beats two
    // dependent ones; measured 12.10 vs 12.17-12.29 ms per config-2 pass, 173.2 vs 165.4 ms at 40 x 4000 x 24, profiles/r5_ab_kept_cams.txt)
    const unsigned* kc = cams && n > kCamScanMin ? cams + src->kept_base : nullptr;
    const int stride = gridDim.x * blockDim.x;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (kc) {
        // the side array four entries at a time: one 4-byte load in flight per thread was 1.1 TB/s of a scan that has nothing else to wait for
        for (; i + 3 * stride < n; i += 4 * stride) {
            const unsigned c0 = kc[i], c1 = kc[i + stride], c2 = kc[i + 2 * stride], c3 = kc[i + 3 * stride];
            if (c0 == view_id) { const Match r = kept[i]; if ((int)r.segID2 < S) atomicAdd(&rowcnt[r.segID2 * N + cam], 1); }
            if (c1 == view_id) { const Match r = kept[i + stride]; if ((int)r.segID2 < S) atomicAdd(&rowcnt[r.segID2 * N + cam], 1); }
            if (c2 == view_id) { const Match r = kept[i + 2 * stride]; if ((int)r.segID2 < S) atomicAdd(&rowcnt[r.segID2 * N + cam], 1); }
            if (c3 == view_id) { const Match r = kept[i + 3 * stride]; if ((int)r.segID2 < S) atomicAdd(&rowcnt[r.segID2 * N + cam], 1); }
        }
    }
    for (; i < n; i += stride) {
        if (kc && kc[i] != view_id) continue;
        const Match r = kept[i];
        if (r.camID2 == view_id && (int)r.segID2 < S) atomicAdd(&rowcnt[r.segID2 * N + cam], 1);
    }
}
// Round 6, run tables (l3d_runtable.hpp): the records of source w that point at this view are the runs rt_w[slot][s] .. rt_w[slot + 1][s] of its list
// (slot = this view's local camera number in w's neighbour list) -- read directly, through the packed side array (4 bytes per record: the target
// segment is all the count needs), instead of scanning the source's whole list for them: 1/N of it is touched, twelve sources' lists are not.
// g (a power of two) lanes share a run.
// (round 6, late) NO GLOBAL ATOMICS: 2 M scattered atomicAdds per view -- every one on its own 64-byte line, the rows being N x 4 bytes apart -- ran at
// 14 G/s whatever fed them (144 us per view at 40 x 4000 x 24, with or without the run tables).  A source's segments are cut into kExistChunks chunks; a
// workgroup per (chunk, source) counts its runs' targets in LDS and leaves its S counters in `part`; k_exist_combine turns every (source, target segment)'s
// chunk counts into chunk BASES (exclusive sums, in place) and stores the row count; k_place_rt's scatter workgroups -- one per (chunk, source) again -- start
// their LDS cursors at row start + chunk base and place their records without a global cursor.
constexpr int kExistChunks = 16;
constexpr int kExistThreads = 1024;         // (sixteen waves per (chunk, source): a workgroup's critical path is its runs / waves dependent load pairs)
__global__ __launch_bounds__(kExistThreads) void k_exist_count_rt(const unsigned* __restrict__ qt_arena, const RtInfo* __restrict__ info, const ChainResult* __restrict__ res,
                                                        const int* __restrict__ src_index, const int* __restrict__ src_slot, int g, int S, int* __restrict__ part)
{
    extern __shared__ int s_hist[];
    const int j = blockIdx.y, ch = blockIdx.x;
    const int si = src_index[j], slot = src_slot[j];
    const RtInfo w = info[si];
    int* out = part + ((size_t)j * kExistChunks + ch) * S;
    for (int u = threadIdx.x; u < S; u += kExistThreads) s_hist[u] = 0;
    __syncthreads();
    if (w.rt && slot >= 0 && res[si].n_kept > 0) {
        const unsigned* qt = qt_arena + res[si].kept_base;
        const int* r0 = w.rt + (size_t)slot * w.S;
        const int* r1 = r0 + w.S;
        const int s_lo = (int)(((long long)w.S * ch) / kExistChunks), s_hi = (int)(((long long)w.S * (ch + 1)) / kExistChunks);
        const int grp = threadIdx.x / g, gl = threadIdx.x - grp * g, ngrp = kExistThreads / g;
        for (int s = s_lo + grp; s < s_hi; s += ngrp) {
            const int a = r0[s], b = r1[s];
            for (int i = a + gl; i < b; i += g) { const int u = (int)(qt[i] & 0xffffu); if (u < S) atomicAdd(&s_hist[u], 1); }
        }
    }
    __syncthreads();
    for (int u = threadIdx.x; u < S; u += kExistThreads) out[u] = s_hist[u];
}
__global__ __launch_bounds__(256) void k_exist_combine(int* __restrict__ part, const int* __restrict__ src_cam, int N, int S, int* __restrict__ rowcnt)
{
    const int j = blockIdx.y, u = blockIdx.x * 256 + threadIdx.x;
    if (u >= S) return;
    int* p = part + (size_t)j * kExistChunks * S + u;
    int run = 0;
#pragma unroll
    for (int ch = 0; ch < kExistChunks; ++ch) { const int t = p[(size_t)ch * S]; p[(size_t)ch * S] = run; run += t; }
    if (run) atomicAdd(&rowcnt[u * N + src_cam[j]], run);           // (one thread per cell: the atomic only keeps the add whole beside stage 1's rows of other cameras)
}
// the side array of records that did not come from the kept writer (a block's sources taken over from another rank)
__global__ void k_cams_of_records(const Match* __restrict__ arena, long long n, unsigned* __restrict__ cams)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cams[i] = arena[i].camID2;
}

__global__ void k_exist_scatter(const Match* __restrict__ arena, const ChainResult* __restrict__ res, const int* __restrict__ src_index,
                                const int* __restrict__ src_cam, unsigned view_id, int N, int S, const int* __restrict__ row_start,
                                int* __restrict__ cursor, uint2* __restrict__ meta, float4* __restrict__ depths, int cap)
{
    if (row_start[(size_t)S * N] > cap) return;
    const ChainResult* src = res + src_index[blockIdx.y];
    const int cam = src_cam[blockIdx.y];
    const int n = src->n_kept;
    const Match* kept = arena + src->kept_base;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Match r = kept[i];
        if (r.camID2 == view_id && (int)r.segID2 < S) {
            const int row = r.segID2 * N + cam;
            const int slot = row_start[row] + atomicAdd(&cursor[row], 1);
            meta[slot] = make_uint2(r.segID1, (unsigned)cam);
            depths[slot] = make_float4(r.depths[2], r.depths[3], r.depths[0], r.depths[1]);
        }
    }
}

// The scatter order inside a (segment, camera) run is arbitrary.  One wave per run restores the (segment, camera,
// target) order of the reference's list sort: every lane holds up to four entries in registers, ranks them by
// counting (keys are broadcast with shuffles, target ids inside a run are distinct) and writes them to their place.
__global__ __launch_bounds__(256) void k_exist_sort_runs(const int* __restrict__ cams, int n_cams, int N, int S, int seg_begin, int seg_end,
                                                         const int* __restrict__ row_start, uint2* __restrict__ meta,
                                                         float4* __restrict__ depths, int cap, float* stage, long long stage_stride, unsigned* stage_key)
{
    if (row_start[(size_t)S * N] > cap) return;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= (seg_end - seg_begin) * n_cams) return;
    const int seg = seg_begin + t / n_cams, cam = cams[t % n_cams];
    const int b = row_start[seg * N + cam], n = row_start[seg * N + cam + 1] - b;
    sort_exist_run(lane, b, n, cam, meta, depths, stage, stage_stride, stage_key);      // (stage: the verification's scratch + confidence slots, unused until it runs)
}

// Stage-1 candidates of a view are written (k_pair_fill, stage-1 stream, well ahead of the chain) in their own
// (segment, to-be-matched camera) row order; once the reverse matches of the view are counted, each row is moved to its
// place in the combined (segment, camera, target) order -- a 24-byte copy per candidate instead of the triangulation on the
// chain's critical path.  One wave per row (the first workgroups of k_place).
// Both writers of the combined candidate arrays in one launch (independent: stage-1 candidates go to the rows of the cameras
// to be matched, reverse matches to the rows of the source cameras): the first `blocks_move` workgroups move the stage-1 rows, the
// others scatter the reverse matches (as k_exist_scatter, 32 workgroups per source view).
__global__ __launch_bounds__(256) void k_place(int blocks_move, int bps, const int* __restrict__ tbm, int n_tbm, const int* __restrict__ rowA,
                                               const uint2* __restrict__ metaA, const float4* __restrict__ depthsA,
                                               const Match* __restrict__ arena, const unsigned* __restrict__ cams, const ChainResult* __restrict__ res, const int* __restrict__ src_index,
                                               const int* __restrict__ src_cam, unsigned view_id, int N, int S,
                                               const int* __restrict__ row_start, int* __restrict__ cursor,
                                               uint2* __restrict__ meta, float4* __restrict__ depths, int cap)
{
    if (row_start[(size_t)S * N] > cap) return;                  // overflow: the chain is re-run with more room
    if ((int)blockIdx.x < blocks_move) {
        const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (row >= S * n_tbm) return;
        const int y = row / n_tbm, cam = tbm[row % n_tbm];
        const int a = rowA[y * N + cam], b = row_start[y * N + cam], n = row_start[y * N + cam + 1] - b;   // (rowA may have been laid out for an upper bound of the row)
        for (int j = lane; j < n; j += 64) { meta[b + j] = metaA[a + j]; depths[b + j] = depthsA[a + j]; }
        return;
    }
    const int e = (int)blockIdx.x - blocks_move, si = e / bps, bx = e % bps;         // bps workgroups per source view
    const ChainResult* src = res + src_index[si];
    const int cam = src_cam[si];
    const int n = src->n_kept;
    const Match* kept = arena + src->kept_base;
    const unsigned* kc = cams && n > kCamScanMin ? cams + src->kept_base : nullptr;
    auto place = [&](int i) {
        const Match r = kept[i];
        if (r.camID2 == view_id && (int)r.segID2 < S) {
            const int row = r.segID2 * N + cam;
            const int slot = row_start[row] + atomicAdd(&cursor[row], 1);
            meta[slot] = make_uint2(r.segID1, (unsigned)cam);
            depths[slot] = make_float4(r.depths[2], r.depths[3], r.depths[0], r.depths[1]);
        }
    };
    const int stride = bps * 256;
    int i = bx * 256 + (int)threadIdx.x;
    if (kc) {                                                    // (four entries of the side array in flight per thread: k_exist_count)
        for (; i + 3 * stride < n; i += 4 * stride) {
            const unsigned c0 = kc[i], c1 = kc[i + stride], c2 = kc[i + 2 * stride], c3 = kc[i + 3 * stride];
            if (c0 == view_id) place(i);
            if (c1 == view_id) place(i + stride);
            if (c2 == view_id) place(i + 2 * stride);
            if (c3 == view_id) place(i + 3 * stride);
        }
    }
    for (; i < n; i += stride) {
        if (kc && kc[i] != view_id) continue;
        place(i);
    }
}
// k_place with run tables (default): the stage-1 rows are moved as in k_place, the reverse matches come from the sources' runs towards this view through the
// global row cursors -- 6000 small workgroups in one launch with the move: faster than 192 (or 768) big ones with LDS cursors in a launch of their own
// (cand_move 12.0 against 17.1 / 13.2 ms at 40 x 4000 x 24, NOTEBOOK 12.f)
__global__ __launch_bounds__(256) void k_place_rt(int blocks_move, int bps, const int* __restrict__ tbm, int n_tbm, const int* __restrict__ rowA,
                                                  const uint2* __restrict__ metaA, const float4* __restrict__ depthsA,
                                                  const Match* __restrict__ arena, const ChainResult* __restrict__ res, const int* __restrict__ src_index,
                                                  const int* __restrict__ src_cam, int N, int S, const int* __restrict__ row_start, int* __restrict__ cursor,
                                                  uint2* __restrict__ meta, float4* __restrict__ depths, int cap,
                                                  const RtInfo* __restrict__ info, const int* __restrict__ src_slot, int g)
{
    if (row_start[(size_t)S * N] > cap) return;
    if ((int)blockIdx.x < blocks_move) {
        const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (row >= S * n_tbm) return;
        const int y = row / n_tbm, cam = tbm[row % n_tbm];
        const int a = rowA[y * N + cam], b = row_start[y * N + cam], n = row_start[y * N + cam + 1] - b;
        for (int j = lane; j < n; j += 64) { meta[b + j] = metaA[a + j]; depths[b + j] = depthsA[a + j]; }
        return;
    }
    const int e = (int)blockIdx.x - blocks_move, sj = e / bps, bx = e % bps;
    const int si = src_index[sj], cam = src_cam[sj], slot = src_slot[sj];
    const RtInfo w = info[si];
    if (!w.rt || slot < 0 || res[si].n_kept == 0) return;
    const Match* kept = arena + res[si].kept_base;
    const int* r0 = w.rt + (size_t)slot * w.S;
    const int* r1 = r0 + w.S;
    const int grp = threadIdx.x / g, gl = threadIdx.x - grp * g, ngrp = 256 / g;
    for (int sg = bx * ngrp + grp; sg < w.S; sg += bps * ngrp) {
        const int a = r0[sg], b = r1[sg];
        for (int i = a + gl; i < b; i += g) {
            const Match r = kept[i];
            if ((int)r.segID2 < S) {
                const int row = r.segID2 * N + cam;
                const int sl = row_start[row] + atomicAdd(&cursor[row], 1);
                meta[sl] = make_uint2(r.segID1, (unsigned)cam);
                depths[sl] = make_float4(r.depths[2], r.depths[3], r.depths[0], r.depths[1]);
            }
        }
    }
}
// (A/B, option rt_place_lds) k_place with run tables: the stage-1 rows are moved by k_place itself (launched without sources); the reverse matches by a launch of their own -- one workgroup
// of kExistThreads per (chunk, source), LDS cursors from the row starts and the chunk bases of k_exist_combine: no global cursor, no global atomic
__global__ __launch_bounds__(kExistThreads) void k_place_scatter_rt(const Match* __restrict__ arena, const ChainResult* __restrict__ res, const int* __restrict__ src_index,
                                                                    const int* __restrict__ src_cam, int N, int S, const int* __restrict__ row_start,
                                                                    uint2* __restrict__ meta, float4* __restrict__ depths, int cap,
                                                                    const RtInfo* __restrict__ info, const int* __restrict__ src_slot, int g, const int* __restrict__ part)
{
    extern __shared__ int s_cur[];
    if (row_start[(size_t)S * N] > cap) return;
    const int sj = blockIdx.y, bx = blockIdx.x;
    const int si = src_index[sj], cam = src_cam[sj], slot = src_slot[sj];
    const RtInfo w = info[si];
    if (!w.rt || slot < 0 || res[si].n_kept == 0) return;
    const int* base = part + ((size_t)sj * kExistChunks + bx) * S;
    for (int u = threadIdx.x; u < S; u += kExistThreads) s_cur[u] = row_start[u * N + cam] + base[u];
    __syncthreads();
    const Match* kept = arena + res[si].kept_base;
    const int* r0 = w.rt + (size_t)slot * w.S;
    const int* r1 = r0 + w.S;
    const int s_lo = (int)(((long long)w.S * bx) / kExistChunks), s_hi = (int)(((long long)w.S * (bx + 1)) / kExistChunks);
    const int grp = threadIdx.x / g, gl = threadIdx.x - grp * g, ngrp = kExistThreads / g;
    for (int sg = s_lo + grp; sg < s_hi; sg += ngrp) {
        const int a = r0[sg], b = r1[sg];
        for (int i = a + gl; i < b; i += g) {
            const Match r = kept[i];
            if ((int)r.segID2 < S) {
                const int sl = atomicAdd(&s_cur[r.segID2], 1);
                meta[sl] = make_uint2(r.segID1, (unsigned)cam);
                depths[sl] = make_float4(r.depths[2], r.depths[3], r.depths[0], r.depths[1]);
            }
        }
    }
}

// raw candidate total and the largest per-segment count of one view's segment range (phase 1 statistics), by one
// workgroup; out2 = {total, max} lives in host-mapped pinned memory: no copy, the host reads it after the stage-1 event
__global__ __launch_bounds__(1024) void k_raw_stats(const int* __restrict__ rowcnt, int N, int seg_begin, int seg_end, int* __restrict__ out2)
{
    __shared__ int s_t[16], s_m[16];
    int tot = 0, mx = 0;
    for (int s = seg_begin + (int)threadIdx.x; s < seg_end; s += 1024) {
        int c = 0;
        for (int k = 0; k < N; ++k) c += rowcnt[s * N + k];
        tot += c; mx = max(mx, c);
    }
    for (int o = 32; o > 0; o >>= 1) { tot += __shfl_down(tot, o); mx = max(mx, __shfl_down(mx, o)); }
    if ((threadIdx.x & 63) == 0) { s_t[threadIdx.x >> 6] = tot; s_m[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) { tot += s_t[w]; mx = max(mx, s_m[w]); }
        out2[0] = tot; out2[1] = mx;
    }
}

// Kept records of a view into its slice of the arena, in ONE launch behind the verification: each workgroup (one segment) sums
// the kept counts in front of its segment itself (at most a few thousand ints out of L2) instead of waiting for a scan
// launch, and the slice starts where the previous verified view's ended (its result record) -- no cursor, no atomics.
// Workgroup 0 also writes the view's result record (device copy for later views, host-mapped copy for the host).
__global__ __launch_bounds__(256) void k_kept_write_chain(VerifyArgs a, const int* __restrict__ kept_cnt, int nrow, const ChainResult* __restrict__ prev,
                                                          unsigned long long arena_cap, ChainResult* __restrict__ res, ChainResult* __restrict__ res_host,
                                                          const unsigned* __restrict__ local2global, Match* __restrict__ arena, int* __restrict__ best_pos, unsigned* __restrict__ cams,
                                                          int* __restrict__ rt, int rt_stride)
{
    __shared__ int s_red[8];
    __shared__ int s_cnt[32];
    __shared__ int s_qcnt[256];
    __shared__ unsigned long long s_best[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nseg = a.seg_end - a.seg_begin;
    const int yl = blockIdx.x;
    int before = 0, total = 0;
    for (int i = tid; i < nseg; i += 256) { const int v = kept_cnt[a.seg_begin + i]; total += v; if (i < yl) before += v; }
    for (int o = 32; o > 0; o >>= 1) { before += __shfl_down(before, o); total += __shfl_down(total, o); }
    if (lane == 0) { s_red[wave] = before; s_red[4 + wave] = total; }
    __syncthreads();
    before = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    total = s_red[4] + s_red[5] + s_red[6] + s_red[7];
    ChainResult r;
    r.R = a.row_start[nrow];
    r.overflow = r.R > a.cand_cap ? 1 : 0;
    r.n_kept = r.overflow ? 0 : total;
    r.kept_base = prev ? prev->kept_base + prev->n_kept : 0;
    if ((unsigned long long)r.kept_base + (unsigned long long)r.n_kept > arena_cap) { r.overflow |= 2; r.n_kept = 0; }
    if (blockIdx.x == 0 && tid == 0) { *res = r; *res_host = r; }
    if (yl >= nseg || r.overflow) return;
    write_kept_segment_wg(a, a.seg_begin + yl, before, local2global, arena + r.kept_base, s_cnt, best_pos ? best_pos + a.seg_begin + yl : nullptr, s_best, cams ? cams + r.kept_base : nullptr,
                          rt, rt_stride, s_qcnt);
}

void launch_exist_count(const Match* arena, const ChainResult* res, const int* src_index, const int* src_cam, int n_src, unsigned view_id,
                        int N, int S, int* rowcnt, hipStream_t st, const unsigned* cams, int bps)
{
    if (n_src > 0) hipLaunchKernelGGL(k_exist_count, dim3(std::max(1, bps), n_src), dim3(256), 0, st, arena, cams, res, src_index, src_cam, view_id, N, S, rowcnt);
}
void launch_exist_scatter(const Match* arena, const ChainResult* res, const int* src_index, const int* src_cam, int n_src, unsigned view_id,
                          int N, int S, const int* row_start, int* cursor, uint2* meta, float4* depths, int cap, hipStream_t st)
{
    if (n_src > 0) hipLaunchKernelGGL(k_exist_scatter, dim3(32, n_src), dim3(256), 0, st, arena, res, src_index, src_cam, view_id, N, S, row_start, cursor, meta, depths, cap);
}
void launch_exist_sort_runs(const int* cams, int n_cams, int N, int S, const int* row_start, uint2* meta, float4* depths, int cap, hipStream_t st,
                            int seg_begin, int seg_end, float* stage, long long stage_stride, unsigned* stage_key)
{
    if (seg_end < 0) seg_end = S;
    const int runs = (seg_end - seg_begin) * n_cams;
    if (runs > 0) hipLaunchKernelGGL(k_exist_sort_runs, dim3((runs + 3) / 4), dim3(256), 0, st, cams, n_cams, N, S, seg_begin, seg_end, row_start, meta, depths, cap, stage, stage_stride, stage_key);
}
void launch_place(const int* tbm, int n_tbm, int N, int S, const int* rowA, const uint2* metaA, const float4* depthsA,
                  const Match* arena, const ChainResult* res, const int* src_index, const int* src_cam, int n_src, unsigned view_id,
                  const int* row_start, int* cursor, int cand_cap, uint2* meta, float4* depths, hipStream_t st, const unsigned* cams, int bps,
                  const RtInfo* info, const int* src_slot, int g, const int* part)
{
    bps = std::max(1, bps);
    const int blocks_move = (S * n_tbm + 3) / 4;
    const int blocks = blocks_move + bps * n_src;
    if (info && part) {     // (option rt_place_lds, A/B) the move by k_place without sources, the scatter by kExistChunks workgroups per source with LDS cursors of S ints
        if (blocks_move > 0) hipLaunchKernelGGL(k_place, dim3(blocks_move), dim3(256), 0, st, blocks_move, 1, tbm, n_tbm, rowA, metaA, depthsA, arena, (const unsigned*)nullptr, res, src_index, src_cam,
                                                view_id, N, S, row_start, cursor, meta, depths, cand_cap);
        if (n_src > 0 && S > 0) hipLaunchKernelGGL(k_place_scatter_rt, dim3(kExistChunks, n_src), dim3(kExistThreads), (size_t)S * 4, st, arena, res, src_index, src_cam, N, S, row_start, meta, depths,
                                                   cand_cap, info, src_slot, std::max(1, g), part);
    }
    else if (info) { if (blocks > 0) hipLaunchKernelGGL(k_place_rt, dim3(blocks), dim3(256), 0, st, blocks_move, bps, tbm, n_tbm, rowA, metaA, depthsA, arena, res, src_index, src_cam,
                                                         N, S, row_start, cursor, meta, depths, cand_cap, info, src_slot, std::max(1, g)); }
    else if (blocks > 0) hipLaunchKernelGGL(k_place, dim3(blocks), dim3(256), 0, st, blocks_move, bps, tbm, n_tbm, rowA, metaA, depthsA, arena, cams, res, src_index, src_cam,
                                            view_id, N, S, row_start, cursor, meta, depths, cand_cap);
}
void launch_exist_count_rt(const unsigned* qt_arena, const RtInfo* info, const ChainResult* res, const int* src_index, const int* src_cam, const int* src_slot, int n_src, int g,
                           int N, int S, int* rowcnt, int* part, hipStream_t st)
{
    if (n_src <= 0 || S <= 0) return;
    hipLaunchKernelGGL(k_exist_count_rt, dim3(kExistChunks, n_src), dim3(kExistThreads), (size_t)S * 4, st, qt_arena, info, res, src_index, src_slot, std::max(1, g), S, part);
    hipLaunchKernelGGL(k_exist_combine, dim3((S + 255) / 256, n_src), dim3(256), 0, st, part, src_cam, N, S, rowcnt);
}
int exist_chunks() { return kExistChunks; }
void launch_raw_stats(const int* rowcnt, int N, int seg_begin, int seg_end, int* out2_host, hipStream_t st)
{
    hipLaunchKernelGGL(k_raw_stats, dim3(1), dim3(1024), 0, st, rowcnt, N, seg_begin, seg_end, out2_host);
}
void launch_kept_write_chain(const VerifyArgs& a, const int* kept_cnt, int nrow, const ChainResult* prev, unsigned long long arena_cap, ChainResult* res,
                             ChainResult* res_host, const unsigned* l2g, Match* arena, hipStream_t st, int* best_pos, unsigned* cams, int* rt, int rt_stride)
{
    hipLaunchKernelGGL(k_kept_write_chain, dim3(std::max(1, a.seg_end - a.seg_begin)), dim3(256), 0, st, a, kept_cnt, nrow, prev, arena_cap, res, res_host, l2g, arena, best_pos, cams, rt, rt_stride);
}

}  // namespace l3d

namespace {

typedef l3d::ChainViewDev ViewDev;

size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

}  // namespace

// cb: per-view delivery of the kept lists to the host (l3d_match_chain); map: products built on the device at the end of the chain
// (l3d_match_chain_resident) -- either or both
// [k_begin, k_end): the views this call computes (k_end < 0: all).  Views outside the range are treated as if they had never run: they
// launch nothing and their result records stay zero, so a view inside the range finds no kept matches of a source in front of it
// (l3d_match_chain_blocks: a block of views started cold).  With a range and neither cb nor map the call only fills the kept arena and
// the per-view result records (c->ch_pin_res).
// pre: the views [pre->k0, pre->k1 = k_begin) taken over from another rank (their kept lists, best depth pairs and positions): they are put at
// the head of the arena with their result records, so the range's views find their TRUE sources -- a block of views re-run warm after its
// cold-started speculation failed.
namespace {
struct ChainPreload {
    int k0 = 0, k1 = 0;
    const Match* records = nullptr;        // device: the views' kept lists, back to back
    const float2* best = nullptr;          // device: best depth pairs of the verified views among them, back to back (S_src each)
    const int* bestpos = nullptr;          // device: ... and the positions of the best kept matches
    std::vector<int> n_kept, R;            // per view
};
}
static int run_chain(l3d_ctx* c, const l3d_chain_view* views, int n_views, l3d_chain_callback cb, void* user, const l3d_dense_map* map,
                     l3d_chain_summary* summary, int64_t* n_pot, int k_begin = 0, int k_end = -1, const ChainPreload* pre = nullptr)
{
    if (!c) return L3D_ERR_INVALID;
    const bool ranged = k_end >= 0;
    if (!ranged) { k_begin = 0; k_end = n_views; }
    if (n_views < 0 || (n_views > 0 && (!views || (!cb && !map && !ranged))) || k_begin < 0 || k_end > n_views || k_begin > k_end) return fail(c, L3D_ERR_INVALID, "l3d_match_chain: bad argument");
    c->products.valid = false;
    if (n_views == 0) return L3D_OK;
    const double t_enter = now_s();
    HIPCHK(c, hipSetDevice(c->device));
    const double t_setup0 = now_s();
    c->pin_arena.reset();
    hipStream_t st = c->stream;         // phase 2 (the chain proper)
    hipStream_t s1 = c->stage1_stream;  // stage 1 runs ahead here, concurrently with the latency-bound kernels of phase 2
    const bool serial = c->opt.chain_serial != 0;   // diagnostic: one stream, kernels one at a time (isolated durations)
    if (serial) s1 = st;
    // option mask_stream: k_pair_mask of view k+1 next to k_pair_fill of view k instead of behind it
    hipStream_t sm = s1;
    if (c->opt.mask_stream && !serial) {
        if (!c->mask_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->mask_stream, hipStreamNonBlocking));
        sm = c->mask_stream;
    }
    (void)hipGetLastError();            // errors of earlier, already reported calls are not ours

    // ---- validation, table layout and upload, per-view slices of the whole-run arenas (l3d_chain_common.hip: shared with the sharded chain)
    std::vector<ViewDev> vd;
    ChainLayout L;
    if (int rc = chain_plan_views(c, views, n_views, 0, 1, vd, L, "l3d_match_chain")) return rc;
    const bool rays_env = c->opt.tgt_rays != 0;      // (0: k_pair_fill normalises per candidate, A/B)
    if (int rc = chain_upload_tables(c, views, n_views, vd, L, rays_env, st)) return rc;
    // bit rows: a ring that covers every view between the one being collected and the newest stage 1 (kRing below: after a capacity
    // overflow the candidates of all of them are re-formed from their bit rows)
    // run tables (round 6): the kept writer fills one per view and packs (local camera, target) into the side array; later views and the products
    // read runs instead of scanning lists.  L3D_RUN_TABLES=0 / L3D_KEPT_CAMS=0: the A/B paths (side array of global camera ids / none)
    const bool use_rt = c->opt.run_tables != 0 && c->opt.kept_cams != 0 && c->chain_ring != 0;
    if (int rc = chain_assign_arenas(c, views, n_views, vd, L, true, true, c->chain_ring != 0 ? L3D_AHEAD + L3D_S1AHEAD + 3 : 0, st, use_rt)) return rc;
    if (use_rt) {
        std::vector<RtInfo>& info = c->rtinfo_host;     // (lives in the context: the upload is asynchronous)
        info.resize((size_t)n_views);
        for (int k = 0; k < n_views; ++k) info[(size_t)k] = RtInfo{ vd[(size_t)k].rt, views[k].S_src, views[k].N };
        HIPCHK(c, c->ch_rtinfo.reserve((size_t)n_views * sizeof(RtInfo) + 64));
        HIPCHK(c, c->ch_existpart.reserve((size_t)std::max(1, L.maxN) * exist_chunks() * (size_t)std::max(1, L.maxS) * 4 + 256));     // chunk counts / bases of a view's sources
        HIPCHK(c, hipMemcpyAsync(c->ch_rtinfo.p, info.data(), info.size() * sizeof(RtInfo), hipMemcpyHostToDevice, st));
    }
    const unsigned char* dtab = L.dtab;
    const int maxN = L.maxN;
    if (ranged) {
        // (after the arenas are laid out for all views: a view keeps its slices whatever the range is)
        double p_range = 0, p_max = 0;
        for (int k = 0; k < n_views; ++k) {
            if (k < k_begin || k >= k_end) { vd[(size_t)k].verified = false; continue; }
            if (!vd[(size_t)k].verified) continue;
            double p = 0;
            for (int j = 0; j < views[k].n_tbm; ++j) p += (double)views[k].S_src * views[k].offsets[2 * views[k].to_be_matched[j] + 1];
            p_range += p; p_max = std::max(p_max, p);
        }
        L.pairs = p_range; L.max_pairs = p_max;
    }
    HIPCHK(c, c->ch_res.reserve((size_t)n_views * sizeof(ChainResult) + 16));
    HIPCHK(c, c->ch_flags.reserve(64));
    HIPCHK(c, c->ch_pin_res.reserve((size_t)n_views * (sizeof(ChainResult) + 8) + 64));
    HIPCHK(c, hipMemsetAsync(c->ch_res.p, 0, (size_t)n_views * sizeof(ChainResult), st));
    HIPCHK(c, hipMemsetAsync(c->ch_flags.p, 0, 64, st));
    {   // stage 1 starts after the tables and the zeroed row counts are in place
        hipEvent_t ready = get_local_event(c);
        HIPCHK(c, hipEventRecord(ready, st));
        HIPCHK(c, hipStreamWaitEvent(s1, ready, 0));
        if (sm != s1) HIPCHK(c, hipStreamWaitEvent(sm, ready, 0));
        put_local_event(c, ready);
    }
    // per-view results are written by the kernels straight into host-mapped pinned memory (no copy operations on the streams)
    ChainResult* hres = c->ch_pin_res.as<ChainResult>();
    int* hstats = reinterpret_cast<int*>(c->ch_pin_res.as<unsigned char>() + (size_t)n_views * sizeof(ChainResult));
    if (ranged) memset(hres, 0, (size_t)n_views * sizeof(ChainResult));          // (views outside the range: no records, whatever an earlier chain left here)
    ChainResult* hres_dev = nullptr;
    HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&hres_dev), hres, 0));
    int* hstats_dev = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(hres_dev) + (size_t)n_views * sizeof(ChainResult));
    // the four depths of a stage-1 pair are triangulated once, by k_pair_fill (ring scheme only: the fill runs ahead, its true row counts
    // are in place before the chain counts the view's reverse matches on top); L3D_DEPTH_IN_FILL=0: A/B, k_pair_mask triangulates too
    const bool depth_in_fill_env = c->opt.depth_in_fill != 0;
    const bool depth_in_fill = depth_in_fill_env && c->chain_ring != 0;
    auto pair_args = [&](int k) {
        PairArgs pa = chain_pair_args(c, views[k], vd[(size_t)k], dtab);
        pa.depth_in_fill = depth_in_fill ? 1 : 0;
        return pa;
    };

    { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(c, L3D_ERR_HIP, std::string("chain setup: ") + hipGetErrorString(e_)); }
    // ---- phase 1 (stage 1 of a view: pair test -> bit rows -> row counts -> statistics) is independent of the
    // chain; it is enqueued a window ahead of phase 2 so that the GPU always has work while the host trails behind
    const double pairs = L.pairs, max_pairs = L.max_pairs;
    std::vector<hipEvent_t> ev1((size_t)n_views, nullptr), evm((size_t)n_views, nullptr);
    int k_p1 = 0;                       // next view whose stage 1 is enqueued
    c->stats[0] = pairs;
    double raw_sum = 0;

    // ---- capacities (guarded on the device; an overflow restarts the chain at that view with more room)
    // first guess from the pair counts (raw density ~6 %, kept ~0.2 % of the pairs on the synthetic scenes)
    size_t cand_cap = chain_first_cand_cap(max_pairs);
    // The kept arena's first guess is GENEROUS (5 % of the pairs: the densest synthetic scene keeps 4.5 %, config 2 0.16 %) within 35 % of the free HBM: memory that is
    // never written costs an allocation of a millisecond when it is the process's first big one, while growing later means a hipMalloc of tens of GB in a process that
    // has freed big buffers before -- measured at 256 x 4000 x 24: 0.5 ms for the first 59 GB, 1.8-2.2 s for 75 GB at the second regrow, 25 ms per GB at 40 views in one
    // run of three (profiles/r6_first_pass_allocations.txt).  A job sized by memory gives its capacities (l3d_set_chain_capacities) or the old guess (L3D_ARENA_GUESS=4)
    size_t arena_cap = (size_t)(pairs * 0.004) + 1048576;
    std::vector<hipEvent_t> ev((size_t)n_views, nullptr);
    int k_enq = 0;                      // next view whose phase 2 is enqueued
    // run-ahead depths, A/B measured on one box (ms per config-2 pass): (12, 24) 19.1, (6, 12) 18.7, (4, 8) 18.4, (2, 4) 18.3,
    // (24, 40) 20.0 -- a shallow queue keeps the stage-1 candidates of a view cache-warm until its chain consumes them
    // the ring covers every view that can be in flight between the one being collected and the newest stage 1: after an
    // overflow ALL of them are refilled before any of their chains runs again
    const int kAhead = L3D_AHEAD, kStage1Ahead = L3D_S1AHEAD, kRing = kAhead + kStage1Ahead + 3;
    int rc_final = L3D_OK;
    // a pass over the same scene (same number of views, same pair count) starts with what the previous one ended up needing: no
    // overflow, no restart, no allocation after the first pass
    const bool same_scene = c->chain_seen_views == n_views && c->chain_seen_pairs == pairs;
    if (same_scene) { cand_cap = std::max(cand_cap, c->chain_seen_cand_cap); arena_cap = std::max(arena_cap, c->chain_seen_arena_cap); }
    else if (c->opt.arena_guess > 4 && !ranged && !pre) {      // (a block of views of a partitioned job is sized by memory: it keeps the small guess and grows)
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
            const size_t want = (size_t)(pairs * 0.001 * c->opt.arena_guess) + 1048576, fits = (size_t)((double)fr * 0.35 / 40.0);
            arena_cap = std::max(arena_cap, std::min(want, fits));
        }
        (void)hipGetLastError();
        if (arena_cap > 0xfffffff0ull) arena_cap = 0xfffffff0ull;
    }
    if (c->test_cand_cap) cand_cap = c->test_cand_cap;      // tests: force the overflow / restart path
    if (c->test_arena_cap) arena_cap = c->test_arena_cap;
    // views that own a slice of the arena: the range's verified views and the preloaded ones (a view's slice starts where the previous such view's ended)
    std::vector<char> has_rec((size_t)n_views, 0);
    for (int k = 0; k < n_views; ++k) has_rec[(size_t)k] = vd[(size_t)k].verified ? 1 : 0;
    long long pre_records = 0;
    if (pre) {
        if (pre->k1 != k_begin || pre->k0 < 0 || pre->k0 > pre->k1 || (int)pre->n_kept.size() != pre->k1 - pre->k0 || (int)pre->R.size() != pre->k1 - pre->k0) return fail(c, L3D_ERR_INVALID, "match_chain: bad preload");
        for (int k = pre->k0; k < pre->k1; ++k) { pre_records += pre->n_kept[(size_t)(k - pre->k0)]; has_rec[(size_t)k] = views[k].n_tbm > 0 ? 1 : 0; }
        arena_cap += (size_t)pre_records;
    }

    const bool use_cams = c->opt.kept_cams != 0;         // (0: A/B -- the sources' lists are scanned record by record)
    // early pair transposes (round 6, l3d_products.hip): a (view, camera) pair of the potential-correspondence build depends on that view's kept list alone, so
    // it is transposed on a side stream right behind the view's kept writer -- next to the following views' chains -- and the end of matchViews finds only the rows
    // left.  Entries live in an array aligned with the kept arena (a view's pairs hold exactly its records).  L3D_PROD_EARLY=0: all transposes at the end (A/B)
    Products& PE = c->products;
    // Long lists only (> 2^18 records per view: what the last pass over this scene kept, or the arena's first guess), view by view.  At config 2 (36 k records per view)
    // two launches and an event per view cost more than the 0.2 ms of transposes they take off the end (12.60 vs 12.41 ms per pass); eight views per launch still lose
    // (12.52 vs 12.41; the first pass pays 5 ms for the side stream and its buffers).  Option values 2 / 3 force a view / eight views per launch (tests)
    bool early = map && use_rt && !ranged && !pre && !cb && c->opt.prod_early != 0 && c->opt.prod_transpose != 0 && maxN > 0;
    int early_batch = c->opt.prod_early == 3 ? 8 : 1;
    if (early && c->opt.prod_early == 1) {
        const double per_view = (same_scene && c->chain_seen_kept > 0 ? c->chain_seen_kept : pairs * 0.004) / std::max(1, n_views);
        early = per_view > (double)kCamScanMin;
    }
    int early_next = 0;                 // views [0, early_next) have their transposes launched
    hipStream_t sp = nullptr;
    int early_maxSt = 1;
    if (early) {
        std::vector<int> tab((size_t)n_views * maxN * 2 + (size_t)n_views * (sizeof(EarlyView) / 4), 0);       // [St | boff_off], n_views x maxN each; the views' descriptors behind them
        PE.e_boff_off.assign((size_t)n_views * maxN, 0);
        long long nb = 0;
        for (int k = 0; k < n_views && early; ++k) {
            if (!vd[(size_t)k].verified) continue;
            for (int q = 0; q < views[k].N; ++q) {
                const uint32_t* it = std::lower_bound(map->view_ids, map->view_ids + map->n_views, views[k].local2global[q]);
                if (it == map->view_ids + map->n_views || *it != views[k].local2global[q]) continue;
                const int t = (int)(it - map->view_ids), St = map->seg_base[t + 1] - map->seg_base[t];
                if (St <= 0) continue;
                tab[(size_t)k * maxN + q] = St;
                tab[(size_t)n_views * maxN + (size_t)k * maxN + q] = (int)nb;
                PE.e_boff_off[(size_t)k * maxN + q] = (int)nb;
                nb += St + 1;
                early_maxSt = std::max(early_maxSt, St);
                if (nb > 0x7ffffff0ll) early = false;
            }
        }
        if (early) {
            if (!c->prod_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->prod_stream, hipStreamNonBlocking));
            sp = c->prod_stream;
            HIPCHK(c, PE.e_tab.reserve(tab.size() * 4 + 64));
            static_assert(sizeof(EarlyView) % 8 == 0, "descriptor size");
            {
                const int* dt = PE.e_tab.as<int>();
                EarlyView* hv = reinterpret_cast<EarlyView*>(tab.data() + (size_t)n_views * maxN * 2);       // (n_views * maxN * 2 ints: 8-byte aligned)
                for (int k = 0; k < n_views; ++k) {
                    EarlyView e;
                    e.rt = vd[(size_t)k].verified ? vd[(size_t)k].rt : nullptr;
                    e.St = dt + (size_t)k * maxN; e.boff_off = dt + (size_t)n_views * maxN + (size_t)k * maxN;
                    e.k = k; e.S = views[k].S_src; e.N = views[k].N; e.pad = 0;
                    hv[k] = e;
                }
            }
            HIPCHK(c, PE.e_cnt.reserve((size_t)n_views * maxN * 4 + 64));
            HIPCHK(c, PE.e_poff.reserve((size_t)n_views * maxN * 4 + 64));
            HIPCHK(c, PE.e_boff.reserve((size_t)nb * 4 + 64));
            HIPCHK(c, hipMemcpyAsync(PE.e_tab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipStreamSynchronize(st));                    // (`tab` is pageable and leaves scope)
        }
    }
    auto reserve_caps = [&]() -> int {
        if (int rc = chain_reserve_candidates(c, L, cand_cap, c->chain_ring ? kRing : 0)) return rc;
        HIPCHK(c, c->ch_kept.reserve(arena_cap * sizeof(Match)));
        if (use_cams) HIPCHK(c, c->ch_keptcam.reserve(arena_cap * 4 + 64));
        if (early && arena_cap < 0x7ffffff0ull) { HIPCHK(c, PE.e_E.reserve(arena_cap * 4 + 64)); HIPCHK(c, PE.e_T.reserve(arena_cap * 4 + 64)); }
        return L3D_OK;
    };
    { int rc = reserve_caps(); if (rc) return rc; }
    if (pre && pre->k1 > pre->k0) {
        long long base = 0;
        size_t so = 0;
        for (int k = pre->k0; k < pre->k1; ++k) {
            ChainResult r;
            r.kept_base = (unsigned)base; r.n_kept = has_rec[(size_t)k] ? pre->n_kept[(size_t)(k - pre->k0)] : 0; r.R = pre->R[(size_t)(k - pre->k0)]; r.overflow = 0;
            hres[k] = r;
            base += r.n_kept;
            if (views[k].n_tbm > 0) {
                // (chain_assign_arenas gave every verified view of the schedule its slices, whatever the range)
                ChainViewDev& d = vd[(size_t)k];
                if (views[k].S_src > 0 && d.best && pre->best) HIPCHK(c, hipMemcpyAsync(d.best, pre->best + so, (size_t)views[k].S_src * 8, hipMemcpyDeviceToDevice, st));
                if (views[k].S_src > 0 && d.bestpos && pre->bestpos) HIPCHK(c, hipMemcpyAsync(d.bestpos, pre->bestpos + so, (size_t)views[k].S_src * 4, hipMemcpyDeviceToDevice, st));
                so += (size_t)views[k].S_src;
            }
        }
        if (base > 0) HIPCHK(c, hipMemcpyAsync(c->ch_kept.p, pre->records, (size_t)base * sizeof(Match), hipMemcpyDeviceToDevice, st));
        if (base > 0 && use_cams && !use_rt) hipLaunchKernelGGL(k_cams_of_records, dim3((unsigned)((base + 255) / 256)), dim3(256), 0, st, c->ch_kept.as<Match>(), base, c->ch_keptcam.as<unsigned>());
        if (base > 0 && use_rt) {
            // the taken-over lists did not come out of this chain's kept writer: their side array and run tables are rebuilt from the records
            std::vector<RtJob> jobs;
            std::vector<unsigned> ids;
            std::vector<int> qs;
            std::vector<size_t> at;
            for (int k = pre->k0; k < pre->k1; ++k) {
                if (!has_rec[(size_t)k] || hres[k].n_kept == 0 || !vd[(size_t)k].rt) continue;
                std::vector<std::pair<unsigned, int>> byid;
                for (int q = 0; q < views[k].N; ++q) byid.push_back({ views[k].local2global[q], q });
                std::sort(byid.begin(), byid.end());
                at.push_back(ids.size());
                for (auto& e : byid) { ids.push_back(e.first); qs.push_back(e.second); }
                RtJob j;
                j.recs = c->ch_kept.as<Match>() + hres[k].kept_base; j.qt = c->ch_keptcam.as<unsigned>() + hres[k].kept_base; j.rt = vd[(size_t)k].rt;
                j.ids = nullptr; j.qs = nullptr; j.n = hres[k].n_kept; j.S = views[k].S_src; j.N = views[k].N; j.pad = 0;
                jobs.push_back(j);
            }
            if (!jobs.empty()) {
                auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
                const size_t o_ids = al(jobs.size() * sizeof(RtJob)), o_qs = o_ids + al(ids.size() * 4), o_err = o_qs + al(qs.size() * 4), o_key = o_err + 256;
                size_t n_keys = 0;
                for (const RtJob& j : jobs) n_keys += (size_t)j.n;
                HIPCHK(c, c->ch_rtjobs.reserve(o_key + n_keys * 4 + 256));       // (not ch_stage: the taken-over records may live there)
                unsigned char* sb = c->ch_rtjobs.as<unsigned char>();
                int max_n = 0, max_cells = 0;
                size_t ko = 0;
                for (size_t i = 0; i < jobs.size(); ++i) {
                    jobs[i].ids = reinterpret_cast<const unsigned*>(sb + o_ids) + at[i]; jobs[i].qs = reinterpret_cast<const int*>(sb + o_qs) + at[i];
                    jobs[i].skey = reinterpret_cast<unsigned*>(sb + o_key) + ko; ko += (size_t)jobs[i].n;
                    max_n = std::max(max_n, jobs[i].n); max_cells = std::max(max_cells, (jobs[i].N + 1) * jobs[i].S);
                }
                HIPCHK(c, hipMemcpyAsync(sb, jobs.data(), jobs.size() * sizeof(RtJob), hipMemcpyHostToDevice, st));
                HIPCHK(c, hipMemcpyAsync(sb + o_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, st));
                HIPCHK(c, hipMemcpyAsync(sb + o_qs, qs.data(), qs.size() * 4, hipMemcpyHostToDevice, st));
                HIPCHK(c, hipMemsetAsync(sb + o_err, 0, 4, st));
                launch_qt_from_records(reinterpret_cast<const RtJob*>(sb), (int)jobs.size(), max_n, reinterpret_cast<int*>(sb + o_err), st);
                launch_rt_from_qt(reinterpret_cast<const RtJob*>(sb), (int)jobs.size(), max_cells, st);
                int bad = 0;
                HIPCHK(c, hipMemcpyAsync(&bad, sb + o_err, 4, hipMemcpyDeviceToHost, st));
                HIPCHK(c, hipStreamSynchronize(st));
                if (bad) return fail(c, L3D_ERR_INVALID, "match_chain: the lists taken over from another rank are not ordered (segment, camera) -- no run tables");
            }
        }
        HIPCHK(c, hipMemcpyAsync(c->ch_res.as<ChainResult>() + pre->k0, hres + pre->k0, (size_t)(pre->k1 - pre->k0) * sizeof(ChainResult), hipMemcpyHostToDevice, st));
    }
    auto ringA_meta = [&](int k) { return c->ch_ringA_meta.as<uint2>() + (size_t)(k % kRing) * cand_cap; };
    auto ringA_depths = [&](int k) { return c->ch_ringA_depths.as<float4>() + (size_t)(k % kRing) * cand_cap; };
    // row starts + depth records of a view's stage-1 candidates alone (its reverse matches are not known yet)
    // L3D_CHAIN_RING=0 (A/B): triangulation on the chain stream, straight into the combined order -- measured 5 % slower on
    // config 2 than the ring scheme, which keeps the chain stream short.  (Also measured: letting the verification read the
    // stage-1 candidates in place instead of copying them (k_place) -- 20 % SLOWER: the copy is a streaming pass that
    // leaves the candidates cache-hot for the latency-bound kernels that follow.)
    const bool use_ring = c->chain_ring != 0;
    // the row starts of the stage-1 candidates are formed inside k_pair_fill from k_pair_mask's counters and their block sums: no
    // scan launch on the stage-1 stream (the longer of the two), no statistics for the host to wait for.  L3D_FUSED_ROWS=0: A/B.
    const bool fused_rows_env = c->opt.fused_rows != 0;
    const bool fused_rows = fused_rows_env && use_ring && depth_in_fill && maxN <= 96;      // (k_pair_mask's LDS block sums: 64 rows N apart span <= 32 blocks)
    auto enqueue_fillA = [&](int k, hipStream_t s) {
        if (!use_ring) return;
        const ViewDev& d = vd[(size_t)k];
        PairArgs pa = pair_args(k);
        pa.cand_cap = (int)cand_cap;
        pa.rowcnt = d.rowcnt;
        if (fused_rows) { pa.rowub = d.rowub; pa.rowblk = d.rowblk; pa.rowstart_out = d.rowA; }
        { ProfScope p(c, "pair_fill", s); launch_pair_fill(pa, d.rowA, ringA_meta(k), ringA_depths(k), s); }
    };
    auto enqueue_stage1 = [&](int k) -> int {
        if (!vd[(size_t)k].verified) return L3D_OK;
        hstats[2 * k] = hstats[2 * k + 1] = 0;
        if (views[k].S_src > 0) {
            const PairArgs pa = pair_args(k);
            {   // bit rows + row counts (added into the rows zeroed at chain start) in one launch
                PairArgs pm = pa;
                pm.rowcnt = fused_rows ? vd[(size_t)k].rowub : vd[(size_t)k].rowcnt;
                if (fused_rows) pm.rowblk = vd[(size_t)k].rowblk;
                // (own stream: the bit rows' ring slot was last used by view k - kRing, whose chain may still re-form its candidates from them)
                if (sm != s1) for (int j = k - kRing; j >= 0; j -= kRing) if (ev[(size_t)j]) { HIPCHK(c, hipStreamWaitEvent(sm, ev[(size_t)j], 0)); break; }
                { ProfScope p(c, "pair_mask", sm); launch_pair_mask(pm, vd[(size_t)k].maxW, sm, c->opt.pair_spb); }
                if (sm != s1) {
                    if (!evm[(size_t)k]) evm[(size_t)k] = get_local_event(c);
                    HIPCHK(c, hipEventRecord(evm[(size_t)k], sm));
                    HIPCHK(c, hipStreamWaitEvent(s1, evm[(size_t)k], 0));
                }
            }
            // row starts of the stage-1 candidates + their statistics straight into host-mapped memory (one launch)
            if (fused_rows) {}
            else if (use_ring) { ProfScope p(c, "scan", s1); launch_scan(vd[(size_t)k].rowcnt, vd[(size_t)k].rowA, views[k].S_src * views[k].N, nullptr, s1, nullptr, views[k].N, 0, views[k].S_src, hstats_dev + 2 * k); }
            else launch_raw_stats(vd[(size_t)k].rowcnt, views[k].N, 0, views[k].S_src, hstats_dev + 2 * k, s1);
            // the ring slot was last used by view k - kRing: wait until its chain has consumed it
            for (int j = k - kRing; j >= 0; j -= kRing) if (ev[(size_t)j]) { HIPCHK(c, hipStreamWaitEvent(s1, ev[(size_t)j], 0)); break; }
            enqueue_fillA(k, s1);
        }
        ev1[(size_t)k] = get_local_event(c);
        HIPCHK(c, hipEventRecord(ev1[(size_t)k], s1));
        return L3D_OK;
    };

    double t_ev1 = 0;                   // host time spent waiting for stage-1 statistics
    double kept_seen = pre ? (double)pre_records : 0.0;     // kept records / views whose result record the host has read (sizes the source scans)
    int views_seen = pre ? pre->k1 - pre->k0 : 0;
    auto enqueue_view = [&](int k) -> int {
        const l3d_chain_view& v = views[k];
        const ViewDev& d = vd[(size_t)k];
        while (k_p1 < n_views && k_p1 <= k + kStage1Ahead) { int rc = enqueue_stage1(k_p1); if (rc) return rc; ++k_p1; }
        if (!d.verified) return L3D_OK;
        const double te0 = now_s();
        if (!fused_rows) HIPCHK(c, hipEventSynchronize(ev1[(size_t)k]));          // its stage-1 statistics (enqueued a window earlier)
        t_ev1 += now_s() - te0;
        HIPCHK(c, hipStreamWaitEvent(st, ev1[(size_t)k], 0));
        PairArgs pa = pair_args(k);
        pa.cand_cap = (int)cand_cap;
        const int S = v.S_src, N = v.N;
        const size_t nrow = (size_t)S * N;
        Match* arena = c->ch_kept.as<Match>();
        ChainResult* dres = c->ch_res.as<ChainResult>();
        const int* d_sc = reinterpret_cast<const int*>(dtab + d.o_sc);
        (void)hipGetLastError();
        const int* d_si = reinterpret_cast<const int*>(dtab + d.o_si);
        unsigned* cams = use_cams ? c->ch_keptcam.as<unsigned>() : nullptr;
        // workgroups per source view of the two scans of the sources' lists: a thread walks its stride of a list with dependent loads, so the list
        // must be spread over enough of them -- 32 x 256 threads take config 2's 36 k records in 5 steps, but 3.5 M records (4000 x 24) in 430:
        // sized from the lists the chain has seen so far (the host trails a few views behind)
        const int bps = (int)std::min(512.0, std::max(32.0, (views_seen > 0 ? kept_seen / views_seen : 0.0) / 4096.0));
        // run tables: g lanes per run (the average run of the lists seen so far, rounded up to a power of two), workgroups per source to cover its segments
        const int* d_ss = reinterpret_cast<const int*>(dtab + d.o_ss);
        const RtInfo* d_info = use_rt ? c->ch_rtinfo.as<RtInfo>() : nullptr;
        int rt_g = 1;
        if (use_rt) { const double avg_run = views_seen > 0 ? kept_seen / views_seen / std::max(1.0, (double)S * std::max(1, N)) : 1.0; while (rt_g < 64 && rt_g < avg_run) rt_g <<= 1; }
        if (c->opt.rt_g > 0) rt_g = c->opt.rt_g;
        const int rt_bps = std::max(1, std::min(512, (L.maxS * std::max(1, rt_g) + 255) / 256));
        // short lists (config 2 keeps 36 k matches per view) are scanned record by record as before: one pass over cache-resident data beats the run
        // tables' extra level of dependent loads (12.2 vs 12.9 ms per config-2 pass); the side array holds (camera, target) words then, not camera ids
        const bool rt_exist = use_rt && views_seen > 0 && kept_seen / views_seen > (double)kCamScanMin;
        const unsigned* scan_cams = use_rt ? nullptr : cams;
        if (rt_exist) { ProfScope p(c, "exist"); launch_exist_count_rt(cams, d_info, dres, d_si, d_sc, d_ss, v.n_sources, rt_g, N, S, d.rowcnt, c->ch_existpart.as<int>(), st); }
        else { ProfScope p(c, "exist"); launch_exist_count(arena, dres, d_si, d_sc, v.n_sources, v.view_id, N, S, d.rowcnt, st, scan_cams, bps); }
        // combined row starts (+ zeroed scatter cursors, + the segments ordered longest first for the verification launch)
        { ProfScope p(c, "scan"); launch_scan(d.rowcnt, c->row_start.as<int>(), (int)nrow, c->ch_cursor.as<int>(), st, c->ch_segorder.as<int>(), N, 0, S); }
        if (use_ring) {
            {
                ProfScope p(c, "cand_move");
                launch_place(pa.tbm, v.n_tbm, N, S, d.rowA, ringA_meta(k), ringA_depths(k), arena, dres, d_si, d_sc, v.n_sources, v.view_id,
                             c->row_start.as<int>(), c->ch_cursor.as<int>(), (int)cand_cap, c->cand_meta.as<uint2>(), c->cand_depths.as<float4>(), st, scan_cams, rt_exist ? rt_bps : bps,
                             rt_exist ? d_info : nullptr, d_ss, rt_g, c->opt.rt_place_lds ? c->ch_existpart.as<int>() : nullptr);
            }
            if (v.n_sources && !(c->verify_mode == 0 && verify_window_supported(N))) {     // (the window kernel orders the runs itself)
                ProfScope p(c, "exist");
                launch_exist_sort_runs(d_sc, v.n_sources, N, S, c->row_start.as<int>(), c->cand_meta.as<uint2>(), c->cand_depths.as<float4>(), (int)cand_cap, st, 0, -1,
                                       c->vw_scratch.as<float>(), (long long)cand_cap + kVWSlack, c->cand_conf.as<unsigned>());
            }
        } else {
            if (S > 0) { ProfScope p(c, "pair_fill"); launch_pair_fill(pa, c->row_start.as<int>(), c->cand_meta.as<uint2>(), c->cand_depths.as<float4>(), st); }
            ProfScope p(c, "exist");
            launch_exist_scatter(arena, dres, d_si, d_sc, v.n_sources, v.view_id, N, S, c->row_start.as<int>(),
                                 c->ch_cursor.as<int>(), c->cand_meta.as<uint2>(), c->cand_depths.as<float4>(), (int)cand_cap, st);
            if (v.n_sources && !(c->verify_mode == 0 && verify_window_supported(N)))
                launch_exist_sort_runs(d_sc, v.n_sources, N, S, c->row_start.as<int>(), c->cand_meta.as<uint2>(), c->cand_depths.as<float4>(), (int)cand_cap, st, 0, -1,
                                       c->vw_scratch.as<float>(), (long long)cand_cap + kVWSlack, c->cand_conf.as<unsigned>());
        }
        VerifyArgs va = chain_verify_args(c, v, d, dtab, cand_cap);
        va.res = dres + k;
        // (fused row starts: no statistics -- the largest LDS image the budget allows)
        chain_launch_verify(c, va, d, d_sc, v.n_sources, fused_rows ? -1 : hstats[2 * k + 1], cand_cap, st);
        {
            ProfScope p(c, "kept_write");
            int pv = k - 1;
            while (pv >= 0 && !has_rec[(size_t)pv]) --pv;                    // the arena slice starts where the previous verified (or preloaded) view's ended
            launch_kept_write_chain(va, c->kept_cnt.as<int>(), (int)nrow, pv >= 0 ? dres + pv : nullptr, (unsigned long long)arena_cap, dres + k, hres_dev + k,
                                    reinterpret_cast<const unsigned*>(dtab + d.o_l2g), arena, st, (map || ranged) ? d.bestpos : nullptr, cams, use_rt ? d.rt : nullptr, S);
        }
        { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(c, L3D_ERR_HIP, std::string("chain launch, view ") + std::to_string(k) + ": " + hipGetErrorString(e_)); }
        // (with a delivery callback the host starts D2H copies of device memory once it has seen this event: a default, fenced event then)
        if (!ev[(size_t)k]) ev[(size_t)k] = cb ? get_event(c) : get_local_event(c);
        HIPCHK(c, hipEventRecord(ev[(size_t)k], st));
        if (early && arena_cap < 0x7ffffff0ull && k + 1 - early_next >= early_batch) {
            HIPCHK(c, hipStreamWaitEvent(sp, ev[(size_t)k], 0));
            const double avg_run = views_seen > 0 ? kept_seen / views_seen / std::max(1.0, (double)S * std::max(1, N)) : 1.0;
            launch_early_transposes(c, reinterpret_cast<const EarlyView*>(PE.e_tab.as<int>() + (size_t)n_views * maxN * 2), early_next, k + 1 - early_next, maxN, early_maxSt, dres, cams, PE.e_cnt.as<int>(),
                                    PE.e_poff.as<unsigned>(), PE.e_boff.as<int>(), PE.e_E.as<unsigned>(), PE.e_T.as<unsigned>(), avg_run, sp);
            early_next = k + 1;
        }
        return L3D_OK;
    };

    // ---- phase 2 + trailing result loop.  This thread enqueues and watches the per-view result records (overflow ->
    // grow and restart at that view); finished views are handed, in order, to a delivery thread that copies the kept slice
    // and the depth pairs (copy stream, SDMA) and runs the caller's bookkeeping -- so the ~13 launches per view of this
    // thread are never held up by host work.
    const double t_loop0 = now_s();
    double kept_total = 0, t_cb = 0, t_wait = 0, t_d2h = 0;
    struct Item { int k; int verified; ChainResult r; };
    std::mutex mu;
    std::condition_variable cv_work, cv_idle;
    std::deque<Item> work;
    bool done = false, busy = false;
    int deliver_rc = L3D_OK;
    std::string deliver_err;
    std::thread deliverer([&]() {
        if (!cb) return;                                // resident run: nothing is handed to the host
        (void)hipSetDevice(c->device);
        for (;;) {
            Item it;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&]() { return done || !work.empty(); });
                if (work.empty()) return;
                it = work.front();
                work.pop_front();
                busy = true;
            }
            int rc = L3D_OK;
            std::string err;
            if (deliver_rc == L3D_OK) {
                const l3d_chain_view& v = views[it.k];
                const ViewDev& d = vd[(size_t)it.k];
                if (!it.verified) {                     // cudawrapper.cu:877-878: nothing to match, the caller keeps its list
                    if (cb(user, it.k, 0, nullptr, 0, nullptr, 0, 0)) { rc = L3D_ERR_INVALID; err = "callback failed"; }
                } else {
                    const double td0 = now_s();
                    // the kept list lands in the pinned arena, where it stays valid for the caller until the next chain starts
                    hipError_t e = hipSuccess;
                    l3d_match* kept_host = static_cast<l3d_match*>(c->pin_arena.alloc((size_t)it.r.n_kept * sizeof(Match) + 16, &e));
                    if (e == hipSuccess) e = c->ch_pin_best.reserve((size_t)v.S_src * 8 + 16);
                    if (e == hipSuccess && it.r.n_kept)
                        e = hipMemcpyAsync(kept_host, c->ch_kept.as<Match>() + it.r.kept_base, (size_t)it.r.n_kept * sizeof(Match), hipMemcpyDeviceToHost, c->copy_stream);
                    if (e == hipSuccess && v.S_src) e = hipMemcpyAsync(c->ch_pin_best.p, d.best, (size_t)v.S_src * 8, hipMemcpyDeviceToHost, c->copy_stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(c->copy_stream);
                    if (e != hipSuccess) { rc = L3D_ERR_HIP; err = std::string("chain delivery, view ") + std::to_string(it.k) + ": " + hipGetErrorString(e); }
                    else {
                        float* best = c->ch_pin_best.as<float>();
                        int nb = 0;
                        if (it.r.R > 0)
                            for (int s = 0; s < v.S_src; ++s)
                                if (best[2 * s] != -1.0f) { best[2 * nb] = best[2 * s]; best[2 * nb + 1] = best[2 * s + 1]; ++nb; }   // in place: nb <= s
                        kept_total += it.r.n_kept;
                        const double tc0 = now_s();
                        t_d2h += tc0 - td0;
                        if (cb(user, it.k, 1, kept_host, it.r.n_kept, best, nb, it.r.R)) { rc = L3D_ERR_INVALID; err = "callback failed"; }
                        t_cb += now_s() - tc0;
                    }
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (rc && deliver_rc == L3D_OK) { deliver_rc = rc; deliver_err = err; }
                busy = false;
            }
            cv_idle.notify_all();
        }
    });
    auto hand_over = [&](int k, int verified, const ChainResult& r) {
        if (!cb) { if (verified) kept_total += r.n_kept; return; }
        { std::lock_guard<std::mutex> lk(mu); work.push_back(Item{ k, verified, r }); }
        cv_work.notify_one();
    };
    auto wait_delivered = [&]() {                       // every handed-over view has left the device arena
        if (!cb) return;
        std::unique_lock<std::mutex> lk(mu);
        cv_idle.wait(lk, [&]() { return work.empty() && !busy; });
    };
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        rc_final = fail(c, L3D_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
        return false;
    };
    for (int k = 0; k < n_views && rc_final == L3D_OK; ++k) {
        while (k_enq < n_views && k_enq <= k + kAhead) { int rc = enqueue_view(k_enq); if (rc) { rc_final = rc; break; } ++k_enq; }
        if (rc_final) break;
        { std::lock_guard<std::mutex> lk(mu); if (deliver_rc) break; }
        const ViewDev& d = vd[(size_t)k];
        if (!d.verified) { hand_over(k, 0, ChainResult()); continue; }
        const double tw0 = now_s();
        if (!hip_ok(hipEventSynchronize(ev[(size_t)k]), "hipEventSynchronize")) break;
        t_wait += now_s() - tw0;
        const ChainResult r = hres[k];
        if (r.overflow) {
            // not enough room for this view's candidates / kept matches: everything before it is valid and stays
            // in the arena; wait for the queue (and the delivery of earlier views) to drain, grow, and re-enqueue from this view
            if (!hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize") || !hip_ok(hipStreamSynchronize(sm), "hipStreamSynchronize") || !hip_ok(hipStreamSynchronize(s1), "hipStreamSynchronize")) break;
            if (sp && !hip_ok(hipStreamSynchronize(sp), "hipStreamSynchronize")) break;
            const double t_restart0 = now_s();
            wait_delivered();
            if (r.overflow & 1) cand_cap = (size_t)r.R + (size_t)r.R / 4 + 65536;
            if (r.overflow & 2) {
                // the arena cannot be reallocated without losing earlier lists that later views still read:
                // copy it over
                // (projected from the views done so far when there are enough of them: dense scenes keep 10x the first guess)
                size_t new_cap = arena_cap * 2;
                // (views of THIS call: a ranged chain -- a block of views -- must not size its arena for the whole scene)
                const int n_pre = pre ? pre->k1 - pre->k0 : 0, done = k - k_begin + n_pre, span = k_end - k_begin + n_pre;
                // the first views of a chain have few sources yet and keep less than the later ones (40 x 4000 x 24: the projection at view 8 x 1.3 fell 2 % short,
                // the arena overflowed again at view 38 and DOUBLED to 9.8 GB -- an allocation that took 3 to 250 ms): 1.5 early on; past the
                // middle the projection is good and replaces the doubling
                if (done >= 4) {
                    const size_t proj = (size_t)((double)r.kept_base / done * span * (2 * done < span ? 1.5 : 1.15)) + 1048576;
                    new_cap = 2 * done < span ? std::max(new_cap, proj) : std::max(proj, arena_cap + arena_cap / 8);
                }
                // (records are indexed with 32 bits: the arena ends at 2^32 records = 137 GB; doubling must not run past it)
                const size_t kMaxRecords = 0xfffffff0u;
                if (new_cap > kMaxRecords) {
                    if (arena_cap >= kMaxRecords) { rc_final = fail(c, L3D_ERR_UNSUPPORTED, "match_chain: more than 2^32 kept matches in one chain (view " + std::to_string(k) + " of " + std::to_string(n_views) + ")"); break; }
                    new_cap = kMaxRecords;
                }
                void* np = nullptr;
                const double t_m0 = now_s();
                {
                    const hipError_t me = hipMalloc(&np, new_cap * sizeof(Match));
                    if (me != hipSuccess) {
                        size_t fr = 0, tot = 0;
                        (void)hipMemGetInfo(&fr, &tot);
                        rc_final = fail(c, L3D_ERR_NOMEM, "match_chain: growing the kept arena to " + std::to_string(new_cap * sizeof(Match) >> 20) + " MB at view " + std::to_string(k) + " of " + std::to_string(n_views) +
                                                          " (" + std::to_string((size_t)r.kept_base * sizeof(Match) >> 20) + " MB in use, " + std::to_string(fr >> 20) + " of " + std::to_string(tot >> 20) + " MB free): " + hipGetErrorString(me));
                        break;
                    }
                }
                const double t_m1 = now_s();
                if (!hip_ok(hipMemcpy(np, c->ch_kept.p, (size_t)r.kept_base * sizeof(Match), hipMemcpyDeviceToDevice), "hipMemcpy")) break;
                const double t_m2 = now_s();
                (void)hipFree(c->ch_kept.p);
                if (c->opt.timing) fprintf(stderr, "[l3d match_chain]   arena of %zu MB: hipMalloc %.2f ms, copy of %zu MB %.2f ms, hipFree of the old one %.2f ms\n", new_cap * sizeof(Match) >> 20, (t_m1 - t_m0) * 1e3,
                                           (size_t)r.kept_base * sizeof(Match) >> 20, (t_m2 - t_m1) * 1e3, (now_s() - t_m2) * 1e3);
                c->ch_kept.p = np; c->ch_kept.cap = new_cap * sizeof(Match);
                if (use_cams) {                             // (the side array grows with it, its used part kept)
                    void* nc = nullptr;
                    if (!hip_ok(hipMalloc(&nc, new_cap * 4 + 64), "hipMalloc (target cameras of the kept arena)")) break;
                    if (!hip_ok(hipMemcpy(nc, c->ch_keptcam.p, (size_t)r.kept_base * 4, hipMemcpyDeviceToDevice), "hipMemcpy")) { (void)hipFree(nc); break; }
                    (void)hipFree(c->ch_keptcam.p);
                    c->ch_keptcam.p = nc; c->ch_keptcam.cap = new_cap * 4 + 64;
                }
                if (early && new_cap < 0x7ffffff0ull) {     // (the earlier views' transposed entries move with their records; beyond 2^31 records the end transposes block by block)
                    void* ne = nullptr;
                    if (!hip_ok(hipMalloc(&ne, new_cap * 4 + 64), "hipMalloc (transposed entries of the kept arena)")) break;
                    if (!hip_ok(hipMemcpy(ne, PE.e_E.p, (size_t)r.kept_base * 4, hipMemcpyDeviceToDevice), "hipMemcpy")) { (void)hipFree(ne); break; }
                    (void)hipFree(PE.e_E.p);
                    PE.e_E.p = ne; PE.e_E.cap = new_cap * 4 + 64;
                }
                arena_cap = new_cap;
            }
            { int rc = reserve_caps(); if (rc) { rc_final = rc; break; } }
            if (c->opt.timing) fprintf(stderr, "[l3d match_chain] view %d of %d overflowed (%d): candidates %zu, arena %zu records; regrown in %.2f ms, %.2f ms into the loop\n", k, n_views, r.overflow, cand_cap, arena_cap,
                                       (now_s() - t_restart0) * 1e3, (now_s() - t_loop0) * 1e3);
            // the row counts of the views enqueued after k were already incremented by their reverse matches: rebuild
            // and the stage-1 candidate buffers of every view in flight live in the (re-sized) ring: refill them
            for (int j = k; j < k_p1; ++j) {
                if (!vd[(size_t)j].verified || views[j].S_src == 0) continue;
                if (j < k_enq) {
                    if (!hip_ok(hipMemsetAsync(vd[(size_t)j].rowcnt, 0, (size_t)views[j].S_src * views[j].N * 4, st), "hipMemsetAsync")) break;
                    if (!fused_rows) launch_row_count(pair_args(j), vd[(size_t)j].rowcnt, st);      // (fused: the upper bounds live in rowub, untouched)
                }
                enqueue_fillA(j, st);
            }
            if (rc_final) break;
            k_enq = k;
            early_next = std::min(early_next, k);           // (the views run again are transposed again)
            --k;
            continue;
        }
        raw_sum += r.R;                     // candidates verified (stage-1 + existing), counted when the view is final (a restart enqueues views twice)
        kept_seen += r.n_kept; views_seen += 1;
        hand_over(k, 1, r);
    }
    { std::lock_guard<std::mutex> lk(mu); done = true; }
    cv_work.notify_one();
    deliverer.join();
    if (rc_final == L3D_OK && deliver_rc) rc_final = fail(c, deliver_rc, deliver_err);
    const double t_prod0 = now_s();
    if (rc_final == L3D_OK && map) {
        // ---- the products of matchViews, on the device, from the arena (l3d_products.hip); enqueued behind the last view
        std::vector<ProdChainView> pv((size_t)n_views);
        for (int k = 0; k < n_views; ++k) pv[(size_t)k] = ProdChainView{ vd[(size_t)k].verified ? vd[(size_t)k].best : nullptr, vd[(size_t)k].verified ? vd[(size_t)k].bestpos : nullptr, vd[(size_t)k].verified ? 1 : 0,
                                                                         use_rt && vd[(size_t)k].verified ? vd[(size_t)k].rt : nullptr };
        ProdEarly pe;
        const bool early_done = early && arena_cap < 0x7ffffff0ull;
        if (early_done) {
            if (early_next < n_views) {                     // the last batch's remainder: on the chain's stream, behind the last view
                const double avg_run = views_seen > 0 ? kept_seen / views_seen / std::max(1.0, (double)L.maxS * std::max(1, maxN)) : 1.0;
                launch_early_transposes(c, reinterpret_cast<const EarlyView*>(PE.e_tab.as<int>() + (size_t)n_views * maxN * 2), early_next, n_views - early_next, maxN, early_maxSt, c->ch_res.as<ChainResult>(),
                                        c->ch_keptcam.as<unsigned>(), PE.e_cnt.as<int>(), PE.e_poff.as<unsigned>(), PE.e_boff.as<int>(), PE.e_E.as<unsigned>(), PE.e_T.as<unsigned>(), avg_run, st);
                early_next = n_views;
            }
            hipEvent_t e = get_local_event(c);
            (void)hipEventRecord(e, sp);
            (void)hipStreamWaitEvent(st, e, 0);
            put_local_event(c, e);
            pe.pcnt_kq = PE.e_cnt.as<int>(); pe.poff_kq = PE.e_poff.as<unsigned>(); pe.boff = PE.e_boff.as<int>(); pe.E = PE.e_E.as<unsigned>();
            pe.boff_off_host = PE.e_boff_off.data(); pe.maxN = maxN;
        }
        rc_final = build_products(c, views, n_views, pv.data(), hres, map, summary, n_pot, 0, -1, nullptr, use_rt ? c->ch_keptcam.as<unsigned>() : nullptr, early_done ? &pe : nullptr);
    }
    if (c->opt.timing)
        fprintf(stderr, "[l3d match_chain] setup %.2f ms | enqueue + watch loop %.2f ms (waiting: view results %.2f, stage-1 statistics %.2f) | delivery thread: d2h %.2f, callback %.2f\n",
                (t_loop0 - t_setup0) * 1e3, (t_prod0 - t_loop0) * 1e3, t_wait * 1e3, t_ev1 * 1e3, t_d2h * 1e3, t_cb * 1e3);
    if (c->opt.timing && map) fprintf(stderr, "[l3d match_chain] products on the device %.2f ms\n", (now_s() - t_prod0) * 1e3);
    const double t_tail0 = now_s();
    if (sm != s1) (void)hipStreamSynchronize(sm);
    (void)hipStreamSynchronize(s1);
    if (sp) (void)hipStreamSynchronize(sp);
    (void)hipStreamSynchronize(st);
    if (c->opt.timing) fprintf(stderr, "[l3d match_chain] hipSetDevice %.3f ms, final syncs %.3f ms\n", (t_setup0 - t_enter) * 1e3, (now_s() - t_tail0) * 1e3);
    for (hipEvent_t e : ev) { if (cb) put_event(c, e); else put_local_event(c, e); }
    for (hipEvent_t e : ev1) put_local_event(c, e);
    for (hipEvent_t e : evm) put_local_event(c, e);
    c->stats[1] = raw_sum;
    c->stats[3] = kept_total;
    if (rc_final == L3D_OK && !c->test_cand_cap && !c->test_arena_cap) {
        c->chain_seen_views = n_views; c->chain_seen_pairs = pairs; c->chain_seen_cand_cap = cand_cap; c->chain_seen_arena_cap = arena_cap; c->chain_seen_kept = kept_seen;
    }
    return rc_final;
}

extern "C" int l3d_match_chain(l3d_ctx* c, const l3d_chain_view* views, int n_views, l3d_chain_callback cb, void* user)
{
    if (c && n_views > 0 && !cb) return fail(c, L3D_ERR_INVALID, "l3d_match_chain: bad argument");
    return run_chain(c, views, n_views, cb, user, nullptr, nullptr, nullptr);
}

extern "C" int l3d_match_chain_resident(l3d_ctx* c, const l3d_chain_view* views, int n_views, const l3d_dense_map* map, l3d_chain_summary* summary, int64_t* n_pot)
{
    if (!c) return L3D_ERR_INVALID;
    if (!map || !map->view_ids || !map->seg_base || map->n_views < 0 || (n_views > 0 && !summary)) return fail(c, L3D_ERR_INVALID, "l3d_match_chain_resident: bad argument");
    if (n_pot) *n_pot = 0;
    return run_chain(c, views, n_views, nullptr, nullptr, map, summary, n_pot);
}

// =================================================================================================================================
// matchViews sharded by BLOCKS OF VIEWS, speculatively, with exact verification (round 4).
//
// The chain over views has a short memory: started cold at view B - L (nothing known about earlier views), its kept lists become
// bit-identical to the true chain's after about three neighbour windows (measured: scripts/speculate_blocks.py).  So rank r of `world`
// runs the ordinary single-GPU chain -- full-width kernels, no per-view collective -- on views [B_r - warmup, B_{r+1}) only, and the ranks
// then CHECK the speculation: every rank publishes a 64-bit digest of every kept list it computed; rank r's block is exact if rank r-1's is
// and the `window` views in front of B_r came out of rank r's warm-up exactly as rank r-1 (whose block they belong to) computed them --
// from B_r on every view then has the same inputs as in the one chain, and the same arithmetic.  All ranks read the same gathered table,
// so all reach the same verdict without another collective.  When it holds, the ranks all-gather their blocks' kept records (+ best depth
// pairs / positions), lay them out as the one chain's arena and build matchViews' products from it (l3d_products.hip, unchanged).  When it
// does not, nothing is committed and the caller takes the segment-sharded run (l3d_shard_chain_run), whose result needs no speculation.
namespace l3d {

struct BlockDigest { unsigned long long hash; int n_kept, R; };      // per view of the chain; zero = not computed by this rank
static_assert(sizeof(BlockDigest) == 16, "digest entry");

// order-sensitive 64-bit digest of a view's kept records: sum over records of a mix of (index, the record's eight words)
__global__ __launch_bounds__(256) void k_block_digest(const Match* __restrict__ arena, const ChainResult* __restrict__ res, int k_begin, BlockDigest* __restrict__ out)
{
    const int k = k_begin + blockIdx.y;
    const ChainResult r = res[k];
    unsigned long long h = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < r.n_kept; i += gridDim.x * 256) {
        const uint4* w = reinterpret_cast<const uint4*>(arena + r.kept_base + i);
        const uint4 a = w[0], b = w[1];
        unsigned long long x = 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
        const unsigned v[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
#pragma unroll
        for (int q = 0; q < 8; ++q) { x ^= v[q]; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; }
        h += x;
    }
    for (int o = 32; o > 0; o >>= 1) h += __shfl_down(h, o);
    if ((threadIdx.x & 63) == 0 && h) atomicAdd(&out[k].hash, h);
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[k].n_kept = r.n_kept; out[k].R = r.R; }
}

}  // namespace l3d

namespace l3d {

// one source's records that point at an early-return view, in list order (stable): out == nullptr counts only
__global__ __launch_bounds__(256) void k_early_pack(const Match* __restrict__ arena, const ChainResult* __restrict__ res, const int* __restrict__ group_src,
                                                    const unsigned* __restrict__ early_ids, int n_early, const long long* __restrict__ out_off, Match* __restrict__ out, int* __restrict__ counts)
{
    __shared__ int s_w[4];
    const ChainResult r = res[group_src[blockIdx.x]];
    const Match* kept = arena + r.kept_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long off = out ? out_off[blockIdx.x] : 0;
    int total = 0;
    for (int b = 0; b < r.n_kept; b += 256) {
        const int i = b + tid;
        Match m;
        bool on = false;
        if (i < r.n_kept) { m = kept[i]; for (int e = 0; e < n_early; ++e) on = on || m.camID2 == early_ids[e]; }
        const unsigned long long bal = __ballot(on);
        if (lane == 0) s_w[wave] = __popcll(bal);
        __syncthreads();
        int before = 0, all = 0;
        for (int w = 0; w < 4; ++w) { if (w < wave) before += s_w[w]; all += s_w[w]; }
        if (on && out) out[off + before + __popcll(bal & ((1ull << lane) - 1ull))] = m;
        off += all; total += all;
        __syncthreads();
    }
    if (tid == 0 && !out) counts[blockIdx.x] = total;
}

// one view's best matches as a package other ranks can build its hypotheses from: [view, R, n, S | position of every segment's best match in the
// compact list or -1 | best depth pairs | the best records, compact, in segment order (room for S)]
__global__ __launch_bounds__(256) void k_alias_pack(const Match* __restrict__ arena, const ChainResult* __restrict__ res, const int* __restrict__ pk_view, const int* __restrict__ pk_S,
                                                    const long long* __restrict__ pk_best_off, const long long* __restrict__ pk_out_off, const float2* __restrict__ best_all,
                                                    const int* __restrict__ bestpos_all, unsigned char* __restrict__ out)
{
    __shared__ int s_w[4];
    const int k = pk_view[blockIdx.x], S = pk_S[blockIdx.x];
    const ChainResult r = res[k];
    const float2* best = best_all + pk_best_off[blockIdx.x];
    const int* bestpos = bestpos_all + pk_best_off[blockIdx.x];
    unsigned char* o = out + pk_out_off[blockIdx.x];
    int* o_pos = reinterpret_cast<int*>(o + 16);
    float2* o_best = reinterpret_cast<float2*>(o + 16 + (size_t)S * 4);
    Match* o_rec = reinterpret_cast<Match*>(o + 16 + (size_t)S * 12);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int off = 0;
    for (int b = 0; b < S; b += 256) {
        const int s = b + tid;
        const int p = s < S ? bestpos[s] : -1;
        const bool on = p >= 0 && r.n_kept > 0;
        const unsigned long long bal = __ballot(on);
        if (lane == 0) s_w[wave] = __popcll(bal);
        __syncthreads();
        int before = 0, all = 0;
        for (int w = 0; w < 4; ++w) { if (w < wave) before += s_w[w]; all += s_w[w]; }
        if (s < S) {
            const int q = on ? off + before + __popcll(bal & ((1ull << lane) - 1ull)) : -1;
            o_pos[s] = q; o_best[s] = best[s];
            if (on) o_rec[q] = arena[r.kept_base + p];
        }
        off += all;
        __syncthreads();
    }
    if (tid == 0) { int* h = reinterpret_cast<int*>(o); h[0] = k; h[1] = r.R; h[2] = off; h[3] = S; }
}

}  // namespace l3d

namespace {

// `used` bytes of the kept arena kept across a growth (DevBuf::reserve drops the content)
int arena_grow_keep(l3d_ctx* c, size_t records, size_t used_records)
{
    if (records * sizeof(Match) <= c->ch_kept.cap) return L3D_OK;
    const size_t want = records * sizeof(Match) + records * sizeof(Match) / 8 + 4096;
    void* np = nullptr;
    if (hipMalloc(&np, want) != hipSuccess) { (void)hipGetLastError(); return fail(c, L3D_ERR_NOMEM, "growing the kept arena to " + std::to_string(want >> 20) + " MB"); }
    if (used_records && c->ch_kept.p) HIPCHK(c, hipMemcpy(np, c->ch_kept.p, used_records * sizeof(Match), hipMemcpyDeviceToDevice));
    if (c->ch_kept.p) (void)hipFree(c->ch_kept.p);
    c->ch_kept.p = np; c->ch_kept.cap = want;
    return L3D_OK;
}

}  // namespace

// partition = 0: every rank ends up with the whole arena and the whole table (replicas for the finishing stages: l3d_match_chain_blocks).
// partition = 1: nothing is replicated (l3d_match_chain_partition, include/line3d_amd.h).
static int chain_blocks_impl(l3d_ctx* c, const l3d_chain_view* views, int n_views, const l3d_dense_map* map, l3d_chain_summary* summary, int64_t* n_pot,
                             int rank, int world, int warmup_views, int window, l3d_exchange_fn exchange, void* exchange_user, int* verdict, int partition)
{
    if (!c) return L3D_ERR_INVALID;
    if (!views || n_views <= 0 || !map || !summary || !exchange || !verdict || world < 1 || rank < 0 || rank >= world || warmup_views < 0 || window < 0)
        return fail(c, L3D_ERR_INVALID, "l3d_match_chain_blocks: bad argument");
    *verdict = 1;
    if (n_pot) *n_pot = 0;
    c->products.valid = false;
    c->products.part = ProductsPart();
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    auto block_begin = [&](int r) { return (int)(((long long)n_views * r) / world); };
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    // reach: the largest distance, in chain positions, between a view and one of its neighbours (every rank computes the same number)
    int reach = window;
    {
        std::vector<std::pair<unsigned, int>> idx((size_t)n_views);
        for (int k = 0; k < n_views; ++k) idx[(size_t)k] = { views[k].view_id, k };
        std::sort(idx.begin(), idx.end());
        for (int k = 0; k < n_views; ++k)
            for (int q = 0; q < views[k].N; ++q) {
                if (!views[k].local2global) continue;
                auto it = std::lower_bound(idx.begin(), idx.end(), std::make_pair(views[k].local2global[q], -1));
                if (it != idx.end() && it->first == views[k].local2global[q]) reach = std::max(reach, std::abs(it->second - k));
            }
    }
    // partitioned: a rank runs its chain 2 x reach views PAST its block and must be exact 2 x reach views in front of it -- everything the rows of
    // its block, their flags and the hypotheses they name can depend on is then computed locally (DESIGN.md section 6 iv)
    const int tail = partition ? 2 * reach : 0;
    const int check = partition ? std::max(window, 2 * reach) : window;
    const int own0 = block_begin(rank), own1 = block_begin(rank + 1);
    int first = rank == 0 ? 0 : std::max(0, own0 - warmup_views);
    const int last = std::min(n_views, own1 + tail);
    const double t0 = now_s();
    // ---- this rank's chain: its block and the warm-up views in front of it, started cold
    // (a rank whose chain fails must not leave the others waiting in the first collective: it still publishes its table, with a mark that
    // every rank reads -- they all return an error then, without entering another collective)
    int chain_rc = run_chain(c, views, n_views, nullptr, nullptr, nullptr, nullptr, nullptr, first, last);
    std::string chain_err;
    if (chain_rc) { std::lock_guard<std::mutex> lk(c->err_mu); chain_err = c->err; }
    // partitioned: the job is sized by memory, not by the time of a second pass -- what only a running chain needs (candidate store and its ring,
    // window scratch, bit rows, viewing rays, row counters: 10-15 GB at 4000 segments x 24 neighbours) is given back before the products are built
    auto release_chain_scratch = [&]() {
        if (!partition || c->opt.part_release == 0) return;
        (void)hipStreamSynchronize(st); (void)hipStreamSynchronize(c->stage1_stream);
        DevBuf* b[] = { &c->ch_ringA_meta, &c->ch_ringA_depths, &c->cand_meta, &c->cand_depths, &c->cand_conf, &c->vw_scratch, &c->ch_mask, &c->ch_rays, &c->ch_rowcnt, &c->ch_rowA };
        for (DevBuf* x : b) x->release();
    };
    release_chain_scratch();
    const ChainResult* hres = c->ch_pin_res.as<ChainResult>();
    double t1 = now_s();
    // useful work of this rank = its own block (the warm-up is the price of the speculation)
    {
        double p = 0;
        for (int k = own0; k < own1; ++k)
            for (int j = 0; j < views[k].n_tbm; ++j) p += (double)views[k].S_src * views[k].offsets[2 * views[k].to_be_matched[j] + 1];
        c->stats[0] = p;
    }
    // Everything a rank can fail in ON ITS OWN between two collectives (an allocation, a copy, a launch) is collected in local_rc and travels
    // with the next exchange: a failed rank still enters it, with a mark every rank reads, and then all of them return -- nobody is left waiting.
    const size_t tab_bytes = al((size_t)n_views * sizeof(BlockDigest));
    int local_rc = chain_rc;
    std::string local_err = chain_err;
    auto note = [&](int rc) { if (rc && !local_rc) { local_rc = rc; std::lock_guard<std::mutex> lk(c->err_mu); local_err = c->err; } };
#define L3D_SOFT(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) note(fail(c, L3D_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_))); } while (0)
    {   // (sized once for the digest tables AND the status words: no reallocation between collectives)
        hipError_t e = c->ch_hdr.reserve(tab_bytes * (size_t)(world + 1) + 512 * (size_t)(world + 2) + 256);
        if (e != hipSuccess) return fail(c, L3D_ERR_NOMEM, "l3d_match_chain_blocks: digest tables");     // (before the first collective of this call: every rank of the job is configured alike)
    }
    BlockDigest* dtab_own = c->ch_hdr.as<BlockDigest>();
    BlockDigest* dtab_all = reinterpret_cast<BlockDigest*>(c->ch_hdr.as<unsigned char>() + tab_bytes);
    long long* st_own = reinterpret_cast<long long*>(c->ch_hdr.as<unsigned char>() + tab_bytes * (size_t)(world + 1));
    long long* st_all = reinterpret_cast<long long*>(c->ch_hdr.as<unsigned char>() + tab_bytes * (size_t)(world + 1) + 256);
    std::vector<BlockDigest> tab((size_t)world * (size_t)n_views);
    std::vector<long long> words((size_t)world, 0);
    // publishes `mine` (negative = this rank failed with code -mine), reads everybody's; non-zero return: somebody failed (this rank's own message is kept)
    auto all_gather_word = [&](long long mine, const char* what) -> int {
        if (local_rc && mine >= 0) mine = -(long long)local_rc;
        hipError_t e = hipMemcpyAsync(st_own, &mine, 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);                   // (`mine` is a stack word)
        if (e != hipSuccess) { note(fail(c, L3D_ERR_HIP, std::string("l3d_match_chain_blocks: status word: ") + hipGetErrorString(e))); (void)hipMemsetAsync(st_own, 0xff, 8, st); }   // (all ones = -1: failed)
        if (exchange(exchange_user, -3, st_own, st_all, 256, world, (void*)st)) return fail(c, L3D_ERR_HIP, std::string("l3d_match_chain_blocks: the exchange of the status words failed (") + what + ")");
        e = hipSuccess;
        for (int r = 0; r < world && e == hipSuccess; ++r) e = hipMemcpyAsync(&words[(size_t)r], reinterpret_cast<const unsigned char*>(st_all) + (size_t)r * 256, 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return fail(c, L3D_ERR_HIP, std::string("l3d_match_chain_blocks: status word: ") + hipGetErrorString(e));     // (after the collective: nobody waits for this rank any more)
        if (local_rc) return fail(c, local_rc, local_err);
        for (int r = 0; r < world; ++r)
            if (words[(size_t)r] < 0) return fail(c, L3D_ERR_HIP, "l3d_match_chain_blocks: rank " + std::to_string(r) + " failed (code " + std::to_string(-words[(size_t)r]) + ") while " + what);
        return L3D_OK;
    };
    auto owner = [&](int k) { int r = (int)(((long long)k * world) / n_views); while (r + 1 < world && block_begin(r + 1) <= k) ++r; while (r > 0 && block_begin(r) > k) --r; return r; };
    // offsets of the views' slices in the whole-run arrays of best pairs / positions (chain_assign_arenas: verified views back to back)
    std::vector<long long> best_off((size_t)n_views + 1, 0);
    for (int k = 0; k < n_views; ++k) best_off[(size_t)k + 1] = best_off[(size_t)k] + (views[k].n_tbm > 0 ? views[k].S_src : 0);

    // ---- digests of every list this rank computed, all-gathered; the verdict; a block whose speculation failed is re-run WARM from its
    // predecessor's true lists (the first one that failed: everything in front of it is exact), then the digests are exchanged again -- a miss
    // costs one block, not the pass
    int arena_first = first;                // first view whose records sit in this rank's arena
    std::vector<int> comp_from((size_t)world, 0);       // per rank: the first view it computed (or took over); every rank keeps the same table
    for (int r = 1; r < world; ++r) comp_from[(size_t)r] = std::max(0, block_begin(r) - warmup_views);
    double t2 = t1;
    int rounds = 0, blocks_rerun = 0;
    for (;; ++rounds) {
        L3D_SOFT(hipMemsetAsync(dtab_own, 0, tab_bytes, st));
        if (local_rc) {
            BlockDigest mark; mark.hash = ~0ull; mark.n_kept = -1; mark.R = local_rc;
            (void)hipMemcpyAsync(dtab_own, &mark, sizeof(mark), hipMemcpyHostToDevice, st);
            (void)hipStreamSynchronize(st);
        } else if (last > arena_first) {
            L3D_SOFT(hipMemcpyAsync(c->ch_res.p, hres, (size_t)n_views * sizeof(ChainResult), hipMemcpyHostToDevice, st));     // (the final records: a restart rewrites them)
            hipLaunchKernelGGL(k_block_digest, dim3(16, last - arena_first), dim3(256), 0, st, c->ch_kept.as<Match>(), c->ch_res.as<ChainResult>(), arena_first, dtab_own);
        }
        if (exchange(exchange_user, -1, dtab_own, dtab_all, tab_bytes, world, (void*)st)) return fail(c, L3D_ERR_HIP, "l3d_match_chain_blocks: the exchange of the digests failed");
        {
            hipError_t e = hipSuccess;
            for (int r = 0; r < world && e == hipSuccess; ++r)
                e = hipMemcpyAsync(tab.data() + (size_t)r * n_views, reinterpret_cast<const unsigned char*>(dtab_all) + (size_t)r * tab_bytes, (size_t)n_views * sizeof(BlockDigest), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) {
                // this rank cannot read the table: it cannot know the verdict the others reach.  The status-word exchange below (entered by every
                // rank after every digest exchange) carries the failure to them.
                note(fail(c, L3D_ERR_HIP, std::string("l3d_match_chain_blocks: reading the digest tables: ") + hipGetErrorString(e)));
            }
        }
        t2 = now_s();
        if (local_rc == L3D_OK)
            for (int r = 0; r < world; ++r)
                if (tab[(size_t)r * n_views].n_kept == -1 && tab[(size_t)r * n_views].hash == ~0ull) { note(fail(c, L3D_ERR_HIP, "l3d_match_chain_blocks: the chain of rank " + std::to_string(r) + " failed (code " + std::to_string(tab[(size_t)r * n_views].R) + ")")); break; }
        if (int a_rc = all_gather_word(0, "computing its block")) return a_rc;
        // ---- the verdict (the same on every rank: same table).  miss[r]: the `check` views in front of rank r's block did not come out of its warm-up as
        // its predecessor computed them.  No miss anywhere = every rank exact (rank 0 is; rank r's check views equal rank r-1's, which are then the
        // one chain's, and from them on rank r's chain had the one chain's inputs).
        bool hopeless = false;
        int n_miss = 0;
        std::vector<char> miss((size_t)world, 0);
        for (int r = 1; r < world; ++r) {
            const int b = block_begin(r), lo = std::max(0, b - check);
            // a warm-up shorter than the check leaves views the rank never computed: a miss (the warm re-run below takes them over)
            bool ok = lo >= comp_from[(size_t)r];
            // (replicas: a block shorter than the window cannot vouch for its successor's sources -- the blocks are what gets gathered; partitioned: the
            // predecessor's exact range reaches back as far as this rank's check)
            if (!partition && b - check < block_begin(r - 1)) { ok = false; hopeless = true; }
            for (int k = lo; k < b && ok; ++k) {
                const BlockDigest &x = tab[(size_t)r * n_views + k], &y = tab[(size_t)(r - 1) * n_views + k];
                // (the kept LIST must be the same; the number of candidates it was chosen from may differ while the warm-up converges)
                if (x.hash != y.hash || x.n_kept != y.n_kept) {
                    ok = false;
                    if (c->opt.timing) fprintf(stderr, "[l3d chain_blocks] rank %d's warm-up view %d differs from rank %d's (%d vs %d kept)\n", r, k, r - 1, x.n_kept, y.n_kept);
                }
            }
            if (ok && partition) {          // the predecessor's tail against this rank's block: both are the one chain's, or something is wrong
                const int e = std::min(std::min(n_views, b + tail), block_begin(r + 1));
                for (int k = b; k < e && ok; ++k) {
                    const BlockDigest &x = tab[(size_t)r * n_views + k], &y = tab[(size_t)(r - 1) * n_views + k];
                    if (x.hash != y.hash || x.n_kept != y.n_kept) ok = false;
                }
            }
            miss[(size_t)r] = ok ? 0 : 1;
            n_miss += ok ? 0 : 1;
        }
        if (c->opt.timing) fprintf(stderr, "[l3d chain_blocks rank %d/%d] round %d: views %d..%d (block from %d): chain %.2f ms, digests + exchange %.2f ms, %d rank(s) missed\n",
                                   rank, world, rounds, arena_first, last - 1, own0, (t1 - t0) * 1e3, (t2 - t1) * 1e3, n_miss);
        if (n_miss == 0) break;
        // option block_recover = 0 (A/B, tests of the fall-through): the round-4 behaviour -- any miss sends the pass to the caller's other mode
        if (hopeless || c->opt.block_recover == 0 || rounds >= world) return L3D_OK;                  // *verdict = 1: nothing committed
        // ---- recovery, all missed blocks at once: every rank that missed takes over its predecessor's last `check` views (records, best depth
        // pairs, best positions: one all-gather, the one primitive of the protocol) and re-runs its block WARM from them.  A predecessor that missed
        // too hands over what its cold start produced -- usually already the one chain's by the end of a block; if not, the next round's digests
        // show it and that rank runs again.  After round j rank j is exact whatever happened (it took over from rank j-1, exact since round j-1).
        blocks_rerun += n_miss;
        long long max_rec = 0, max_seg = 0;
        std::vector<long long> t_rec((size_t)world, 0), t_seg((size_t)world, 0);
        std::vector<int> t_first((size_t)world, -1);               // (a tail's records are contiguous in the sender's arena from its first verified view on)
        for (int r = 1; r < world; ++r) {
            if (!miss[(size_t)r]) continue;
            const int s_ = r - 1, fb = block_begin(r), k0 = std::max(0, fb - check);
            for (int k = k0; k < fb; ++k) { t_rec[(size_t)s_] += tab[(size_t)s_ * n_views + k].n_kept; if (views[k].n_tbm > 0) { t_seg[(size_t)s_] += views[k].S_src; if (t_first[(size_t)s_] < 0) t_first[(size_t)s_] = k; } }
            max_rec = std::max(max_rec, t_rec[(size_t)s_]); max_seg = std::max(max_seg, t_seg[(size_t)s_]);
        }
        // a tail as one byte stream [records | best depth pairs | best positions], the same offsets on every rank; it travels in CHUNKS (option
        // handover_chunk_kb, 256 MB): an all-gather hands every rank every slot, and a whole tail per slot would cost world x tail bytes on every
        // rank (52 GB at 4000 x 24 x 8 ranks) for the one slot a rank reads -- a chunk per slot costs world x 256 MB
        const size_t o_best = al((size_t)max_rec * sizeof(Match)), o_bpos = o_best + al((size_t)max_seg * 8), slot = o_bpos + al((size_t)max_seg * 4) + 256;
        const size_t chunk = std::min(slot, al((size_t)std::max(1, c->opt.handover_chunk_kb) << 10));
        const bool sends = rank + 1 < world && miss[(size_t)rank + 1], takes = miss[(size_t)rank] != 0;
        {
            hipError_t e = c->ch_send.reserve(chunk + 256);
            if (e == hipSuccess) e = c->ch_gathered.reserve(chunk * (size_t)world + 256);
            if (e == hipSuccess && takes) e = c->ch_stage.reserve(slot + 256);
            if (e != hipSuccess) note(fail(c, L3D_ERR_NOMEM, "l3d_match_chain_blocks: the hand-over of a block's sources"));
        }
        if (int a_rc = all_gather_word(0, "staging the hand-over of a block's sources")) return a_rc;
        for (size_t off = 0; off < slot; off += chunk) {
            const size_t n = std::min(chunk, slot - off);
            if (sends && !local_rc) {
                const int k0 = std::max(0, block_begin(rank + 1) - check);
                struct Piece { size_t at, len; const unsigned char* src; } pieces[3] = {
                    { 0, (size_t)t_rec[(size_t)rank] * sizeof(Match), t_first[(size_t)rank] >= 0 ? reinterpret_cast<const unsigned char*>(c->ch_kept.as<Match>() + hres[t_first[(size_t)rank]].kept_base) : nullptr },
                    { o_best, (size_t)t_seg[(size_t)rank] * 8, reinterpret_cast<const unsigned char*>(c->ch_best.as<float2>() + best_off[(size_t)k0]) },
                    { o_bpos, (size_t)t_seg[(size_t)rank] * 4, reinterpret_cast<const unsigned char*>(c->ch_bestpos.as<int>() + best_off[(size_t)k0]) } };
                for (const Piece& q : pieces) {
                    const size_t lo = std::max(q.at, off), hi = std::min(q.at + q.len, off + n);
                    if (q.src && hi > lo) L3D_SOFT(hipMemcpyAsync(c->ch_send.as<unsigned char>() + (lo - off), q.src + (lo - q.at), hi - lo, hipMemcpyDeviceToDevice, st));
                }
            }
            if (exchange(exchange_user, -5, c->ch_send.p, c->ch_gathered.p, n, world, (void*)st)) return fail(c, L3D_ERR_HIP, "l3d_match_chain_blocks: the hand-over exchange failed");
            if (takes && !local_rc) L3D_SOFT(hipMemcpyAsync(c->ch_stage.as<unsigned char>() + off, c->ch_gathered.as<unsigned char>() + (size_t)(rank - 1) * n, n, hipMemcpyDeviceToDevice, st));
            if (off + chunk < slot) L3D_SOFT(hipStreamSynchronize(st));       // (the next chunk reuses the send and gathered buffers)
        }
        for (int r = 1; r < world; ++r) if (miss[(size_t)r]) comp_from[(size_t)r] = std::max(0, block_begin(r) - check);
        if (miss[(size_t)rank]) {
            const int src_rank = rank - 1, fb = own0, k0 = std::max(0, fb - check);
            ChainPreload pre;
            pre.k0 = k0; pre.k1 = fb;
            const unsigned char* G = c->ch_stage.as<unsigned char>();
            pre.records = reinterpret_cast<const Match*>(G);
            pre.best = reinterpret_cast<const float2*>(G + o_best);
            pre.bestpos = reinterpret_cast<const int*>(G + o_bpos);
            for (int k = k0; k < fb; ++k) { pre.n_kept.push_back(tab[(size_t)src_rank * n_views + k].n_kept); pre.R.push_back(tab[(size_t)src_rank * n_views + k].R); }
            const double tr0 = now_s();
            const int rc = run_chain(c, views, n_views, nullptr, nullptr, nullptr, nullptr, nullptr, fb, last, &pre);
            if (rc) note(rc);
            release_chain_scratch();
            hres = c->ch_pin_res.as<ChainResult>();
            arena_first = k0;
            t1 += now_s() - tr0;
            if (c->opt.timing) fprintf(stderr, "[l3d chain_blocks rank %d/%d] block re-run warm from rank %d's last %d views: %.2f ms\n", rank, world, src_rank, fb - k0, (now_s() - tr0) * 1e3);
        }
    }
    c->products.part.recovery_rounds = rounds; c->products.part.blocks_rerun = blocks_rerun;
    // first view from which this rank's lists are the one chain's
    const int exact_from = rank == 0 ? 0 : std::max(arena_first, own0 - check);
    (void)comp_from;

    const int nvd = map->n_views;
    auto dense_of = [&](int k) {                                   // the dense view a chain view is (ids ascend in both)
        if (k >= n_views) return nvd;
        const uint32_t* it = std::lower_bound(map->view_ids, map->view_ids + nvd, views[k].view_id);
        return (int)(it - map->view_ids);
    };
    Products& P = c->products;

    if (partition) {
        // =========================================================================================================================
        // Nothing is replicated.  This rank holds the exact records of the views [exact_from, last) and builds from them, locally: the rows
        // of the table for the views within `reach` of its block, best matches and medians for every view it holds.  What the early-return
        // quirk (cudawrapper.cu:877-878) files under LOCAL camera numbers read as view ids can name any view of the scene: the records that
        // point at an early-return view -- a sliver of their sources' lists -- are all-gathered, each source's by the rank that owns it.
        std::vector<ChainResult> hloc((size_t)n_views);
        for (int k = 0; k < n_views; ++k) { hloc[(size_t)k] = ChainResult(); if (k >= exact_from && k < last) hloc[(size_t)k] = hres[k]; }
        long long used = 0;
        for (int k = exact_from; k < last; ++k) used = std::max(used, (long long)hres[k].kept_base + hres[k].n_kept);
        // ---- early-return views: groups = their sources; sender = the owner of the source
        std::vector<unsigned> early_ids;
        std::vector<int> groups_all;                               // every source of an early-return view, ascending, unique
        for (int k = 0; k < n_views; ++k) {
            if (views[k].n_tbm != 0 || views[k].n_sources == 0) continue;
            early_ids.push_back(views[k].view_id);
            for (int q = 0; q < views[k].n_sources; ++q) if (views[k].source_index[q] >= 0 && views[k].source_index[q] < k && views[views[k].source_index[q]].n_tbm > 0) groups_all.push_back(views[k].source_index[q]);
        }
        std::sort(groups_all.begin(), groups_all.end());
        groups_all.erase(std::unique(groups_all.begin(), groups_all.end()), groups_all.end());
        if (early_ids.size() > 64 || groups_all.size() > 480) return L3D_OK;    // (a schedule full of early returns: the replicated mode takes it; the same on every rank)
        if (!groups_all.empty() && world > 1) {
            std::vector<int> mine;
            for (int si : groups_all) if (owner(si) == rank) mine.push_back(si);
            const size_t o_ids = 0, o_grp = 256, o_off = o_grp + 2048, o_cnt = o_off + 4096, ctl = o_cnt + 2048;
            std::vector<int> cnt(mine.size(), 0);
            long long total = 0;
            {
                hipError_t e = c->ch_send.reserve(ctl + 256);
                if (e != hipSuccess) note(fail(c, L3D_ERR_NOMEM, "l3d_match_chain_partition: early-return control block"));
                else if (!mine.empty()) {
                    unsigned char* S0 = c->ch_send.as<unsigned char>();
                    L3D_SOFT(hipMemcpyAsync(S0 + o_ids, early_ids.data(), early_ids.size() * 4, hipMemcpyHostToDevice, st));
                    L3D_SOFT(hipMemcpyAsync(S0 + o_grp, mine.data(), mine.size() * 4, hipMemcpyHostToDevice, st));
                    hipLaunchKernelGGL(k_early_pack, dim3((unsigned)mine.size()), dim3(256), 0, st, c->ch_kept.as<Match>(), c->ch_res.as<ChainResult>(), reinterpret_cast<const int*>(S0 + o_grp),
                                       reinterpret_cast<const unsigned*>(S0 + o_ids), (int)early_ids.size(), (const long long*)nullptr, (Match*)nullptr, reinterpret_cast<int*>(S0 + o_cnt));
                    L3D_SOFT(hipMemcpyAsync(cnt.data(), S0 + o_cnt, mine.size() * 4, hipMemcpyDeviceToHost, st));
                    L3D_SOFT(hipStreamSynchronize(st));
                    for (int x : cnt) total += x;
                }
            }
            if (int a_rc = all_gather_word(total, "counting the records that point at early-return views")) return a_rc;
            long long max_total = 0;
            for (int r = 0; r < world; ++r) max_total = std::max(max_total, words[(size_t)r]);
            const size_t hdr = 4096, eslot = hdr + al((size_t)max_total * sizeof(Match)) + 256;
            {
                // slot: [int n_groups | (int source, int count) x n_groups] [records of the groups, back to back]; staged in ch_stage (ch_send holds the control block)
                hipError_t e = c->ch_stage.reserve(eslot + 256);
                if (e == hipSuccess) e = c->ch_gathered.reserve(eslot * (size_t)world + 256);
                if (e != hipSuccess) note(fail(c, L3D_ERR_NOMEM, "l3d_match_chain_partition: early-return slots"));
                else {
                    std::vector<int> h((size_t)1 + 2 * mine.size(), 0);
                    std::vector<long long> offs(mine.size(), 0);
                    h[0] = (int)mine.size();
                    long long o = 0;
                    for (size_t g = 0; g < mine.size(); ++g) { h[1 + 2 * g] = mine[g]; h[2 + 2 * g] = cnt[g]; offs[g] = o; o += cnt[g]; }
                    unsigned char* S0 = c->ch_send.as<unsigned char>();
                    unsigned char* E0 = c->ch_stage.as<unsigned char>();
                    L3D_SOFT(hipMemcpyAsync(E0, h.data(), h.size() * 4, hipMemcpyHostToDevice, st));
                    if (total > 0) {
                        L3D_SOFT(hipMemcpyAsync(S0 + o_off, offs.data(), offs.size() * 8, hipMemcpyHostToDevice, st));
                        hipLaunchKernelGGL(k_early_pack, dim3((unsigned)mine.size()), dim3(256), 0, st, c->ch_kept.as<Match>(), c->ch_res.as<ChainResult>(), reinterpret_cast<const int*>(S0 + o_grp),
                                           reinterpret_cast<const unsigned*>(S0 + o_ids), (int)early_ids.size(), reinterpret_cast<const long long*>(S0 + o_off), reinterpret_cast<Match*>(E0 + hdr), (int*)nullptr);
                    }
                    L3D_SOFT(hipStreamSynchronize(st));        // (h / offs are stack vectors)
                }
            }
            if (int a_rc = all_gather_word(0, "staging the records that point at early-return views")) return a_rc;
            if (exchange(exchange_user, -6, c->ch_stage.p, c->ch_gathered.p, eslot, world, (void*)st)) return fail(c, L3D_ERR_HIP, "l3d_match_chain_partition: the exchange of the early-return records failed");
            // a source this rank does not hold gets a list of its own: just those records, in list order
            const unsigned char* G = c->ch_gathered.as<unsigned char>();
            std::vector<int> hh(1024);
            for (int r = 0; r < world && !local_rc; ++r) {
                if (r == rank) continue;
                L3D_SOFT(hipMemcpyAsync(hh.data(), G + (size_t)r * eslot, hdr, hipMemcpyDeviceToHost, st));
                L3D_SOFT(hipStreamSynchronize(st));
                if (local_rc) break;
                long long o = 0;
                for (int g = 0; g < hh[0] && g < 480; ++g) {
                    const int si = hh[1 + 2 * g], n = hh[2 + 2 * g];
                    if (si >= 0 && si < n_views && n > 0 && !(si >= exact_from && si < last)) {
                        if (int rc = arena_grow_keep(c, (size_t)(used + n) + 64, (size_t)used)) { note(rc); break; }
                        L3D_SOFT(hipMemcpyAsync(c->ch_kept.as<Match>() + used, G + (size_t)r * eslot + hdr + (size_t)o * sizeof(Match), (size_t)n * sizeof(Match), hipMemcpyDeviceToDevice, st));
                        ChainResult& x = hloc[(size_t)si];
                        x.kept_base = (unsigned)used; x.n_kept = n; x.R = n; x.overflow = 0;
                        used += n;
                    }
                    o += n;
                }
            }
            if (used > 0xfffffff0ll) note(fail(c, L3D_ERR_UNSUPPORTED, "l3d_match_chain_partition: more than 2^32 kept matches on one rank"));
        }
        // ---- the views an early return's LOCAL camera numbers name (cudawrapper.cu:877-878 hands the list back with local numbers, line3D.cc:861-865
        // files the entries under them read as view ids): rows of an early-return view point at their segments, whoever holds them.  The affinity fill
        // only asks whether such a segment has a hypothesis and where it stands in the order -- the views' best matches (one record per segment:
        // a few hundred KB per view) are all-gathered, each view's by the rank that owns it
        std::vector<char> alias_known((size_t)n_views, 0);
        if (!early_ids.empty() && world > 1) {
            std::vector<std::pair<unsigned, int>> idx((size_t)n_views);
            for (int k = 0; k < n_views; ++k) idx[(size_t)k] = { views[k].view_id, k };
            std::sort(idx.begin(), idx.end());
            std::vector<int> alias;
            for (int k = 0; k < n_views; ++k) {
                if (views[k].n_tbm != 0 || views[k].n_sources == 0) continue;
                for (int q = 0; q < views[k].n_sources; ++q) {
                    auto it = std::lower_bound(idx.begin(), idx.end(), std::make_pair((unsigned)views[k].source_cam[q], -1));
                    if (it != idx.end() && it->first == (unsigned)views[k].source_cam[q] && views[it->second].n_tbm > 0) alias.push_back(it->second);
                }
            }
            std::sort(alias.begin(), alias.end());
            alias.erase(std::unique(alias.begin(), alias.end()), alias.end());
            auto pk_bytes = [&](int b) { return al(16 + (size_t)views[b].S_src * (12 + sizeof(Match))); };
            std::vector<size_t> slot_of((size_t)world, 0);
            for (int b : alias) slot_of[(size_t)owner(b)] += pk_bytes(b);
            size_t aslot = 256;
            for (int r = 0; r < world; ++r) aslot = std::max(aslot, slot_of[(size_t)r] + 256);
            if (!alias.empty()) {
                std::vector<int> pk_view, pk_S;
                std::vector<long long> pk_bo, pk_oo;
                size_t o = 0;
                for (int b : alias) if (owner(b) == rank) { pk_view.push_back(b); pk_S.push_back(views[b].S_src); pk_bo.push_back(best_off[(size_t)b]); pk_oo.push_back((long long)o); o += pk_bytes(b); }
                {
                    const size_t n = pk_view.size(), ctl = al(n * 4) * 2 + al(n * 8) * 2 + 256;
                    hipError_t e = c->ch_stage.reserve(aslot + 256);
                    if (e == hipSuccess) e = c->ch_gathered.reserve(aslot * (size_t)world + 256);
                    if (e == hipSuccess) e = c->ch_send.reserve(ctl);
                    if (e != hipSuccess) note(fail(c, L3D_ERR_NOMEM, "l3d_match_chain_partition: alias-view packages"));
                    else if (n > 0) {
                        unsigned char* S0 = c->ch_send.as<unsigned char>();
                        const size_t o1 = al(n * 4), o2 = 2 * al(n * 4), o3 = o2 + al(n * 8);
                        L3D_SOFT(hipMemcpyAsync(S0, pk_view.data(), n * 4, hipMemcpyHostToDevice, st));
                        L3D_SOFT(hipMemcpyAsync(S0 + o1, pk_S.data(), n * 4, hipMemcpyHostToDevice, st));
                        L3D_SOFT(hipMemcpyAsync(S0 + o2, pk_bo.data(), n * 8, hipMemcpyHostToDevice, st));
                        L3D_SOFT(hipMemcpyAsync(S0 + o3, pk_oo.data(), n * 8, hipMemcpyHostToDevice, st));
                        hipLaunchKernelGGL(k_alias_pack, dim3((unsigned)n), dim3(256), 0, st, c->ch_kept.as<Match>(), c->ch_res.as<ChainResult>(), reinterpret_cast<const int*>(S0), reinterpret_cast<const int*>(S0 + o1),
                                           reinterpret_cast<const long long*>(S0 + o2), reinterpret_cast<const long long*>(S0 + o3), c->ch_best.as<float2>(), c->ch_bestpos.as<int>(), c->ch_stage.as<unsigned char>());
                        L3D_SOFT(hipStreamSynchronize(st));        // (the control vectors are on the stack)
                    }
                }
                if (int a_rc = all_gather_word(0, "packing the best matches of the views early returns name")) return a_rc;
                if (exchange(exchange_user, -11, c->ch_stage.p, c->ch_gathered.p, aslot, world, (void*)st)) return fail(c, L3D_ERR_HIP, "l3d_match_chain_partition: the exchange of the alias views' best matches failed");
                const unsigned char* G = c->ch_gathered.as<unsigned char>();
                std::vector<size_t> at((size_t)world, 0);
                for (int b : alias) {
                    const int r = owner(b);
                    const unsigned char* pk = G + (size_t)r * aslot + at[(size_t)r];
                    at[(size_t)r] += pk_bytes(b);
                    if (local_rc || (b >= exact_from && b < last)) continue;              // (held: its own records say it all)
                    int h[4] = { 0, 0, 0, 0 };
                    L3D_SOFT(hipMemcpyAsync(h, pk, 16, hipMemcpyDeviceToHost, st));
                    L3D_SOFT(hipStreamSynchronize(st));
                    const int S = views[b].S_src;
                    if (local_rc) break;
                    if (h[0] != b || h[3] != S || h[2] < 0 || h[2] > S) { note(fail(c, L3D_ERR_INVALID, "l3d_match_chain_partition: a package of best matches does not name the view it should")); break; }
                    //  a view this rank does not hold can be BOTH a source of an early-return view -- its list here is then the all-gathered sliver of
                    // records that point at that view -- and a view an early return's local numbers name: one list cannot be both, and the best matches
                    // would replace the sliver silently.  No scene of the tests does this; a job that does is refused here, on every rank alike (the
                    // status words carry it), and takes the segment-sharded partition (l3d_shard_chain_partition), which keeps such views whole
                    if (hloc[(size_t)b].n_kept > 0) { note(fail(c, L3D_ERR_UNSUPPORTED, "l3d_match_chain_partition: view " + std::to_string(views[b].view_id) + " is a source of an early-return view and named by an early return's local camera number: use the segment-sharded partition")); break; }
                    if (int rc = arena_grow_keep(c, (size_t)(used + h[2]) + 64, (size_t)used)) { note(rc); break; }
                    if (h[2] > 0) L3D_SOFT(hipMemcpyAsync(c->ch_kept.as<Match>() + used, pk + 16 + (size_t)S * 12, (size_t)h[2] * sizeof(Match), hipMemcpyDeviceToDevice, st));
                    L3D_SOFT(hipMemcpyAsync(c->ch_bestpos.as<int>() + best_off[(size_t)b], pk + 16, (size_t)S * 4, hipMemcpyDeviceToDevice, st));
                    L3D_SOFT(hipMemcpyAsync(c->ch_best.as<float2>() + best_off[(size_t)b], pk + 16 + (size_t)S * 4, (size_t)S * 8, hipMemcpyDeviceToDevice, st));
                    ChainResult& x = hloc[(size_t)b];
                    x.kept_base = (unsigned)used; x.n_kept = h[2]; x.R = h[1]; x.overflow = 0;
                    used += h[2];
                    alias_known[(size_t)b] = 1;
                }
            }
        }
        // ---- the local products
        std::vector<ProdChainView> pvh((size_t)n_views);
        for (int k = 0; k < n_views; ++k) {
            const bool ver = views[k].n_tbm > 0, held = (k >= exact_from && k < last) || alias_known[(size_t)k];
            pvh[(size_t)k].verified = ver ? 1 : 0;
            pvh[(size_t)k].best = ver && held ? c->ch_best.as<float2>() + best_off[(size_t)k] : nullptr;
            pvh[(size_t)k].bestpos = ver && held ? c->ch_bestpos.as<int>() + best_off[(size_t)k] : nullptr;
        }
        const int row0 = std::max(rank == 0 ? 0 : exact_from, own0 - reach), row1 = std::min(last, own1 + reach);
        ProductsPart part;
        part.active = true; part.rank = rank; part.world = world;
        part.own_dv0 = dense_of(own0); part.own_dv1 = dense_of(own1);
        part.row_dv0 = dense_of(row0); part.row_dv1 = dense_of(row1);
        part.held_dv0 = dense_of(exact_from); part.held_dv1 = dense_of(last);
        part.recovery_rounds = rounds; part.blocks_rerun = blocks_rerun;
        int64_t n_local = 0;
        const double t3 = now_s();
        if (!local_rc) {
            std::vector<char> held((size_t)n_views, 0);
            for (int k = exact_from; k < last; ++k) held[(size_t)k] = 1;
            note(build_products(c, views, n_views, pvh.data(), hloc.data(), map, summary, &n_local, part.row_dv0, part.row_dv1, held.data()));
        }
        if (int a_rc = all_gather_word(n_local, "building its rows of the products")) return a_rc;
        part.n_pot_all = 0;
        for (int r = 0; r < world; ++r) part.n_pot_all += words[(size_t)r];
        P.part = part;
        P.n_pot = n_local;
        P.valid = true;
        if (n_pot) *n_pot = n_local;
        memcpy(c->ch_pin_res.as<ChainResult>(), hloc.data(), (size_t)n_views * sizeof(ChainResult));       // (what l3d_chain_kept_list reads)
        { double kept = 0, raw = 0; for (int k = own0; k < own1; ++k) { kept += hloc[(size_t)k].n_kept; raw += hloc[(size_t)k].R; } c->stats[3] = kept; c->stats[1] = raw; }      // (this rank's block)
        if (c->opt.timing) fprintf(stderr, "[l3d chain_partition rank %d/%d] exact from view %d, chain to view %d, rows of views %d..%d, %lld potential correspondences here of %lld: products %.2f ms\n",
                                   rank, world, exact_from, last - 1, row0, row1 - 1, (long long)n_local, part.n_pot_all, (now_s() - t3) * 1e3);
        *verdict = 0;
        return L3D_OK;
    }

    // ---- all-gather of the blocks: [records of the block's views][best depth pairs][best positions], padded to the largest block
    std::vector<long long> rec_of((size_t)world, 0), seg_of((size_t)world, 0);
    for (int k = 0; k < n_views; ++k) {
        const int r = owner(k);
        rec_of[(size_t)r] += tab[(size_t)r * n_views + k].n_kept;
        if (views[k].n_tbm > 0) seg_of[(size_t)r] += views[k].S_src;
    }
    long long max_rec = 0, max_seg = 0, total = 0;
    for (int r = 0; r < world; ++r) { max_rec = std::max(max_rec, rec_of[(size_t)r]); max_seg = std::max(max_seg, seg_of[(size_t)r]); total += rec_of[(size_t)r]; }
    if (total > 0xfffffff0ll) return fail(c, L3D_ERR_UNSUPPORTED, "l3d_match_chain_blocks: more than 2^32 kept matches (l3d_match_chain_partition keeps every rank's records where they are)");     // (the same on every rank)
    const size_t o_best = al((size_t)max_rec * sizeof(Match)), o_bpos = o_best + al((size_t)max_seg * 8), slot = o_bpos + al((size_t)max_seg * 4);
    {
        hipError_t e = c->ch_send.reserve(slot + 256);
        if (e == hipSuccess) e = c->ch_gathered.reserve(slot * (size_t)world + 256);
        if (e != hipSuccess) note(fail(c, L3D_ERR_NOMEM, "l3d_match_chain_blocks: the blocks' slots"));
        else {
            unsigned char* send = c->ch_send.as<unsigned char>();
            long long own_start = 0;                                    // (this rank's arena: the views it holds, back to back from arena_first)
            for (int k = arena_first; k < own0; ++k) own_start += hres[k].n_kept;
            if (rec_of[(size_t)rank] > 0)
                L3D_SOFT(hipMemcpyAsync(send, c->ch_kept.as<Match>() + own_start, (size_t)rec_of[(size_t)rank] * sizeof(Match), hipMemcpyDeviceToDevice, st));
            if (seg_of[(size_t)rank] > 0) {
                L3D_SOFT(hipMemcpyAsync(send + o_best, c->ch_best.as<float2>() + best_off[(size_t)own0], (size_t)seg_of[(size_t)rank] * 8, hipMemcpyDeviceToDevice, st));
                L3D_SOFT(hipMemcpyAsync(send + o_bpos, c->ch_bestpos.as<int>() + best_off[(size_t)own0], (size_t)seg_of[(size_t)rank] * 4, hipMemcpyDeviceToDevice, st));
            }
        }
    }
    if (int a_rc = all_gather_word(0, "staging its block")) return a_rc;
    if (exchange(exchange_user, -2, c->ch_send.p, c->ch_gathered.p, slot, world, (void*)st)) return fail(c, L3D_ERR_HIP, "l3d_match_chain_blocks: the exchange of the kept lists failed");
    // ---- the one chain's arena: blocks in rank order = views in order; then matchViews' products: every rank builds the rows of its OWN block
    // of views (sort + unique of the keys whose source lies in the block: its views' records and their neighbours', all of them in the arena now),
    // the pieces are all-gathered and put together -- 1/world of the sort per rank instead of all of it on every rank
    std::vector<ChainResult> hres_all((size_t)n_views);
    std::vector<ProdChainView> pvh((size_t)n_views);
    std::vector<int> dvb((size_t)world + 1);
    for (int r = 0; r <= world; ++r) dvb[(size_t)r] = r == 0 ? 0 : (r == world ? nvd : dense_of(block_begin(r)));
    for (int r = 1; r <= world; ++r) if (dvb[(size_t)r] < dvb[(size_t)r - 1]) return fail(c, L3D_ERR_INVALID, "l3d_match_chain_blocks: the chain's views do not ascend with the dense map");     // (the same on every rank)
    int64_t n_local = 0;
    double t3 = 0;
    const auto assemble_and_build = [&]() -> int {
        HIPCHK(c, hipStreamSynchronize(st));            // (the arena below may be reallocated: everything that reads the old one is done)
        HIPCHK(c, c->ch_kept.reserve(((size_t)total + 64) * sizeof(Match)));
        long long base = 0;
        for (int k = 0; k < n_views; ++k) {
            const BlockDigest& e = tab[(size_t)owner(k) * n_views + k];
            ChainResult& r = hres_all[(size_t)k];
            r.kept_base = (unsigned)base; r.n_kept = e.n_kept; r.R = e.R; r.overflow = 0;
            base += e.n_kept;
            const bool ver = views[k].n_tbm > 0;
            pvh[(size_t)k].verified = ver ? 1 : 0;
            pvh[(size_t)k].best = ver ? c->ch_best.as<float2>() + best_off[(size_t)k] : nullptr;
            pvh[(size_t)k].bestpos = ver ? c->ch_bestpos.as<int>() + best_off[(size_t)k] : nullptr;
        }
        long long at = 0;
        const unsigned char* G = c->ch_gathered.as<unsigned char>();
        for (int r = 0; r < world; ++r) {
            const int b0 = block_begin(r);
            if (rec_of[(size_t)r]) HIPCHK(c, hipMemcpyAsync(c->ch_kept.as<Match>() + at, G + (size_t)r * slot, (size_t)rec_of[(size_t)r] * sizeof(Match), hipMemcpyDeviceToDevice, st));
            if (seg_of[(size_t)r]) {
                HIPCHK(c, hipMemcpyAsync(c->ch_best.as<float2>() + best_off[(size_t)b0], G + (size_t)r * slot + o_best, (size_t)seg_of[(size_t)r] * 8, hipMemcpyDeviceToDevice, st));
                HIPCHK(c, hipMemcpyAsync(c->ch_bestpos.as<int>() + best_off[(size_t)b0], G + (size_t)r * slot + o_bpos, (size_t)seg_of[(size_t)r] * 4, hipMemcpyDeviceToDevice, st));
            }
            at += rec_of[(size_t)r];
        }
        t3 = now_s();
        return build_products(c, views, n_views, pvh.data(), hres_all.data(), map, summary, &n_local, dvb[(size_t)rank], dvb[(size_t)rank + 1]);
    };
    // counts first (a piece is padded to the largest; a negative count = this rank failed), then [row starts of the block, numbered from 0 | entries]
    { note(assemble_and_build()); if (int a_rc = all_gather_word((long long)n_local, "building its rows of the products")) return a_rc; }
    const std::vector<long long> cnts = words;
    long long max_cnt = 0, max_rows = 0, n_pot_all = 0;
    for (int r = 0; r < world; ++r) {
        max_cnt = std::max(max_cnt, cnts[(size_t)r]); n_pot_all += cnts[(size_t)r];
        max_rows = std::max(max_rows, (long long)map->seg_base[dvb[(size_t)r + 1]] - map->seg_base[dvb[(size_t)r]]);
    }
    const size_t o_ent = al((size_t)max_rows * 8), pslot = o_ent + al((size_t)max_cnt * 4 + 4);
    {
        hipError_t e = c->ch_send.reserve(pslot + 256);
        if (e == hipSuccess) e = c->ch_gathered.reserve(pslot * (size_t)world + 256);
        if (e != hipSuccess) note(fail(c, L3D_ERR_NOMEM, "l3d_match_chain_blocks: the pieces' slots"));
        else {
            const long long r0 = map->seg_base[dvb[(size_t)rank]], nr = (long long)map->seg_base[dvb[(size_t)rank + 1]] - r0;
            unsigned char* sp = c->ch_send.as<unsigned char>();
            if (nr > 0) L3D_SOFT(hipMemcpyAsync(sp, P.pot_start.as<long long>() + r0, (size_t)nr * 8, hipMemcpyDeviceToDevice, st));
            if (n_local > 0) L3D_SOFT(hipMemcpyAsync(sp + o_ent, P.pot_tgt.p, (size_t)n_local * 4, hipMemcpyDeviceToDevice, st));
        }
    }
    if (int a_rc = all_gather_word(0, "staging its piece of the products")) return a_rc;
    if (exchange(exchange_user, -4, c->ch_send.p, c->ch_gathered.p, pslot, world, (void*)st)) return fail(c, L3D_ERR_HIP, "l3d_match_chain_blocks: the exchange of the table pieces failed");
    // (past the last collective: a failure from here on is this rank's alone)
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, P.pot_tgt.reserve(((size_t)n_pot_all + 2) * 4));
    {
        const unsigned char* G = c->ch_gathered.as<unsigned char>();
        long long base = 0;
        for (int r = 0; r < world; ++r) {
            const long long r0 = map->seg_base[dvb[(size_t)r]], nr = (long long)map->seg_base[dvb[(size_t)r + 1]] - r0;
            launch_prod_shift_rows(reinterpret_cast<const long long*>(G + (size_t)r * pslot), nr, base, P.pot_start.as<long long>() + r0, st);
            if (cnts[(size_t)r]) HIPCHK(c, hipMemcpyAsync(P.pot_tgt.as<int>() + base, G + (size_t)r * pslot + o_ent, (size_t)cnts[(size_t)r] * 4, hipMemcpyDeviceToDevice, st));
            base += cnts[(size_t)r];
        }
        HIPCHK(c, hipMemcpyAsync(P.pot_start.as<long long>() + map->seg_base[nvd], &n_pot_all, 8, hipMemcpyHostToDevice, st));     // the closing row start
        HIPCHK(c, hipStreamSynchronize(st));
        HIPCHK(c, hipGetLastError());
    }
    P.n_pot = n_pot_all;
    P.valid = true;
    if (n_pot) *n_pot = n_pot_all;
    memcpy(c->ch_pin_res.as<ChainResult>(), hres_all.data(), (size_t)n_views * sizeof(ChainResult));       // (what l3d_chain_kept_list reads)
    c->stats[3] = (double)total;
    { double raw = 0; for (int k = own0; k < own1; ++k) raw += tab[(size_t)rank * n_views + k].R; c->stats[1] = raw; }      // (this rank's useful share)
    if (c->opt.timing) fprintf(stderr, "[l3d chain_blocks rank %d/%d] gather of the blocks %.2f ms, products (own rows + gather of the pieces) %.2f ms\n", rank, world, (t3 - t2) * 1e3, (now_s() - t3) * 1e3);
    *verdict = 0;
    return L3D_OK;
#undef L3D_SOFT
}

extern "C" int l3d_match_chain_blocks(l3d_ctx* c, const l3d_chain_view* views, int n_views, const l3d_dense_map* map, l3d_chain_summary* summary, int64_t* n_pot,
                                      int rank, int world, int warmup_views, int window, l3d_exchange_fn exchange, void* exchange_user, int* verdict)
{
    return chain_blocks_impl(c, views, n_views, map, summary, n_pot, rank, world, warmup_views, window, exchange, exchange_user, verdict, 0);
}

extern "C" int l3d_match_chain_partition(l3d_ctx* c, const l3d_chain_view* views, int n_views, const l3d_dense_map* map, l3d_chain_summary* summary, int64_t* n_pot,
                                         int rank, int world, int warmup_views, int window, l3d_exchange_fn exchange, void* exchange_user, int* verdict)
{
    return chain_blocks_impl(c, views, n_views, map, summary, n_pot, rank, world, warmup_views, window, exchange, exchange_user, verdict, 1);
}

extern "C" int l3d_partition_info(l3d_ctx* c, int info[10], int64_t* n_pot_all)
{
    if (!c) return L3D_ERR_INVALID;
    const ProductsPart& q = c->products.part;
    if (info) { const int v[10] = { q.rank, q.world, q.own_dv0, q.own_dv1, q.row_dv0, q.row_dv1, q.held_dv0, q.held_dv1, q.recovery_rounds, q.blocks_rerun }; memcpy(info, v, sizeof(v)); }
    if (n_pot_all) *n_pot_all = q.active ? q.n_pot_all : c->products.n_pot;
    return L3D_OK;                      // (info[1] = world of the partition; not partitioned: the defaults, world 1)
}

void l3d::warm_chain() { touch_kernel(reinterpret_cast<const void*>(&k_exist_count)); }
