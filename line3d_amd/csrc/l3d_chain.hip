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

#ifndef L3D_AHEAD
#define L3D_AHEAD 4
#define L3D_S1AHEAD 8
#endif

using namespace l3d;

namespace l3d {

// reverse matches for view `view_id` out of the kept lists of earlier views (blockIdx.y = source): count per
// (segment, camera) row.  (seg, tgt) swap roles and the depth pairs swap, line3D.cc:847-856.
__global__ void k_exist_count(const Match* __restrict__ arena, const ChainResult* __restrict__ res, const int* __restrict__ src_index,
                              const int* __restrict__ src_cam, unsigned view_id, int N, int S, int* __restrict__ rowcnt)
{
    const ChainResult* src = res + src_index[blockIdx.y];
    const int cam = src_cam[blockIdx.y];
    const int n = src->n_kept;
    const Match* kept = arena + src->kept_base;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Match r = kept[i];
        if (r.camID2 == view_id && (int)r.segID2 < S) atomicAdd(&rowcnt[r.segID2 * N + cam], 1);
    }
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
__global__ __launch_bounds__(256) void k_place(int blocks_move, const int* __restrict__ tbm, int n_tbm, const int* __restrict__ rowA,
                                               const uint2* __restrict__ metaA, const float4* __restrict__ depthsA,
                                               const Match* __restrict__ arena, const ChainResult* __restrict__ res, const int* __restrict__ src_index,
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
    const int e = (int)blockIdx.x - blocks_move, si = e / 32, bx = e % 32;
    const ChainResult* src = res + src_index[si];
    const int cam = src_cam[si];
    const int n = src->n_kept;
    const Match* kept = arena + src->kept_base;
    for (int i = bx * 256 + (int)threadIdx.x; i < n; i += 32 * 256) {
        const Match r = kept[i];
        if (r.camID2 == view_id && (int)r.segID2 < S) {
            const int row = r.segID2 * N + cam;
            const int slot = row_start[row] + atomicAdd(&cursor[row], 1);
            meta[slot] = make_uint2(r.segID1, (unsigned)cam);
            depths[slot] = make_float4(r.depths[2], r.depths[3], r.depths[0], r.depths[1]);
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
                                                          const unsigned* __restrict__ local2global, Match* __restrict__ arena, int* __restrict__ best_pos)
{
    __shared__ int s_red[8];
    __shared__ int s_cnt[32];
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
    write_kept_segment_wg(a, a.seg_begin + yl, before, local2global, arena + r.kept_base, s_cnt, best_pos ? best_pos + a.seg_begin + yl : nullptr, s_best);
}

void launch_exist_count(const Match* arena, const ChainResult* res, const int* src_index, const int* src_cam, int n_src, unsigned view_id,
                        int N, int S, int* rowcnt, hipStream_t st)
{
    if (n_src > 0) hipLaunchKernelGGL(k_exist_count, dim3(32, n_src), dim3(256), 0, st, arena, res, src_index, src_cam, view_id, N, S, rowcnt);
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
                  const int* row_start, int* cursor, int cand_cap, uint2* meta, float4* depths, hipStream_t st)
{
    const int blocks_move = (S * n_tbm + 3) / 4;
    const int blocks = blocks_move + 32 * n_src;
    if (blocks > 0) hipLaunchKernelGGL(k_place, dim3(blocks), dim3(256), 0, st, blocks_move, tbm, n_tbm, rowA, metaA, depthsA, arena, res, src_index, src_cam,
                                       view_id, N, S, row_start, cursor, meta, depths, cand_cap);
}
void launch_raw_stats(const int* rowcnt, int N, int seg_begin, int seg_end, int* out2_host, hipStream_t st)
{
    hipLaunchKernelGGL(k_raw_stats, dim3(1), dim3(1024), 0, st, rowcnt, N, seg_begin, seg_end, out2_host);
}
void launch_kept_write_chain(const VerifyArgs& a, const int* kept_cnt, int nrow, const ChainResult* prev, unsigned long long arena_cap, ChainResult* res,
                             ChainResult* res_host, const unsigned* l2g, Match* arena, hipStream_t st, int* best_pos)
{
    hipLaunchKernelGGL(k_kept_write_chain, dim3(std::max(1, a.seg_end - a.seg_begin)), dim3(256), 0, st, a, kept_cnt, nrow, prev, arena_cap, res, res_host, l2g, arena, best_pos);
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
static int run_chain(l3d_ctx* c, const l3d_chain_view* views, int n_views, l3d_chain_callback cb, void* user, const l3d_dense_map* map,
                     l3d_chain_summary* summary, int64_t* n_pot, int k_begin = 0, int k_end = -1)
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
    if (int rc = chain_assign_arenas(c, views, n_views, vd, L, true, true, c->chain_ring != 0 ? L3D_AHEAD + L3D_S1AHEAD + 3 : 0, st)) return rc;
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
    if (c->test_cand_cap) cand_cap = c->test_cand_cap;      // tests: force the overflow / restart path
    if (c->test_arena_cap) arena_cap = c->test_arena_cap;

    auto reserve_caps = [&]() -> int {
        if (int rc = chain_reserve_candidates(c, L, cand_cap, c->chain_ring ? kRing : 0)) return rc;
        HIPCHK(c, c->ch_kept.reserve(arena_cap * sizeof(Match)));
        return L3D_OK;
    };
    { int rc = reserve_caps(); if (rc) return rc; }
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
        { ProfScope p(c, "exist"); launch_exist_count(arena, dres, d_si, d_sc, v.n_sources, v.view_id, N, S, d.rowcnt, st); }
        // combined row starts (+ zeroed scatter cursors, + the segments ordered longest first for the verification launch)
        { ProfScope p(c, "scan"); launch_scan(d.rowcnt, c->row_start.as<int>(), (int)nrow, c->ch_cursor.as<int>(), st, c->ch_segorder.as<int>(), N, 0, S); }
        if (use_ring) {
            {
                ProfScope p(c, "cand_move");
                launch_place(pa.tbm, v.n_tbm, N, S, d.rowA, ringA_meta(k), ringA_depths(k), arena, dres, d_si, d_sc, v.n_sources, v.view_id,
                             c->row_start.as<int>(), c->ch_cursor.as<int>(), (int)cand_cap, c->cand_meta.as<uint2>(), c->cand_depths.as<float4>(), st);
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
            while (pv >= 0 && !vd[(size_t)pv].verified) --pv;                // the arena slice starts where the previous verified view's ended
            launch_kept_write_chain(va, c->kept_cnt.as<int>(), (int)nrow, pv >= 0 ? dres + pv : nullptr, (unsigned long long)arena_cap, dres + k, hres_dev + k,
                                    reinterpret_cast<const unsigned*>(dtab + d.o_l2g), arena, st, (map || ranged) ? d.bestpos : nullptr);
        }
        { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(c, L3D_ERR_HIP, std::string("chain launch, view ") + std::to_string(k) + ": " + hipGetErrorString(e_)); }
        // (with a delivery callback the host starts D2H copies of device memory once it has seen this event: a default, fenced event then)
        if (!ev[(size_t)k]) ev[(size_t)k] = cb ? get_event(c) : get_local_event(c);
        HIPCHK(c, hipEventRecord(ev[(size_t)k], st));
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
            wait_delivered();
            if (r.overflow & 1) cand_cap = (size_t)r.R + (size_t)r.R / 4 + 65536;
            if (r.overflow & 2) {
                // the arena cannot be reallocated without losing earlier lists that later views still read:
                // copy it over
                // (projected from the views done so far when there are enough of them: dense scenes keep 10x the first guess)
                size_t new_cap = arena_cap * 2;
                if (k >= 4) new_cap = std::max(new_cap, (size_t)((double)r.kept_base / k * n_views * 1.3) + 1048576);
                // (records are indexed with 32 bits: the arena ends at 2^32 records = 137 GB; doubling must not run past it)
                const size_t kMaxRecords = 0xfffffff0u;
                if (new_cap > kMaxRecords) {
                    if (arena_cap >= kMaxRecords) { rc_final = fail(c, L3D_ERR_UNSUPPORTED, "match_chain: more than 2^32 kept matches in one chain (view " + std::to_string(k) + " of " + std::to_string(n_views) + ")"); break; }
                    new_cap = kMaxRecords;
                }
                void* np = nullptr;
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
                if (!hip_ok(hipMemcpy(np, c->ch_kept.p, (size_t)r.kept_base * sizeof(Match), hipMemcpyDeviceToDevice), "hipMemcpy")) break;
                (void)hipFree(c->ch_kept.p);
                c->ch_kept.p = np; c->ch_kept.cap = new_cap * sizeof(Match);
                arena_cap = new_cap;
            }
            { int rc = reserve_caps(); if (rc) { rc_final = rc; break; } }
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
            --k;
            continue;
        }
        raw_sum += r.R;                     // candidates verified (stage-1 + existing), counted when the view is final (a restart enqueues views twice)
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
        for (int k = 0; k < n_views; ++k) pv[(size_t)k] = ProdChainView{ vd[(size_t)k].verified ? vd[(size_t)k].best : nullptr, vd[(size_t)k].verified ? vd[(size_t)k].bestpos : nullptr, vd[(size_t)k].verified ? 1 : 0 };
        rc_final = build_products(c, views, n_views, pv.data(), hres, map, summary, n_pot);
    }
    if (c->opt.timing)
        fprintf(stderr, "[l3d match_chain] setup %.2f ms | enqueue + watch loop %.2f ms (waiting: view results %.2f, stage-1 statistics %.2f) | delivery thread: d2h %.2f, callback %.2f\n",
                (t_loop0 - t_setup0) * 1e3, (t_prod0 - t_loop0) * 1e3, t_wait * 1e3, t_ev1 * 1e3, t_d2h * 1e3, t_cb * 1e3);
    if (c->opt.timing && map) fprintf(stderr, "[l3d match_chain] products on the device %.2f ms\n", (now_s() - t_prod0) * 1e3);
    const double t_tail0 = now_s();
    if (sm != s1) (void)hipStreamSynchronize(sm);
    (void)hipStreamSynchronize(s1);
    (void)hipStreamSynchronize(st);
    if (c->opt.timing) fprintf(stderr, "[l3d match_chain] hipSetDevice %.3f ms, final syncs %.3f ms\n", (t_setup0 - t_enter) * 1e3, (now_s() - t_tail0) * 1e3);
    for (hipEvent_t e : ev) { if (cb) put_event(c, e); else put_local_event(c, e); }
    for (hipEvent_t e : ev1) put_local_event(c, e);
    for (hipEvent_t e : evm) put_local_event(c, e);
    c->stats[1] = raw_sum;
    c->stats[3] = kept_total;
    if (rc_final == L3D_OK && !c->test_cand_cap && !c->test_arena_cap) {
        c->chain_seen_views = n_views; c->chain_seen_pairs = pairs; c->chain_seen_cand_cap = cand_cap; c->chain_seen_arena_cap = arena_cap;
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

extern "C" int l3d_match_chain_blocks(l3d_ctx* c, const l3d_chain_view* views, int n_views, const l3d_dense_map* map, l3d_chain_summary* summary, int64_t* n_pot,
                                      int rank, int world, int warmup_views, int window, l3d_exchange_fn exchange, void* exchange_user, int* verdict)
{
    if (!c) return L3D_ERR_INVALID;
    if (!views || n_views <= 0 || !map || !summary || !exchange || !verdict || world < 1 || rank < 0 || rank >= world || warmup_views < 0 || window < 0)
        return fail(c, L3D_ERR_INVALID, "l3d_match_chain_blocks: bad argument");
    *verdict = 1;
    if (n_pot) *n_pot = 0;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    auto block_begin = [&](int r) { return (int)(((long long)n_views * r) / world); };
    const int own0 = block_begin(rank), own1 = block_begin(rank + 1);
    const int first = rank == 0 ? 0 : std::max(0, own0 - warmup_views);
    const double t0 = now_s();
    // ---- this rank's chain: its block and the warm-up views in front of it, started cold
    // (a rank whose chain fails must not leave the others waiting in the first collective: it still publishes its table, with a mark that
    // every rank reads -- they all return an error then, without entering another collective)
    const int chain_rc = run_chain(c, views, n_views, nullptr, nullptr, nullptr, nullptr, nullptr, first, own1);
    std::string chain_err;
    if (chain_rc) { std::lock_guard<std::mutex> lk(c->err_mu); chain_err = c->err; }
    const ChainResult* hres = c->ch_pin_res.as<ChainResult>();
    const double t1 = now_s();
    // useful work of this rank = its own block (the warm-up is the price of the speculation)
    {
        double p = 0;
        for (int k = own0; k < own1; ++k)
            for (int j = 0; j < views[k].n_tbm; ++j) p += (double)views[k].S_src * views[k].offsets[2 * views[k].to_be_matched[j] + 1];
        c->stats[0] = p;
    }
    // ---- digests of every list this rank computed, all-gathered
    const size_t tab_bytes = (((size_t)n_views * sizeof(BlockDigest)) + 255) & ~(size_t)255;
    HIPCHK(c, c->ch_hdr.reserve(tab_bytes * (size_t)(world + 1) + 256));
    BlockDigest* dtab_own = c->ch_hdr.as<BlockDigest>();
    BlockDigest* dtab_all = reinterpret_cast<BlockDigest*>(c->ch_hdr.as<unsigned char>() + tab_bytes);
    HIPCHK(c, hipMemsetAsync(dtab_own, 0, tab_bytes, st));
    if (chain_rc) {
        BlockDigest mark; mark.hash = ~0ull; mark.n_kept = -1; mark.R = chain_rc;
        (void)hipMemcpyAsync(dtab_own, &mark, sizeof(mark), hipMemcpyHostToDevice, st);
        (void)hipStreamSynchronize(st);
    } else if (own1 > first) {
        HIPCHK(c, hipMemcpyAsync(c->ch_res.p, hres, (size_t)n_views * sizeof(ChainResult), hipMemcpyHostToDevice, st));     // (the final records: a restart rewrites them)
        hipLaunchKernelGGL(k_block_digest, dim3(16, own1 - first), dim3(256), 0, st, c->ch_kept.as<Match>(), c->ch_res.as<ChainResult>(), first, dtab_own);
    }
    if (exchange(exchange_user, -1, dtab_own, dtab_all, tab_bytes, world, (void*)st)) return fail(c, L3D_ERR_HIP, "l3d_match_chain_blocks: the exchange of the digests failed");
    std::vector<BlockDigest> tab((size_t)world * (size_t)n_views);
    for (int r = 0; r < world; ++r)
        HIPCHK(c, hipMemcpyAsync(tab.data() + (size_t)r * n_views, reinterpret_cast<const unsigned char*>(dtab_all) + (size_t)r * tab_bytes, (size_t)n_views * sizeof(BlockDigest), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const double t2 = now_s();
    if (chain_rc) return fail(c, chain_rc, "l3d_match_chain_blocks: this rank's chain failed: " + chain_err);
    for (int r = 0; r < world; ++r)
        if (tab[(size_t)r * n_views].n_kept == -1 && tab[(size_t)r * n_views].hash == ~0ull)
            return fail(c, L3D_ERR_HIP, "l3d_match_chain_blocks: the chain of rank " + std::to_string(r) + " failed (code " + std::to_string(tab[(size_t)r * n_views].R) + ")");
    // ---- the verdict (the same on every rank: same table)
    bool ok = true;
    for (int r = 1; r < world && ok; ++r) {
        const int b = block_begin(r), fr = std::max(0, b - warmup_views), lo = b - window;
        if (lo < fr || lo < block_begin(r - 1)) { ok = false; break; }       // warm-up shorter than the window, or a block shorter than the window
        for (int k = lo; k < b; ++k) {
            const BlockDigest &x = tab[(size_t)r * n_views + k], &y = tab[(size_t)(r - 1) * n_views + k];
            // (the kept LIST must be the same; the number of candidates it was chosen from may differ while the warm-up converges)
            if (x.hash != y.hash || x.n_kept != y.n_kept) { ok = false; if (c->opt.timing) fprintf(stderr, "[l3d chain_blocks] rank %d's warm-up view %d differs from rank %d's (%d vs %d kept)\n", r, k, r - 1, x.n_kept, y.n_kept); break; }
        }
    }
    if (c->opt.timing) fprintf(stderr, "[l3d chain_blocks rank %d/%d] views %d..%d (block from %d): chain %.2f ms, digests + exchange %.2f ms, speculation %s\n",
                               rank, world, first, own1 - 1, own0, (t1 - t0) * 1e3, (t2 - t1) * 1e3, ok ? "exact" : "NOT exact");
    if (!ok) return L3D_OK;                                                    // *verdict = 1: nothing committed
    // From here on a rank that fails on its own (an allocation, a launch) must not leave the others waiting in the next collective: every
    // step that only this rank can fail in is followed by a small all-gather of status words, and either all ranks enter the big collective
    // behind it or none does.
    HIPCHK(c, c->ch_hdr.reserve(tab_bytes * (size_t)(world + 1) + 512 * (size_t)(world + 2)));     // (the digest tables are on the host by now)
    long long* st_own = reinterpret_cast<long long*>(c->ch_hdr.as<unsigned char>());
    long long* st_all = reinterpret_cast<long long*>(c->ch_hdr.as<unsigned char>() + 256);
    std::vector<long long> words((size_t)world, 0);
    // publishes `mine` (negative = this rank failed with code -mine), reads everybody's; non-zero return: somebody failed (this rank's own message is kept)
    auto all_gather_word = [&](long long mine, const char* what) -> int {
        std::string own_err;
        if (mine < 0) { std::lock_guard<std::mutex> lk(c->err_mu); own_err = c->err; }
        hipError_t e = hipMemcpyAsync(st_own, &mine, 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);                   // (`mine` is a stack word)
        if (e != hipSuccess) return fail(c, L3D_ERR_HIP, std::string("l3d_match_chain_blocks: status word: ") + hipGetErrorString(e));
        if (exchange(exchange_user, -3, st_own, st_all, 256, world, (void*)st)) return fail(c, L3D_ERR_HIP, std::string("l3d_match_chain_blocks: the exchange of the status words failed (") + what + ")");
        for (int r = 0; r < world; ++r) {
            e = hipMemcpyAsync(&words[(size_t)r], reinterpret_cast<const unsigned char*>(st_all) + (size_t)r * 256, 8, hipMemcpyDeviceToHost, st);
            if (e != hipSuccess) return fail(c, L3D_ERR_HIP, std::string("l3d_match_chain_blocks: status word: ") + hipGetErrorString(e));
        }
        e = hipStreamSynchronize(st);
        if (e != hipSuccess) return fail(c, L3D_ERR_HIP, std::string("l3d_match_chain_blocks: status word: ") + hipGetErrorString(e));
        if (mine < 0) return fail(c, (int)-mine, own_err);
        for (int r = 0; r < world; ++r)
            if (words[(size_t)r] < 0) return fail(c, L3D_ERR_HIP, "l3d_match_chain_blocks: rank " + std::to_string(r) + " failed (code " + std::to_string(-words[(size_t)r]) + ") while " + what);
        return L3D_OK;
    };
    // ---- all-gather of the blocks: [records of the block's views][best depth pairs][best positions], padded to the largest block
    auto owner = [&](int k) { int r = (int)(((long long)k * world) / n_views); while (r + 1 < world && block_begin(r + 1) <= k) ++r; while (r > 0 && block_begin(r) > k) --r; return r; };
    std::vector<long long> rec_of((size_t)world, 0), seg_of((size_t)world, 0);
    for (int k = 0; k < n_views; ++k) {
        const int r = owner(k);
        rec_of[(size_t)r] += tab[(size_t)r * n_views + k].n_kept;
        if (views[k].n_tbm > 0) seg_of[(size_t)r] += views[k].S_src;
    }
    long long max_rec = 0, max_seg = 0, total = 0;
    for (int r = 0; r < world; ++r) { max_rec = std::max(max_rec, rec_of[(size_t)r]); max_seg = std::max(max_seg, seg_of[(size_t)r]); total += rec_of[(size_t)r]; }
    if (total > 0xfffffff0ll) return fail(c, L3D_ERR_UNSUPPORTED, "l3d_match_chain_blocks: more than 2^32 kept matches");     // (the same on every rank)
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_best = al((size_t)max_rec * sizeof(Match)), o_bpos = o_best + al((size_t)max_seg * 8), slot = o_bpos + al((size_t)max_seg * 4);
    // offsets of the views' slices in the whole-run arrays of best pairs / positions (chain_assign_arenas: verified views back to back)
    std::vector<long long> best_off((size_t)n_views + 1, 0);
    for (int k = 0; k < n_views; ++k) best_off[(size_t)k + 1] = best_off[(size_t)k] + (views[k].n_tbm > 0 ? views[k].S_src : 0);
    const auto stage_block = [&]() -> int {
        HIPCHK(c, c->ch_send.reserve(slot + 256));
        HIPCHK(c, c->ch_gathered.reserve(slot * (size_t)world + 256));
        unsigned char* send = c->ch_send.as<unsigned char>();
        long long own_start = 0;                                    // (this rank's arena: the views it computed, back to back from its cold start)
        for (int k = first; k < own0; ++k) own_start += hres[k].n_kept;
        if (rec_of[(size_t)rank] > 0)
            HIPCHK(c, hipMemcpyAsync(send, c->ch_kept.as<Match>() + own_start, (size_t)rec_of[(size_t)rank] * sizeof(Match), hipMemcpyDeviceToDevice, st));
        if (seg_of[(size_t)rank] > 0) {
            HIPCHK(c, hipMemcpyAsync(send + o_best, c->ch_best.as<float2>() + best_off[(size_t)own0], (size_t)seg_of[(size_t)rank] * 8, hipMemcpyDeviceToDevice, st));
            HIPCHK(c, hipMemcpyAsync(send + o_bpos, c->ch_bestpos.as<int>() + best_off[(size_t)own0], (size_t)seg_of[(size_t)rank] * 4, hipMemcpyDeviceToDevice, st));
        }
        return L3D_OK;
    };
    { const int rc = stage_block(); if (int a_rc = all_gather_word(rc ? -(long long)rc : 0, "staging its block")) return a_rc; }
    if (exchange(exchange_user, -2, c->ch_send.p, c->ch_gathered.p, slot, world, (void*)st)) return fail(c, L3D_ERR_HIP, "l3d_match_chain_blocks: the exchange of the kept lists failed");
    // ---- the one chain's arena: blocks in rank order = views in order; then matchViews' products: every rank builds the rows of its OWN block
    // of views (sort + unique of the keys whose source lies in the block: its views' records and their neighbours', all of them in the arena now),
    // the pieces are all-gathered and put together -- 1/world of the sort per rank instead of all of it on every rank
    std::vector<ChainResult> hres_all((size_t)n_views);
    std::vector<ProdChainView> pvh((size_t)n_views);
    const int nvd = map->n_views;
    auto dense_of = [&](int k) {                                   // the dense view a chain view is (ids ascend in both)
        if (k >= n_views) return nvd;
        const uint32_t* it = std::lower_bound(map->view_ids, map->view_ids + nvd, views[k].view_id);
        return (int)(it - map->view_ids);
    };
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
    Products& P = c->products;
    // counts first (a piece is padded to the largest; a negative count = this rank failed), then [row starts of the block, numbered from 0 | entries]
    { const int rc = assemble_and_build(); if (int a_rc = all_gather_word(rc ? -(long long)rc : (long long)n_local, "building its rows of the products")) return a_rc; }
    const std::vector<long long> cnts = words;
    long long max_cnt = 0, max_rows = 0, n_pot_all = 0;
    for (int r = 0; r < world; ++r) {
        max_cnt = std::max(max_cnt, cnts[(size_t)r]); n_pot_all += cnts[(size_t)r];
        max_rows = std::max(max_rows, (long long)map->seg_base[dvb[(size_t)r + 1]] - map->seg_base[dvb[(size_t)r]]);
    }
    const size_t o_ent = al((size_t)max_rows * 8), pslot = o_ent + al((size_t)max_cnt * 4 + 4);
    const auto stage_piece = [&]() -> int {
        HIPCHK(c, c->ch_send.reserve(pslot + 256));
        HIPCHK(c, c->ch_gathered.reserve(pslot * (size_t)world + 256));
        const long long r0 = map->seg_base[dvb[(size_t)rank]], nr = (long long)map->seg_base[dvb[(size_t)rank + 1]] - r0;
        unsigned char* sp = c->ch_send.as<unsigned char>();
        if (nr > 0) HIPCHK(c, hipMemcpyAsync(sp, P.pot_start.as<long long>() + r0, (size_t)nr * 8, hipMemcpyDeviceToDevice, st));
        if (n_local > 0) HIPCHK(c, hipMemcpyAsync(sp + o_ent, P.pot_tgt.p, (size_t)n_local * 4, hipMemcpyDeviceToDevice, st));
        return L3D_OK;
    };
    { const int rc = stage_piece(); if (int a_rc = all_gather_word(rc ? -(long long)rc : 0, "staging its piece of the products")) return a_rc; }
    if (exchange(exchange_user, -4, c->ch_send.p, c->ch_gathered.p, pslot, world, (void*)st)) return fail(c, L3D_ERR_HIP, "l3d_match_chain_blocks: the exchange of the table pieces failed");
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
}

void l3d::warm_chain() { touch_kernel(reinterpret_cast<const void*>(&k_exist_count)); }
