// l3d_kernels.hpp -- argument blocks and launchers shared by l3d_kernels.hip and l3d_capi.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace l3d {

#ifndef L3D_SRC_PER_BLOCK
#define L3D_SRC_PER_BLOCK 64
#endif
static_assert(L3D_SRC_PER_BLOCK >= 1 && L3D_SRC_PER_BLOCK <= 256, "k_pair_mask: ring keys hold the source row in 8 bits, one thread stages one source segment");
constexpr int kSrcPerBlock = L3D_SRC_PER_BLOCK;    // source segments walked by one k_pair_mask workgroup (A/B, ms per config-2 pass: 32 -> +0.3, 48 -> +0.9, 64 best, 96 -> +0.4, 128 -> +0.8)
constexpr int kMaxW64 = 256;        // bit-row words per camera: up to 16384 segments per view
constexpr int kVerifyTile = 256;
constexpr int kVWSlack = 8;          // entries readable past the end of a k_verify_window image (the scan loads a group ahead)    // witnesses staged in LDS per step of k_verify

struct Match {                      // == l3d_match (include/line3d_amd.h)
    uint32_t segID1, camID2, segID2;
    float depths[4];
    float confidence;
};

struct ExistRec {                   // an existing (reverse) match, localized and ranked by the host
    int seg, cam;                   // source segment, LOCAL camera
    uint32_t tgt;
    int rank;                       // index inside its (seg,cam) run, ascending tgt
    float d[4];
};

struct Hypothesis {                 // == l3d_hypothesis
    double P1[3], P2[3], dir[3];
    float depth_p1, depth_p2;
    float k_lower, k_upper, median_depth;
    uint32_t pad;
};

// per view of a chain on the device: its run table (null: none) and sizes -- what a later view needs to find the records that point at it (l3d_runtable.hpp)
struct RtInfo { const int* rt; int S, N; };

struct ChainResult {                // per view of the resident chain (device -> pinned host)
    unsigned kept_base;             // first record of the view's slice of the kept arena (records of 32 bytes: 32 bits reach 137 GB)
    int n_kept, R, overflow;
};

struct PairArgs {
    const float4* src_segs;         // S_src
    const float4* tgt_segs;         // concatenated neighbours
    const int2* offsets;            // N x (start,count)
    const float* F;                 // N x 9
    const float* RtKinv;            // N x 9
    const float* centers;           // N x 3
    const float* RtKinv_src;        // 9
    const float* C_src;             // 3
    const int* tbm;                 // n_tbm local camera ids
    unsigned long long* mask;       // [n_tbm][S_src][W64]
    int S_src, N, n_tbm, W64;
    int seg_begin, seg_end;
    int cand_cap;                   // candidate capacity guard of the resident chain (0: none)
    unsigned long long* dbg;        // diagnostic (L3D_PAIR_STATS=1): {pairs, level-1 survivors, level-2 survivors, set bits}, else null
    int src_per_block;              // set by launch_pair_mask (<= kSrcPerBlock)
    int* rowcnt;                    // chains: the per-(segment, camera) candidate counts are added here by k_pair_mask itself (rows zeroed at
                                    // chain start); null: a separate k_row_count launch (per-view seam call, chain restarts)
    int dbg_view;                   // (diagnostics: the view id of this launch)
    int wedge_pretest;              // conservative filters in front of the exact test: bit 0 wedge test, bit 1 overlap-bound test, bit 2 set = no accepts at level 2 (default 3)
    // resident chain, row starts without a scan launch: k_pair_mask also adds its counts to rowblk[(row) >> 8] (sums of 256 rows; row =
    // segment * N + camera).  k_pair_fill then reads the (upper-bound) counts from rowub -- the array k_pair_mask filled, never
    // written again -- and forms a row's start itself: the block sums in front of its block + the counts in front of it inside the
    // block; it writes the start to rowstart_out and the row's true count to rowcnt (a different array: the chain's final counts).
    int* rowblk = nullptr;
    const int* rowub = nullptr;
    int* rowstart_out = nullptr;
    const float4* tgt_rays = nullptr;   // resident chain: unit viewing rays of the target endpoints, 2 per entry of tgt_segs (k_tgt_rays); null: k_pair_fill
                                    // normalises them itself, per candidate, with the same operations
    const float4* src_rays = nullptr;   // ... and of the view's own end points (2 per source segment)
    int depth_in_fill = 0;          // resident chain: k_pair_mask stops after the exact overlap test (its bits and row counts are then an UPPER
                                    // bound), the four depths are triangulated ONCE, in k_pair_fill, which drops the pairs without four positive
                                    // depths (cudawrapper.cu:931), packs the row and writes its true count back into rowcnt; 0: the bit already
                                    // says "four positive depths" (per-view seam call, sharded chain)
};

struct VerifyArgs {
    const float4* src_segs;
    const float4* tgt_segs;
    const int2* offsets;
    const float* P;                 // N x 12
    const float* RtKinv_src;
    const float* C_src;
    const int* row_start;           // [S_src*N + 1]
    const uint2* cand_meta;         // (tgt, cam)
    const float4* cand_depths;
    float* cand_conf;
    int N, seg_begin, seg_end;
    int nrow_total;                 // S_src * N: row_start[nrow_total] = number of candidates
    int mmax;                       // max candidates of one segment (LDS sizing)
    int only_above;                 // k_verify (all-pairs): process only segments with more than this many candidates (-1: all)
    int skip_above;                 // k_verify_window: leave segments with more than mmax candidates to the `big` launch (0/1)
    int big;                        // k_verify_window: 0 = segments with at most mmax candidates (LDS image), 1 = only the bigger ones
                                    // (arrays in `scratch`), 2 = both in one launch (grid 2 x segments)
    const int* seg_order;           // k_verify_window: workgroup i takes segment seg_order[i] (longest first); null = seg_begin + i
    int* kept_cnt;                  // fused per-segment epilogue of k_verify_window (k_seg_post): number of kept matches ...
    float2* best_depths;            // ... and depths of the first best hypothesis; null = separate k_seg_post launch
    const int* exist_cams;          // k_verify_window: local ids of the cameras whose rows hold reverse matches in arbitrary order; the
    int n_exist_cams;               // workgroup of a segment puts those runs into ascending target order itself (0: already ordered)
    float* scratch;                 // 4 arrays of scratch_stride floats (candidate capacity + 2), global memory
    long long scratch_stride;
    int cand_cap;                   // candidate capacity guard of the resident chain (0: none)
    const struct ChainResult* res;  // optional device record of the resident chain (kept base/count)
    int debug;                      // timing-only ablations (L3D_VW_DEBUG), 0 in production
    unsigned long long* stamps;     // per-phase cycle sums of k_verify_window (diagnostic build: L3D_VW_STAMPS=1), else null
    float sigma_p, sigma_a, spatial_k;
    int* bstart_g;                  // k_verify_window_gb (more than 16 neighbours): [segments][kVWBuckets + 1] bucket starts of the built images, read by the rounds from
                                    // global memory -- the LDS they took (with the build's cursors: 16.4 KB) is what kept a workgroup per CU out at 24 neighbours; else null
};
// k_verify_window_build + k_vw_walk (big == 2 only): a segment that outgrows the LDS image is only BUILT by its scratch block (bucketed image, bucket
// starts, header) and its hypotheses are verified in units of split_unit by the workgroups of a second launch -- a launch of FEW segments (one rank's
// slice of a view on a dense scene) no longer lasts as long as its longest segment.  A block of its own, passed to those two kernels only: the
// one-launch kernel keeps its argument block (and its scalar registers: 2.6 % more VALU instructions were SGPR spills when these fields sat in VerifyArgs)
struct VWSplitArgs {
    int split_unit = 0, units_max = 0;      // hypotheses per unit (a multiple of 256; 0: no split) | upper bound of the number of units (grid of the second launch)
    int* unit_start = nullptr;              // [segments + 1] exclusive prefix of the units per segment, in seg_order order (written by the first launch)
    int* bstart_g = nullptr;                // [segments][kVWBuckets + 1] bucket starts of the built images
    int4* seg_hdr = nullptr;                // [segments] (bucket base, bits of the largest |depth|, -, -)
    unsigned long long* best64 = nullptr;   // [segments] (confidence bits << 32) | (0x7fffffff - candidate index): atomicMax = the first strict maximum in candidate order
    int* done = nullptr;                    // [segments] units finished: the last one writes best_depths
};
constexpr int kVWBuckets = 2048;

// code-object warm-up, one function per translation unit (l3d_warm_up)
void warm_kernels(); void warm_verify_window(); void warm_rdd(); void warm_affinity(); void warm_linefit(); void warm_chain(); void warm_chain_sharded(); void warm_products(); void warm_sort();
inline void touch_kernel(const void* f) { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, f); }

// forced_spb / wide_max / vw_lds_opt: the context's own switches (options pair_spb, vw_wide_max, vw_lds) -- per context, not process-wide
void launch_pair_mask(const PairArgs& a, int maxW, hipStream_t st, int forced_spb = 0);
void launch_row_count(const PairArgs& a, int* rowcnt, hipStream_t st);
void launch_exist_hist(const ExistRec* ex, int n, int N, int* rowcnt, hipStream_t st);
struct RayJob {                     // one view of k_tgt_rays
    const float4* tgt; const int2* offsets; const float* RtKinv; float4* out; int n_tgt, N;
};
void launch_tgt_rays(const RayJob* jobs, int n_jobs, int max_n_tgt, hipStream_t st);
void launch_scan(const int* in, int* out, int n, int* zero, hipStream_t st, int* seg_order = nullptr, int N = 0, int seg_begin = 0, int seg_end = 0,
                 int* stats_out = nullptr);
void launch_scan_range(const int* rowcnt, int* row_start, int N, int seg_begin, int seg_end, int nrow_total, int* zero, int* seg_order, hipStream_t st,
                       int* stats_out = nullptr);
void launch_pair_fill(const PairArgs& a, const int* row_start, uint2* meta, float4* depths, hipStream_t st);
void launch_exist_place(const ExistRec* ex, int n, int N, const int* row_start, uint2* meta, float4* depths, hipStream_t st);
void launch_verify(const VerifyArgs& a, hipStream_t st);
void launch_verify_window(const VerifyArgs& a, hipStream_t st, int wide_max = 640, const VWSplitArgs* split = nullptr);
size_t verify_window_lds_bytes(int mmax, int N);
size_t verify_window_max_lds(int vw_lds_opt = 0);
bool verify_window_supported(int N);
void verify_window_set_lds_budget(size_t bytes);
void launch_seg_mmax(const int* row_start, int N, int seg_begin, int seg_end, int* out, hipStream_t st);
void launch_seg_post(const VerifyArgs& a, int* kept_cnt, float2* best, hipStream_t st);
void launch_kept_write(const VerifyArgs& a, const int* kept_start, const unsigned* l2g, Match* out, hipStream_t st);
// resident chain (l3d_chain.hip)
void launch_exist_count(const Match* arena, const ChainResult* res, const int* src_index, const int* src_cam, int n_src, unsigned view_id,
                        int N, int S, int* rowcnt, hipStream_t st, const unsigned* cams = nullptr, int bps = 32);
void launch_exist_scatter(const Match* arena, const ChainResult* res, const int* src_index, const int* src_cam, int n_src, unsigned view_id,
                          int N, int S, const int* row_start, int* cursor, uint2* meta, float4* depths, int cap, hipStream_t st);
void launch_exist_sort_runs(const int* cams, int n_cams, int N, int S, const int* row_start, uint2* meta, float4* depths, int cap, hipStream_t st,
                            int seg_begin = 0, int seg_end = -1, float* stage = nullptr, long long stage_stride = 0, unsigned* stage_key = nullptr);
void launch_place(const int* tbm, int n_tbm, int N, int S, const int* rowA, const uint2* metaA, const float4* depthsA,
                  const Match* arena, const ChainResult* res, const int* src_index, const int* src_cam, int n_src, unsigned view_id,
                  const int* row_start, int* cursor, int cand_cap, uint2* meta, float4* depths, hipStream_t st, const unsigned* cams = nullptr, int bps = 32,
                  const struct RtInfo* info = nullptr, const int* src_slot = nullptr, int g = 1, const int* part = nullptr);
int exist_chunks();
// (run tables, l3d_runtable.hpp: the sources' runs towards this view instead of scans of their lists)
void launch_exist_count_rt(const unsigned* qt_arena, const struct RtInfo* info, const ChainResult* res, const int* src_index, const int* src_cam, const int* src_slot, int n_src, int g,
                           int N, int S, int* rowcnt, int* part, hipStream_t st);
void launch_raw_stats(const int* rowcnt, int N, int seg_begin, int seg_end, int* out2_host, hipStream_t st);
void launch_kept_write_chain(const VerifyArgs& a, const int* kept_cnt, int nrow, const ChainResult* prev, unsigned long long arena_cap, ChainResult* res,
                             ChainResult* res_host, const unsigned* l2g, Match* arena, hipStream_t st, int* best_pos = nullptr, unsigned* cams = nullptr,
                             int* rt = nullptr, int rt_stride = 0);
void launch_collinearity(const float4* segs, int S, float sigma_sqr, unsigned long long* mask, int W64, int* rowcnt, hipStream_t st);
void launch_collinearity_fill(const float4* segs, int S, float sigma_sqr, const unsigned long long* mask, int W64,
                              const int* row_start, int* oi, int* oj, float* ow, hipStream_t st);
void launch_similarity(const Hypothesis* hyp, const int2* pairs, int n, float sigma_a, float two_log, float* sim, hipStream_t st);
void launch_test_sqthr(const float* u, int n, float* walk, float* fast, hipStream_t st);
void launch_test_math(const float* x, int n, float* e, float* ac, double* acd, hipStream_t st);

}  // namespace l3d
