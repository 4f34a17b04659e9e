// l3d_kernels.hip -- HIP kernels for gfx950 (CDNA4, wave64) of the Line3D matching path.
// No MFMA: there is no dense contraction on this path; the kernels are
// FP32-VALU bound, inputs are tiny and L2/LDS resident.
//
//   k_pair_mask      stage 1a: epipolar/overlap test + triangulation, one bit per (src,tgt) pair
//   k_row_count      stage 1b: per (src segment, camera) candidate counts from the bit rows
//   k_exist_hist     counts of the already existing (reverse) matches
//   k_scan           exclusive scan (independent tile workgroups, l3d_scan.hpp)
//   k_pair_fill      stage 1c: depth records for the set bits, written in (seg,cam,tgt) order
//   k_exist_place    existing matches into their candidate slots
//   k_verify         stage 2: multi-view support score of every candidate (K_verify_matches)
//   k_seg_post       per segment: best hypothesis depths, number of kept matches
//   k_kept_write     ordered compaction of the kept matches
//   k_collinearity   per-view 2-D collinearity relation (upper triangle, bit + value)
//   (replicator dynamics diffusion: l3d_rdd.hip)
//   k_similarity     batched similarity_coll3D for the affinity fill
#include <algorithm>
#include <cstdlib>

#include "l3d_geometry.hpp"
#include "l3d_kernels.hpp"
#include "l3d_options.hpp"
#include "l3d_scan.hpp"
#include "l3d_kept.hpp"
#include "l3d_similarity.hpp"

namespace l3d {

// =================================================================================================
// Stage 1a.  grid = (tgt tiles of 256, src blocks of src_per_block <= kSrcPerBlock, n_tbm); block = 256.
// Lane <-> target segment, the block walks kSrcPerBlock source segments whose invariants are staged in LDS and read as
// broadcasts.  One wave ballot = one 64-bit word of the (camera, src) bit row.
//
// The reference evaluates, for EVERY pair, four line intersections with divisions, two overlap ratios with square
// roots and four triangulations (K_pairwise_matches, cudawrapper.cu:537-611) and keeps 0.7 % of the pairs.  Here the
// exact float sequence runs only on pairs that two cheap CONSERVATIVE filters could not reject; a filter only ever
// rejects a pair the exact sequence rejects too, so the bit rows are identical with the filters on or off
// (tests/: A/B on whole scenes).  Each level runs with (nearly) full waves: survivors are compacted through per-wave
// LDS rings and processed 64 at a time.
//   1. wedge test on ALL pairs (~30 VALU ops, FMA): the epipolar lines e1, e2 of the source endpoints p1, p2 cut the
//      target image into four sectors around the epipole.  A target point q lies on the epipolar line of the source point
//      p(s) = (1-s) p1 + s p2 with s = a/(a-b), a = e1.q, b = e2.q; with a and b of one sign s is outside [0,1].  A target
//      segment whose two endpoints lie strictly (by a safety margin) in ONE same-sign sector and on ONE side of the line
//      e_d = e1 - e2 (the epipolar line of the source line's vanishing point, where s jumps from -inf to +inf) has
//      s(q1), s(q2) both < 0 or both > 1: the intersection points l1_q1, l1_q2 (cudawrapper.cu:573-574) lie on the same
//      side outside [p1, p2], no endpoint of either segment lies on the other, D_segment_overlap_2D returns 0 and the pair
//      is rejected at :586.  The e_d condition is needed: WITHOUT it a segment that crosses e_d inside a same-sign sector
//      has [l1_q1, l1_q2] strictly containing [p1, p2] (the transfer wraps through infinity), both overlaps can pass and
//      the reference keeps the pair whenever its four triangulated depths happen to be positive -- which they are for
//      cameras that face each other (the 3-D segment then lies behind one camera for one endpoint, which :931 does not
//      look at; measured: 1 % of the reference's candidates of an opposing camera pair).  e_d is tested once per source
//      segment against the bounding box of the tile's target endpoints (it nearly always misses it unless the epipole is
//      in or near the image): then no target segment of the tile can cross it and the plain sector test is exact; otherwise
//      the sector test of that source segment is switched off (its lines are zeroed for level 1) and level 2, which works
//      with the finite interval between the intersection points like the reference, decides.  Symmetrically for the source
//      segment and each target's lines.  The lines are pre-divided by their margins (1e-4 of
//      the term magnitudes, ~0.2 px, against ~1e-6 relative float error), so the test is min/max chains against +-1.
//      ~12 % of the pairs survive.
//   2. overlap-bound test on the survivors (~110 ops, FMA, reciprocals, no square root): D_segment_overlap_2D of
//      collinear points is the 1-D intersection-over-union of two intervals; with t = a/(a-b) (a, b: the two endpoints
//      against one epipolar line -- the numbers level 1 already looked at) the interval of the intersection points is
//      known in the segment's own parameter without computing the points.  The test evaluates an UPPER bound of both
//      ratios -- t widened by e = 1e-2 (1+|t|)/|a-b| (conditioning of the intersection, in margin units) + coordinate
//      rounding -- and drops the pair when the bound misses the thresholds (min > 0.1, max > 0.3, cudawrapper.cu:586-588)
//      or the intersection pair is surely shorter than a pixel (:211).  Ill-conditioned pairs (e >= 10) are kept.
//      ~8 % of the pairs survive (6 % are candidates).
//   3. the exact overlap test and triangulation on those.
// =================================================================================================
constexpr int kPairQueue = 128;
constexpr float kWedgeTau = 1.0e-4f;
constexpr float kIouCond = 1.0e-2f;           // error of t = a/(a-b) in units of (1+|t|)/|a-b| (a, b in margin units)
constexpr float kIouSlack = 1.0e-3f;

// LDS images of the block's source segments and the tile's target segments: odd strides (gathers by segment index spread over
// the banks) and as small as possible -- with the 25-word target image the workgroup stays under 40 KB, i.e. FOUR workgroups
// per CU instead of three (the scaled lines of level 2 are re-formed from the exact line and its inverse margin: the same
// product, the same bits).
template <bool kDepth> struct SrcBlockInvT {  // 25 words
    SrcPairInv s;                             // exact invariants of the pair test
    float im1, im2;                           // 1 / wedge margin of the epipolar lines of p1 / p2
    f3 ray1, ray2;                            // normalize(RtKinv_src * p1), (* p2): the reference's float sequence
    float pad;
};
template <> struct SrcBlockInvT<false> {      // 19 words: without the rays when the triangulation happens in k_pair_fill
    SrcPairInv s;
    float im1, im2;
    float pad;
};
struct __attribute__((aligned(16))) SrcLevel1 {   // what the level-1 loop reads per source segment, as three aligned 16-byte broadcasts
    float e1x, e1y, e1z, p1x;                 // the line of p1 over its margin for the sector test (zero where e_d = e1 - e2 may cross the tile) | p1
    float e2x, e2y, e2z, p1y;                 // the line of p2 likewise
    float p2x, p2y, pad0, pad1;
};
template <bool kDepth> struct TgtBlockInvT {  // 25 words
    TgtPairInv t;
    float im1, im2;                           // 1 / margin of the epipolar lines of q1 / q2 (in the source image)
    f3 ray1, ray2;                            // normalize(RtKinv_tgt * q1), (* q2)
    float pad;
};
template <> struct TgtBlockInvT<false> {      // 19 words: 32 KB per workgroup, five per CU
    TgtPairInv t;
    float im1, im2;
    float pad;
};

__device__ __forceinline__ float fdot(f3 a, f3 b) { return __builtin_fmaf(a.x, b.x, __builtin_fmaf(a.y, b.y, a.z * b.z)); }
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_line(f3 l, v2f x, v2f y)
{
    const v2f lx = { l.x, l.x }, ly = { l.y, l.y }, lz = { l.z, l.z };
    return __builtin_elementwise_fma(lx, x, __builtin_elementwise_fma(ly, y, lz));
}
__device__ __forceinline__ float fline(f3 l, float x, float y) { return __builtin_fmaf(l.x, x, __builtin_fmaf(l.y, y, l.z)); }
// Bounds of D_segment_overlap_2D(segment [0,1] of length len, intersection points at parameters t1, t2) where
// ti = ai/(ai - bi) and ri = 1/(ai - bi); ext_over_len = (largest coordinate)/len.  For collinear points the reference's case
// analysis (:209-251) is the 1-D intersection over union of the two intervals; the float intersection points it works on lie within
// e of t1, t2: conditioning of the intersection (kIouCond, in margin units) + coordinate rounding + the segment's own line, whose offset
// term x1 y2 - x2 y1 cancels from products of the size ext^2 (2^-24 ext^2 / len pixels beside the true line, kLineCond (ext/len)^2 in t: a 3-pixel
// segment at x = 1000 has a line that is only good to 0.05 px).  upper: 2.0f = "cannot tell"; lower: -1.0f = "cannot tell" -- ill-conditioned,
// or the intersection pair not SURELY a pixel long (:211).  BOTH are "cannot tell" when one of the intersection points lies within 4 e of an END
// POINT of the segment: there the reference's point-on-segment tests (:135-141, a dot product against 1e-12) flip on float noise and
// D_segment_overlap_2D returns 0 -- or thousands -- whatever the intervals are (measured with the diagnostic build -DL3D_BOUND_CHECK: accepts AND
// round 1's rejects decided a dozen pairs in 1e11 against the exact test before this guard; tests/golden/endpoint_quirk_pairs.npz).
constexpr float kLineCond = 5.0e-7f;
struct IouBounds { float upper, lower; };
__device__ __forceinline__ IouBounds iou_bounds(float t1, float r1, float t2, float r2, float len, float ext_over_len)
{
    const float e0 = __builtin_fmaf(kLineCond * ext_over_len, ext_over_len, 1.0e-6f * ext_over_len);
    const float e1 = __builtin_fmaf(kIouCond * (1.0f + __builtin_fabsf(t1)), __builtin_fabsf(r1), __builtin_fmaf(1.0e-6f, __builtin_fabsf(t1), e0));
    const float e2 = __builtin_fmaf(kIouCond * (1.0f + __builtin_fabsf(t2)), __builtin_fabsf(r2), __builtin_fmaf(1.0e-6f, __builtin_fabsf(t2), e0));
    const float e = __builtin_fmaxf(e1, e2);
    if (!(e < 10.0f)) return { 2.0f, -1.0f };                             // ill-conditioned, infinite or NaN
    const float lo = __builtin_fminf(t1, t2), hi = __builtin_fmaxf(t1, t2);
    if ((hi - lo + 2.0f * e) * len < 1.0f - kIouSlack) return { 0.0f, -1.0f };   // the intersection pair is shorter than a pixel (:211)
    // an intersection point within 4 e of an end point of the segment: the reference's case analysis may land anywhere (0, or a ratio of thousands
    // when the "rest" it divides by is the distance of two nearly coincident points) -- neither bound means anything, the exact test decides.
    // (Both points beyond the same end by more than that stay decidable: no point lies on the other segment, the overlap is 0.)
    const float edge = __builtin_fminf(__builtin_fminf(__builtin_fabsf(lo), __builtin_fabsf(lo - 1.0f)), __builtin_fminf(__builtin_fabsf(hi), __builtin_fabsf(hi - 1.0f)));
    if (!(edge > 4.0f * e)) return { 2.0f, -1.0f };
    const float in0 = __builtin_fminf(hi, 1.0f) - __builtin_fmaxf(lo, 0.0f), un0 = __builtin_fmaxf(hi, 1.0f) - __builtin_fminf(lo, 0.0f);
    const float uni_lo = un0 - 2.0f * e;
    IouBounds b;
    b.upper = uni_lo > 0.0f ? (in0 + 2.0f * e) * __builtin_amdgcn_rcpf(uni_lo) * (1.0f + 1.0e-5f) : 2.0f;
    // the lower bound only where the pair is well conditioned (e < 0.1) and the pair of intersection points surely at least a pixel long
    const bool sure = e < 0.1f && (hi - lo - 2.0f * e) * len > 1.0f + kIouSlack && in0 - 2.0f * e > 0.0f;
    b.lower = sure ? (in0 - 2.0f * e) * __builtin_amdgcn_rcpf(un0 + 2.0f * e) * (1.0f - 1.0e-5f) : -1.0f;
    return b;
}

#ifndef L3D_PM_WAVES
#define L3D_PM_WAVES 0
#endif
template <bool kDepth>
__global__ __launch_bounds__(256)
#if L3D_PM_WAVES
__attribute__((amdgpu_waves_per_eu(L3D_PM_WAVES, L3D_PM_WAVES)))
#endif
void k_pair_mask(PairArgs a)
{
    typedef SrcBlockInvT<kDepth> SrcBlockInv;
    typedef TgtBlockInvT<kDepth> TgtBlockInv;
    __shared__ SrcBlockInv s_src[kSrcPerBlock];
    __shared__ SrcLevel1 s_l1[kSrcPerBlock];
    __shared__ TgtBlockInv s_tgt[256];
    __shared__ float s_cam[9 + 9 + 3 + 9];   // F, RtKinv_tgt, C_tgt of this camera, RtKinv_src
    __shared__ float s_boxw[4][8];       // per wave: {min x, max x, min y, max y} of its target endpoints, of its source endpoints
    __shared__ unsigned short s_qa[4][kPairQueue];
    __shared__ unsigned short s_qb[4][kPairQueue];
    __shared__ unsigned long long s_bits[kSrcPerBlock * 4];
    __shared__ int s_blk[kSrcPerBlock * 96 / 256 + 2];   // block sums of this workgroup's rows (fused row starts: N <= 96, rows N apart)

    const int j = blockIdx.z;
    const int cam = a.tbm[j];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int x = blockIdx.x * 256 + tid;
    const int width = a.offsets[cam].y;
    const int toff = a.offsets[cam].x;
    const int y0 = a.seg_begin + blockIdx.y * a.src_per_block;

    if (blockIdx.x * 256 >= width) return;   // whole tile beyond this camera's segments (uniform)

    if (tid < 9) s_cam[tid] = a.F[cam * 9 + tid];
    else if (tid < 18) s_cam[tid] = a.RtKinv[cam * 9 + (tid - 9)];
    else if (tid < 21) s_cam[tid] = a.centers[cam * 3 + (tid - 18)];
    else if (tid < 30) s_cam[tid] = a.RtKinv_src[tid - 21];
    for (int i = tid; i < kSrcPerBlock * 4; i += 256) s_bits[i] = 0ull;
    __syncthreads();

    const f3 C_tgt = mk3(s_cam[18], s_cam[19], s_cam[20]);
    const f3 C_src = mk3(a.C_src[0], a.C_src[1], a.C_src[2]);
    const float* Rt = s_cam + 9;
    const float* Rs = s_cam + 21;

    const bool valid = x < width;
    const float4 tseg = valid ? a.tgt_segs[toff + x] : make_float4(0.f, 0.f, 1.f, 1.f);
    const TgtPairInv t = make_tgt_inv(tseg, s_cam);
    s_tgt[tid].t = t;
    if constexpr (kDepth) {
        s_tgt[tid].ray1 = normalize(mat3_apply(Rt, t.q1));
        s_tgt[tid].ray2 = normalize(mat3_apply(Rt, t.q2));
    }
    const int ny = min(a.src_per_block, a.seg_end - y0);
    {   // bounding boxes of this tile's target endpoints and this block's source endpoints: wave reductions, combined by every
        // thread after the barrier (the source segments sit in the first waves only)
        const float big = 3.0e38f;
        float bx0 = valid ? __builtin_fminf(tseg.x, tseg.z) : big, bx1 = valid ? __builtin_fmaxf(tseg.x, tseg.z) : -big;
        float by0 = valid ? __builtin_fminf(tseg.y, tseg.w) : big, by1 = valid ? __builtin_fmaxf(tseg.y, tseg.w) : -big;
        for (int o = 32; o > 0; o >>= 1) {
            bx0 = __builtin_fminf(bx0, __shfl_down(bx0, o)); bx1 = __builtin_fmaxf(bx1, __shfl_down(bx1, o));
            by0 = __builtin_fminf(by0, __shfl_down(by0, o)); by1 = __builtin_fmaxf(by1, __shfl_down(by1, o));
        }
        float cx0 = big, cx1 = -big, cy0 = big, cy1 = -big;
        if (wave * 64 < ny) {                                                       // (wave-uniform)
            if (tid < ny) {
                const float4 sg = a.src_segs[y0 + tid];
                cx0 = __builtin_fminf(sg.x, sg.z); cx1 = __builtin_fmaxf(sg.x, sg.z);
                cy0 = __builtin_fminf(sg.y, sg.w); cy1 = __builtin_fmaxf(sg.y, sg.w);
            }
            for (int o = 32; o > 0; o >>= 1) {
                cx0 = __builtin_fminf(cx0, __shfl_down(cx0, o)); cx1 = __builtin_fmaxf(cx1, __shfl_down(cx1, o));
                cy0 = __builtin_fminf(cy0, __shfl_down(cy0, o)); cy1 = __builtin_fmaxf(cy1, __shfl_down(cy1, o));
            }
        }
        if (lane == 0) {
            s_boxw[wave][0] = bx0; s_boxw[wave][1] = bx1; s_boxw[wave][2] = by0; s_boxw[wave][3] = by1;
            s_boxw[wave][4] = cx0; s_boxw[wave][5] = cx1; s_boxw[wave][6] = cy0; s_boxw[wave][7] = cy1;
        }
    }
    __syncthreads();
    float box[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v0 = s_boxw[0][i], v1 = s_boxw[1][i], v2 = s_boxw[2][i], v3 = s_boxw[3][i];
        box[i] = (i & 1) ? __builtin_fmaxf(__builtin_fmaxf(v0, v1), __builtin_fmaxf(v2, v3)) : __builtin_fminf(__builtin_fminf(v0, v1), __builtin_fminf(v2, v3));
    }
    // coordinate extents for the margins: max |x|, |y| of the target endpoints (0, 1) / of the source endpoints (2, 3)
    const float ext0 = __builtin_fmaxf(__builtin_fabsf(box[0]), __builtin_fabsf(box[1])), ext1 = __builtin_fmaxf(__builtin_fabsf(box[2]), __builtin_fabsf(box[3]));
    const float ext2 = __builtin_fmaxf(__builtin_fabsf(box[4]), __builtin_fabsf(box[5])), ext3 = __builtin_fmaxf(__builtin_fabsf(box[6]), __builtin_fabsf(box[7]));
    if (tid < ny) {
        SrcBlockInv& b = s_src[tid];
        const SrcPairInv si = make_src_inv(a.src_segs[y0 + tid], s_cam);
        // a zero margin (degenerate line) gives inf/nan below: comparisons fail, nothing is culled
        const float m1 = kWedgeTau * (__builtin_fabsf(si.epi_p1.x) * ext0 + __builtin_fabsf(si.epi_p1.y) * ext1 + __builtin_fabsf(si.epi_p1.z));
        const float m2 = kWedgeTau * (__builtin_fabsf(si.epi_p2.x) * ext0 + __builtin_fabsf(si.epi_p2.y) * ext1 + __builtin_fabsf(si.epi_p2.z));
        const float i1 = 1.0f / m1, i2 = 1.0f / m2;
        b.s = si;
        b.im1 = i1; b.im2 = i2;
        const f3 e1s = i1 * si.epi_p1, e2s = i2 * si.epi_p2;
        if constexpr (kDepth) { b.ray1 = normalize(mat3_apply(Rs, si.p1)); b.ray2 = normalize(mat3_apply(Rs, si.p2)); }
        // e_d = e1 - e2 against the bounding box of the tile's target endpoints (e_d cancels: its margin is the sum of the two
        // lines' margins, not 1e-4 of its own small terms).  Where it may cross the box the sector test of this source segment
        // is switched off (zero lines: all four values 0, never > 1 or < -1); level 2 decides those pairs.
        const f3 ed = si.epi_p1 - si.epi_p2;
        const float md = m1 + m2;
        const float lo = ed.z + __builtin_fminf(ed.x * box[0], ed.x * box[1]) + __builtin_fminf(ed.y * box[2], ed.y * box[3]);
        const float hi = ed.z + __builtin_fmaxf(ed.x * box[0], ed.x * box[1]) + __builtin_fmaxf(ed.y * box[2], ed.y * box[3]);
        const bool dsafe = lo > md || hi < -md;                // (NaN/inf: comparisons fail -> not safe)
        if (a.dbg && !dsafe) atomicAdd(&a.dbg[4], 256ull);    // (diagnostic: pairs whose source-side sector test is off)
        const f3 e1w = dsafe ? e1s : mk3(0.0f, 0.0f, 0.0f), e2w = dsafe ? e2s : mk3(0.0f, 0.0f, 0.0f);
        SrcLevel1 l1;
        l1.e1x = e1w.x; l1.e1y = e1w.y; l1.e1z = e1w.z; l1.p1x = si.p1.x;
        l1.e2x = e2w.x; l1.e2y = e2w.y; l1.e2z = e2w.z; l1.p1y = si.p1.y;
        l1.p2x = si.p2.x; l1.p2y = si.p2.y; l1.pad0 = 0.0f; l1.pad1 = 0.0f;
        s_l1[tid] = l1;
    }
    f3 eq1s, eq2s;                           // this lane's epipolar lines (in the source image) over their margins, for level 1
    {
        const float mq1 = kWedgeTau * (__builtin_fabsf(t.epi_q1.x) * ext2 + __builtin_fabsf(t.epi_q1.y) * ext3 + __builtin_fabsf(t.epi_q1.z));
        const float mq2 = kWedgeTau * (__builtin_fabsf(t.epi_q2.x) * ext2 + __builtin_fabsf(t.epi_q2.y) * ext3 + __builtin_fabsf(t.epi_q2.z));
        const float iq1 = 1.0f / mq1, iq2 = 1.0f / mq2;
        eq1s = iq1 * t.epi_q1; eq2s = iq2 * t.epi_q2;
        s_tgt[tid].im1 = iq1; s_tgt[tid].im2 = iq2;                                // (level 2 re-forms the real lines from LDS)
        const f3 ed = t.epi_q1 - t.epi_q2;
        const float md = mq1 + mq2;
        const float lo = ed.z + __builtin_fminf(ed.x * box[4], ed.x * box[5]) + __builtin_fminf(ed.y * box[6], ed.y * box[7]);
        const float hi = ed.z + __builtin_fmaxf(ed.x * box[4], ed.x * box[5]) + __builtin_fmaxf(ed.y * box[6], ed.y * box[7]);
        if (!(lo > md || hi < -md)) {                                                                   // this target's sector test is off
            eq1s = mk3(0.0f, 0.0f, 0.0f); eq2s = mk3(0.0f, 0.0f, 0.0f);
            if (a.dbg && valid) atomicAdd(&a.dbg[5], (unsigned long long)ny);
        }
    }
    const float ext = __builtin_fmaxf(__builtin_fmaxf(ext0, ext1), __builtin_fmaxf(ext2, ext3));
    __syncthreads();

    unsigned short* qa = s_qa[wave];
    unsigned short* qb = s_qb[wave];
    const TgtBlockInv* tw = s_tgt + wave * 64;
    int ha = 0, ca = 0, hb = 0, cb = 0;            // wave-uniform ring states
    int n_l1 = 0, n_l2 = 0;                        // diagnostic counters
    // (level 2 only inside the validated range of coordinates: beyond 2^15 pixels -- no test image is a quarter of that -- every pair level 1 keeps takes the exact test;
    // iou_bounds abstains by itself where kLineCond (ext / len)^2 >= 10, i.e. for segments shorter than ext / 4472: DESIGN.md section 4, tests/test_gpu_bound_check.py)
    const bool use_wedge = (a.wedge_pretest & 1) != 0, use_iou = (a.wedge_pretest & 2) != 0 && ext < 32768.0f;
    // level 2 also ACCEPTS (bit set, no exact test) where the lower bounds of both ratios clear the thresholds: the exact test's only product
    // here is the bit (the depths and their sign test are k_pair_fill's).  Off with bit 2 of the switch (A/B: identical bit rows, tests/)
    const bool use_accept = !kDepth && use_iou && (a.wedge_pretest & 4) == 0;

    // One extra iteration (k == ny) flushes the rings.  Each level's body appears exactly once (a single loop drains
    // whichever ring is due), so nothing is outlined and no state lives in scratch.
    for (int k = 0; k <= ny; ++k) {
        const bool last = k == ny;
        if (!last) {
            // level 1: wedge test (broadcast reads of the source invariants)
            const SrcLevel1 sl = s_l1[k];
            bool cand = valid;
            if (use_wedge) {
                // both endpoints against one line at once: packed FP32 FMAs (v_pk_fma_f32), two lanes of work per instruction
                const v2f qx = { t.q1.x, t.q2.x }, qy = { t.q1.y, t.q2.y };
                const v2f a12 = pk_line(mk3(sl.e1x, sl.e1y, sl.e1z), qx, qy), a34 = pk_line(mk3(sl.e2x, sl.e2y, sl.e2z), qx, qy);
                const float lo2 = __builtin_fminf(__builtin_fminf(a12.x, a12.y), __builtin_fminf(a34.x, a34.y));
                const float hi2 = __builtin_fmaxf(__builtin_fmaxf(a12.x, a12.y), __builtin_fmaxf(a34.x, a34.y));
                const bool out2 = lo2 > 1.0f || hi2 < -1.0f;        // target segment strictly inside one same-sign sector of e1, e2 (and off e_d)
                const v2f px = { sl.p1x, sl.p2x }, py = { sl.p1y, sl.p2y };
                const v2f b12 = pk_line(eq1s, px, py), b34 = pk_line(eq2s, px, py);
                const float lo1 = __builtin_fminf(__builtin_fminf(b12.x, b12.y), __builtin_fminf(b34.x, b34.y));
                const float hi1 = __builtin_fmaxf(__builtin_fmaxf(b12.x, b12.y), __builtin_fmaxf(b34.x, b34.y));
                const bool out1 = lo1 > 1.0f || hi1 < -1.0f;        // the same for the source segment and the target's lines
                cand = valid && !out1 && !out2;
            }
            const unsigned long long cm = __ballot(cand);
            if (cm) {
                if (cand) qa[(ha + ca + __popcll(cm & ((1ull << lane) - 1ull))) & (kPairQueue - 1)] = (unsigned short)(k | (lane << 8));
                ca += __popcll(cm);
                n_l1 += __popcll(cm);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
        }
        for (;;) {
            if (cb >= 64 || (last && ca == 0 && cb > 0)) {
                // level 3: exact overlap test + triangulation (the reference's float sequence) for up to 64 queued pairs
                const int n = min(cb, 64);
                const unsigned key = qb[(hb + lane) & (kPairQueue - 1)];
                const int kk = key & 0xff, origin = (key >> 8) & 63;
                if (lane < n) {
                    const SrcBlockInv& sb = s_src[kk];
                    const TgtBlockInv& tb = tw[origin];
                    f3 l2_p1, l2_p2, l1_q1, l1_q2;
                    if (pair_overlap_test(sb.s, tb.t, l2_p1, l2_p2, l1_q1, l1_q2)) {
                        if constexpr (kDepth) {
                            const float4 d = pair_depths_pre(sb.ray1, sb.ray2, tb.ray1, tb.ray2, l2_p1, l2_p2, l1_q1, l1_q2, Rs, Rt, C_src, C_tgt);
                            if (d.x > 0.0f && d.y > 0.0f && d.z > 0.0f && d.w > 0.0f)     // cudawrapper.cu:931
                                atomicOr(&s_bits[kk * 4 + wave], 1ull << origin);
                        } else {
                            atomicOr(&s_bits[kk * 4 + wave], 1ull << origin);               // (the depths and their sign test: k_pair_fill)
                        }
                    }
                }
                hb = (hb + n) & (kPairQueue - 1);
                cb -= n;
                continue;
            }
            if (ca >= 64 || (last && ca > 0)) {
                // level 2: overlap-bound test for up to 64 queued pairs; survivors move on to the exact ring (cb < 64 here)
                const int n = min(ca, 64);
                const unsigned key = qa[(ha + lane) & (kPairQueue - 1)];
                const int kk = key & 0xff, origin = (key >> 8) & 63;
                bool pass = lane < n;
                if (pass && use_iou) {
                    const SrcBlockInv& sb = s_src[kk];
                    const TgtBlockInv& tb = tw[origin];
                    const f3 p1 = sb.s.p1, p2 = sb.s.p2, q1 = tb.t.q1, q2 = tb.t.q2;
                    // interval of the epipolar lines of q1/q2 on the source segment, of p1/p2 on the target segment
                    const f3 te1 = tb.im1 * tb.t.epi_q1, te2 = tb.im2 * tb.t.epi_q2, se1 = sb.im1 * sb.s.epi_p1, se2 = sb.im2 * sb.s.epi_p2;
                    const float b1 = fline(te1, p1.x, p1.y), b2 = fline(te1, p2.x, p2.y);
                    const float b3 = fline(te2, p1.x, p1.y), b4 = fline(te2, p2.x, p2.y);
                    const float a1 = fline(se1, q1.x, q1.y), a2 = fline(se1, q2.x, q2.y);
                    const float a3 = fline(se2, q1.x, q1.y), a4 = fline(se2, q2.x, q2.y);
                    const float rb1 = __builtin_amdgcn_rcpf(b1 - b2), rb2 = __builtin_amdgcn_rcpf(b3 - b4);
                    const float ra1 = __builtin_amdgcn_rcpf(a1 - a2), ra2 = __builtin_amdgcn_rcpf(a3 - a4);
                    const float ls = sb.s.len, lt = tb.t.len;
                    const IouBounds o1 = iou_bounds(b1 * rb1, rb1, b3 * rb2, rb2, ls, ext * __builtin_amdgcn_rcpf(ls));
                    const IouBounds o2 = iou_bounds(a1 * ra1, ra1, a3 * ra2, ra2, lt, ext * __builtin_amdgcn_rcpf(lt));
                    const bool rej = __builtin_fmaxf(o1.upper, o2.upper) < kMinOverlapUpper - kIouSlack || __builtin_fminf(o1.upper, o2.upper) < kMinOverlapLower - kIouSlack;
                    // (segment lengths of at least a pixel: the reference's first exit, :211, on the exact invariants)
                    const bool acc = use_accept && ls >= 1.0f && lt >= 1.0f && __builtin_fminf(o1.lower, o2.lower) > kMinOverlapLower + kIouSlack &&
                                     __builtin_fmaxf(o1.lower, o2.lower) > kMinOverlapUpper + kIouSlack;
                    if (acc) atomicOr(&s_bits[kk * 4 + wave], 1ull << origin);
#ifdef L3D_BOUND_CHECK
                    if ((acc || rej) && a.dbg) {
                        f3 c1, c2, c3, c4; bool v1, v2, v3, v4;
                        c1 = hom_normalize(cross(tb.t.line2, sb.s.epi_p1), v1); c2 = hom_normalize(cross(tb.t.line2, sb.s.epi_p2), v2);
                        c3 = hom_normalize(cross(sb.s.line1, tb.t.epi_q1), v3); c4 = hom_normalize(cross(sb.s.line1, tb.t.epi_q2), v4);
                        const float ov1 = segment_overlap(sb.s.p1, sb.s.p2, sb.s.len, c3, c4, seglen2d(c3, c4));
                        const float ov2 = segment_overlap(tb.t.q1, tb.t.q2, tb.t.len, c1, c2, seglen2d(c1, c2));
                        const bool ex = v1 && v2 && v3 && v4 && __builtin_fminf(ov1, ov2) > kMinOverlapLower && __builtin_fmaxf(ov1, ov2) > kMinOverlapUpper;
                        if (acc ? !ex : ex) {                      // (an accept the exact test rejects / a reject it keeps)
                            const unsigned long long idx = atomicAdd(&a.dbg[6], 1ull);
                            if (idx < 48) {
                                unsigned long long* r = a.dbg + 8 + idx * 8;
                                r[0] = ((unsigned long long)a.dbg_view << 32) | (unsigned)(y0 + kk);
                                r[1] = ((unsigned long long)cam << 32) | (unsigned)(blockIdx.x * 256 + wave * 64 + origin);
                                float* f = reinterpret_cast<float*>(r + 4);
                                f[0] = acc ? o1.lower : o1.upper; f[1] = acc ? o2.lower : o2.upper; f[2] = ov1; f[3] = ov2; f[4] = b1 * rb1; f[5] = b3 * rb2; f[6] = a1 * ra1; f[7] = a3 * ra2;
                            }
                        }
                    }
#endif
                    pass = !rej && !acc;
                }
                const unsigned long long pm = __ballot(pass);
                if (pm) {
                    if (pass) qb[(hb + cb + __popcll(pm & ((1ull << lane) - 1ull))) & (kPairQueue - 1)] = (unsigned short)key;
                    cb += __popcll(pm);
                    n_l2 += __popcll(pm);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                }
                ha = (ha + n) & (kPairQueue - 1);
                ca -= n;
                continue;
            }
            break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    for (int r = lane; r < ny; r += 64) a.mask[((size_t)j * a.S_src + (y0 + r)) * a.W64 + blockIdx.x * 4 + wave] = s_bits[r * 4 + wave];
    if (a.rowcnt) {
        // this tile's share of the row counts (stage 1b folded in: one atomic per source segment and workgroup)
        __syncthreads();                                       // every wave's bits of the tile are in s_bits
        if (tid < ny) {
            const int cnt = __popcll(s_bits[tid * 4]) + __popcll(s_bits[tid * 4 + 1]) + __popcll(s_bits[tid * 4 + 2]) + __popcll(s_bits[tid * 4 + 3]);
            if (cnt) atomicAdd(&a.rowcnt[(y0 + tid) * a.N + cam], cnt);
        }
        if (a.rowblk && wave == 0) {
            // block sums (256 rows): this workgroup's rows lie N apart, i.e. in a handful of consecutive blocks -- summed in LDS
            // first, one global atomic per block and workgroup (one per row made every workgroup of the launch queue up on the
            // same ~100 counters: k_pair_mask 6.8 -> 18.7 ms)
            const int b0 = (y0 * a.N + cam) >> 8;
            const int nb = (((y0 + ny - 1) * a.N + cam) >> 8) - b0 + 1;        // <= kSrcPerBlock * N / 256 + 2
            for (int l = lane; l < nb; l += 64) s_blk[l] = 0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            for (int r = lane; r < ny; r += 64) {                             // (every row of the workgroup, whatever kSrcPerBlock is)
                const int cnt = __popcll(s_bits[r * 4]) + __popcll(s_bits[r * 4 + 1]) + __popcll(s_bits[r * 4 + 2]) + __popcll(s_bits[r * 4 + 3]);
                if (cnt) atomicAdd(&s_blk[(((y0 + r) * a.N + cam) >> 8) - b0], cnt);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            for (int l = lane; l < nb; l += 64) { const int v = s_blk[l]; if (v) atomicAdd(&a.rowblk[b0 + l], v); }
        }
    }
    if (a.dbg) {
        int nbits = 0;
        for (int r = lane; r < ny; r += 64) nbits += __popcll(s_bits[r * 4 + wave]);
        for (int o = 32; o > 0; o >>= 1) nbits += __shfl_down(nbits, o);
        if (lane == 0) {
            atomicAdd(&a.dbg[0], (unsigned long long)ny * (unsigned long long)min(64, width - (blockIdx.x * 256 + wave * 64) > 0 ? width - (blockIdx.x * 256 + wave * 64) : 0));
            atomicAdd(&a.dbg[1], (unsigned long long)n_l1); atomicAdd(&a.dbg[2], (unsigned long long)n_l2); atomicAdd(&a.dbg[3], (unsigned long long)nbits);
        }
    }
}

// Stage 1b.  One wave per (src segment, tbm camera) row.
__global__ __launch_bounds__(256) void k_row_count(PairArgs a, int* __restrict__ rowcnt)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int nrows = (a.seg_end - a.seg_begin) * a.n_tbm;
    if (row >= nrows) return;
    const int j = row % a.n_tbm;
    const int y = a.seg_begin + row / a.n_tbm;
    const int cam = a.tbm[j];
    const int nw = (a.offsets[cam].y + 63) >> 6;
    const unsigned long long* m = a.mask + ((size_t)j * a.S_src + y) * a.W64;
    int c = 0;
    for (int w = lane; w < nw; w += 64) c += __popcll(m[w]);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if (lane == 0) rowcnt[y * a.N + cam] = c;
}

__global__ void k_exist_hist(const ExistRec* __restrict__ ex, int n, int N, int* __restrict__ rowcnt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&rowcnt[ex[i].seg * N + ex[i].cam], 1);
}

// Exclusive scan of n ints, one independent workgroup per 4096-int tile (l3d_scan.hpp); out has n+1 entries, `zero`
// (optional) gets n zeros; one more workgroup orders the segments longest first (optional); total_out (optional) gets the total.
__global__ __launch_bounds__(kTileThreads) void k_scan(const int* __restrict__ in, int* __restrict__ out, int n, int* __restrict__ zero,
                                                       int* __restrict__ seg_order, int N, int seg_begin, int seg_end,
                                                       const int* __restrict__ rowcnt_all, int* __restrict__ total_out, int* __restrict__ stats_out)
{
    __shared__ int s_w[8];
    __shared__ int s_hist[130];
    const int n_tiles = max(1, (n + kTileInts - 1) / kTileInts);
    const int b = (int)blockIdx.x;
    if (b < n_tiles) wg_scan_excl_tile(in, out, n, zero, b, s_w, total_out);                          // one independent workgroup per tile
    else if (seg_order && b == n_tiles) wg_segment_order(rowcnt_all, N, seg_begin, seg_end, seg_order, s_hist);   // rows -> segments, longest first
    else if (stats_out) wg_raw_stats(rowcnt_all, N, seg_begin, seg_end, stats_out, s_w);              // total and per-segment maximum
}

// r-th (0-based) set bit of a 64-bit word
__device__ __forceinline__ int select_bit(unsigned long long m, int r)
{
    int pos = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        const unsigned long long lo = m & ((1ull << w) - 1ull);
        const int c = __popcll(lo);
        if (r >= c) { r -= c; m >>= w; pos += w; } else { m = lo; }
    }
    return pos;
}

// Unit viewing rays of the target endpoints, normalize(RtKinv_cam * (x, y, 1)) -- what K_pairwise_matches recomputes for every pair
// (cudawrapper.cu:590-601 through D_get_ray_tgt:288) depends on the neighbour's camera and segment only: once per chain for every
// entry of every view's target array, read (2 x 16 bytes) by k_pair_fill instead of two mat-vecs, two square roots and six
// correctly rounded divisions per candidate.  Same operations, same bits.
__global__ __launch_bounds__(256) void k_tgt_rays(const RayJob* __restrict__ jobs)
{
    const RayJob jb = jobs[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= jb.n_tgt) return;
    int cam = jb.offsets ? -1 : 0;                       // (no offsets: one camera for all entries -- a view's own segments under its own camera)
    if (jb.offsets) for (int c = 0; c < jb.N; ++c) { const int2 o = jb.offsets[c]; if (i >= o.x && i < o.x + o.y) cam = c; }
    if (cam < 0) return;
    const float4 t = jb.tgt[i];
    const f3 r1 = normalize(mat3_apply(jb.RtKinv + cam * 9, mk3(t.x, t.y, 1.0f)));
    const f3 r2 = normalize(mat3_apply(jb.RtKinv + cam * 9, mk3(t.z, t.w, 1.0f)));
    jb.out[2 * (size_t)i] = make_float4(r1.x, r1.y, r1.z, 0.0f);
    jb.out[2 * (size_t)i + 1] = make_float4(r2.x, r2.y, r2.z, 0.0f);
}

// Stage 1c.  One wave per (src segment, tbm camera) row: the row's set bits are enumerated in
// ascending target order, 64 at a time with all lanes busy, and the depth record of each is written
// to slot row_start + rank -> candidates come out sorted (seg, cam, tgt) with no sort pass.
#ifndef L3D_PF_WAVES
#define L3D_PF_WAVES 0
#endif
__global__ __launch_bounds__(256)
#if L3D_PF_WAVES
__attribute__((amdgpu_waves_per_eu(L3D_PF_WAVES, L3D_PF_WAVES)))
#endif
void k_pair_fill(PairArgs a, const int* __restrict__ row_start,
                                                   uint2* __restrict__ cand_meta, float4* __restrict__ cand_depths)
{
    __shared__ unsigned long long s_words[4][kMaxW64];
    __shared__ int s_pref[4][kMaxW64 + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    const int nrows = (a.seg_end - a.seg_begin) * a.n_tbm;
    if (row >= nrows) return;
    const int j = row % a.n_tbm;
    const int y = a.seg_begin + row / a.n_tbm;
    const int cam = a.tbm[j];
    const int nw = (a.offsets[cam].y + 63) >> 6;
    const int toff = a.offsets[cam].x;
    const unsigned long long* m = a.mask + ((size_t)j * a.S_src + y) * a.W64;

    // words + exclusive prefix of popcounts (nw <= kMaxW64; chunks of 64 words)
    int base = 0;
    for (int w0 = 0; w0 < nw; w0 += 64) {
        const int w = w0 + lane;
        const unsigned long long word = w < nw ? m[w] : 0ull;
        int c = __popcll(word);
        int incl = c;
        for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        if (w < nw) { s_words[wave][w] = word; s_pref[wave][w] = base + incl - c; }
        base += __shfl(incl, 63);
    }
    const int total = base;
    if (lane == 0) s_pref[wave][nw] = total;
    if (total == 0) return;
    int slot0;
    if (a.rowub) {
        // the row's start from k_pair_mask's counters: 256-row block sums in front of the row's block + the rows in front of it
        // inside the block (about 350 ints per wave, L2 resident) -- no scan launch between k_pair_mask and this kernel
        const int idx = y * a.N + cam, blk = idx >> 8, nblk = (a.S_src * a.N + 255) >> 8;
        int before = 0, all = 0;
        for (int b = lane; b < nblk; b += 64) { const int v = a.rowblk[b]; all += v; if (b < blk) before += v; }
        for (int i = (blk << 8) + lane; i < idx; i += 64) before += a.rowub[i];
        for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o); all += __shfl_xor(all, o); }
        if (a.cand_cap && all > a.cand_cap) {                                  // overflow: the chain is re-run with more room -- it finds out
            if (lane == 0) a.rowcnt[idx] = total;                              // from the final counts, which then carry the upper bounds
            return;
        }
        slot0 = before;
        if (lane == 0) a.rowstart_out[idx] = before;
    } else {
        if (a.cand_cap && row_start[(size_t)a.S_src * a.N] > a.cand_cap) return;   // overflow: the chain is re-run with more room
        slot0 = row_start[y * a.N + cam];
    }

    const SrcPairInv s = make_src_inv(a.src_segs[y], a.F + cam * 9);
    const f3 C_tgt = mk3(a.centers[cam * 3], a.centers[cam * 3 + 1], a.centers[cam * 3 + 2]);
    const f3 C_src = mk3(a.C_src[0], a.C_src[1], a.C_src[2]);
    f3 ray_p1, ray_p2;                                                                                                  // row invariants
    if (a.src_rays) {                      // (the same table for the view's own segments: the same operations once per chain instead of once per row)
        const float4 r1 = a.src_rays[2 * (size_t)y], r2 = a.src_rays[2 * (size_t)y + 1];
        ray_p1 = mk3(r1.x, r1.y, r1.z); ray_p2 = mk3(r2.x, r2.y, r2.z);
    } else {
        ray_p1 = normalize(mat3_apply(a.RtKinv_src, s.p1)); ray_p2 = normalize(mat3_apply(a.RtKinv_src, s.p2));
    }

    int written = 0;                       // depth_in_fill: records of the row so far (wave-uniform)
    for (int k0 = 0; k0 < total; k0 += 64) {
        const int k = k0 + lane;
        const bool kv = k < total;
        int x = 0;
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kv) {
            int lo = 0, hi = nw;               // largest w with pref[w] <= k
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_pref[wave][mid] <= k) lo = mid; else hi = mid; }
            x = lo * 64 + select_bit(s_words[wave][lo], k - s_pref[wave][lo]);
            const TgtPairInv t = make_tgt_inv(a.tgt_segs[toff + x], a.F + cam * 9);
            // the bit is set, so the overlap test passed: only its intersection points are needed again
            f3 l2_p1, l2_p2, l1_q1, l1_q2;
            pair_intersections(s, t, l2_p1, l2_p2, l1_q1, l1_q2);
            f3 ray_q1, ray_q2;
            if (a.tgt_rays) {
                const float4 r1 = a.tgt_rays[2 * (size_t)(toff + x)], r2 = a.tgt_rays[2 * (size_t)(toff + x) + 1];
                ray_q1 = mk3(r1.x, r1.y, r1.z); ray_q2 = mk3(r2.x, r2.y, r2.z);
            } else {
                ray_q1 = normalize(mat3_apply(a.RtKinv + cam * 9, t.q1)); ray_q2 = normalize(mat3_apply(a.RtKinv + cam * 9, t.q2));
            }
            d = pair_depths_pre(ray_p1, ray_p2, ray_q1, ray_q2, l2_p1, l2_p2, l1_q1, l1_q2, a.RtKinv_src, a.RtKinv + cam * 9, C_src, C_tgt);
        }
        if (a.depth_in_fill) {
            // the bit only says "overlap test passed": a candidate needs four positive depths (cudawrapper.cu:931); the row is packed
            const bool ok = kv && d.x > 0.0f && d.y > 0.0f && d.z > 0.0f && d.w > 0.0f;
            const unsigned long long om = __ballot(ok);
            if (ok) {
                const int pos = slot0 + written + __popcll(om & ((1ull << lane) - 1ull));
                cand_meta[pos] = make_uint2((unsigned)x, (unsigned)cam);
                cand_depths[pos] = d;
            }
            written += __popcll(om);
        } else if (kv) {
            cand_meta[slot0 + k] = make_uint2((unsigned)x, (unsigned)cam);
            cand_depths[slot0 + k] = d;
        }
    }
    if (a.depth_in_fill && lane == 0) a.rowcnt[y * a.N + cam] = written;           // the row's true count replaces the upper bound
}

__global__ void k_exist_place(const ExistRec* __restrict__ ex, int n, int N, const int* __restrict__ row_start,
                              uint2* __restrict__ cand_meta, float4* __restrict__ cand_depths)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ExistRec r = ex[i];
    const int slot = row_start[r.seg * N + r.cam] + r.rank;
    cand_meta[slot] = make_uint2(r.tgt, r.cam);
    cand_depths[slot] = make_float4(r.d[0], r.d[1], r.d[2], r.d[3]);
}

// =================================================================================================
// Stage 2.  One workgroup per source segment.  Every candidate record is both a hypothesis (its own
// 3-D segment X1,X2) and a witness for the other hypotheses of the same source segment.
// Candidate tiles are staged in LDS together with everything that depends on the candidate only
// (3-D endpoints, unit direction, normalised target line); the per-(hypothesis, camera) projections
// are refreshed when the (wave-uniform) witness camera changes.
// =================================================================================================
struct Witness {            // 20 floats
    float X1[3], X2[3], v[3];
    float l2[3], den2;      // target line and sqrt(l.x^2+l.y^2)
    float q[4];
    int cam;
    int pad;
};

__global__ __launch_bounds__(256) void k_verify(VerifyArgs a)
{
    __shared__ Witness s_w[kVerifyTile];
    __shared__ float s_ray[6];

    const int y = a.seg_begin + blockIdx.x;
    const int tid = threadIdx.x;
    const int start = a.row_start[y * a.N];
    const int m = a.row_start[(y + 1) * a.N] - start;
    if (m == 0 || m <= a.only_above) return;
    if (a.cand_cap && a.row_start[a.nrow_total] > a.cand_cap) return;   // candidate overflow: the chain is re-run

    const f3 C = mk3(a.C_src[0], a.C_src[1], a.C_src[2]);
    if (tid == 0) {
        const float4 s = a.src_segs[y];
        const f3 r1 = normalize(mat3_apply(a.RtKinv_src, mk3(s.x, s.y, 1.0f)));
        const f3 r2 = normalize(mat3_apply(a.RtKinv_src, mk3(s.z, s.w, 1.0f)));
        s_ray[0] = r1.x; s_ray[1] = r1.y; s_ray[2] = r1.z;
        s_ray[3] = r2.x; s_ray[4] = r2.y; s_ray[5] = r2.z;
    }
    __syncthreads();
    const f3 ray1 = mk3(s_ray[0], s_ray[1], s_ray[2]);
    const f3 ray2 = mk3(s_ray[3], s_ray[4], s_ray[5]);

    const float two_sig_d = 2.0f * (a.sigma_p * a.sigma_p);
    const float two_sig_a = 2.0f * (a.sigma_a * a.sigma_a);

    for (int h0 = 0; h0 < m; h0 += 256) {
        const int h = h0 + tid;
        const bool hv = h < m;
        // hypothesis registers
        f3 X1 = mk3(0, 0, 0), X2 = mk3(0, 0, 0), v1 = mk3(0, 0, 0);
        float T1 = -1.0f, T2 = -1.0f;
        int cam_h = -1;
        if (hv) {
            const float4 d = a.cand_depths[start + h];
            cam_h = (int)a.cand_meta[start + h].y;
            X1 = C + d.x * ray1;                 // D_unproject_point_src, cudawrapper.cu:338-344
            X2 = C + d.y * ray2;
            v1 = normalize(X1 - X2);
            if (a.spatial_k > 0.0f) {
                T1 = sq_threshold(a.spatial_k * length(C - X1));
                T2 = sq_threshold(a.spatial_k * length(C - X2));
            }
        }
        float conf_sum = 0.0f, cur_max = 0.0f;
        int cur_cam = -1;                       // wave-uniform
        f3 pr1 = mk3(0, 0, 0), pr2 = mk3(0, 0, 0), line1 = mk3(0, 0, 0);
        float den1 = 1.0f;
        bool pvalid = false;

        for (int c0 = 0; c0 < m; c0 += kVerifyTile) {
            __syncthreads();
            // stage witnesses c0 .. c0+tile
            for (int i = tid; i < kVerifyTile && c0 + i < m; i += 256) {
                const uint2 meta = a.cand_meta[start + c0 + i];
                const float4 d = a.cand_depths[start + c0 + i];
                const f3 Q1 = C + d.x * ray1;
                const f3 Q2 = C + d.y * ray2;
                const f3 v2 = normalize(Q1 - Q2);
                const float4 tq = a.tgt_segs[a.offsets[meta.y].x + meta.x];
                const f3 q1 = mk3(tq.x, tq.y, 1.0f), q2 = mk3(tq.z, tq.w, 1.0f);
                const f3 l2 = cross(q1, q2);
                Witness w;
                w.X1[0] = Q1.x; w.X1[1] = Q1.y; w.X1[2] = Q1.z;
                w.X2[0] = Q2.x; w.X2[1] = Q2.y; w.X2[2] = Q2.z;
                w.v[0] = v2.x; w.v[1] = v2.y; w.v[2] = v2.z;
                w.l2[0] = l2.x; w.l2[1] = l2.y; w.l2[2] = l2.z;
                w.den2 = line_norm2d(l2);
                w.q[0] = tq.x; w.q[1] = tq.y; w.q[2] = tq.z; w.q[3] = tq.w;
                w.cam = (int)meta.y;
                w.pad = 0;
                s_w[i] = w;
            }
            __syncthreads();
            const int nt = min(kVerifyTile, m - c0);
            for (int i = 0; i < nt; ++i) {
                const Witness& w = s_w[i];
                const int cam2 = w.cam;                       // uniform
                if (cam2 != cur_cam) {                         // cudawrapper.cu:677-687
                    conf_sum += cur_max;
                    cur_max = 0.0f;
                    cur_cam = cam2;
                    bool va, vb;
                    pr1 = project(a.P + cam2 * 12, X1, va);   // :690-693
                    pr2 = project(a.P + cam2 * 12, X2, vb);
                    pvalid = va && vb;
                    line1 = cross(pr1, pr2);
                    den1 = line_norm2d(line1);
                }
                if (!hv || cam2 == cam_h || !pvalid) continue;   // :658,:674 (own slot has cam2 == cam_h)
                // 3-D gate, :388-401
                if (a.spatial_k > 0.0f) {
                    const f3 e1 = X1 - mk3(w.X1[0], w.X1[1], w.X1[2]);
                    const f3 e2 = X2 - mk3(w.X2[0], w.X2[1], w.X2[2]);
                    if (dot(e1, e1) > T1 || dot(e2, e2) > T2) continue;
                }
                // 2-D distance, :404-417 (x/d is monotone for d>0: max of quotients = quotient of max)
                const f3 l2 = mk3(w.l2[0], w.l2[1], w.l2[2]);
                const f3 q1 = mk3(w.q[0], w.q[1], 1.0f), q2 = mk3(w.q[2], w.q[3], 1.0f);
                const float d1 = __builtin_fmaxf(__builtin_fabsf(line_numer(l2, pr1) / w.den2),
                                                 __builtin_fabsf(line_numer(l2, pr2) / w.den2));
                const float d2 = __builtin_fmaxf(__builtin_fabsf(line_numer(line1, q1) / den1),
                                                 __builtin_fabsf(line_numer(line1, q2) / den1));
                const float dist = __builtin_fmaxf(d1, d2);
                // angle, :118-130 (double arithmetic: CUDART_PI is a double literal)
                const float cs = __builtin_fmaxf(__builtin_fminf(dot(v1, mk3(w.v[0], w.v[1], w.v[2])), 1.0f), -1.0f);
                float angle = (float)((double)c_acosf(cs) / 3.1415926535897931e+0 * (double)180.0f);
                if (angle > 90.0f) angle = 180.0f - angle;
                const float cd = c_expf(-dist * dist / two_sig_d);
                const float conf = __builtin_fminf(cd, c_expf(-angle * angle / two_sig_a));
                if (conf > 0.5f && conf > cur_max) cur_max = conf;   // :699-704
            }
        }
        conf_sum += cur_max;                                    // :709
        if (hv) a.cand_conf[start + h] = conf_sum;
        __syncthreads();
    }
}

// Per segment (one wave each): first strict maximum of the confidences -> best depths for the
// median (cudawrapper.cu:1037-1062) and the number of kept matches (conf > 1, :1096).
__global__ __launch_bounds__(256) void k_seg_post(VerifyArgs a, int* __restrict__ kept_cnt, float2* __restrict__ best_depths)
{
    const int y = a.seg_begin + blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (y >= a.seg_end) return;
    if (a.cand_cap && a.row_start[a.nrow_total] > a.cand_cap) { if (lane == 0) { kept_cnt[y] = 0; best_depths[y] = make_float2(-1.0f, -1.0f); } return; }
    const int start = a.row_start[y * a.N];
    const int m = a.row_start[(y + 1) * a.N] - start;
    float best = 0.0f;
    int best_i = 0x7fffffff;
    int kept = 0;
    for (int i = lane; i < m; i += 64) {
        const float c = a.cand_conf[start + i];
        kept += c > 1.0f;
        if (c > best) { best = c; best_i = i; }      // strided ascending: first index of this lane's max
    }
    for (int o = 32; o > 0; o >>= 1) {
        kept += __shfl_down(kept, o);
        const float ob = __shfl_down(best, o);
        const int oi = __shfl_down(best_i, o);
        if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
    }
    if (lane == 0) {
        kept_cnt[y] = kept;
        float2 bd = make_float2(-1.0f, -1.0f);         // marker: not part of the median list
        if (best > 0.5f) {                             // conf_t/2.0f
            const float4 d = a.cand_depths[start + best_i];
            bd = make_float2(d.x, d.y);
        }
        best_depths[y] = bd;
    }
}

__global__ __launch_bounds__(256) void k_kept_write(VerifyArgs a, const int* __restrict__ kept_start,
                                                    const unsigned* __restrict__ local2global, Match* __restrict__ out)
{
    __shared__ int s_cnt[32];
    const int y = a.seg_begin + blockIdx.x;                  // one workgroup per segment
    write_kept_segment_wg(a, y, kept_start[y], local2global, out, s_cnt);
}

// =================================================================================================
// Collinearity (K_collinearity, cudawrapper.cu:476-535).  One thread per (x<y) pair inside 64x... tiles;
// output: dense upper-triangle bit rows + a compact value written in a second pass on the host side
// from the few set bits.  Here: row y, lanes sweep x<y; ballot -> bit words; values recomputed on fill.
// =================================================================================================
__device__ __forceinline__ float collin_pair(float4 sx, float4 sy, float sigma_sqr)
{
    const f3 p1 = mk3(sx.x, sx.y, 1.0f), p2 = mk3(sx.z, sx.w, 1.0f);
    const f3 q1 = mk3(sy.x, sy.y, 1.0f), q2 = mk3(sy.z, sy.w, 1.0f);
    const f3 line1 = cross(p1, p2), line2 = cross(q1, q2);
    const float d1 = __builtin_fmaxf(dist_p2l(line2, p1), dist_p2l(line2, p2));
    const float d2 = __builtin_fmaxf(dist_p2l(line1, q1), dist_p2l(line1, q2));
    const float d = __builtin_fmaxf(d1, d2);
    const float aff = c_expf(-d * d / (2.0f * sigma_sqr));
    if (!(aff > kCollinAffT)) return 0.0f;
    const float pos1 = (q1.x - p1.x) * (q2.x - p1.x) + (q1.y - p1.y) * (q2.y - p1.y);
    const float pos2 = (q1.x - p2.x) * (q2.x - p2.x) + (q1.y - p2.y) * (q2.y - p2.y);
    const float pos3 = (p1.x - q1.x) * (p2.x - q1.x) + (p1.y - q1.y) * (p2.y - q1.y);
    const float pos4 = (p1.x - q2.x) * (p2.x - q2.x) + (p1.y - q2.y) * (p2.y - q2.y);
    if (pos1 > -kEpsG && pos2 > -kEpsG && pos3 > -kEpsG && pos4 > -kEpsG) return aff;
    return 0.0f;
}

// grid = (ceil(S/256), S): row = smaller index i, lanes = larger index j (> i).  mask[i][word].
__global__ __launch_bounds__(256) void k_collinearity(const float4* __restrict__ segs, int S, float sigma_sqr,
                                                      unsigned long long* __restrict__ mask, int W64, int* __restrict__ rowcnt)
{
    const int i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 + 255 <= i) return;               // whole tile at or below the diagonal
    bool ok = false;
    if (j < S && j > i) ok = collin_pair(segs[i], segs[j], sigma_sqr) > 0.0f;   // x=i < y=j
    const unsigned long long b = __ballot(ok);
    if ((threadIdx.x & 63) == 0) {
        mask[(size_t)i * W64 + blockIdx.x * 4 + (threadIdx.x >> 6)] = b;
        if (b) atomicAdd(&rowcnt[i], __popcll(b));
    }
}

// one wave per row i: write (i, j, w) for the set bits in ascending j
__global__ __launch_bounds__(256) void k_collinearity_fill(const float4* __restrict__ segs, int S, float sigma_sqr,
                                                           const unsigned long long* __restrict__ mask, int W64,
                                                           const int* __restrict__ row_start,
                                                           int* __restrict__ out_i, int* __restrict__ out_j, float* __restrict__ out_w)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= S) return;
    int o = row_start[i];
    const int nw = (S + 63) >> 6;
    for (int w = (i >> 6); w < nw; ++w) {
        const unsigned long long b = mask[(size_t)i * W64 + w];
        if (!b) continue;
        if ((b >> lane) & 1ull) {
            const int j = w * 64 + lane;
            const int pos = o + __popcll(b & ((1ull << lane) - 1ull));
            out_i[pos] = i; out_j[pos] = j;
            out_w[pos] = collin_pair(segs[i], segs[j], sigma_sqr);
        }
        o += __popcll(b);
    }
}

// =================================================================================================
// batched similarity_coll3D (l3d_similarity.hpp)
// =================================================================================================
__global__ void k_similarity(const Hypothesis* __restrict__ hyp, const int2* __restrict__ pairs, int n,
                             float sigma_a, float two_log, float* __restrict__ sim)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    sim[k] = similarity_coll3D(hyp[pairs[k].x], hyp[pairs[k].y], sigma_a, two_log);
}

__global__ void k_test_sqthr(const float* __restrict__ u, int n, float* __restrict__ walk, float* __restrict__ fast)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { walk[i] = sq_threshold_walk(u[i]); fast[i] = sq_threshold(u[i]); }
}

__global__ void k_test_math(const float* __restrict__ x, int n, float* __restrict__ e, float* __restrict__ ac, double* __restrict__ acd)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    e[i] = c_expf(x[i]);
    const float cl = __builtin_fmaxf(__builtin_fminf(x[i], 1.0f), -1.0f);
    ac[i] = c_acosf(cl);
    acd[i] = c_acos((double)cl);
}

// ---- launchers (host) ---------------------------------------------------------------------------
// Source segments per workgroup: kSrcPerBlock when that still fills the GPU, fewer for short source ranges (one rank's share
// of a view in the sharded chain: 250 segments at 8 GPUs would be 192 workgroups walking 64 sources each -- a launch bound by
// the latency of one workgroup).
int pair_mask_src_per_block(int n_src, int maxW, int n_tbm, int forced)
{
    if (forced > 0) return std::min(forced, kSrcPerBlock);
    const int tiles = (maxW + 255) / 256;
    int spb = kSrcPerBlock;
    while (spb > 8 && (long long)tiles * ((n_src + spb - 1) / spb) * n_tbm < 768) spb /= 2;
    return spb;
}
void launch_pair_mask(const PairArgs& a0, int maxW, hipStream_t st, int forced_spb)
{
    PairArgs a = a0;
    a.src_per_block = pair_mask_src_per_block(a.seg_end - a.seg_begin, maxW, a.n_tbm, forced_spb);
    dim3 grid((maxW + 255) / 256, (a.seg_end - a.seg_begin + a.src_per_block - 1) / a.src_per_block, a.n_tbm);
    if (a.depth_in_fill) hipLaunchKernelGGL(k_pair_mask<false>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_pair_mask<true>, grid, dim3(256), 0, st, a);
}
void launch_row_count(const PairArgs& a, int* rowcnt, hipStream_t st)
{
    const int nrows = (a.seg_end - a.seg_begin) * a.n_tbm;
    hipLaunchKernelGGL(k_row_count, dim3((nrows + 3) / 4), dim3(256), 0, st, a, rowcnt);
}
void launch_exist_hist(const ExistRec* ex, int n, int N, int* rowcnt, hipStream_t st)
{
    if (n) hipLaunchKernelGGL(k_exist_hist, dim3((n + 255) / 256), dim3(256), 0, st, ex, n, N, rowcnt);
}
void launch_scan(const int* in, int* out, int n, int* zero, hipStream_t st, int* seg_order, int N, int seg_begin, int seg_end, int* stats_out)
{
    const int n_tiles = std::max(1, (n + kTileInts - 1) / kTileInts);
    hipLaunchKernelGGL(k_scan, dim3(n_tiles + (seg_order ? 1 : 0) + (stats_out ? 1 : 0)), dim3(kTileThreads), 0, st, in, out, n, zero, seg_order, N, seg_begin, seg_end,
                       in, nullptr, stats_out);
}
// the rows of segments [seg_begin, seg_end) only (one rank's range of a view): out[row] for those rows, out[nrow_total] = their total
void launch_scan_range(const int* rowcnt, int* row_start, int N, int seg_begin, int seg_end, int nrow_total, int* zero, int* seg_order, hipStream_t st, int* stats_out)
{
    const int n = (seg_end - seg_begin) * N;
    const size_t o = (size_t)seg_begin * N;
    const int n_tiles = std::max(1, (n + kTileInts - 1) / kTileInts);
    hipLaunchKernelGGL(k_scan, dim3(n_tiles + (seg_order ? 1 : 0) + (stats_out ? 1 : 0)), dim3(kTileThreads), 0, st, rowcnt + o, row_start + o, n, zero ? zero + o : nullptr,
                       seg_order, N, seg_begin, seg_end, rowcnt, row_start + nrow_total, stats_out);
}
void launch_pair_fill(const PairArgs& a, const int* row_start, uint2* meta, float4* depths, hipStream_t st)
{
    const int nrows = (a.seg_end - a.seg_begin) * a.n_tbm;
    hipLaunchKernelGGL(k_pair_fill, dim3((nrows + 3) / 4), dim3(256), 0, st, a, row_start, meta, depths);
}
void launch_exist_place(const ExistRec* ex, int n, int N, const int* row_start, uint2* meta, float4* depths, hipStream_t st)
{
    if (n) hipLaunchKernelGGL(k_exist_place, dim3((n + 255) / 256), dim3(256), 0, st, ex, n, N, row_start, meta, depths);
}
void launch_verify(const VerifyArgs& a, hipStream_t st)
{
    hipLaunchKernelGGL(k_verify, dim3(a.seg_end - a.seg_begin), dim3(256), 0, st, a);
}
void launch_seg_post(const VerifyArgs& a, int* kept_cnt, float2* best, hipStream_t st)
{
    hipLaunchKernelGGL(k_seg_post, dim3((a.seg_end - a.seg_begin + 3) / 4), dim3(256), 0, st, a, kept_cnt, best);
}
void launch_kept_write(const VerifyArgs& a, const int* kept_start, const unsigned* l2g, Match* out, hipStream_t st)
{
    if (a.seg_end > a.seg_begin) hipLaunchKernelGGL(k_kept_write, dim3(a.seg_end - a.seg_begin), dim3(256), 0, st, a, kept_start, l2g, out);
}
void launch_tgt_rays(const RayJob* jobs, int n_jobs, int max_n_tgt, hipStream_t st)
{
    if (n_jobs > 0 && max_n_tgt > 0) hipLaunchKernelGGL(k_tgt_rays, dim3((max_n_tgt + 255) / 256, n_jobs), dim3(256), 0, st, jobs);
}
void launch_collinearity(const float4* segs, int S, float sigma_sqr, unsigned long long* mask, int W64, int* rowcnt, hipStream_t st)
{
    hipLaunchKernelGGL(k_collinearity, dim3((S + 255) / 256, S), dim3(256), 0, st, segs, S, sigma_sqr, mask, W64, rowcnt);
}
void launch_collinearity_fill(const float4* segs, int S, float sigma_sqr, const unsigned long long* mask, int W64,
                              const int* row_start, int* oi, int* oj, float* ow, hipStream_t st)
{
    hipLaunchKernelGGL(k_collinearity_fill, dim3((S + 3) / 4), dim3(256), 0, st, segs, S, sigma_sqr, mask, W64, row_start, oi, oj, ow);
}
void launch_similarity(const Hypothesis* hyp, const int2* pairs, int n, float sigma_a, float two_log, float* sim, hipStream_t st)
{
    if (n) hipLaunchKernelGGL(k_similarity, dim3((n + 255) / 256), dim3(256), 0, st, hyp, pairs, n, sigma_a, two_log, sim);
}
void launch_test_sqthr(const float* u, int n, float* walk, float* fast, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(k_test_sqthr, dim3((n + 255) / 256), dim3(256), 0, st, u, n, walk, fast);
}
void launch_test_math(const float* x, int n, float* e, float* ac, double* acd, hipStream_t st)
{
    if (n) hipLaunchKernelGGL(k_test_math, dim3((n + 255) / 256), dim3(256), 0, st, x, n, e, ac, acd);
}

}  // namespace l3d

// first use of a kernel loads its translation unit's code object (milliseconds): l3d_warm_up does that ahead of the first matchViews
void l3d::warm_kernels() { touch_kernel(reinterpret_cast<const void*>(&k_pair_mask<false>)); }
