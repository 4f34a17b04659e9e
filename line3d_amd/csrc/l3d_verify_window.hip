// l3d_verify_window.hip -- stage 2 (K_verify_matches, cudawrapper.cu:614-714) as a depth-window search.
//
// Observation: hypothesis y and witness i of the same source segment are unprojected along the SAME two
// source rays (Q1 = C + d1_i*ray1, P1 = C + d1_y*ray1, cudawrapper.cu:644-645,671-672), so the reference's
// 3-D gate |P1-Q1| <= k*depth1 && |P2-Q2| <= k*depth2 (:388-401) is, up to float rounding, a 1-D interval
// test on the depths.  Bucketing a segment's candidates by d1 turns the O(m^2) all-pairs loop into
// O(m * window) with the EXACT reference gate and confidence evaluated only inside a conservative window
// (the window margin provably covers the rounding of the 3-D computation, see window_margin below;
// DESIGN.md section 4).  Results are bit-identical to the all-pairs kernel (k_verify in l3d_kernels.hip,
// kept as the A/B reference and the fallback for huge segments).
//
// One workgroup per source segment.  The only global traffic is one coalesced read of the segment's candidate
// records (24 B each), L2-resident gathers of target segments for the few gate survivors, and the confidence
// store; everything else (3-D endpoints, directions, target lines) is recomputed from the depths with the same
// float operations the reference uses, hence the same bits.
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "l3d_geometry.hpp"
#include "l3d_kernels.hpp"
#include "l3d_options.hpp"
#include "l3d_kept.hpp"
#include "l3d_verify_eval.hpp"

namespace l3d {

// LDS image of one source segment: ALL its candidates bucketed by the first depth d1.  Depths are positive floats,
// whose bit patterns are monotone in the value and roughly logarithmic, so (bits >> kBucketShift) is an order
// preserving bucket id of relative width 2^-8 .. 2^-7 (0.4-0.8 %) -- about the width of the gate window
// (spatial_k ~ 0.5 % of the depth).  A counting sort on that id (LDS atomics, O(m), no comparison sort) makes every
// depth window a contiguous range of a few buckets; the order inside a bucket is arbitrary, which cannot change
// the result (per-camera maxima, summed in camera order).
#ifndef L3D_BUCKET_SHIFT
#define L3D_BUCKET_SHIFT 15
#define L3D_BUCKETS 2048
#endif
static_assert(L3D_BUCKETS == kVWBuckets, "the bucket starts of built images (VerifyArgs::bstart_g) are sized by kVWBuckets");
constexpr int kBucketShift = L3D_BUCKET_SHIFT;               // 256 buckets per octave of depth
constexpr int kBuckets = L3D_BUCKETS;                 // 8 octaves; anything beyond is clamped into the last bucket
#ifndef L3D_VQ
#define L3D_VQ 128
#endif
constexpr int kVQ = L3D_VQ;                       // per-wave ring of gate candidates
__device__ __forceinline__ int bucket_of(float d, int base)
{
    if (!(d > 0.0f)) return 0;                                        // windows may reach below zero
    const int raw = (int)(__float_as_uint(d) >> kBucketShift) - base;
    return raw < 0 ? 0 : (raw > kBuckets - 1 ? kBuckets - 1 : raw);
}

struct VWLds {
    float* sd1; float* sd2; unsigned* sci; unsigned* stgt;     // [cap] bucket-grouped: depths, (cam<<24)|index, target id
};

// Evaluate up to 64 queued (hypothesis lane, witness position) pairs with all lanes busy: exact 3-D gate first
// (cudawrapper.cu:388-401), then projection validity (:690-693) and the 2-D/angle confidence (:404-426); the
// per-camera maximum goes to LDS with an integer atomic max (confidences are positive floats).
__device__ __forceinline__ void vw_drain(const VerifyArgs& a, const float* sP, const int* sOff, const unsigned* q, int head, int n, int lane,
                                         f3 C, f3 ray1, f3 ray2, float d1y, float d2y, bool gate,
                                         float* smax_wave, int* dirty, float two_sig_d, float two_sig_a)
{
    // a ring entry carries everything the witness contributes (camera, target id, both depths): the only global access of
    // the evaluation is the gather of the target segment, issued straight after the ring read.  The hypothesis is rebuilt here
    // from its two depths (3-D endpoints, direction, gate thresholds; the reference's operations, cudawrapper.cu:644-645,390-394):
    // only the few pairs that pass the 1-D tests need it, the window walk itself works on depths alone.
    unsigned key = 0, tgt = 0;
    float wd1 = 0.0f, wd2 = 0.0f;
    if (lane < n) {
        const uint4 e = reinterpret_cast<const uint4*>(q)[(head + lane) & (kVQ - 1)];
        key = e.x; tgt = e.y; wd1 = __uint_as_float(e.z); wd2 = __uint_as_float(e.w);
    }
    const int origin = key & 63, cam = (int)(key >> 8);
    float4 tq = make_float4(0.f, 0.f, 1.f, 1.f);
    if (lane < n) tq = a.tgt_segs[sOff[cam] + tgt];
    const float hd1 = __shfl(d1y, origin), hd2 = __shfl(d2y, origin);
    if (lane >= n) return;
    const f3 hX1 = C + hd1 * ray1;                                       // D_unproject_point_src, cudawrapper.cu:644-645
    const f3 hX2 = C + hd2 * ray2;
    const f3 hv = normalize(hX1 - hX2);
    float hT1 = 0.0f, hT2 = 0.0f;
    if (gate) {
        hT1 = sq_threshold(a.spatial_k * length(C - hX1));              // cudawrapper.cu:390-394
        hT2 = sq_threshold(a.spatial_k * length(C - hX2));
    }
    const float conf = witness_conf(C, ray1, ray2, hX1, hX2, hv, hT1, hT2, gate, wd1, wd2, sP + cam * 12, tq, two_sig_d, two_sig_a);
    if (a.stamps) {                                                      // diagnostic: how far the drained pairs get
        const f3 Q1 = C + wd1 * ray1, Q2 = C + wd2 * ray2;
        const f3 e1 = hX1 - Q1, e2 = hX2 - Q2;
        const bool g = !gate || !(dot(e1, e1) > hT1 || dot(e2, e2) > hT2);
        const unsigned long long m0 = __ballot(true), m1 = __ballot(g), m2 = __ballot(conf > 0.0f), m3 = __ballot(conf > 0.5f);
        if (lane == 0) { atomicAdd(&a.stamps[6], (unsigned long long)__popcll(m0)); atomicAdd(&a.stamps[7], (unsigned long long)__popcll(m1));
                         atomicAdd(&a.stamps[8], (unsigned long long)__popcll(m2)); atomicAdd(&a.stamps[9], (unsigned long long)__popcll(m3)); }
    }
    if (conf > 0.5f) {                                                   // :699-704 (max over the camera's witnesses)
        atomicMax(reinterpret_cast<int*>(&smax_wave[cam * 64 + origin]), __float_as_int(conf));
        *dirty = 1;                                                      // (this wave's maxima are no longer all zero)
    }
}

// The rounds of a segment's hypotheses [h_begin, h_end) of the bucketed image L (positions in bucket order): NT hypotheses per round, the window walk,
// the evaluation of the survivors, the per-hypothesis sums.  A whole segment (k_verify_window) or one unit of a long one (k_vw_walk) -- the confidence of
// a hypothesis does not depend on who else is processed with it.  kept_l / best_l / besti_l: this thread's share of the segment's epilogue.
#define VW_STAMP(k) do { if (a.stamps) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); t_acc[k] += t_ - t_prev; t_prev = t_; } } while (0)
template <int NT>
__device__ __forceinline__ void vw_rounds(const VerifyArgs& a, const VWLds& L, const int* s_bstart, int base, float dabs_max, int start, int h_begin, int h_end,
                                          unsigned* q, const float* sP, const int* sOff, float* smax_wave, int* dirty, f3 C, f3 ray1, f3 ray2, float c_inf,
                                          float two_sig_d, float two_sig_a, bool gate, int& kept_l, float& best_l, int& besti_l,
                                          unsigned long long* t_acc, unsigned long long& t_prev)
{
    const int tid = threadIdx.x, lane = tid & 63;

    // ---- hypotheses in bucket order: the lanes of a wave have neighbouring depths, hence nearly the same window
    for (int h0 = h_begin; h0 < h_end; h0 += NT) {
        const int h = h0 + tid;
        const bool hv = h < h_end;
        float d1y = 0.0f, d2y = 0.0f, w1 = 0.0f, w2 = 0.0f;
        unsigned cam_h = 0xffu, idx_h = 0;
        if (hv) {
            d1y = L.sd1[h]; d2y = L.sd2[h];
            cam_h = L.sci[h] >> 24; idx_h = L.sci[h] & 0xffffffu;
            if (gate) {
                // the gate's uncertainty spatial_k * |C - X| (cudawrapper.cu:390-394) is at most spatial_k * |d| * (1 + 1e-5): |ray| = 1 +- 3u
                // and X = C + d*ray carries a few ulps of |C| + |d| -- far inside the margin window_margin adds.  The walk needs nothing else
                // of the hypothesis; its 3-D endpoints and thresholds are formed in vw_drain for the pairs that get that far.
                w1 = window_margin(a.spatial_k * __builtin_fabsf(d1y) * 1.00001f, __builtin_fabsf(d1y), dabs_max, c_inf);
                w2 = window_margin(a.spatial_k * __builtin_fabsf(d2y) * 1.00001f, __builtin_fabsf(d2y), dabs_max, c_inf);
            } else {
                w1 = w2 = __builtin_inff();
            }
        }
        int head = 0, count = 0;                                       // wave-uniform ring state
        const float lo1 = d1y - w1, hi1 = d1y + w1;
        int j = 0, jend = 0;
        if (hv) { j = s_bstart[bucket_of(lo1, base)]; jend = s_bstart[bucket_of(hi1, base) + 1]; }
        if (a.stamps && a.debug == 9) {                                  // diagnostic (L3D_VW_DEBUG=9 on top of the stamps: this loop distorts the phase split): entries walked, of them inside the d1 window / inside both windows
            int n_in = 0, n_in1 = 0, n_in2 = 0, n_oth = 0;
            for (int e = j; e < jend; ++e) {
                ++n_in;
                const bool i1 = L.sd1[e] >= lo1 && L.sd1[e] <= hi1;
                const bool i2 = __builtin_fabsf(L.sd2[e] - d2y) <= w2;
                n_in1 += i1; n_in2 += i1 && i2; n_oth += i1 && i2 && (L.sci[e] >> 24) != cam_h;
            }
            for (int o = 32; o > 0; o >>= 1) { n_in += __shfl_down(n_in, o); n_in1 += __shfl_down(n_in1, o); n_in2 += __shfl_down(n_in2, o); n_oth += __shfl_down(n_oth, o); }
            const unsigned long long nh = __popcll(__ballot(hv));
            if (lane == 0) { atomicAdd(&a.stamps[10], (unsigned long long)n_in); atomicAdd(&a.stamps[11], (unsigned long long)n_in1); atomicAdd(&a.stamps[12], (unsigned long long)n_in2);
                             atomicAdd(&a.stamps[13], (unsigned long long)n_oth); atomicAdd(&a.stamps[14], nh); }
        }
        // the window is walked in groups of kG entries: the next group's loads (3 per entry, LDS or L2) are in flight while
        // the current one is tested, so a wave pays one memory round trip per group instead of one per entry
#ifndef L3D_KG
#define L3D_KG 4
#endif
        constexpr int kG = L3D_KG;
        float c1[kG], c2[kG];
        unsigned cc[kG], ct[kG];
#pragma unroll
        for (int g = 0; g < kG; ++g) { c1[g] = L.sd1[j + g]; c2[g] = L.sd2[j + g]; cc[g] = L.sci[j + g]; ct[g] = L.stgt[j + g]; }
        VW_STAMP(1);
        for (;;) {
            if (!__any(j < jend)) break;
            float n1[kG], n2[kG];
            unsigned nc[kG], nt[kG];
            const int jn = j < jend ? j + kG : j;                          // (lanes that are done keep re-reading in range)
#pragma unroll
            for (int g = 0; g < kG; ++g) { n1[g] = L.sd1[jn + g]; n2[g] = L.sd2[jn + g]; nc[g] = L.sci[jn + g]; nt[g] = L.stgt[jn + g]; }
#pragma unroll
            for (int g = 0; g < kG; ++g) {
                // :674 (other cameras only) and the 1-D tests every gate-passing witness satisfies; the exact 3-D gate
                // and the confidence run on the compacted survivors (vw_drain)
                const bool push = j + g < jend && (cc[g] >> 24) != cam_h && c1[g] >= lo1 && c1[g] <= hi1 && __builtin_fabsf(c2[g] - d2y) <= w2;
                const unsigned long long pm = __ballot(push);
                if (pm) {
                    if (push) {
                        const int pos = (head + count + __popcll(pm & ((1ull << lane) - 1ull))) & (kVQ - 1);
                        reinterpret_cast<uint4*>(q)[pos] = make_uint4((unsigned)lane | ((cc[g] >> 24) << 8), ct[g], __float_as_uint(c1[g]), __float_as_uint(c2[g]));
                    }
                    count += __popcll(pm);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    if (count >= 64) {
                        VW_STAMP(2);
                        vw_drain(a, sP, sOff, q, head, 64, lane, C, ray1, ray2, d1y, d2y, gate, smax_wave, dirty, two_sig_d, two_sig_a);
                        head = (head + 64) & (kVQ - 1);
                        count -= 64;
                        VW_STAMP(3);
                    }
                }
            }
            j = jn;
#pragma unroll
            for (int g = 0; g < kG; ++g) { c1[g] = n1[g]; c2[g] = n2[g]; cc[g] = nc[g]; ct[g] = nt[g]; }
        }
        VW_STAMP(2);
        if (count > 0) vw_drain(a, sP, sOff, q, head, count, lane, C, ray1, ray2, d1y, d2y, gate, smax_wave, dirty, two_sig_d, two_sig_a);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        VW_STAMP(3);
        float conf_sum = 0.0f;
        if (__builtin_amdgcn_readfirstlane(*dirty)) {
            for (int c = 0; c < a.N; ++c) { conf_sum += smax_wave[c * 64 + lane]; smax_wave[c * 64 + lane] = 0.0f; }   // ascending camera order; +0.0f is exact
            if (lane == 0) *dirty = 0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        if (hv) {
            a.cand_conf[start + idx_h] = conf_sum;
            kept_l += conf_sum > 1.0f;
            if (conf_sum > best_l || (conf_sum == best_l && (int)idx_h < besti_l)) { best_l = conf_sum; besti_l = (int)idx_h; }
        }
        VW_STAMP(4);
    }
}

// units of the split launch: segment seg_order[i] gets ceil(m / split_unit) units when it outgrows the LDS image (and the candidates did not overflow),
// none otherwise; unit_start = exclusive prefix in that order (heaviest segments first).  One workgroup.
template <int NT>
__device__ __forceinline__ void vw_unit_table(const VerifyArgs& a, const VWSplitArgs& sp, int* s_w)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nseg = a.seg_end - a.seg_begin;
    const bool overflow = a.cand_cap && a.row_start[a.nrow_total] > a.cand_cap;
    const int per = (nseg + NT - 1) / NT, i0 = min(nseg, tid * per), i1 = min(nseg, i0 + per);
    auto units_of = [&](int i) {
        const int y = a.seg_order ? a.seg_order[i] : a.seg_begin + i;
        const int m = a.row_start[(y + 1) * a.N] - a.row_start[y * a.N];
        return (!overflow && m > a.mmax) ? (m + sp.split_unit - 1) / sp.split_unit : 0;
    };
    int tot = 0;
    for (int i = i0; i < i1; ++i) tot += units_of(i);
    int incl = tot;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int run = incl - tot;
    for (int w = 0; w < wave; ++w) run += s_w[w];
    for (int i = i0; i < i1; ++i) { sp.unit_start[i] = run; run += units_of(i); }
    if (tid == NT - 1) sp.unit_start[nseg] = run;
}

// Second launch of a split verification: workgroup b = unit (segment, part) by binary search in unit_start; the segment's image was built by its scratch
// block of k_verify_window (bucketed arrays in the scratch, bucket starts and header in global memory).  The unit verifies its split_unit hypotheses
// (vw_rounds: the very code of the one-launch kernel) and adds its share of the segment's epilogue with atomics -- kept count, and the first strict
// maximum in candidate order as ONE 64-bit maximum --; the unit that finishes last writes the best hypothesis' depths (cudawrapper.cu:1037-1062).
template <int NT, bool kGB>
__global__ __launch_bounds__(NT) void k_vw_walk(VerifyArgs a, VWSplitArgs sp)
{
    constexpr int NW = NT / 64;
    extern __shared__ __align__(16) unsigned char s_raw[];
    __shared__ int s_bstart[kGB ? 1 : kBuckets + 1];          // (kGB -- more than 16 neighbours --: the rounds read the starts where the build left them)
    __shared__ int s_dirty[NW];
    __shared__ int s_rk[NW], s_ri[NW];
    __shared__ float s_rb[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nseg = a.seg_end - a.seg_begin;
    const int b = (int)blockIdx.x;
    if (b >= sp.unit_start[nseg]) return;
    int lo = 0, hi = nseg;                                    // largest i with unit_start[i] <= b (it owns at least one unit)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sp.unit_start[mid] <= b) lo = mid; else hi = mid; }
    const int part = b - sp.unit_start[lo], units = sp.unit_start[lo + 1] - sp.unit_start[lo];
    const int y = a.seg_order ? a.seg_order[lo] : a.seg_begin + lo;
    const int ys = y - a.seg_begin;
    const int start = a.row_start[y * a.N];
    const int m = a.row_start[(y + 1) * a.N] - start;
    const int h_begin = part * sp.split_unit, h_end = min(m, h_begin + sp.split_unit);
    VWLds L;
    {
        float* g = a.scratch;
        const size_t stride = (size_t)a.scratch_stride;
        L.sd1 = g + start; L.sd2 = g + stride + start;
        L.sci = reinterpret_cast<unsigned*>(g + 2 * stride) + start; L.stgt = reinterpret_cast<unsigned*>(g + 3 * stride) + start;
    }
    float* smax = reinterpret_cast<float*>(s_raw);                           // [NT][N] per-(hypothesis lane, camera) maxima
    unsigned* qall = reinterpret_cast<unsigned*>(smax + NT * a.N);
    unsigned* q = qall + wave * kVQ * 4;
    float* sP = reinterpret_cast<float*>(qall + NW * kVQ * 4);
    int* sOff = reinterpret_cast<int*>(sP + a.N * 12);
    for (int i = tid; i < a.N * 12; i += NT) sP[i] = a.P[i];
    for (int i = tid; i < a.N; i += NT) sOff[i] = a.offsets[i].x;
    const int* bs = sp.bstart_g + (size_t)ys * (kBuckets + 1);
    if constexpr (!kGB) for (int i = tid; i <= kBuckets; i += NT) s_bstart[i] = bs[i];
    float* smax_wave = smax + wave * 64 * a.N;
    for (int c = 0; c < a.N; ++c) smax_wave[c * 64 + lane] = 0.0f;
    if (lane == 0) s_dirty[wave] = 0;
    int* dirty = &s_dirty[wave];
    const int4 hdr = sp.seg_hdr[ys];
    const int base = hdr.x;
    const float dabs_max = __int_as_float(hdr.y);
    __syncthreads();
    const f3 C = mk3(a.C_src[0], a.C_src[1], a.C_src[2]);
    const float4 sseg = a.src_segs[y];
    const f3 ray1 = normalize(mat3_apply(a.RtKinv_src, mk3(sseg.x, sseg.y, 1.0f)));
    const f3 ray2 = normalize(mat3_apply(a.RtKinv_src, mk3(sseg.z, sseg.w, 1.0f)));
    const float c_inf = __builtin_fmaxf(__builtin_fabsf(C.x), __builtin_fmaxf(__builtin_fabsf(C.y), __builtin_fabsf(C.z)));
    const float two_sig_d = 2.0f * (a.sigma_p * a.sigma_p);
    const float two_sig_a = 2.0f * (a.sigma_a * a.sigma_a);
    const bool gate = a.spatial_k > 0.0f;
    int kept_l = 0, besti_l = 0x7fffffff;
    float best_l = 0.0f;
    unsigned long long t_prev = 0ull, t_acc[5] = { 0, 0, 0, 0, 0 };
    vw_rounds<NT>(a, L, kGB ? bs : s_bstart, base, dabs_max, start, h_begin, h_end, q, sP, sOff, smax_wave, dirty, C, ray1, ray2, c_inf, two_sig_d, two_sig_a, gate, kept_l, best_l, besti_l, t_acc, t_prev);
    for (int o = 32; o > 0; o >>= 1) {
        kept_l += __shfl_down(kept_l, o);
        const float ob = __shfl_down(best_l, o);
        const int oi = __shfl_down(besti_l, o);
        if (ob > best_l || (ob == best_l && oi < besti_l)) { best_l = ob; besti_l = oi; }
    }
    if (lane == 0) { s_rk[wave] = kept_l; s_rb[wave] = best_l; s_ri[wave] = besti_l; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NW; ++w) {
            kept_l += s_rk[w];
            if (s_rb[w] > best_l || (s_rb[w] == best_l && s_ri[w] < besti_l)) { best_l = s_rb[w]; besti_l = s_ri[w]; }
        }
        if (kept_l) atomicAdd(&a.kept_cnt[y], kept_l);
        atomicMax(&sp.best64[ys], ((unsigned long long)__float_as_uint(best_l) << 32) | (unsigned long long)(unsigned)(0x7fffffff - besti_l));     // (confidences are >= +0)
        __threadfence();
        if (atomicAdd(&sp.done[ys], 1) == units - 1) {         // the last unit of the segment: every unit's maximum is in
            __threadfence();
            const unsigned long long bb = atomicMax(&sp.best64[ys], 0ull);
            const float bconf = __uint_as_float((unsigned)(bb >> 32));
            float2 bd = make_float2(-1.0f, -1.0f);            // marker: not part of the median list
            if (bconf > 0.5f) {                                // conf_t/2.0f
                const float4 d = a.cand_depths[start + (0x7fffffff - (int)(unsigned)(bb & 0xffffffffull))];
                bd = make_float2(d.x, d.y);
            }
            a.best_depths[y] = bd;
        }
    }
}

// NT threads per workgroup: 256 when the grid fills the chip, 512 when only a few segments are verified per launch
// (one rank's slice of a view in the sharded chain): the segment's hypotheses then run 8 waves wide instead of 4.
#ifndef L3D_VW_WAVES
#define L3D_VW_WAVES 0
#endif
// kGB: the bucket starts live in GLOBAL memory during the rounds (VerifyArgs::bstart_g) and the build's two bucket tables (starts, cursors: 16.4 KB)
// alias the per-lane maxima + rings, which the build does not use: 16.4 KB of static LDS less -- at 24 neighbours (24.5 KB of maxima) a workgroup
// drops from 51.7 to 35.3 KB: four per CU instead of three.  The rounds read two bucket starts per hypothesis, once per round: L2 latency that a
// round of tens of microseconds does not notice.  Launches of up to 16 neighbours keep the tables in LDS (four workgroups per CU either way).
template <int NT, bool kSplit, bool kGB>
__device__ __forceinline__ void vw_segment_block(const VerifyArgs& a, const VWSplitArgs& sp)
{
    constexpr int NW = NT / 64;
    extern __shared__ __align__(16) unsigned char s_raw[];
    __shared__ int s_dmax, s_base;
    __shared__ int s_tables[kGB ? 4 : 2 * kBuckets + 4];     // bucket starts [kBuckets + 1] | cursors [kBuckets]  (kGB: in the dynamic region, below)
    int* s_bstart = s_tables;
    int* s_cursor = s_tables + kBuckets + 2;
    __shared__ int s_wtot[NW];
    __shared__ int s_dirty[NW];              // per wave: a confidence was recorded in the current group of hypotheses
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // big == false: segments whose candidates fit the LDS image (m <= mmax); big == true: the rest, same algorithm with the
    // bucketed arrays in a global scratch (L2) instead of LDS -- still O(m*window), never the all-pairs loop.
    // a.big == 2 runs both kinds in one launch: blocks [0, nseg) are the LDS blocks, [nseg, 2 nseg) the scratch blocks.
    const int nseg = a.seg_end - a.seg_begin;
    if constexpr (kSplit) { if ((int)blockIdx.x == 2 * nseg) { vw_unit_table<NT>(a, sp, s_wtot); return; } }     // (one more workgroup: the units of the second launch)
    const bool big = a.big == 2 ? (int)blockIdx.x >= nseg : a.big != 0;
    const int widx = (int)blockIdx.x >= nseg ? (int)blockIdx.x - nseg : (int)blockIdx.x;
#ifdef L3D_NO_SEG_ORDER
    const int y = a.seg_begin + widx;
#else
    const int y = a.seg_order ? a.seg_order[widx] : a.seg_begin + widx;
#endif
    const int start = a.row_start[y * a.N];
    const int m = a.row_start[(y + 1) * a.N] - start;
    const bool overflow = a.cand_cap && a.row_start[a.nrow_total] > a.cand_cap;   // candidate overflow: the chain is re-run
    const bool mine = big ? m > a.mmax : !((a.skip_above || a.big == 2) && m > a.mmax);
    if (m == 0 || overflow || !mine) {
        // nothing to verify: the LDS block of the segment still owns its epilogue (cudawrapper.cu:1037-1062, :1096)
        if (!big && (m == 0 || overflow) && a.kept_cnt && tid == 0) { a.kept_cnt[y] = 0; a.best_depths[y] = make_float2(-1.0f, -1.0f); }
        return;
    }
    if (a.debug == 4) return;
    if (a.n_exist_cams > 0) {
        // the reverse matches of this segment were scattered into their rows in arbitrary order: a wave per (source camera) run
        // ranks them by target id (k_exist_sort_runs folded in -- one launch less on the per-view path)
        uint2* meta_w = const_cast<uint2*>(a.cand_meta);
        float4* depths_w = const_cast<float4*>(a.cand_depths);
        for (int t = wave; t < a.n_exist_cams; t += NW) {
            const int cam = a.exist_cams[t];
            const int b = a.row_start[y * a.N + cam], n = a.row_start[y * a.N + cam + 1] - b;
            // (the segment's slices of the scratch and of the confidence array are not in use yet: the staging area of a long run)
            sort_exist_run(lane, b, n, cam, meta_w, depths_w, a.scratch, a.scratch_stride, a.scratch ? reinterpret_cast<unsigned*>(a.cand_conf) : nullptr);
        }
        __threadfence_block();
        __syncthreads();
    }
    unsigned long long t_prev = a.stamps ? __builtin_amdgcn_s_memtime() : 0ull, t_acc[5] = { 0, 0, 0, 0, 0 };

    const int cap = big ? 0 : a.mmax + kVWSlack;                       // slack: the scan loads a group of entries ahead
    VWLds L;
    L.sd1 = reinterpret_cast<float*>(s_raw);
    L.sd2 = L.sd1 + cap;
    L.sci = reinterpret_cast<unsigned*>(L.sd2 + cap);
    L.stgt = L.sci + cap;
    float* smax = reinterpret_cast<float*>(L.stgt + cap);                // [NT][N] per-(hypothesis lane, camera) maxima
    if (big) {                                                           // the segment's own slice of the scratch (+2 per array)
        float* g = a.scratch;
        const size_t stride = (size_t)a.scratch_stride;
        L.sd1 = g + start; L.sd2 = g + stride + start;
        L.sci = reinterpret_cast<unsigned*>(g + 2 * stride) + start; L.stgt = reinterpret_cast<unsigned*>(g + 3 * stride) + start;
    }
    unsigned* qall = reinterpret_cast<unsigned*>(smax + NT * a.N);           // 16-byte entries (the image sizes keep it aligned)
    unsigned* q = qall + wave * kVQ * 4;
    float* sP = reinterpret_cast<float*>(qall + NW * kVQ * 4);               // projection matrices and target offsets of the N cameras
    int* sOff = reinterpret_cast<int*>(sP + a.N * 12);
    for (int i = tid; i < a.N * 12; i += NT) sP[i] = a.P[i];
    for (int i = tid; i < a.N; i += NT) sOff[i] = a.offsets[i].x;
    float* smax_wave = smax + wave * 64 * a.N;
    if constexpr (kGB) {                                                 // (the launcher guarantees NT * N * 4 + rings >= 2 * kBuckets + 4 ints)
        s_bstart = reinterpret_cast<int*>(smax);
        s_cursor = s_bstart + kBuckets + 2;
    }
    // the per-(camera, lane) maxima are zero whenever a group of hypotheses starts: zeroed here once, and again only after a group that recorded
    // something (most groups record nothing: a fifth of the hypotheses has a witness at all, and those cluster on few segments)
    if constexpr (!kGB) for (int c = 0; c < a.N; ++c) smax_wave[c * 64 + lane] = 0.0f;       // [camera][lane]: conflict-free rows   (kGB: after the build)
    if (lane == 0) s_dirty[wave] = 0;
    int* dirty = &s_dirty[wave];

    // ---- one coalesced pass over the segment's candidates (kept in registers), counting sort on the depth bucket
    constexpr int kMaxPerThread = 2048 / NT;                             // m <= 2048 in registers, more is re-read
    if (tid == 0) { s_dmax = 0; s_base = 0x7fffffff; }
    for (int b = tid; b < kBuckets; b += NT) s_cursor[b] = 0;
    float4 rd[kMaxPerThread];
    uint2 rm[kMaxPerThread];
    float dm = 0.0f;
    int rmin = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < kMaxPerThread; ++k) {
        const int i = tid + k * NT;
        if (i < m) { rd[k] = a.cand_depths[start + i]; rm[k] = a.cand_meta[start + i]; }
    }
#pragma unroll
    for (int k = 0; k < kMaxPerThread; ++k) {
        const int i = tid + k * NT;
        if (i < m) {
            dm = __builtin_fmaxf(dm, __builtin_fmaxf(__builtin_fabsf(rd[k].x), __builtin_fabsf(rd[k].y)));
            rmin = min(rmin, (int)(__float_as_uint(rd[k].x) >> kBucketShift));
        }
    }
    for (int i = tid + kMaxPerThread * NT; i < m; i += NT) {
        const float4 d = a.cand_depths[start + i];
        dm = __builtin_fmaxf(dm, __builtin_fmaxf(__builtin_fabsf(d.x), __builtin_fabsf(d.y)));
        rmin = min(rmin, (int)(__float_as_uint(d.x) >> kBucketShift));
    }
    for (int o = 32; o > 0; o >>= 1) { dm = __builtin_fmaxf(dm, __shfl_down(dm, o)); rmin = min(rmin, __shfl_down(rmin, o)); }
    __syncthreads();
    if (lane == 0) { atomicMax(&s_dmax, __float_as_int(dm)); atomicMin(&s_base, rmin); }   // non-negative floats order like ints
    __syncthreads();
    const int base = s_base;
#pragma unroll
    for (int k = 0; k < kMaxPerThread; ++k) if (tid + k * NT < m) atomicAdd(&s_cursor[bucket_of(rd[k].x, base)], 1);
    for (int i = tid + kMaxPerThread * NT; i < m; i += NT) atomicAdd(&s_cursor[bucket_of(a.cand_depths[start + i].x, base)], 1);
    __syncthreads();
    {   // exclusive scan of the bucket counts: kBuckets/NT per thread + wave scan + wave totals
        constexpr int BPT = kBuckets / NT;
        int c[BPT], tot = 0;
#pragma unroll
        for (int k = 0; k < BPT; ++k) { c[k] = s_cursor[BPT * tid + k]; tot += c[k]; }
        int incl = tot;
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        int run = incl - tot;
        for (int w = 0; w < wave; ++w) run += s_wtot[w];
#pragma unroll
        for (int k = 0; k < BPT; ++k) { s_bstart[BPT * tid + k] = run; s_cursor[BPT * tid + k] = run; run += c[k]; }
        if (tid == NT - 1) s_bstart[kBuckets] = run;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kMaxPerThread; ++k) {
        const int i = tid + k * NT;
        if (i < m) {
            const int pos = atomicAdd(&s_cursor[bucket_of(rd[k].x, base)], 1);
            L.sd1[pos] = rd[k].x; L.sd2[pos] = rd[k].y;
            L.sci[pos] = (rm[k].y << 24) | (unsigned)i;
            L.stgt[pos] = rm[k].x;
        }
    }
    for (int i = tid + kMaxPerThread * NT; i < m; i += NT) {
        const float4 d = a.cand_depths[start + i];
        const uint2 mt = a.cand_meta[start + i];
        const int pos = atomicAdd(&s_cursor[bucket_of(d.x, base)], 1);
        L.sd1[pos] = d.x; L.sd2[pos] = d.y;
        L.sci[pos] = (mt.y << 24) | (unsigned)i;
        L.stgt[pos] = mt.x;
    }
    if (tid < kVWSlack && !big) { L.sd1[m + tid] = 0.0f; L.sd2[m + tid] = 0.0f; L.sci[m + tid] = 0u; L.stgt[m + tid] = 0u; }   // (global slices: the prefetch reads a neighbour's entries, never uses them)
    __syncthreads();
    const int* bst = s_bstart;                                           // what the rounds read
    if constexpr (kGB) {
        int* gb = a.bstart_g + (size_t)(y - a.seg_begin) * (kBuckets + 1);
        for (int b = tid; b <= kBuckets; b += NT) gb[b] = s_bstart[b];
        __threadfence_block();
        __syncthreads();                                                 // the tables are copied: their LDS is the maxima's again
        for (int c = 0; c < a.N; ++c) smax_wave[c * 64 + lane] = 0.0f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        bst = gb;
    }
    if (a.debug == 1) return;
    VW_STAMP(0);
    if constexpr (kSplit) {
        if (big) {
            // split: this block only built the image; bucket starts and header go to global memory, the segment's epilogue state is reset, and the units
            // of k_vw_walk (the next launch on this stream) verify the hypotheses
            const int ys = y - a.seg_begin;
            int* bs = sp.bstart_g + (size_t)ys * (kBuckets + 1);
            for (int b = tid; b <= kBuckets; b += NT) bs[b] = bst[b];
            if (tid == 0) { sp.seg_hdr[ys] = make_int4(base, s_dmax, 0, 0); a.kept_cnt[y] = 0; sp.best64[ys] = 0ull; sp.done[ys] = 0; }
            return;
        }
    }

    const f3 C = mk3(a.C_src[0], a.C_src[1], a.C_src[2]);
    const float4 sseg = a.src_segs[y];
    const f3 ray1 = normalize(mat3_apply(a.RtKinv_src, mk3(sseg.x, sseg.y, 1.0f)));
    const f3 ray2 = normalize(mat3_apply(a.RtKinv_src, mk3(sseg.z, sseg.w, 1.0f)));
    const float c_inf = __builtin_fmaxf(__builtin_fabsf(C.x), __builtin_fmaxf(__builtin_fabsf(C.y), __builtin_fabsf(C.z)));
    const float dabs_max = __int_as_float(s_dmax);
    const float two_sig_d = 2.0f * (a.sigma_p * a.sigma_p);
    const float two_sig_a = 2.0f * (a.sigma_a * a.sigma_a);
    const bool gate = a.spatial_k > 0.0f;

    // fused per-segment epilogue (k_seg_post): kept count and the first strict maximum in candidate order
    int kept_l = 0, besti_l = 0x7fffffff;
    float best_l = 0.0f;
    vw_rounds<NT>(a, L, bst, base, dabs_max, start, 0, m, q, sP, sOff, smax_wave, dirty, C, ray1, ray2, c_inf, two_sig_d, two_sig_a, gate, kept_l, best_l, besti_l, t_acc, t_prev);
    if (a.kept_cnt) {
        __shared__ int s_rk[NW], s_ri[NW];
        __shared__ float s_rb[NW];
        for (int o = 32; o > 0; o >>= 1) {
            kept_l += __shfl_down(kept_l, o);
            const float ob = __shfl_down(best_l, o);
            const int oi = __shfl_down(besti_l, o);
            if (ob > best_l || (ob == best_l && oi < besti_l)) { best_l = ob; besti_l = oi; }
        }
        if (lane == 0) { s_rk[wave] = kept_l; s_rb[wave] = best_l; s_ri[wave] = besti_l; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NW; ++w) {
                kept_l += s_rk[w];
                if (s_rb[w] > best_l || (s_rb[w] == best_l && s_ri[w] < besti_l)) { best_l = s_rb[w]; besti_l = s_ri[w]; }
            }
            a.kept_cnt[y] = kept_l;
            float2 bd = make_float2(-1.0f, -1.0f);         // marker: not part of the median list
            if (best_l > 0.5f) {                           // conf_t/2.0f
                const float4 d = a.cand_depths[start + besti_l];
                bd = make_float2(d.x, d.y);
            }
            a.best_depths[y] = bd;
        }
    }
    if (a.stamps && lane == 0) {
        for (int k = 0; k < 5; ++k) atomicAdd(&a.stamps[k], t_acc[k]);
        atomicAdd(&a.stamps[5], 1ull);
        atomicMax(&a.stamps[15], t_acc[0] + t_acc[1] + t_acc[2] + t_acc[3] + t_acc[4]);       // the longest wave: a launch lasts as long as its longest segment
    }
#undef VW_STAMP
}

template <int NT>
__global__ __launch_bounds__(NT)
#if L3D_VW_WAVES
__attribute__((amdgpu_waves_per_eu(L3D_VW_WAVES, L3D_VW_WAVES)))
#endif
void k_verify_window(VerifyArgs a)
{
    vw_segment_block<NT, false, false>(a, VWSplitArgs());
}
// more than 16 neighbours: bucket starts in global memory during the rounds (four workgroups per CU instead of three at 24 neighbours)
template <int NT>
__global__ __launch_bounds__(NT) void k_verify_window_gb(VerifyArgs a)
{
    vw_segment_block<NT, false, true>(a, VWSplitArgs());
}
// first launch of a split verification: LDS blocks as ever, scratch blocks build only, one more workgroup writes the unit table
template <int NT>
__global__ __launch_bounds__(NT) void k_verify_window_build(VerifyArgs a, VWSplitArgs sp)
{
    vw_segment_block<NT, true, false>(a, sp);
}

// max candidates per segment (LDS sizing of k_verify_window)
__global__ void k_seg_mmax(const int* __restrict__ row_start, int N, int seg_begin, int seg_end, int* __restrict__ out)
{
    const int y = seg_begin + blockIdx.x * blockDim.x + threadIdx.x;
    int m = 0;
    if (y < seg_end) m = row_start[(y + 1) * N] - row_start[y * N];
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_down(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, m);
}

size_t verify_window_lds_bytes_nt(int mmax, int N, int nt) { return (size_t)(mmax + kVWSlack) * 16 + (size_t)nt * N * 4 + (size_t)(nt / 64) * kVQ * 16 + (size_t)N * 52 + 16; }
size_t verify_window_lds_bytes(int mmax, int N) { return verify_window_lds_bytes_nt(mmax, N, 256); }
size_t verify_window_lds_bytes_big(int N, int nt) { return (size_t)nt * N * 4 + (size_t)(nt / 64) * kVQ * 16 + (size_t)N * 52 + 64; }
// Largest dynamic LDS a k_verify_window launch may ask for on this device/runtime (queried once): up to 160 KB per
// workgroup on gfx950 once the kernel has opted in; runtimes that refuse the opt-in stay at the 48/64 KB default.
static size_t g_lds_budget_override = 0;
void verify_window_set_lds_budget(size_t bytes) { g_lds_budget_override = bytes; }
size_t verify_window_max_lds(int vw_lds_opt)
{
    static size_t limit = 0;
    if (g_lds_budget_override) return g_lds_budget_override;
    if (vw_lds_opt > 0) return (size_t)vw_lds_opt;     // diagnostic (L3D_VW_LDS): dynamic LDS budget in bytes
    if (limit) return limit;
    // Measured on MI355X (config 2): the kernel is latency bound and gains more from resident workgroups than from a
    // large LDS image -- 24 KB of dynamic LDS (+8 KB static -> 5 workgroups per CU, the VGPR limit) beats 48 KB by 8 %,
    // and segments that outgrow the image lose nothing on the global-scratch (L2) variant.  So the budget is small.
    size_t want = 24000;            // (+ 16.4 KB static: just under 40 KB, four workgroups per CU now that the kernel needs 121 VGPRs)
    limit = want;
    return limit;
}
// The per-(hypothesis lane, camera) maxima alone need 1 KB per camera: beyond ~50 neighbours the kernel does not fit the
// 64 KB a workgroup may ask for and the caller takes the all-pairs kernel.
bool verify_window_supported(int N) { return verify_window_lds_bytes(64, N) <= 60 * 1024; }
// hipFuncAttributeMaxDynamicSharedMemorySize applies to the CURRENT device: the opt-in is tracked per (device, kernel
// instantiation), under a mutex (several contexts on several GPUs may launch from different threads).  A refused opt-in
// leaves the launch error for the caller's hipGetLastError check: nothing is launched with an LDS request the device
// would reject silently.
constexpr int kWideLdsMax = 112 * 1024;
static bool lds_opt_in(const void* fn, int which)
{
    static std::mutex mu;
    static unsigned char done[4][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lk(mu);
    if (done[which][dev]) return true;
    // (the 8-wave instantiation runs few workgroups per launch -- one rank's slice of a view --: its per-lane maxima at 24 neighbours need 66 KB)
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, which == 1 ? kWideLdsMax : 60 * 1024) != hipSuccess) return false;   // (error stays pending)
    done[which][dev] = 1;
    return true;
}
void launch_verify_window(const VerifyArgs& a, hipStream_t st, int wide_max, const VWSplitArgs* sp)
{
    const int nseg = a.seg_end - a.seg_begin;
    if (nseg <= 0) return;
    if (sp && sp->split_unit > 0 && a.big == 2) {
        // split (the caller's decision: few segments per launch on a dense scene -- one rank's slice of a view): the launch of whole segments lasts as
        // long as its longest one; built by the 4-wave kernel, verified in units that keep every CU busy
        const size_t lds = std::max(verify_window_lds_bytes(a.mmax, a.N), verify_window_lds_bytes_big(a.N, 256));
        if (!lds_opt_in(reinterpret_cast<const void*>(k_verify_window_build<256>), 2)) return;
        hipLaunchKernelGGL(k_verify_window_build<256>, dim3(2 * nseg + 1), dim3(256), lds, st, a, *sp);
        if (a.N > 16) hipLaunchKernelGGL((k_vw_walk<256, true>), dim3((unsigned)std::max(1, sp->units_max)), dim3(256), verify_window_lds_bytes_big(a.N, 256), st, a, *sp);
        else hipLaunchKernelGGL((k_vw_walk<256, false>), dim3((unsigned)std::max(1, sp->units_max)), dim3(256), verify_window_lds_bytes_big(a.N, 256), st, a, *sp);
        return;
    }
    const dim3 grid(a.big == 2 ? 2 * nseg : nseg);
    // few segments (up to about two workgroups per CU): 8 waves per segment, if the wider per-lane maxima still fit
    // (replayed ranks, 2000 segments per view: 250 segments 90 -> 74 us per view, 500 segments 98 -> 94, 1000 segments 135 -> 196)
    const size_t lds512 = a.big == 1 ? verify_window_lds_bytes_big(a.N, 512) : std::max(verify_window_lds_bytes_nt(a.mmax, a.N, 512), verify_window_lds_bytes_big(a.N, 512));
    if (nseg <= wide_max && lds512 <= (size_t)kWideLdsMax) {
        if (!lds_opt_in(reinterpret_cast<const void*>(k_verify_window<512>), 1)) return;
        hipLaunchKernelGGL(k_verify_window<512>, grid, dim3(512), lds512, st, a);
    } else {
        const size_t lds = a.big == 1 ? verify_window_lds_bytes_big(a.N, 256) : std::max(verify_window_lds_bytes(a.mmax, a.N), verify_window_lds_bytes_big(a.N, 256));
        if (a.bstart_g && a.N > 16 && a.big == 2) {         // (the maxima + rings of 17+ cameras hold the build's two tables: 256 * 17 * 4 + 8192 > 16.4 KB)
            if (!lds_opt_in(reinterpret_cast<const void*>(k_verify_window_gb<256>), 3)) return;
            hipLaunchKernelGGL(k_verify_window_gb<256>, grid, dim3(256), lds, st, a);
            return;
        }
        if (!lds_opt_in(reinterpret_cast<const void*>(k_verify_window<256>), 0)) return;
        hipLaunchKernelGGL(k_verify_window<256>, grid, dim3(256), lds, st, a);
    }
}
void launch_seg_mmax(const int* row_start, int N, int seg_begin, int seg_end, int* out, hipStream_t st)
{
    const int n = seg_end - seg_begin;
    if (n > 0) hipLaunchKernelGGL(k_seg_mmax, dim3((n + 255) / 256), dim3(256), 0, st, row_start, N, seg_begin, seg_end, out);
}

}  // namespace l3d

void l3d::warm_verify_window() { touch_kernel(reinterpret_cast<const void*>(&k_seg_mmax)); }
