// l3d_verify_window.hip -- stage 2 (K_verify_matches, cudawrapper.cu:614-714) as a depth-window search.
//
// Observation: hypothesis y and witness i of the same source segment are unprojected along the SAME two
// source rays (Q1 = C + d1_i*ray1, P1 = C + d1_y*ray1, cudawrapper.cu:644-645,671-672), so the reference's
// 3-D gate |P1-Q1| <= k*depth1 && |P2-Q2| <= k*depth2 (:388-401) is, up to float rounding, a 1-D interval
// test on the depths.  Bucketing a segment's candidates by d1 turns the O(m^2) all-pairs loop into
// O(m * window) with the EXACT reference gate and confidence evaluated only inside a
// conservative window (the window margin provably covers the rounding of the 3-D computation, see
// window_margin below).  Results are bit-identical to the all-pairs kernel
// (k_verify in l3d_kernels.hip, kept as the A/B reference and the fallback for huge segments).
//
//   k_cand_prep      one workgroup per source segment: per-candidate records (3-D endpoints, unit direction,
//                    target line + norm, target segment, camera) in 5 float4 arrays (80 B / candidate)
//   k_verify_window  one workgroup per source segment: LDS holds all its candidates bucketed by d1;
//                    lane <-> hypothesis; window = a few contiguous buckets; gate passers go to a per-wave
//                    ring and are evaluated 64 at a time
#include "l3d_geometry.hpp"
#include "l3d_kernels.hpp"

namespace l3d {

// rec0 = (X1, d1)  rec1 = (X2, d2)  rec2 = (v, den2)  rec3 = (l2, cam)  rec4 = q
__global__ __launch_bounds__(256) void k_cand_prep(VerifyArgs a)
{
    const int y = a.seg_begin + blockIdx.x;
    const int start = a.row_start[y * a.N];
    const int m = a.row_start[(y + 1) * a.N] - start;
    if (m == 0) return;
    const f3 C = mk3(a.C_src[0], a.C_src[1], a.C_src[2]);
    const float4 s = a.src_segs[y];
    const f3 ray1 = normalize(mat3_apply(a.RtKinv_src, mk3(s.x, s.y, 1.0f)));
    const f3 ray2 = normalize(mat3_apply(a.RtKinv_src, mk3(s.z, s.w, 1.0f)));
    for (int i = threadIdx.x; i < m; i += 256) {
        const uint2 meta = a.cand_meta[start + i];
        const float4 d = a.cand_depths[start + i];
        const f3 X1 = C + d.x * ray1;                 // D_unproject_point_src, cudawrapper.cu:338-344
        const f3 X2 = C + d.y * ray2;
        const f3 v = normalize(X1 - X2);
        const float4 tq = a.tgt_segs[a.offsets[meta.y].x + meta.x];
        const f3 l2 = cross(mk3(tq.x, tq.y, 1.0f), mk3(tq.z, tq.w, 1.0f));
        a.rec[0][start + i] = make_float4(X1.x, X1.y, X1.z, d.x);
        a.rec[1][start + i] = make_float4(X2.x, X2.y, X2.z, d.y);
        a.rec[2][start + i] = make_float4(v.x, v.y, v.z, line_norm2d(l2));
        a.rec[3][start + i] = make_float4(l2.x, l2.y, l2.z, __int_as_float((int)meta.y));
        a.rec[4][start + i] = tq;
    }
}

// Window half-width for depth d_y: any witness that passes the reference gate sqrtf(|X_y - X_i|^2) <= unc
// satisfies |d_y - d_i| <= unc*(1+8u) + 3.5u*(|d_y| + |d_i| + |C|_inf) with u = 2^-24 (two roundings per
// coordinate of X = C + d*ray, one for the difference, dot/sqrt relative 3u, |ray| = 1 +- 3u).  The margin
// below is > 5x that bound; dabs_max bounds |d_i| for the whole segment.
__device__ __forceinline__ float window_margin(float unc, float d_y, float dabs_max, float c_inf)
{
    return unc * 1.00001f + 2.0e-6f * (d_y + dabs_max + c_inf);
}

// Gate-passing (hypothesis, witness) pairs are ~0.5 per (hypothesis, camera): evaluating the confidence in place
// would run ~200 instructions with a third of the lanes.  They are pushed to a per-wave LDS ring as
// (origin lane, camera, witness) and evaluated 64 at a time; the origin lane's 3-D segment comes back over the
// wave (shuffles), the per-camera maxima are collected with LDS atomic max (confidences are positive floats,
// which order like ints) and summed in ascending camera order at the end (cudawrapper.cu:677-709).
constexpr int kVQ = 128;

__device__ __forceinline__ void vw_drain(const VerifyArgs& a, int start, unsigned* q, int head, int n, int lane,
                                         f3 X1, f3 X2, f3 v1, float* smax_wave, float two_sig_d, float two_sig_a)
{
    unsigned key = 0, wi = 0;
    if (lane < n) { key = q[((head + lane) & (kVQ - 1)) * 2]; wi = q[((head + lane) & (kVQ - 1)) * 2 + 1]; }
    const int origin = key & 63, cam = (int)(key >> 8);
    const f3 hX1 = mk3(__shfl(X1.x, origin), __shfl(X1.y, origin), __shfl(X1.z, origin));
    const f3 hX2 = mk3(__shfl(X2.x, origin), __shfl(X2.y, origin), __shfl(X2.z, origin));
    const f3 hv = mk3(__shfl(v1.x, origin), __shfl(v1.y, origin), __shfl(v1.z, origin));
    if (lane < n) {
        bool va, vb;
        const f3 pr1 = project(a.P + cam * 12, hX1, va);                 // :690-693
        const f3 pr2 = project(a.P + cam * 12, hX2, vb);
        if (va && vb) {
            const f3 line1 = cross(pr1, pr2);
            const float den1 = line_norm2d(line1);
            const float4 r2 = a.rec[2][start + wi], r3 = a.rec[3][start + wi], tq = a.rec[4][start + wi];
            const f3 l2 = mk3(r3.x, r3.y, r3.z);
            const f3 q1 = mk3(tq.x, tq.y, 1.0f), q2 = mk3(tq.z, tq.w, 1.0f);
            const float dd1 = __builtin_fmaxf(__builtin_fabsf(line_numer(l2, pr1) / r2.w), __builtin_fabsf(line_numer(l2, pr2) / r2.w));
            const float dd2 = __builtin_fmaxf(__builtin_fabsf(line_numer(line1, q1) / den1), __builtin_fabsf(line_numer(line1, q2) / den1));
            const float dist = __builtin_fmaxf(dd1, dd2);
            const float cs = __builtin_fmaxf(__builtin_fminf(dot(hv, mk3(r2.x, r2.y, r2.z)), 1.0f), -1.0f);
            float angle = (float)((double)c_acosf(cs) / 3.1415926535897931e+0 * (double)180.0f);
            if (angle > 90.0f) angle = 180.0f - angle;
            const float cd = c_expf(-dist * dist / two_sig_d);
            const float conf = __builtin_fminf(cd, c_expf(-angle * angle / two_sig_a));
            if (conf > 0.5f)                                              // :699-704 (max over the camera's witnesses)
                atomicMax(reinterpret_cast<int*>(&smax_wave[origin * a.N + cam]), __float_as_int(conf));
        }
    }
}

// LDS image of one source segment: ALL its candidates bucketed by the first depth d1.  Depths are positive floats,
// whose bit patterns are monotone in the value and roughly logarithmic, so (bits >> kBucketShift) is an order
// preserving bucket id of relative width 2^-7 .. 2^-6 (0.8-1.6 %) -- about the width of the gate window
// (spatial_k ~ 0.5 % of the depth).  A counting sort on that id (LDS atomics, O(m), no comparison sort) makes every
// depth window a contiguous range of <= 3-4 buckets; the order inside a bucket is arbitrary, which cannot change
// the result (per-camera maxima, summed in camera order).
constexpr int kBucketShift = 17;
constexpr int kBuckets = 512;                  // 8 octaves of depth; anything beyond is clamped into the last bucket
// entry key = (d1 bits << 32) | (camera << 24) | candidate index
__device__ __forceinline__ float key_d1(unsigned long long k) { return __uint_as_float((unsigned)(k >> 32)); }
__device__ __forceinline__ int key_cam(unsigned long long k) { return (int)((k >> 24) & 0xffu); }
__device__ __forceinline__ int key_idx(unsigned long long k) { return (int)(k & 0xffffffu); }
__device__ __forceinline__ int bucket_of(float d, int base)
{
    if (!(d > 0.0f)) return 0;                                        // windows may reach below zero
    const int raw = (int)(__float_as_uint(d) >> kBucketShift) - base;
    return raw < 0 ? 0 : (raw > kBuckets - 1 ? kBuckets - 1 : raw);
}

__global__ __launch_bounds__(256) void k_verify_window(VerifyArgs a)
{
    extern __shared__ __align__(16) unsigned char s_raw[];
    __shared__ int s_dmax, s_base;
    __shared__ int s_bstart[kBuckets + 1];
    __shared__ int s_cursor[kBuckets];
    const int y = a.seg_begin + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int start = a.row_start[y * a.N];
    const int m = a.row_start[(y + 1) * a.N] - start;
    if (m == 0) return;
    if (a.debug == 4) return;

    unsigned long long* sk = reinterpret_cast<unsigned long long*>(s_raw);   // [mmax] keys grouped by bucket
    float* sd2 = reinterpret_cast<float*>(sk + a.mmax);                      // [mmax] d2 in the same order
    float* smax = sd2 + a.mmax;                                              // [256][N] per-(hypothesis lane, camera) maxima
    unsigned* qall = reinterpret_cast<unsigned*>(smax + 256 * a.N);
    unsigned* q = qall + wave * kVQ * 2;
    float* smax_wave = smax + wave * 64 * a.N;

    // ---- counting sort of the candidates on the depth bucket
    if (tid == 0) { s_dmax = 0; s_base = 0x7fffffff; }
    for (int b = tid; b < kBuckets; b += 256) s_cursor[b] = 0;
    __syncthreads();
    float dm = 0.0f;
    int rmin = 0x7fffffff;
    for (int i = tid; i < m; i += 256) {
        const float d1 = a.rec[0][start + i].w, d2 = a.rec[1][start + i].w;
        dm = __builtin_fmaxf(dm, __builtin_fmaxf(__builtin_fabsf(d1), __builtin_fabsf(d2)));
        rmin = min(rmin, (int)(__float_as_uint(d1) >> kBucketShift));
    }
    for (int o = 32; o > 0; o >>= 1) { dm = __builtin_fmaxf(dm, __shfl_down(dm, o)); rmin = min(rmin, __shfl_down(rmin, o)); }
    if (lane == 0) { atomicMax(&s_dmax, __float_as_int(dm)); atomicMin(&s_base, rmin); }   // non-negative floats order like ints
    __syncthreads();
    const int base = s_base;
    for (int i = tid; i < m; i += 256) atomicAdd(&s_cursor[bucket_of(a.rec[0][start + i].w, base)], 1);
    __syncthreads();
    {   // exclusive scan of the 512 bucket counts: 2 per thread + wave scan + 4 wave totals
        const int c0 = s_cursor[2 * tid], c1 = s_cursor[2 * tid + 1];
        int incl = c0 + c1;
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        __shared__ int s_wtot[4];
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wave; ++w) off += s_wtot[w];
        const int excl = off + incl - (c0 + c1);
        s_bstart[2 * tid] = excl;
        s_bstart[2 * tid + 1] = excl + c0;
        if (tid == 255) s_bstart[kBuckets] = excl + c0 + c1;
        __syncthreads();
        s_cursor[2 * tid] = excl;
        s_cursor[2 * tid + 1] = excl + c0;
    }
    __syncthreads();
    for (int i = tid; i < m; i += 256) {
        const float d1 = a.rec[0][start + i].w;
        const unsigned cam = (unsigned)__float_as_int(a.rec[3][start + i].w);
        const int pos = atomicAdd(&s_cursor[bucket_of(d1, base)], 1);
        sk[pos] = ((unsigned long long)__float_as_uint(d1) << 32) | ((unsigned long long)cam << 24) | (unsigned)i;
        sd2[pos] = a.rec[1][start + i].w;
    }
    __syncthreads();
    if (a.debug == 1) return;

    const f3 C = mk3(a.C_src[0], a.C_src[1], a.C_src[2]);
    const float4 sseg = a.src_segs[y];
    const f3 ray1 = normalize(mat3_apply(a.RtKinv_src, mk3(sseg.x, sseg.y, 1.0f)));
    const f3 ray2 = normalize(mat3_apply(a.RtKinv_src, mk3(sseg.z, sseg.w, 1.0f)));
    const float c_inf = __builtin_fmaxf(__builtin_fabsf(C.x), __builtin_fmaxf(__builtin_fabsf(C.y), __builtin_fabsf(C.z)));
    const float dabs_max = __int_as_float(s_dmax);
    const float two_sig_d = 2.0f * (a.sigma_p * a.sigma_p);
    const float two_sig_a = 2.0f * (a.sigma_a * a.sigma_a);
    const bool gate = a.spatial_k > 0.0f;

    for (int h0 = 0; h0 < m; h0 += 256) {
        const int h = h0 + tid;
        const bool hv = h < m;
        f3 X1 = mk3(0, 0, 0), X2 = mk3(0, 0, 0), v1 = mk3(0, 0, 0);
        float d1y = 0.0f, d2y = 0.0f, T1 = 0.0f, T2 = 0.0f, w1 = 0.0f, w2 = 0.0f;
        int cam_h = -1;
        if (hv) {
            const float4 r0 = a.rec[0][start + h], r1 = a.rec[1][start + h], r2 = a.rec[2][start + h];
            X1 = mk3(r0.x, r0.y, r0.z); d1y = r0.w;
            X2 = mk3(r1.x, r1.y, r1.z); d2y = r1.w;
            v1 = mk3(r2.x, r2.y, r2.z);
            cam_h = __float_as_int(a.rec[3][start + h].w);
            if (gate) {
                const float unc1 = a.spatial_k * length(C - X1);      // cudawrapper.cu:390-394
                const float unc2 = a.spatial_k * length(C - X2);
                T1 = sq_threshold(unc1);
                T2 = sq_threshold(unc2);
                w1 = window_margin(unc1, __builtin_fabsf(d1y), dabs_max, c_inf);
                w2 = window_margin(unc2, __builtin_fabsf(d2y), dabs_max, c_inf);
            } else {
                w1 = w2 = __builtin_inff();
            }
        }
        for (int c = 0; c < a.N; ++c) smax_wave[lane * a.N + c] = 0.0f;
        int head = 0, count = 0;                                       // wave-uniform ring state
        const float lo1 = d1y - w1, hi1 = d1y + w1;
        int j = 0, jend = 0;
        if (hv) { j = s_bstart[bucket_of(lo1, base)]; jend = s_bstart[bucket_of(hi1, base) + 1]; }
        unsigned long long cur = sk[min(j, m - 1)];
        float cur2 = sd2[min(j, m - 1)];
        for (;;) {
            const bool in = j < jend;
            if (!__any(in)) break;
            const unsigned long long nxt = sk[min(j + 1, m - 1)];      // issued before cur is consumed
            const float nxt2 = sd2[min(j + 1, m - 1)];
            bool push = false;
            const float cd1 = key_d1(cur);
            // :674 (other cameras only), then the 1-D pre-tests that every gate-passing witness satisfies
            if (in && key_cam(cur) != cam_h && cd1 >= lo1 && cd1 <= hi1 && __builtin_fabsf(cur2 - d2y) <= w2) {
                push = true;
                if (gate) {                                            // exact 3-D gate, :396-400
                    // the witness' 3-D endpoints are recomputed from its depths (same float operations as
                    // k_cand_prep, hence the same bits) instead of being gathered from memory
                    const f3 e1 = X1 - (C + cd1 * ray1);
                    const f3 e2 = X2 - (C + cur2 * ray2);
                    push = !(dot(e1, e1) > T1 || dot(e2, e2) > T2);
                }
            }
            const unsigned long long pm = __ballot(push);
            if (pm) {
                if (push) {
                    const int pos = (head + count + __popcll(pm & ((1ull << lane) - 1ull))) & (kVQ - 1);
                    q[pos * 2] = (unsigned)lane | ((unsigned)key_cam(cur) << 8);
                    q[pos * 2 + 1] = (unsigned)key_idx(cur);
                }
                count += __popcll(pm);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (count >= 64) {
                    vw_drain(a, start, q, head, 64, lane, X1, X2, v1, smax_wave, two_sig_d, two_sig_a);
                    head = (head + 64) & (kVQ - 1);
                    count -= 64;
                }
            }
            if (in) { ++j; cur = nxt; cur2 = nxt2; }
        }
        if (count > 0) vw_drain(a, start, q, head, count, lane, X1, X2, v1, smax_wave, two_sig_d, two_sig_a);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        float conf_sum = 0.0f;
        for (int c = 0; c < a.N; ++c) conf_sum += smax_wave[lane * a.N + c];   // ascending camera order; +0.0f is exact
        if (hv) a.cand_conf[start + h] = conf_sum;
    }
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

void launch_cand_prep(const VerifyArgs& a, hipStream_t st)
{
    hipLaunchKernelGGL(k_cand_prep, dim3(a.seg_end - a.seg_begin), dim3(256), 0, st, a);
}
size_t verify_window_lds_bytes(int mmax, int N) { return (size_t)mmax * 12 + (size_t)256 * N * 4 + 4 * kVQ * 8 + 16; }
void launch_verify_window(const VerifyArgs& a, hipStream_t st)
{
    const size_t lds = verify_window_lds_bytes(a.mmax, a.N);
    static bool attr_set = false;
    if (lds > 48 * 1024 && !attr_set) {       // opt in to > default dynamic LDS only when a launch needs it
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_verify_window), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_verify_window, dim3(a.seg_end - a.seg_begin), dim3(256), lds, st, a);
}
void launch_seg_mmax(const int* row_start, int N, int seg_begin, int seg_end, int* out, hipStream_t st)
{
    const int n = seg_end - seg_begin;
    if (n > 0) hipLaunchKernelGGL(k_seg_mmax, dim3((n + 255) / 256), dim3(256), 0, st, row_start, N, seg_begin, seg_end, out);
}

}  // namespace l3d
