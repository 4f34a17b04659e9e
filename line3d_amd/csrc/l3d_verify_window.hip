// l3d_verify_window.hip -- stage 2 (K_verify_matches, cudawrapper.cu:614-714) as a depth-window search.
//
// Observation: hypothesis y and witness i of the same source segment are unprojected along the SAME two
// source rays (Q1 = C + d1_i*ray1, P1 = C + d1_y*ray1, cudawrapper.cu:644-645,671-672), so the reference's
// 3-D gate |P1-Q1| <= k*depth1 && |P2-Q2| <= k*depth2 (:388-401) is, up to float rounding, a 1-D interval
// test on the depths.  Sorting every (segment, camera) run by d1 turns the O(m^2) all-pairs loop into
// O(m * N * (log n + window)) with the EXACT reference gate and confidence evaluated only inside a
// conservative window (the window margin provably covers the rounding of the 3-D computation, see
// window_margin below).  Results are bit-identical to the all-pairs kernel
// (k_verify in l3d_kernels.hip, kept as the A/B reference and the fallback for huge segments).
//
//   k_cand_prep      one workgroup per source segment: per-candidate records (3-D endpoints, unit direction,
//                    target line + norm, target segment, camera) in 5 float4 arrays (80 B / candidate)
//   k_verify_window  one workgroup per source segment: LDS holds the runs sorted by d1 (d1, d2, index);
//                    lane <-> hypothesis; for each camera: projection, lower_bound, window scan
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

__global__ __launch_bounds__(256) void k_verify_window(VerifyArgs a)
{
    extern __shared__ __align__(16) unsigned char s_raw[];
    __shared__ int s_dmax;
    const int y = a.seg_begin + blockIdx.x;
    const int tid = threadIdx.x;
    const int start = a.row_start[y * a.N];
    const int m = a.row_start[(y + 1) * a.N] - start;
    if (m == 0) return;

    float* sd1 = reinterpret_cast<float*>(s_raw);          // [m] d1 sorted inside each camera run
    float* sd2 = sd1 + a.mmax;                             // [m] d2 in the same order
    int* sidx = reinterpret_cast<int*>(sd2 + a.mmax);      // [m] candidate index (relative to start)
    float* ud1 = reinterpret_cast<float*>(sidx + a.mmax);  // [m] unsorted d1 (staging)

    if (tid == 0) s_dmax = 0;
    __syncthreads();
    float dm = 0.0f;
    for (int i = tid; i < m; i += 256) {
        const float d1 = a.rec[0][start + i].w, d2 = a.rec[1][start + i].w;
        ud1[i] = d1;
        dm = __builtin_fmaxf(dm, __builtin_fmaxf(__builtin_fabsf(d1), __builtin_fabsf(d2)));
    }
    for (int o = 32; o > 0; o >>= 1) dm = __builtin_fmaxf(dm, __shfl_down(dm, o));
    if ((tid & 63) == 0) atomicMax(&s_dmax, __float_as_int(dm));     // non-negative floats order like ints
    __syncthreads();
    // rank of every candidate inside its camera run (ties by index -> a permutation)
    for (int i = tid; i < m; i += 256) {
        const int cam = __float_as_int(a.rec[3][start + i].w);
        const int b = a.row_start[y * a.N + cam] - start, e = a.row_start[y * a.N + cam + 1] - start;
        const float di = ud1[i];
        int r = 0;
        for (int j = b; j < e; ++j) { const float dj = ud1[j]; r += (dj < di) || (dj == di && j < i); }
        sd1[b + r] = di;
        sd2[b + r] = a.rec[1][start + i].w;
        sidx[b + r] = i;
    }
    __syncthreads();

    const f3 C = mk3(a.C_src[0], a.C_src[1], a.C_src[2]);
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
        float conf_sum = 0.0f;
        for (int c = 0; c < a.N; ++c) {                                // ascending camera order, :677-687
            const int b = a.row_start[y * a.N + c] - start, e = a.row_start[y * a.N + c + 1] - start;
            if (b == e) continue;                                      // uniform
            bool act = hv && c != cam_h;                               // :674
            f3 pr1 = mk3(0, 0, 0), pr2 = mk3(0, 0, 0), line1 = mk3(0, 0, 0);
            float den1 = 1.0f;
            if (act) {
                bool va, vb;
                pr1 = project(a.P + c * 12, X1, va);                   // :690-693
                pr2 = project(a.P + c * 12, X2, vb);
                act = va && vb;
                line1 = cross(pr1, pr2);
                den1 = line_norm2d(line1);
            }
            const float lo1 = d1y - w1, hi1 = d1y + w1;
            int lo = b, hi = e;                                        // lower_bound(sd1[b,e), lo1)
            while (__any(act && lo < hi)) {
                if (act && lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (sd1[mid] < lo1) lo = mid + 1; else hi = mid;
                }
            }
            float cur_max = 0.0f;
            int j = lo;
            for (;;) {
                const bool in = act && j < e && sd1[j] <= hi1;
                if (!__any(in)) break;
                if (in) {
                    if (__builtin_fabsf(sd2[j] - d2y) <= w2) {
                        const int i = sidx[j];
                        const float4 q0 = a.rec[0][start + i], q1r = a.rec[1][start + i];
                        bool ok = true;
                        if (gate) {                                    // exact 3-D gate, :396-400
                            const f3 e1 = X1 - mk3(q0.x, q0.y, q0.z);
                            const f3 e2 = X2 - mk3(q1r.x, q1r.y, q1r.z);
                            ok = !(dot(e1, e1) > T1 || dot(e2, e2) > T2);
                        }
                        if (ok) {
                            const float4 r2 = a.rec[2][start + i], r3 = a.rec[3][start + i], tq = a.rec[4][start + i];
                            const f3 l2 = mk3(r3.x, r3.y, r3.z);
                            const f3 q1 = mk3(tq.x, tq.y, 1.0f), q2 = mk3(tq.z, tq.w, 1.0f);
                            const float dd1 = __builtin_fmaxf(__builtin_fabsf(line_numer(l2, pr1) / r2.w),
                                                              __builtin_fabsf(line_numer(l2, pr2) / r2.w));
                            const float dd2 = __builtin_fmaxf(__builtin_fabsf(line_numer(line1, q1) / den1),
                                                              __builtin_fabsf(line_numer(line1, q2) / den1));
                            const float dist = __builtin_fmaxf(dd1, dd2);
                            const float cs = __builtin_fmaxf(__builtin_fminf(dot(v1, mk3(r2.x, r2.y, r2.z)), 1.0f), -1.0f);
                            float angle = (float)((double)c_acosf(cs) / 3.1415926535897931e+0 * (double)180.0f);
                            if (angle > 90.0f) angle = 180.0f - angle;
                            const float cd = c_expf(-dist * dist / two_sig_d);
                            const float conf = __builtin_fminf(cd, c_expf(-angle * angle / two_sig_a));
                            if (conf > 0.5f && conf > cur_max) cur_max = conf;     // :699-704
                        }
                    }
                    ++j;
                }
            }
            conf_sum += cur_max;
        }
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
size_t verify_window_lds_bytes(int mmax) { return (size_t)mmax * 16; }
void launch_verify_window(const VerifyArgs& a, hipStream_t st)
{
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_verify_window), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_verify_window, dim3(a.seg_end - a.seg_begin), dim3(256), verify_window_lds_bytes(a.mmax), st, a);
}
void launch_seg_mmax(const int* row_start, int N, int seg_begin, int seg_end, int* out, hipStream_t st)
{
    const int n = seg_end - seg_begin;
    if (n > 0) hipLaunchKernelGGL(k_seg_mmax, dim3((n + 255) / 256), dim3(256), 0, st, row_start, N, seg_begin, seg_end, out);
}

}  // namespace l3d
