// l3d_similarity.hpp -- Line3D::similarity_coll3D (line3D.cc:1600-1681) as one device function, shared by the batched
// kernel of l3d_similarity_coll3D_batch (l3d_kernels.hip) and the device-side affinity fill (l3d_affinity.hip).
#pragma once

#include "l3d_contract.hpp"
#include "l3d_kernels.hpp"

namespace l3d {

// =================================================================================================
// similarity_coll3D (line3D.cc:1600-1681): double geometry, float Gaussians.
// =================================================================================================
// :1684-1691.  `P1 + (dir * ((X - P1).transpose()) * dir)` (:1689) is, by C++ precedence, the outer product dir * v^T times dir: row i of
// the 3x3 product is dir[i] * v[j], the matrix-vector product adds its three terms in index order (an ulp of a double away from
// dir * (v . dir) -- the association the expression has, not the one it suggests)
__device__ __forceinline__ float p2l_3D(const double* P1, const double* dir, const double* X)
{
    const double v0 = X[0] - P1[0], v1 = X[1] - P1[1], v2 = X[2] - P1[2];
    const double r0 = ((dir[0] * v0) * dir[0] + (dir[0] * v1) * dir[1]) + (dir[0] * v2) * dir[2];
    const double r1 = ((dir[1] * v0) * dir[0] + (dir[1] * v1) * dir[1]) + (dir[1] * v2) * dir[2];
    const double r2 = ((dir[2] * v0) * dir[0] + (dir[2] * v1) * dir[1]) + (dir[2] * v2) * dir[2];
    const double d0 = (P1[0] + r0) - X[0];
    const double d1 = (P1[1] + r1) - X[1];
    const double d2 = (P1[2] + r2) - X[2];
    return (float)__builtin_sqrt(d0 * d0 + d1 * d1 + d2 * d2);
}
__device__ __forceinline__ float lower_unc(const Hypothesis& h, float depth)   // view.cc:353-359
{
    return depth < h.median_depth ? h.k_lower * depth : h.k_lower * h.median_depth;
}
__device__ __forceinline__ float upper_unc(const Hypothesis& h, float depth)
{
    return depth < h.median_depth ? h.k_upper * depth : h.k_upper * h.median_depth;
}
__device__ __forceinline__ float gauss_term(const Hypothesis& h, float depth, float d, float two_log)
{
    const float lo = lower_unc(h, depth);
    if (d < lo) return 1.0f;
    const float up = upper_unc(h, depth);
    const float sig = -(up - lo) * (up - lo) / two_log;         // view.cc:371-377
    return c_expf(-(d - lo) * (d - lo) / (2.0f * sig));
}

// similarity_coll3D(a, b): min of the positional and the angular Gaussian, values <= 0.01 cut to 0 (:1676-1680)
__device__ __forceinline__ float similarity_coll3D(const Hypothesis& a, const Hypothesis& b, float sigma_a, float two_log)
{
    const float d1 = p2l_3D(b.P1, b.dir, a.P1);
    const float d2 = p2l_3D(b.P1, b.dir, a.P2);
    const float w12 = __builtin_fminf(gauss_term(a, a.depth_p1, d1, two_log), gauss_term(a, a.depth_p2, d2, two_log));
    const float d3 = p2l_3D(a.P1, a.dir, b.P1);
    const float d4 = p2l_3D(a.P1, a.dir, b.P2);
    const float w34 = __builtin_fminf(gauss_term(b, b.depth_p1, d3, two_log), gauss_term(b, b.depth_p2, d4, two_log));
    const float w_d = __builtin_fminf(w12, w34);
    const double dd = a.dir[0] * b.dir[0] + a.dir[1] * b.dir[1] + a.dir[2] * b.dir[2];
    float angle = (float)(c_acos(__builtin_fmax(__builtin_fmin(dd, 1.0), -1.0)) / 3.14159265358979323846 * (double)180.0f);
    if (angle > 90.0f) angle = 180.0f - angle;
    const float w_a = c_expf(-angle * angle / (2.0f * sigma_a * sigma_a));
    const float s = __builtin_fminf(w_d, w_a);
    return s <= 0.01f ? 0.0f : s;
}

}  // namespace l3d
