// l3d_verify_eval.hpp -- one (hypothesis, witness) evaluation of K_verify_matches' inner loop (cudawrapper.cu:656-706),
// shared by the depth-window kernel (l3d_verify_window.hip) and the chain kernel (l3d_chain_split.hip): both compact
// the few pairs that pass their 1-D depth pre-tests and evaluate them here with the reference's float operations.
#pragma once

#include "l3d_geometry.hpp"

namespace l3d {

// Hypothesis: 3-D endpoints hX1, hX2 (C + depth * ray of the source segment, :644-645), unit direction hv, squared gate
// thresholds hT1, hT2 (sq_threshold of spatial_k * |C - X|, :390-394).  Witness: depths wd1, wd2 along the SAME rays
// (:669-672), its camera's projection matrix Pc and its 2-D target segment tq.  Returns the confidence
// min(exp(-d^2/2 sigma_p^2), exp(-ang^2/2 sigma_a^2)) (:404-426), or 0 when the 3-D gate (:396-400) or a projection (:690-693)
// fails.  The caller keeps it if > 0.5 (:699-704).
__device__ __forceinline__ float witness_conf(f3 C, f3 ray1, f3 ray2, f3 hX1, f3 hX2, f3 hv, float hT1, float hT2, bool gate,
                                              float wd1, float wd2, const float* Pc, float4 tq, float two_sig_d, float two_sig_a)
{
    const f3 Q1 = C + wd1 * ray1;                                        // D_unproject_point_src, :669-672
    const f3 Q2 = C + wd2 * ray2;
    if (gate) {
        const f3 e1 = hX1 - Q1, e2 = hX2 - Q2;
        if (dot(e1, e1) > hT1 || dot(e2, e2) > hT2) return 0.0f;         // :396-400 on squared distances
    }
    bool va, vb;
    const f3 pr1 = project(Pc, hX1, va);
    const f3 pr2 = project(Pc, hX2, vb);
    if (!(va && vb)) return 0.0f;
    const f3 line1 = cross(pr1, pr2);
    const float den1 = line_norm2d(line1);
    const f3 q1 = mk3(tq.x, tq.y, 1.0f), q2 = mk3(tq.z, tq.w, 1.0f);
    const f3 l2 = cross(q1, q2);
    const float den2 = line_norm2d(l2);
    const float dd1 = __builtin_fmaxf(__builtin_fabsf(line_numer(l2, pr1) / den2), __builtin_fabsf(line_numer(l2, pr2) / den2));
    const float dd2 = __builtin_fmaxf(__builtin_fabsf(line_numer(line1, q1) / den1), __builtin_fabsf(line_numer(line1, q2) / den1));
    const float dist = __builtin_fmaxf(dd1, dd2);
    const f3 v2 = normalize(Q1 - Q2);
    const float cs = __builtin_fmaxf(__builtin_fminf(dot(hv, v2), 1.0f), -1.0f);
    float angle = (float)((double)c_acosf(cs) / 3.1415926535897931e+0 * (double)180.0f);
    if (angle > 90.0f) angle = 180.0f - angle;
    const float cd = c_expf(-dist * dist / two_sig_d);
    return __builtin_fminf(cd, c_expf(-angle * angle / two_sig_a));
}

// Conservative 1-D pre-test half-width for a hypothesis depth d_y against a witness depth d_i: any witness that passes the
// reference gate sqrtf(|X_y - X_i|^2) <= unc satisfies |d_y - d_i| <= unc*(1+8u) + 3.5u*(|d_y| + |d_i| + |C|_inf), u = 2^-24
// (two roundings per coordinate of X = C + d*ray, one for the difference, dot/sqrt relative 3u, |ray| = 1 +- 3u).  The margin
// below is > 5x that bound; dabs bounds |d_i|.
__device__ __forceinline__ float window_margin(float unc, float d_y, float dabs, float c_inf)
{
    return unc * 1.00001f + 2.0e-6f * (d_y + dabs + c_inf);
}

}  // namespace l3d
