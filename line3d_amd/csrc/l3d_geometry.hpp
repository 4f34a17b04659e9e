// l3d_geometry.hpp -- device/host geometry of the matching path, written against the numeric
// contract (l3d_contract.hpp).  Each function names the reference device function whose result
// it must reproduce (cudawrapper.cu line numbers); the code is organised for the HIP kernels
// (per-segment invariants split from per-pair work), not as a transcription.
#pragma once

#include "l3d_contract.hpp"

namespace l3d {

constexpr float kEpsG = 1e-12f;             // L3D_EPS_G, cudawrapper.h:43
constexpr float kCollinAffT = 0.50f;        // cudawrapper.h:44
constexpr float kMinOverlapLower = 0.10f;   // cudawrapper.h:45
constexpr float kMinOverlapUpper = 0.30f;   // cudawrapper.h:46

// |(l.x p.x + l.y p.y + l.z) / sqrt(l.x^2 + l.y^2)|   (D_distance_p2l_2D_f3, :58-61)
L3D_HD float line_norm2d(f3 l) { return __builtin_sqrtf(l.x * l.x + l.y * l.y); }
L3D_HD float line_numer(f3 l, f3 p) { return l.x * p.x + l.y * p.y + l.z; }
L3D_HD float dist_p2l(f3 l, f3 p) { return __builtin_fabsf(line_numer(l, p) / line_norm2d(l)); }

// D_segment_length_2D_f3, :95-99
L3D_HD float seglen2d(f3 a, f3 b)
{
    const float vx = a.x - b.x, vy = a.y - b.y;
    return __builtin_sqrtf(vx * vx + vy * vy);
}

// D_point_on_segment_2D_f3, :135-141
L3D_HD bool on_segment(f3 p1, f3 p2, f3 q)
{
    return ((p1.x - q.x) * (p2.x - q.x) + (p1.y - q.y) * (p2.y - q.y)) < kEpsG;
}

// D_normalize_hom_coords_2D, :255-267.  valid=false <=> the reference returns (0,0,0).
L3D_HD f3 hom_normalize(f3 p, bool& valid)
{
    valid = __builtin_fabsf(p.z) > kEpsG;
    if (!valid) return mk3(0.0f, 0.0f, 0.0f);
    return mk3(p.x / p.z, p.y / p.z, 1.0f);
}

// D_segment_overlap_2D (live body :209-251); lengths of the two segments are passed in because
// one of them is a per-segment invariant in the kernels.
L3D_HD float segment_overlap(f3 s1, f3 s2, float len_src, f3 q1, f3 q2, float len_tgt)
{
    if (len_src < 1.0f || len_tgt < 1.0f) return 0.0f;
    const bool q1_on_s = on_segment(s1, s2, q1);
    const bool q2_on_s = on_segment(s1, s2, q2);
    if (q1_on_s && q2_on_s) return len_tgt / len_src;
    const bool s1_on_q = on_segment(q1, q2, s1);
    const bool s2_on_q = on_segment(q1, q2, s2);
    if (s1_on_q && s2_on_q) return len_src / len_tgt;
    if (q1_on_s) {
        const float len1 = seglen2d(s2, q2);
        const float len2 = seglen2d(s1, q2);
        if (s1_on_q && len1 > kEpsG) return seglen2d(q1, s1) / len1;
        if (len2 > kEpsG) return seglen2d(q1, s2) / len2;
        return 0.0f;
    }
    if (q2_on_s) {
        const float len1 = seglen2d(s1, q1);
        const float len2 = seglen2d(s2, q1);
        if (s2_on_q && len1 > kEpsG) return seglen2d(q2, s2) / len1;
        if (len2 > kEpsG) return seglen2d(q2, s1) / len2;
        return 0.0f;
    }
    return 0.0f;
}

// D_get_triangulation_depth, :306-335, with both rays already normalised by the caller.
L3D_HD float tri_depth(f3 ray1, f3 ray2, f3 w0, bool for_src)
{
    const float a = dot(ray1, ray1);
    const float b = dot(ray1, ray2);
    const float c = dot(ray2, ray2);
    const float d = dot(ray1, w0);
    const float e = dot(ray2, w0);
    const float denom = a * c - b * b;
    if (__builtin_fabsf(denom) > kEpsG) return for_src ? (b * e - c * d) / denom : (a * e - b * d) / denom;
    return -1.0f;
}

// D_project_point_tgt, :355-377 (P 3x4 row-major); valid <=> int(result.z) == 1
L3D_HD f3 project(const float* P, f3 X, bool& valid)
{
    const float p0 = (((0.0f + P[0] * X.x) + P[1] * X.y) + P[2] * X.z) + P[3] * 1.0f;
    const float p1 = (((0.0f + P[4] * X.x) + P[5] * X.y) + P[6] * X.z) + P[7] * 1.0f;
    const float p2 = (((0.0f + P[8] * X.x) + P[9] * X.y) + P[10] * X.z) + P[11] * 1.0f;
    valid = __builtin_fabsf(p2) > kEpsG;
    if (!valid) return mk3(0.0f, 0.0f, 0.0f);
    return mk3(p0 / p2, p1 / p2, 1.0f);
}

// Per-source-segment data of the pair test that does not depend on the target segment.
struct SrcPairInv {
    f3 p1, p2, line1, epi_p1, epi_p2;
    float len;
};
L3D_HD SrcPairInv make_src_inv(float4 s, const float* F)
{
    SrcPairInv r;
    r.p1 = mk3(s.x, s.y, 1.0f);
    r.p2 = mk3(s.z, s.w, 1.0f);
    r.line1 = cross(r.p1, r.p2);
    r.epi_p1 = mat3_apply(F, r.p1);   // D_epipolar_line(p, cam, false), :144-163
    r.epi_p2 = mat3_apply(F, r.p2);
    r.len = seglen2d(r.p1, r.p2);
    return r;
}
struct TgtPairInv {
    f3 q1, q2, line2, epi_q1, epi_q2;
    float len;
};
L3D_HD TgtPairInv make_tgt_inv(float4 t, const float* F)
{
    TgtPairInv r;
    r.q1 = mk3(t.x, t.y, 1.0f);
    r.q2 = mk3(t.z, t.w, 1.0f);
    r.line2 = cross(r.q1, r.q2);
    r.epi_q1 = mat3T_apply(F, r.q1);  // transpose = true
    r.epi_q2 = mat3T_apply(F, r.q2);
    r.len = seglen2d(r.q1, r.q2);
    return r;
}

// The epipolar/overlap part of K_pairwise_matches (:563-588).  On success the four
// intersection points are returned for the triangulation.
L3D_HD bool pair_overlap_test(const SrcPairInv& s, const TgtPairInv& t, f3& l2_p1, f3& l2_p2, f3& l1_q1, f3& l1_q2)
{
    bool v1, v2, v3, v4;
    l2_p1 = hom_normalize(cross(t.line2, s.epi_p1), v1);
    l2_p2 = hom_normalize(cross(t.line2, s.epi_p2), v2);
    l1_q1 = hom_normalize(cross(s.line1, t.epi_q1), v3);
    l1_q2 = hom_normalize(cross(s.line1, t.epi_q2), v4);
    if (!(v1 && v2 && v3 && v4)) return false;
    // the intersection pairs' lengths are per-pair; the segment lengths are invariants
    const float overlap1 = segment_overlap(s.p1, s.p2, s.len, l1_q1, l1_q2, seglen2d(l1_q1, l1_q2));
    const float overlap2 = segment_overlap(t.q1, t.q2, t.len, l2_p1, l2_p2, seglen2d(l2_p1, l2_p2));
    return __builtin_fminf(overlap1, overlap2) > kMinOverlapLower && __builtin_fmaxf(overlap1, overlap2) > kMinOverlapUpper;
}

// Only the four intersection points of pair_overlap_test (:563-575), for pairs already known to pass it.
L3D_HD void pair_intersections(const SrcPairInv& s, const TgtPairInv& t, f3& l2_p1, f3& l2_p2, f3& l1_q1, f3& l1_q2)
{
    bool v;
    l2_p1 = hom_normalize(cross(t.line2, s.epi_p1), v);
    l2_p2 = hom_normalize(cross(t.line2, s.epi_p2), v);
    l1_q1 = hom_normalize(cross(s.line1, t.epi_q1), v);
    l1_q2 = hom_normalize(cross(s.line1, t.epi_q2), v);
}

// The four triangulated depths (:590-601).  RtKinv_src / RtKinv_tgt 3x3 row-major.
L3D_HD float4 pair_depths(const SrcPairInv& s, const TgtPairInv& t, f3 l2_p1, f3 l2_p2, f3 l1_q1, f3 l1_q2,
                          const float* RtKinv_src, const float* RtKinv_tgt, f3 C_src, f3 C_tgt)
{
    const f3 w0 = C_src - C_tgt;
    float4 d;
    d.x = tri_depth(normalize(mat3_apply(RtKinv_src, s.p1)), normalize(mat3_apply(RtKinv_tgt, l2_p1)), w0, true);
    d.y = tri_depth(normalize(mat3_apply(RtKinv_src, s.p2)), normalize(mat3_apply(RtKinv_tgt, l2_p2)), w0, true);
    d.z = tri_depth(normalize(mat3_apply(RtKinv_src, l1_q1)), normalize(mat3_apply(RtKinv_tgt, t.q1)), w0, false);
    d.w = tri_depth(normalize(mat3_apply(RtKinv_src, l1_q2)), normalize(mat3_apply(RtKinv_tgt, t.q2)), w0, false);
    return d;
}

// The same four depths with the rays of the exact endpoints (per-segment invariants) already normalised by the caller
// with the very same operations: ray_p = normalize(mat3_apply(RtKinv_src, p)), ray_q = normalize(mat3_apply(RtKinv_tgt, q)).
L3D_HD float4 pair_depths_pre(f3 ray_p1, f3 ray_p2, f3 ray_q1, f3 ray_q2, f3 l2_p1, f3 l2_p2, f3 l1_q1, f3 l1_q2,
                              const float* RtKinv_src, const float* RtKinv_tgt, f3 C_src, f3 C_tgt)
{
    const f3 w0 = C_src - C_tgt;
    float4 d;
    d.x = tri_depth(ray_p1, normalize(mat3_apply(RtKinv_tgt, l2_p1)), w0, true);
    d.y = tri_depth(ray_p2, normalize(mat3_apply(RtKinv_tgt, l2_p2)), w0, true);
    d.z = tri_depth(normalize(mat3_apply(RtKinv_src, l1_q1)), ray_q1, w0, false);
    d.w = tri_depth(normalize(mat3_apply(RtKinv_src, l1_q2)), ray_q2, w0, false);
    return d;
}

// Largest float x with sqrtf(x) <= u (u >= 0): lets the verification gate compare squared
// distances and stay bit-identical to `length(P-Q) > unc` (cudawrapper.cu:396-400).
L3D_HD float sq_threshold_walk(float u)
{
    union { float f; uint32_t i; } t;
    t.f = u * u;
    if (!(t.f == t.f) || t.f >= 3.0e38f) return t.f;
    // walk to the boundary (at most a few ulps)
    while (t.i > 0 && __builtin_sqrtf(t.f) > u) t.i -= 1;
    for (;;) {
        union { float f; uint32_t i; } n;
        n.i = t.i + 1;
        if (__builtin_sqrtf(n.f) <= u) t.i = n.i; else break;
    }
    return t.f;
}

// The same threshold without the walk.  sqrtf is the correctly rounded square root, so sqrtf(x) <= u  <=>  sqrt(x) lies
// below the midpoint u + h between u and the next float above it (a tie would need x = (u+h)^2, which has ~50
// significant bits and is not a float)  <=>  x < (u+h)^2.  u + h has 25 significant bits, so its square is exact in
// double; the threshold is the largest float strictly below it.  Equal to sq_threshold_walk for every input (tests).
L3D_HD float sq_threshold(float u)
{
    union { float f; uint32_t i; } t, up;
    t.f = u * u;
    if (!(t.f == t.f) || t.f >= 3.0e38f || !(u >= 0.0f)) return sq_threshold_walk(u);
    up.f = u;
    up.i += 1;                                        // next float above u (u >= 0, finite)
    const double m = (double)u + ((double)up.f - (double)u) * 0.5;
    const double M = m * m;
    t.f = (float)M;                                   // round to nearest ...
    if ((double)t.f >= M) t.i -= 1;                   // ... then step below (M > 0, so t.f > 0 here)
    return t.f;
}

}  // namespace l3d
