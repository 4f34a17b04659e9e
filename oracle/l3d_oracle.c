/*
 * oracle/l3d_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, one thread) of the arithmetic on Line3D's
 * matching / affinity hot path.  Every function cites the reference file:line
 * it follows (paths relative to /root/reference).  Nothing here is shipped:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.
 *
 * PARITY PIN STATUS.  Two kinds of evidence, kept apart since round 4:
 *   (A) oracle/_ref/ -- the reference COMPILED HERE: unmodified text behind an extern "C" door (clustering.cc, the texture-free device
 *       functions, the comparators of sparsematrix.h).  These are the pins.
 *   (B) oracle/_spliced/libkernels_spliced.so -- the reference's kernel text with BUILDER-WRITTEN lines where it touches CUDA textures or
 *       launch variables (the diffusion kernels, the three bodies, the two matching kernels, K_collinearity below).  Corroboration only.
 *   And, at the size bench.py times: tests/golden/config2_full.npz, this oracle alone over the whole 64 x 2000 x 12 scene.
 * (A):
 *   - graph segmentation: clustering.cc + universe.h (oracle/Makefile target `ref`);
 *   - the texture-free device functions of cudawrapper.cu (:56-61, 93-99, 116-141, 165-285, 337-344: D_distance_p2l_2D_f3,
 *     D_segment_length_2D_f3, D_angle_between_lines_deg_3D_f3, D_point_on_segment_2D_f3, D_segment_overlap_2D,
 *     D_normalize_hom_coords_2D, D_get_ray_src, D_unproject_point_src) with helper_math.h and the constants of
 *     cudawrapper.h:43-46, against the genuine NVIDIA runtime headers of the triton wheel (target `ref_devfn`,
 *     oracle/make_ref_devfn.py): bit-equal on 10^6 random and adversarial inputs per function, live and as committed
 *     golden vectors (tests/test_oracle_pins.py, tests/golden/devfn_ref.npz); the angle function bit-equal in the libm
 *     build, within 3e-5 degrees in the contract build (acosf is the one transcendental in it);
 * (B):
 *   - the two kernels of replicator_dynamics_diffusion, K_sparseMat_row_normalization and K_sparseMat_diffusion_step
 *     (cudawrapper.cu:717-829; texture-free, every thread independent), compiled the same way (their launch variables, declared by
 *     the genuine <device_launch_parameters.h>, get their storage from oracle/ref_devfn_launch.cc): l3do_rdd_hooked runs them
 *     inside this file's restatement of the host loop -- bit-equal with l3do_rdd, live and as committed vectors
 *     (tests/golden/rdd_ref.npz);
 *   - three bodies BEHIND their texture fetches -- the reference's own lines compiled inside a function of ours whose parameters stand
 *     for the fetched values (make_ref_devfn.py marks which lines are whose): D_hypothesis_confidence (:380-427 without the fetch :407 --
 *     the whole scoring of stage 2: gate, 2-D distances, 3-D angle, two expf), the middle of K_pairwise_matches (:569-588: intersection
 *     points, validity, overlaps against the thresholds -- the stage-1 decision), K_collinearity's body for one pair (:492-529): bit-equal
 *     on 10^6 cases each (the two with transcendentals in the libm build; contract build within a few 1e-6, no decision flips seen);
 *   - the two matching kernels WHOLE: K_pairwise_matches (:538-611) with D_get_triangulation_depth (:304-335), and K_verify_matches (:614-714),
 *     compiled from the reference's text with only their texture-fetch lines replaced by table reads (:551-554, :558, :591-593; :637-641) and
 *     run one thread at a time: l3do_pairwise_dense gives the same dense buffers bit for bit on real scene geometry (helix, narrow baseline,
 *     forward motion, opposing cameras; > 10^6 pairs), l3do_verify the same confidences bit for bit (libm build; contract build within 5e-6 with
 *     the same kept set) on packed candidate lists with clusters, outliers, runs of one camera (tests/golden/pairwise_ref.npz, verify_ref.npz).
 *     Their three texture-reading callees D_epipolar_line, D_get_ray_tgt, D_project_point_tgt -- 3x3 / 3x4 matrix-vector products accumulated
 *     from 0.0f in index order -- are RESTATED over tables in that build (oracle/make_ref_devfn.py says which lines are whose);
 *   - end to end: l3do_set_kernel_hooks plugs those kernels into this file's host orchestration; a whole compute3Dmodel then equals the
 *     un-hooked run bit for bit (tests/test_oracle_pins.py::test_pipeline_with_the_reference_kernels_equals_the_oracle).
 * Still "parity unpinned" by the reference, pinned by restatement, analytic known-answer scenes and committed vectors
 * only: those three matrix-vector products (K_collinearity too is compiled whole, :476-535 with its fetches as table reads, and reproduced
 * bit for bit by l3do_collinearity in the libm build),
 * the host orchestration (cudawrapper.cu:858-1191), sparsematrix.cc's index tables,
 * view.cc and line3D.cc -- they need CUDA texture references, boost, Eigen or OpenCV, which this image lacks; building
 * them would take stand-in headers, so they are treated as unbuildable.  The reference has no tests, golden vectors
 * or fixtures of its own.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no -ffast-math).
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdint.h>

#include "l3d_oracle_math.h"

/* cudawrapper.h:43-46 */
static const float EPS_G = 1e-12;
static const float COLLIN_AFF_T_G = 0.50f;
static const float MIN_OVERLAP_LOWER_T_G = 0.10f;
static const float MIN_OVERLAP_UPPER_T_G = 0.30f;
#define RDD_MAX_ITER 10 /* cudawrapper.h:35 */

typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;

/* helper_math.h:1244-1313,1420 (host definitions: rsqrtf = 1/sqrtf, helper_math.h:61-64) */
static inline f3 mk3(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 scale3(float b, f3 a) { return mk3(b * a.x, b * a.y, b * a.z); }
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float dot2(float ax, float ay, float bx, float by) { return ax * bx + ay * by; }
static inline float length3(f3 v) { return sqrtf(dot3(v, v)); }
static inline f3 normalize3(f3 v)
{
    float inv = 1.0f / sqrtf(dot3(v, v));
    return mk3(v.x * inv, v.y * inv, v.z * inv);
}
static inline f3 cross3(f3 a, f3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

/* cudawrapper.cu:58-61 */
static inline float distance_p2l_2D(f3 line, f3 p)
{
    return fabsf((line.x * p.x + line.y * p.y + line.z) / sqrtf(line.x * line.x + line.y * line.y));
}

/* cudawrapper.cu:95-99 */
static inline float segment_length_2D(f3 p1, f3 p2)
{
    f3 v = sub3(p1, p2);
    return sqrtf(v.x * v.x + v.y * v.y);
}

/* cudawrapper.cu:118-130; acos(float)/CUDART_PI*180.0f is evaluated in double
 * (CUDART_PI is a double literal) and rounded on assignment to float. */
static inline float angle_between_lines_deg_3D(f3 P1, f3 P2, f3 Q1, f3 Q2)
{
    f3 v1 = normalize3(sub3(P1, P2));
    f3 v2 = normalize3(sub3(Q1, Q2));
    float c = fmaxf(fminf(dot3(v1, v2), 1.0f), -1.0f);
    float angle = (float)((double)l3do_acosf(c) / 3.1415926535897931e+0 * (double)180.0f);
    if (angle > 90.0f)
        angle = 180.0f - angle;
    return angle;
}

/* cudawrapper.cu:135-141 */
static inline int point_on_segment_2D(f3 p1, f3 p2, f3 q)
{
    return dot2(p1.x - q.x, p1.y - q.y, p2.x - q.x, p2.y - q.y) < EPS_G;
}

/* cudawrapper.cu:144-163; F is N x 3 x 3 row-major (line3D.cc:745 stores F(r,c) at column c of row cam*3+r) */
static inline f3 epipolar_line(f3 p, const float* F, int cam, int transpose)
{
    float _p[3] = { p.x, p.y, p.z }, _l[3] = { 0.0f, 0.0f, 0.0f };
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            if (!transpose)
                _l[r] += F[cam * 9 + r * 3 + c] * _p[c];
            else
                _l[r] += F[cam * 9 + c * 3 + r] * _p[c];
        }
    return mk3(_l[0], _l[1], _l[2]);
}

/* cudawrapper.cu:209-251 (live body) */
static float segment_overlap_2D(f3 src_p1, f3 src_p2, f3 q1, f3 q2)
{
    float len_src = segment_length_2D(src_p1, src_p2);
    float len_tgt = segment_length_2D(q1, q2);

    if (len_src < 1.0f || len_tgt < 1.0f)
        return 0.0f;

    if (point_on_segment_2D(src_p1, src_p2, q1) && point_on_segment_2D(src_p1, src_p2, q2)) {
        return len_tgt / len_src;
    } else if (point_on_segment_2D(q1, q2, src_p1) && point_on_segment_2D(q1, q2, src_p2)) {
        return len_src / len_tgt;
    } else if (point_on_segment_2D(src_p1, src_p2, q1)) {
        float len1 = segment_length_2D(src_p2, q2);
        float len2 = segment_length_2D(src_p1, q2);
        if (point_on_segment_2D(q1, q2, src_p1) && len1 > EPS_G)
            return segment_length_2D(q1, src_p1) / len1;
        else if (len2 > EPS_G)
            return segment_length_2D(q1, src_p2) / len2;
    } else if (point_on_segment_2D(src_p1, src_p2, q2)) {
        float len1 = segment_length_2D(src_p1, q1);
        float len2 = segment_length_2D(src_p2, q1);
        if (point_on_segment_2D(q1, q2, src_p2) && len1 > EPS_G)
            return segment_length_2D(q2, src_p2) / len1;
        else if (len2 > EPS_G)
            return segment_length_2D(q2, src_p1) / len2;
    }
    return 0.0f;
}

/* cudawrapper.cu:255-267 */
static inline f3 normalize_hom_coords_2D(f3 p)
{
    if (fabsf(p.z) > EPS_G) {
        p.x /= p.z; p.y /= p.z; p.z /= p.z;
        p.z = 1;
        return p;
    }
    return mk3(0, 0, 0);
}

/* cudawrapper.cu:270-303: M is 3x3 row-major */
static inline f3 get_ray(f3 p, const float* M)
{
    float _p[3] = { p.x, p.y, p.z }, _ray[3] = { 0.0f, 0.0f, 0.0f };
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            _ray[r] += M[r * 3 + c] * _p[c];
    return mk3(_ray[0], _ray[1], _ray[2]);
}

/* cudawrapper.cu:306-335 */
static float triangulation_depth(f3 p1, f3 p2, f3 C1, f3 C2, const float* RtKinv2, int for_src,
                                 const float* RtKinv1)
{
    f3 ray1 = normalize3(get_ray(p1, RtKinv1));
    f3 ray2 = normalize3(get_ray(p2, RtKinv2));
    f3 w0 = sub3(C1, C2);

    float a = dot3(ray1, ray1);
    float b = dot3(ray1, ray2);
    float c = dot3(ray2, ray2);
    float d = dot3(ray1, w0);
    float e = dot3(ray2, w0);

    float denom = a * c - b * b;
    if (fabsf(denom) > EPS_G) {
        if (for_src)
            return (b * e - c * d) / denom;
        else
            return (a * e - b * d) / denom;
    }
    return -1.0f;
}

/* cudawrapper.cu:338-344 */
static inline f3 unproject_point_src(f3 p, f3 C, float depth, const float* RtKinv)
{
    f3 ray = normalize3(get_ray(p, RtKinv));
    return add3(C, scale3(depth, ray));
}

/* cudawrapper.cu:355-377: P is 3x4 row-major */
static inline f3 project_point_tgt(f3 X, const float* P)
{
    float _P[4] = { X.x, X.y, X.z, 1.0f };
    float _p[3] = { 0.0f, 0.0f, 0.0f };
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c)
            _p[r] += P[r * 4 + c] * _P[c];
    if (fabsf(_p[2]) > EPS_G)
        return mk3(_p[0] / _p[2], _p[1] / _p[2], 1.0f);
    return mk3(0, 0, 0);
}

/* cudawrapper.cu:380-427 */
static float hypothesis_confidence(f3 p1, f3 p2, f3 P1, f3 P2, f3 Q1, f3 Q2, f3 C, const float* tgtseg,
                                   float sigma_p, float sigma_a, float spatial_k)
{
    if (spatial_k > 0.0f) {
        float depth1 = length3(sub3(C, P1));
        float depth2 = length3(sub3(C, P2));
        float unc1 = spatial_k * depth1;
        float unc2 = spatial_k * depth2;
        float dist1 = length3(sub3(P1, Q1));
        float dist2 = length3(sub3(P2, Q2));
        if (dist1 > unc1 || dist2 > unc2)
            return 0.0f;
    }
    f3 line1 = cross3(p1, p2);
    f3 q1 = mk3(tgtseg[0], tgtseg[1], 1.0f);
    f3 q2 = mk3(tgtseg[2], tgtseg[3], 1.0f);
    f3 line2 = cross3(q1, q2);

    float d1 = fmaxf(distance_p2l_2D(line2, p1), distance_p2l_2D(line2, p2));
    float d2 = fmaxf(distance_p2l_2D(line1, q1), distance_p2l_2D(line1, q2));
    float dist = fmaxf(d1, d2);

    float angle = angle_between_lines_deg_3D(P1, P2, Q1, Q2);
    float sigma_sqr_a = sigma_a * sigma_a;
    float sigma_sqr_d = sigma_p * sigma_p;
    float d = l3do_expf(-dist * dist / (2.0f * sigma_sqr_d));
    return fminf(d, l3do_expf(-angle * angle / (2.0f * sigma_sqr_a)));
}

/* ------------------------------------------------------------------------- */
/* K_collinearity, cudawrapper.cu:476-535 (+ compute_collinearity :833-855:
 * sigma passed squared).  relation is dense S x S, row-major. */
/* ------------------------------------------------------------------------- */
/* Kernel hooks (tests/test_oracle_pins.py): with them set, the host orchestration below runs the reference's kernel text -- oracle/_spliced/
 * libkernels_spliced.so, assembled from cudawrapper.cu's lines with table reads for its texture fetches -- in place of this file's restatements: the whole pipeline with the reference's
 * kernels inside the restated host code must give what it gives without them, bit for bit (libm build).  NULL = restatement. */
typedef void (*l3do_hook_collin)(float* relation, int size, float coll_sigma_sqr, int stride, const float* segs);
typedef void (*l3do_hook_dense)(float* buffer, int width, int height, const float* RtKinv_src, int r_stride, int offset, int cID, const float* C_src,
                                int stride, const float* src_segs, const float* tgt_segs, const float* F, const float* RtKinv_tgt, const float* centers,
                                int y_begin, int y_end);
typedef void (*l3do_hook_verify)(float* matches_data, const float* matches_depths, const int* match_offsets, const int* camera_offsets, int size,
                                 const float* src_segs, const float* RtKinv, int r_stride, const float* C_src, const float* tgt_segs, const float* P,
                                 float sigma_p, float sigma_a, float spatial_k);
typedef void (*l3do_hook_norm)(float* data, const int* start_indices, int num_rows, int num_entries);
typedef void (*l3do_hook_step)(const float* P, const float* W, const int* P_rows, const int* W_cols, float* P_prime, const int* P_prime_rows, int num_entries);
static l3do_hook_collin g_hook_collin = 0;
static l3do_hook_dense g_hook_dense = 0;
static l3do_hook_verify g_hook_verify = 0;
static l3do_hook_norm g_hook_norm = 0;
static l3do_hook_step g_hook_step = 0;
void l3do_set_kernel_hooks(l3do_hook_collin collin, l3do_hook_dense dense, l3do_hook_verify verify, l3do_hook_norm norm, l3do_hook_step step)
{ g_hook_collin = collin; g_hook_dense = dense; g_hook_verify = verify; g_hook_norm = norm; g_hook_step = step; }

/* the kernel's body for one pair of segments, cudawrapper.cu:492-529 behind the four texture fetches (pinned to the reference's own
 * lines: tests/test_oracle_pins.py, `collinearity_pair`) */
static float collinearity_pair(f3 p1, f3 p2, f3 q1, f3 q2, float coll_sigma_sqr)
{
    float result = 0.0f;
    f3 line1 = cross3(p1, p2);
    f3 line2 = cross3(q1, q2);
    float d1 = fmaxf(distance_p2l_2D(line2, p1), distance_p2l_2D(line2, p2));
    float d2 = fmaxf(distance_p2l_2D(line1, q1), distance_p2l_2D(line1, q2));
    float d = fmaxf(d1, d2);
    float aff = l3do_expf(-d * d / (2.0f * coll_sigma_sqr));
    if (aff > COLLIN_AFF_T_G) {
        float pos1 = dot2(q1.x - p1.x, q1.y - p1.y, q2.x - p1.x, q2.y - p1.y);
        float pos2 = dot2(q1.x - p2.x, q1.y - p2.y, q2.x - p2.x, q2.y - p2.y);
        float pos3 = dot2(p1.x - q1.x, p1.y - q1.y, p2.x - q1.x, p2.y - q1.y);
        float pos4 = dot2(p1.x - q2.x, p1.y - q2.y, p2.x - q2.x, p2.y - q2.y);
        if (pos1 > -EPS_G && pos2 > -EPS_G && pos3 > -EPS_G && pos4 > -EPS_G)
            result = aff;
    }
    return result;
}
void l3do_collinearity(const float* segs, int S, float collin_s, float* relation)
{
    float coll_sigma_sqr = collin_s * collin_s;
    if (g_hook_collin) { g_hook_collin(relation, S, coll_sigma_sqr, S, segs); return; }
    for (int y = 0; y < S; ++y)
        for (int x = 0; x < S; ++x) {
            if (x == y) {
                relation[(size_t)y * S + x] = 0.0f;
            } else if (x < y) {
                f3 p1 = mk3(segs[x * 4 + 0], segs[x * 4 + 1], 1.0f);
                f3 p2 = mk3(segs[x * 4 + 2], segs[x * 4 + 3], 1.0f);
                f3 q1 = mk3(segs[y * 4 + 0], segs[y * 4 + 1], 1.0f);
                f3 q2 = mk3(segs[y * 4 + 2], segs[y * 4 + 3], 1.0f);
                float result = collinearity_pair(p1, p2, q1, q2, coll_sigma_sqr);
                relation[(size_t)y * S + x] = result;
                relation[(size_t)x * S + y] = result;
            }
        }
}

/* ------------------------------------------------------------------------- */
/* K_pairwise_matches for ONE thread (y = src segment, x = tgt segment of neighbour
 * `cam`), cudawrapper.cu:538-611.  tgt_segs is the concatenation of all neighbours'
 * segments (line3D.cc:770-781), offset = that neighbour's start. */
/* the middle of the kernel, cudawrapper.cu:569-588: the four intersection points, their validity, the two overlaps against the
 * thresholds (pinned to the reference's own lines: tests/test_oracle_pins.py, `pairwise_overlap`).  1 = potential match. */
static int pairwise_overlap(f3 p1, f3 p2, f3 q1, f3 q2, f3 line1, f3 line2, f3 epi_p1, f3 epi_p2, f3 epi_q1, f3 epi_q2,
                            f3* l2_p1, f3* l2_p2, f3* l1_q1, f3* l1_q2)
{
    *l2_p1 = normalize_hom_coords_2D(cross3(line2, epi_p1));
    *l2_p2 = normalize_hom_coords_2D(cross3(line2, epi_p2));
    *l1_q1 = normalize_hom_coords_2D(cross3(line1, epi_q1));
    *l1_q2 = normalize_hom_coords_2D(cross3(line1, epi_q2));
    if ((int)l2_p1->z == 0 || (int)l2_p2->z == 0 || (int)l1_q1->z == 0 || (int)l1_q2->z == 0)
        return 0;
    float overlap1 = segment_overlap_2D(p1, p2, *l1_q1, *l1_q2);
    float overlap2 = segment_overlap_2D(q1, q2, *l2_p1, *l2_p2);
    return fminf(overlap1, overlap2) > MIN_OVERLAP_LOWER_T_G && fmaxf(overlap1, overlap2) > MIN_OVERLAP_UPPER_T_G;
}

static f4 pairwise_one(const float* src_segs, int y, const float* RtKinv_src, f3 C_src,
                       const float* tgt_segs, int offset, int x, int cam,
                       const float* F, const float* RtKinv, const float* centers)
{
    f4 result = { 0, 0, 0, 0 };
    f3 p1 = mk3(src_segs[y * 4 + 0], src_segs[y * 4 + 1], 1.0f);
    f3 p2 = mk3(src_segs[y * 4 + 2], src_segs[y * 4 + 3], 1.0f);
    f3 line1 = cross3(p1, p2);

    const float* data = tgt_segs + (size_t)(offset + x) * 4;
    f3 q1 = mk3(data[0], data[1], 1.0f);
    f3 q2 = mk3(data[2], data[3], 1.0f);
    f3 line2 = cross3(q1, q2);

    f3 epi_p1 = epipolar_line(p1, F, cam, 0);
    f3 epi_p2 = epipolar_line(p2, F, cam, 0);
    f3 epi_q1 = epipolar_line(q1, F, cam, 1);
    f3 epi_q2 = epipolar_line(q2, F, cam, 1);

    f3 l2_p1, l2_p2, l1_q1, l1_q2;
    if (pairwise_overlap(p1, p2, q1, q2, line1, line2, epi_p1, epi_p2, epi_q1, epi_q2, &l2_p1, &l2_p2, &l1_q1, &l1_q2)) {
        f3 C_tgt = mk3(centers[cam * 3 + 0], centers[cam * 3 + 1], centers[cam * 3 + 2]);
        const float* Rk2 = RtKinv + cam * 9;
        result.x = triangulation_depth(p1, l2_p1, C_src, C_tgt, Rk2, 1, RtKinv_src);
        result.y = triangulation_depth(p2, l2_p2, C_src, C_tgt, Rk2, 1, RtKinv_src);
        result.z = triangulation_depth(l1_q1, q1, C_src, C_tgt, Rk2, 0, RtKinv_src);
        result.w = triangulation_depth(l1_q2, q2, C_src, C_tgt, Rk2, 0, RtKinv_src);
    }
    return result;
}

/* The dense S_src x width float4 buffer of one neighbour (cudawrapper.cu:915-923). */
void l3do_pairwise_dense(const float* src_segs, int S_src, const float* RtKinv_src, const float* C_src,
                         const float* tgt_segs, int offset, int width, int cam,
                         const float* F, const float* RtKinv, const float* centers, float* buffer)
{
    f3 C = mk3(C_src[0], C_src[1], C_src[2]);
    for (int y = 0; y < S_src; ++y)
        for (int x = 0; x < width; ++x) {
            f4 r = pairwise_one(src_segs, y, RtKinv_src, C, tgt_segs, offset, x, cam, F, RtKinv, centers);
            float* o = buffer + ((size_t)y * width + x) * 4;
            o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
        }
}

/* ------------------------------------------------------------------------- */
/* K_verify_matches, cudawrapper.cu:614-714.  matches_data[R][4] = (srcID, camLocal,
 * tgtID, conf) as floats; matches_depths[R][4]; match_offsets[S][2] = (start,count);
 * cam_offsets[N][2] = (start,count) into tgt_segs; P is N x 3 x 4 row-major. */
void l3do_verify(float* matches_data, const float* matches_depths, const int* match_offsets,
                 const int* cam_offsets, int R, const float* src_segs, const float* RtKinv_src,
                 const float* C_src_, const float* tgt_segs, const float* P,
                 float sigma_p, float sigma_a, float spatial_k, int y_begin, int y_end)
{
    f3 C_src = mk3(C_src_[0], C_src_[1], C_src_[2]);
    if (y_end > R) y_end = R;
    for (int y = y_begin; y < y_end; ++y) {
        const float* data = matches_data + (size_t)y * 4;
        int srcID = (int)data[0];
        int camID = (int)data[1];
        const float* depths = matches_depths + (size_t)y * 4;
        float depth_p1 = depths[0];
        float depth_p2 = depths[1];

        f3 p1 = mk3(src_segs[srcID * 4 + 0], src_segs[srcID * 4 + 1], 1.0f);
        f3 p2 = mk3(src_segs[srcID * 4 + 2], src_segs[srcID * 4 + 3], 1.0f);

        f3 P1 = unproject_point_src(p1, C_src, depth_p1, RtKinv_src);
        f3 P2 = unproject_point_src(p2, C_src, depth_p2, RtKinv_src);

        int start = match_offsets[srcID * 2 + 0];
        int end = start + match_offsets[srcID * 2 + 1];

        float confidence = 0.0f;
        int current_cam = -1;
        float current_confidence = 0.0f;

        for (int i = start; i < end; ++i) {
            if (i == y)
                continue;
            const float* data_tgt = matches_data + (size_t)i * 4;
            int camID2 = (int)data_tgt[1];
            int tgtID2 = (int)data_tgt[2];
            int camFeatureOffset = cam_offsets[camID2 * 2 + 0];

            const float* depths_tgt = matches_depths + (size_t)i * 4;
            f3 Q1 = unproject_point_src(p1, C_src, depths_tgt[0], RtKinv_src);
            f3 Q2 = unproject_point_src(p2, C_src, depths_tgt[1], RtKinv_src);

            if (camID2 == camID)
                continue;

            if (camID2 != current_cam) {
                if (current_cam != -1)
                    confidence += current_confidence;
                current_confidence = 0.0f;
                current_cam = camID2;
            }

            f3 proj1 = project_point_tgt(P1, P + camID2 * 12);
            f3 proj2 = project_point_tgt(P2, P + camID2 * 12);

            if ((int)proj1.z == 1 && (int)proj2.z == 1) {
                float conf = hypothesis_confidence(proj1, proj2, P1, P2, Q1, Q2, C_src,
                                                   tgt_segs + (size_t)(tgtID2 + camFeatureOffset) * 4,
                                                   sigma_p, sigma_a, spatial_k);
                if (conf > 0.5f) {
                    if (conf > current_confidence)
                        current_confidence = conf;
                }
            }
        }
        confidence += current_confidence;
        matches_data[(size_t)y * 4 + 3] = confidence;
    }
}

/* ------------------------------------------------------------------------- */
/* L3DMatchingPair (sparsematrix.h:37-65) without active_ (always true on this path). */
typedef struct {
    uint32_t segID1, camID2, segID2;
    float depths[4];
    float confidence;
} l3do_match;

static void stable_sort_matches(l3do_match* m, size_t n);
/* (door for tests/test_oracle_pins.py: the order against the reference's own comparator on a std::list) */
void l3do_sort_matches(l3do_match* m, int n) { stable_sort_matches(m, (size_t)(n > 0 ? n : 0)); }

/* sortMatchingPairs, sparsematrix.h:67-78 */
static int cmp_match(const void* a_, const void* b_)
{
    const l3do_match* a = (const l3do_match*)a_;
    const l3do_match* b = (const l3do_match*)b_;
    if (a->segID1 != b->segID1) return a->segID1 < b->segID1 ? -1 : 1;
    if (a->camID2 != b->camID2) return a->camID2 < b->camID2 ? -1 : 1;
    if (a->segID2 != b->segID2) return a->segID2 < b->segID2 ? -1 : 1;
    return 0;
}

static int cmp_float(const void* a, const void* b)
{
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* stable merge sort on an index permutation by the (seg,cam,tgt) key: std::list::sort is
 * stable; equal keys cannot occur on this path (a camera is either already matched or in
 * toBeMatched, never both) but stability is kept anyway. */
static void stable_sort_matches(l3do_match* m, size_t n)
{
    if (n < 2) return;
    l3do_match* tmp = (l3do_match*)malloc(n * sizeof(l3do_match));
    for (size_t w = 1; w < n; w *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * w) {
            size_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            size_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi)
                tmp[k++] = cmp_match(&m[j], &m[i]) < 0 ? m[j++] : m[i++];
            while (i < mid) tmp[k++] = m[i++];
            while (j < hi) tmp[k++] = m[j++];
        }
        memcpy(m, tmp, n * sizeof(l3do_match));
    }
    free(tmp);
}

/* compute_pairwise_matches, cudawrapper.cu:858-1128.
 *   in_matches   : the localized existing matches (camID2 = LOCAL index), line3D.cc:806
 *   toBeMatched  : local neighbour indices (line3D.cc:732-736)
 *   offsets      : N x (start,count) into tgt_segs
 *   local2global : N entries
 *   best_depths/n_best (optional): the depth pairs entering the median, in segment order, so that
 *     segment ranges can be merged exactly.
 *   seg_begin/seg_end: restrict the SOURCE segments processed (whole view: 0,S_src); used
 *     only by the bounded cpu_baseline sample and the sharding tests. Verification of a
 *     source segment only reads candidates of the same segment, so a range is exact.
 * Outputs (callee-allocated, free with l3do_free): *out_matches, *out_n; *median_depth.
 * Optional stats[4] = {raw_total (incl. existing), num_valid, verify_inner_iterations, pairs}.
 * Returns 0 on success.  With n_tbm == 0 the reference returns before touching anything
 * (877-878): the output is the input list unchanged (local camera ids, confidence 0) and
 * median_depth is left at the caller's value. */
int l3do_compute_pairwise_matches(const float* src_segs, int S_src, const float* RtKinv_src, const float* C_src,
                                  const float* tgt_segs, const int* offsets, int N,
                                  const float* F, const float* RtKinv, const float* centers, const float* P,
                                  const int* toBeMatched, int n_tbm,
                                  const l3do_match* in_matches, int n_in,
                                  const uint32_t* local2global,
                                  float k_upper, float k_lower, float sigma_p, float sigma_a, float spatial_k,
                                  int seg_begin, int seg_end,
                                  l3do_match** out_matches, int* out_n, float* median_depth, double* stats,
                                  float* best_depths /* optional, 2*S_src floats */, int* n_best /* optional */)
{
    (void)k_upper; (void)k_lower; /* only used for a verbose print in the reference (1074-1082) */
    *out_matches = NULL; *out_n = 0;
    if (n_best) *n_best = 0;
    if (n_tbm == 0) {
        l3do_match* o = (l3do_match*)malloc((n_in > 0 ? n_in : 1) * sizeof(l3do_match));
        memcpy(o, in_matches, (size_t)n_in * sizeof(l3do_match));
        *out_matches = o; *out_n = n_in;
        return 0;
    }
    if (seg_begin < 0) seg_begin = 0;
    if (seg_end > S_src) seg_end = S_src;
    f3 C = mk3(C_src[0], C_src[1], C_src[2]);

    size_t cap = (size_t)n_in + 1024, n = 0;
    l3do_match* M = (l3do_match*)malloc(cap * sizeof(l3do_match));
    for (int i = 0; i < n_in; ++i)
        if ((int)in_matches[i].segID1 >= seg_begin && (int)in_matches[i].segID1 < seg_end)
            M[n++] = in_matches[i];

    double pairs = 0;
    /* per-neighbour dense pass + scan, 900-945 */
    for (int t = 0; t < n_tbm; ++t) {
        int localID = toBeMatched[t];
        int feature_offset = offsets[localID * 2 + 0];
        int width = offsets[localID * 2 + 1];
        float* dense = NULL;
        if (g_hook_dense) {                                       /* the reference's kernel fills the S_src x width buffer (915-923; here: the rows asked for) */
            dense = (float*)calloc((size_t)S_src * (size_t)(width > 0 ? width : 1) * 4, sizeof(float));
            g_hook_dense(dense, width, S_src, RtKinv_src, 3, feature_offset, localID, C_src, width, src_segs, tgt_segs, F, RtKinv, centers, seg_begin, seg_end);
        }
        for (int i = seg_begin; i < seg_end; ++i)
            for (int j = 0; j < width; ++j) {
                f4 d;
                if (dense) { const float* q = dense + ((size_t)i * width + j) * 4; d.x = q[0]; d.y = q[1]; d.z = q[2]; d.w = q[3]; }
                else d = pairwise_one(src_segs, i, RtKinv_src, C, tgt_segs, feature_offset, j, localID, F, RtKinv, centers);
                if (d.x > 0.0f && d.y > 0.0f && d.z > 0.0f && d.w > 0.0f) {
                    if (n == cap) { cap *= 2; M = (l3do_match*)realloc(M, cap * sizeof(l3do_match)); }
                    l3do_match mp;
                    mp.segID1 = (uint32_t)i; mp.segID2 = (uint32_t)j; mp.camID2 = (uint32_t)localID;
                    mp.depths[0] = d.x; mp.depths[1] = d.y; mp.depths[2] = d.z; mp.depths[3] = d.w;
                    mp.confidence = 0.0f;
                    M[n++] = mp;
                }
            }
        free(dense);
        pairs += (double)(seg_end - seg_begin) * width;
    }

    /* sort, 951 */
    stable_sort_matches(M, n);
    if (stats) { stats[0] = (double)n; stats[3] = pairs; }
    if (n == 0) { *out_matches = M; *out_n = 0; return 0; } /* 955-956: matches stays empty */

    /* pack, 958-1003 */
    float* data = (float*)malloc(n * 4 * sizeof(float));
    float* dep = (float*)malloc(n * 4 * sizeof(float));
    int* moff = (int*)malloc((size_t)S_src * 2 * sizeof(int));
    for (int i = 0; i < S_src; ++i) { moff[2 * i] = -1; moff[2 * i + 1] = -1; }
    {
        unsigned current_seg = (unsigned)S_src, num_matches = 0, starting_pos = 0;
        for (size_t pos = 0; pos < n; ++pos) {
            if (M[pos].segID1 != current_seg) {
                if (current_seg != (unsigned)S_src) {
                    moff[2 * current_seg] = (int)starting_pos;
                    moff[2 * current_seg + 1] = (int)num_matches;
                    num_matches = 0;
                    starting_pos = (unsigned)pos;
                }
                current_seg = M[pos].segID1;
            }
            data[pos * 4 + 0] = (float)current_seg;
            data[pos * 4 + 1] = (float)M[pos].camID2;
            data[pos * 4 + 2] = (float)M[pos].segID2;
            data[pos * 4 + 3] = 0.0f;
            memcpy(dep + pos * 4, M[pos].depths, 16);
            ++num_matches;
        }
        if (current_seg < (unsigned)S_src && num_matches > 0) {
            moff[2 * current_seg] = (int)starting_pos;
            moff[2 * current_seg + 1] = (int)num_matches;
        }
    }
    if (stats) {
        double it = 0;
        for (int i = 0; i < S_src; ++i)
            if (moff[2 * i] >= 0) it += (double)moff[2 * i + 1] * (double)moff[2 * i + 1];
        stats[2] = it;
    }

    /* verify, 1013 */
    if (g_hook_verify) g_hook_verify(data, dep, moff, offsets, (int)n, src_segs, RtKinv_src, 3, C_src, tgt_segs, P, sigma_p, sigma_a, spatial_k);
    else l3do_verify(data, dep, moff, offsets, (int)n, src_segs, RtKinv_src, C_src, tgt_segs, P,
                     sigma_p, sigma_a, spatial_k, 0, (int)n);

    /* best / median, 1025-1076 */
    float* dlist = (float*)malloc(((size_t)S_src * 2 + 2) * sizeof(float));
    size_t nd = 0;
    float conf_t = 1.00f;
    unsigned num_valid = 0;
    for (int i = 0; i < S_src; ++i) {
        int start = moff[2 * i], end = start + moff[2 * i + 1];
        if (start >= 0) {
            float max_conf = 0.0f, depth_s1 = 0.0f, depth_s2 = 0.0f;
            for (int k = start; k < end; ++k) {
                float conf = data[(size_t)k * 4 + 3];
                if (conf > conf_t) ++num_valid;
                if (conf > max_conf) {
                    max_conf = conf;
                    depth_s1 = dep[(size_t)k * 4 + 0];
                    depth_s2 = dep[(size_t)k * 4 + 1];
                }
            }
            if (max_conf > conf_t / 2.0f) {
                dlist[nd++] = depth_s1;
                dlist[nd++] = depth_s2;
            }
        }
    }
    if (best_depths && n_best) { memcpy(best_depths, dlist, nd * sizeof(float)); *n_best = (int)(nd / 2); }
    *median_depth = -1.0f;
    if (nd > 0) {
        qsort(dlist, nd, sizeof(float), cmp_float);
        *median_depth = dlist[nd / 2];
    }
    if (stats) stats[1] = num_valid;

    /* filter, 1089-1110 */
    float confidence_norm = 2.0f;
    size_t nk = 0;
    l3do_match* K = (l3do_match*)malloc((n > 0 ? n : 1) * sizeof(l3do_match));
    for (size_t i = 0; i < n; ++i) {
        float conf = data[i * 4 + 3];
        if (conf > conf_t) {
            conf /= confidence_norm;
            l3do_match mp;
            mp.segID1 = (uint32_t)data[i * 4 + 0];
            unsigned locID = (unsigned)data[i * 4 + 1];
            mp.camID2 = local2global[locID];
            mp.segID2 = (uint32_t)data[i * 4 + 2];
            memcpy(mp.depths, dep + i * 4, 16);
            mp.confidence = conf;
            K[nk++] = mp;
        }
    }
    free(M); free(data); free(dep); free(moff); free(dlist);
    *out_matches = K; *out_n = (int)nk;
    return 0;
}

void l3do_free(void* p) { free(p); }

/* ------------------------------------------------------------------------- */
/* Replicator dynamics diffusion: SparseMatrix (sparsematrix.cc:63-191) + kernels
 * (cudawrapper.cu:717-829) + driver (cudawrapper.cu:1131-1191). */
typedef struct { int i, j; float w; } l3do_edge;

/* sortCLEdgesByCol / sortCLEdgesByRow, clustering.h:101-121 (stable list sort) */
static int edge_less(const l3do_edge* a, const l3do_edge* b, int by_row)
{
    if (by_row) {
        if (a->i < b->i) return 1;
        if (a->i == b->i && a->j < b->j) return 1;
        return 0;
    }
    if (a->j < b->j) return 1;
    if (a->j == b->j && a->i < b->i) return 1;
    return 0;
}

static void stable_sort_edges(l3do_edge* e, size_t n, int mode /*0 col,1 row,2 weight*/)
{
    if (n < 2) return;
    l3do_edge* tmp = (l3do_edge*)malloc(n * sizeof(l3do_edge));
    for (size_t w = 1; w < n; w *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * w) {
            size_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            size_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                int less = mode == 2 ? (e[j].w < e[i].w) : edge_less(&e[j], &e[i], mode);
                tmp[k++] = less ? e[j++] : e[i++];
            }
            while (i < mid) tmp[k++] = e[i++];
            while (j < hi) tmp[k++] = e[j++];
        }
        memcpy(e, tmp, n * sizeof(l3do_edge));
    }
    free(tmp);
}

/* (door for tests/test_oracle_pins.py: the order against the reference's own std::list::sort with its comparators) */
void l3do_sort_edges(l3do_edge* e, int n, int mode) { stable_sort_edges(e, (size_t)(n > 0 ? n : 0), mode); }

/* entries as float4 (row, col, val, 0) + start index per row/col (-1 if empty), sparsematrix.cc:99-131 */
static void build_sparse(const l3do_edge* sorted, int nnz, int n, int by_row, f4* entries, int* start)
{
    for (int i = 0; i < n; ++i) start[i] = -1;
    int current_rc = -1;
    for (int pos = 0; pos < nnz; ++pos) {
        entries[pos].x = (float)sorted[pos].i;
        entries[pos].y = (float)sorted[pos].j;
        entries[pos].z = sorted[pos].w;   /* normalization_factor = 1 on this path */
        entries[pos].w = 0.0f;
        int rc = by_row ? sorted[pos].i : sorted[pos].j;
        if (current_rc != rc) { start[rc] = pos; current_rc = rc; }
    }
}

/* K_sparseMat_row_normalization, cudawrapper.cu:717-762.  NB: start_indices[y] == -1 for an
 * empty row makes data[-1] read out of bounds in the reference; rows are never empty on this
 * path (every node has at least one edge), the oracle skips them. */
static void row_normalization(f4* data, const int* start_indices, int num_rows, int num_entries)
{
    for (int y = 0; y < num_rows; ++y) {
        int start = start_indices[y];
        if (start < 0) continue;
        float sum = 0.0f;
        int i = start;
        while (i < num_entries) {
            if ((int)data[i].x != y) break;
            sum += data[i].z;
            ++i;
        }
        if (sum < EPS_G) sum = EPS_G;
        i = start;
        while (i < num_entries) {
            if ((int)data[i].x != y) break;
            data[i].z /= sum;
            ++i;
        }
    }
}

/* K_sparseMat_diffusion_step, cudawrapper.cu:765-829: positional lock-step product */
static void diffusion_step(const f4* P, const f4* W, const int* P_rows, const int* W_cols,
                           f4* P_prime, const int* P_prime_rows, int num_entries)
{
    for (int y = 0; y < num_entries; ++y) {
        f4 data = P[y];
        int r = (int)data.y;
        int c = (int)data.x;
        float mul = 0.0f;
        int start_P = P_rows[r];
        int start_W = W_cols[c];
        if (start_P >= 0 && start_W >= 0) {
            while (start_P < num_entries && start_W < num_entries) {
                f4 d1 = P[start_P];
                f4 d2 = W[start_W];
                if ((int)d1.x != r || (int)d2.y != c) break;
                mul += (d1.z * d2.z);
                ++start_P; ++start_W;
            }
        }
        mul *= data.z;
        if (mul < EPS_G) mul = EPS_G;
        int s = P_prime_rows[r];
        int found = 0;
        while (s >= 0 && s < num_entries && !found) {
            f4 dat = P_prime[s];
            if ((int)dat.x != r) break;
            if ((int)dat.y == c) { P_prime[s].z = mul; found = 1; }
            ++s;
        }
    }
}

/* replicator_dynamics_diffusion, cudawrapper.cu:1131-1191 on the matrix built by
 * performDiffusion (line3D.cc:1258: SparseMatrix(A, n) = column-sorted).  in: edges in the
 * order of the list A; out: entries of the returned W (= P after the last swap, row-sorted),
 * as (i,j,w), nnz of them.  iters = L3D_RDD_MAX_ITER in the reference. */
/* The same with the two kernels handed in: tests/test_oracle_pins.py passes the reference's OWN K_sparseMat_row_normalization /
 * K_sparseMat_diffusion_step (oracle/_spliced/libkernels_spliced.so: compiled from cudawrapper.cu:717-829) and requires the result of
 * l3do_rdd bit for bit.  NULL = the restatements above.  float4 records as plain floats. */
typedef void (*l3do_norm_fn)(float* data, const int* start_indices, int num_rows, int num_entries);
typedef void (*l3do_step_fn)(const float* P, const float* W, const int* P_rows, const int* W_cols, float* P_prime, const int* P_prime_rows, int num_entries);
static void norm_default(float* data, const int* s, int nr, int ne) { row_normalization((f4*)data, s, nr, ne); }
static void step_default(const float* P, const float* W, const int* Pr, const int* Wc, float* Pp, const int* Ppr, int ne)
{ diffusion_step((const f4*)P, (const f4*)W, Pr, Wc, (f4*)Pp, Ppr, ne); }
void l3do_rdd_hooked(const l3do_edge* A, int nnz, int n, int iters, l3do_edge* out, l3do_norm_fn norm, l3do_step_fn step);
void l3do_rdd(const l3do_edge* A, int nnz, int n, int iters, l3do_edge* out) { l3do_rdd_hooked(A, nnz, n, iters, out, g_hook_norm, g_hook_step); }
void l3do_rdd_hooked(const l3do_edge* A, int nnz, int n, int iters, l3do_edge* out, l3do_norm_fn norm, l3do_step_fn step)
{
    if (!norm) norm = norm_default;
    if (!step) step = step_default;
    l3do_edge* col = (l3do_edge*)malloc((size_t)nnz * sizeof(l3do_edge));
    memcpy(col, A, (size_t)nnz * sizeof(l3do_edge));
    stable_sort_edges(col, nnz, 0);
    f4* W = (f4*)malloc((size_t)nnz * sizeof(f4));
    int* W_cols = (int*)malloc((size_t)n * sizeof(int));
    build_sparse(col, nnz, n, 0, W, W_cols);

    /* P = SparseMatrix(W, change_sorting=true): re-sort the col-sorted entries by row, stable (157-167) */
    stable_sort_edges(col, nnz, 1);
    f4* P = (f4*)malloc((size_t)nnz * sizeof(f4));
    int* P_rows = (int*)malloc((size_t)n * sizeof(int));
    build_sparse(col, nnz, n, 1, P, P_rows);
    /* P_prime = copy of P (1148), before normalisation */
    f4* Pp = (f4*)malloc((size_t)nnz * sizeof(f4));
    int* Pp_rows = (int*)malloc((size_t)n * sizeof(int));
    memcpy(Pp, P, (size_t)nnz * sizeof(f4));
    memcpy(Pp_rows, P_rows, (size_t)n * sizeof(int));

    norm((float*)P, P_rows, n, nnz);
    for (int it = 0; it < iters; ++it) {
        step((const float*)P, (const float*)W, P_rows, W_cols, (float*)Pp, Pp_rows, nnz);
        f4* t = P; P = Pp; Pp = t;
        int* ti = P_rows; P_rows = Pp_rows; Pp_rows = ti;
        if (it < iters - 1)
            norm((float*)P, P_rows, n, nnz);
    }
    for (int k = 0; k < nnz; ++k) {
        out[k].i = (int)P[k].x; out[k].j = (int)P[k].y; out[k].w = P[k].z;
    }
    free(col); free(W); free(W_cols); free(P); free(P_rows); free(Pp); free(Pp_rows);
}

/* ------------------------------------------------------------------------- */
/* performClustering, clustering.cc:6-47 + CLUniverse, universe.h:59-115.
 * labels[k] = find(k) after all merges, evaluated for k = 0..numNodes-1 in order
 * (processClusteredSegments, line3D.cc:1311-1314, iterates local ids ascending). */
void l3do_clustering(const l3do_edge* edges_in, int E, int numNodes, float c, int* labels)
{
    int* rank = (int*)calloc((size_t)numNodes, sizeof(int));
    int* cid = (int*)malloc((size_t)numNodes * sizeof(int));
    int* size = (int*)malloc((size_t)numNodes * sizeof(int));
    float* threshold = (float*)malloc((size_t)numNodes * sizeof(float));
    for (int i = 0; i < numNodes; ++i) { cid[i] = i; size[i] = 1; threshold[i] = c; }
    l3do_edge* e = (l3do_edge*)malloc((size_t)(E > 0 ? E : 1) * sizeof(l3do_edge));
    memcpy(e, edges_in, (size_t)E * sizeof(l3do_edge));
    stable_sort_edges(e, E, 2);
#define FIND(res, node) do { int y_ = (node); while (y_ != cid[y_]) y_ = cid[y_]; cid[(node)] = y_; (res) = y_; } while (0)
    for (int k = 0; k < E; ++k) {
        int a, b;
        FIND(a, e[k].i);
        FIND(b, e[k].j);
        if (a != b) {
            if (e[k].w <= threshold[a] && e[k].w <= threshold[b]) {
                if (rank[a] > rank[b]) { cid[b] = a; size[a] += size[b]; }
                else { cid[a] = b; size[b] += size[a]; if (rank[a] == rank[b]) rank[b]++; }
                int r; FIND(r, a); a = r;
                threshold[a] = e[k].w + c / (float)size[a];
            }
        }
    }
    for (int k = 0; k < numNodes; ++k) { int r; FIND(r, k); labels[k] = r; }
#undef FIND
    free(rank); free(cid); free(size); free(threshold); free(e);
}

/* ------------------------------------------------------------------------- */
/* View geometry in double (view.cc).  All matrices row-major. */
static void mat3_inverse(const double* m, double* inv)
{
    /* cofactor formula (Eigen's fixed-size 3x3 inverse is cofactor based) */
    double c00 = m[4] * m[8] - m[5] * m[7];
    double c01 = m[5] * m[6] - m[3] * m[8];
    double c02 = m[3] * m[7] - m[4] * m[6];
    double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    double id = 1.0 / det;
    inv[0] = c00 * id;
    inv[1] = (m[2] * m[7] - m[1] * m[8]) * id;
    inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    inv[3] = c01 * id;
    inv[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    inv[6] = c02 * id;
    inv[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}
static void mat3_mul(const double* a, const double* b, double* o)
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            o[r * 3 + c] = a[r * 3 + 0] * b[0 * 3 + c] + a[r * 3 + 1] * b[1 * 3 + c] + a[r * 3 + 2] * b[2 * 3 + c];
}
static void mat3_T(const double* a, double* o)
{
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o[r * 3 + c] = a[c * 3 + r];
}
static void mat3_vec(const double* a, const double* v, double* o)
{
    for (int r = 0; r < 3; ++r) o[r] = a[r * 3 + 0] * v[0] + a[r * 3 + 1] * v[1] + a[r * 3 + 2] * v[2];
}

/* L3DView ctor, view.cc:16-34 (derived quantities) and transform(), view.cc:243-257 (same formulas) */
void l3do_view_derive(const double* K, const double* R, const double* t,
                      double* Kinv, double* RtKinv, double* C, double* P)
{
    double Rt[9];
    mat3_inverse(K, Kinv);
    mat3_T(R, Rt);
    mat3_mul(Rt, Kinv, RtKinv);
    double nt[3] = { -1.0 * t[0], -1.0 * t[1], -1.0 * t[2] };
    mat3_vec(Rt, nt, C);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += K[r * 3 + k] * (c < 3 ? R[k * 3 + c] : t[k]);
            P[r * 4 + c] = s;
        }
}

/* defineSpatialUncertainty / specificSpatialUncertaintyK, view.cc:90-147; returns the
 * double distance, the caller stores it as float (k_upper_/k_lower_ are float members). */
double l3do_spatial_uncertainty_k(const double* RtKinv, const double* C, double pp_x, double pp_y, double dist_px)
{
    double pp[3] = { pp_x, pp_y, 1.0 }, n[3], d[3];
    mat3_vec(RtKinv, pp, n);
    double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    n[0] /= nn; n[1] /= nn; n[2] /= nn;
    double Pl[3] = { C[0] + n[0], C[1] + n[1], C[2] + n[2] };
    double pps[3] = { pp_x + dist_px, pp_y, 1.0 };
    mat3_vec(RtKinv, pps, d);
    double dn = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    d[0] /= dn; d[1] /= dn; d[2] /= dn;
    double Pn = Pl[0] * n[0] + Pl[1] * n[1] + Pl[2] * n[2];
    double nC = n[0] * C[0] + n[1] * C[1] + n[2] * C[2];
    double nd = n[0] * d[0] + n[1] * d[1] + n[2] * d[2];
    double tt = (Pn - nC) / nd;
    double Q[3] = { C[0] + tt * d[0], C[1] + tt * d[1], C[2] + tt * d[2] };
    double dx = Pl[0] - Q[0], dy = Pl[1] - Q[1], dz = Pl[2] - Q[2];
    return sqrt(dx * dx + dy * dy + dz * dz);
}

/* Line3D::fundamental, line3D.cc:1968-1993 */
void l3do_fundamental(const double* K1, const double* R1, const double* t1,
                      const double* K2, const double* R2, const double* t2, double* F)
{
    double R1t[9], R[9], Rt1[3], t[3], T[9], E[9], K2t[9], K2tinv[9], K1inv[9], tmp[9];
    mat3_T(R1, R1t);
    mat3_mul(R2, R1t, R);
    mat3_vec(R, t1, Rt1);
    for (int i = 0; i < 3; ++i) t[i] = t2[i] - Rt1[i];
    T[0] = 0.0;   T[1] = -t[2]; T[2] = t[1];
    T[3] = t[2];  T[4] = 0.0;   T[5] = -t[0];
    T[6] = -t[1]; T[7] = t[0];  T[8] = 0.0;
    mat3_mul(T, R, E);
    mat3_T(K2, K2t);
    mat3_inverse(K2t, K2tinv);
    mat3_inverse(K1, K1inv);
    mat3_mul(K2tinv, E, tmp);
    mat3_mul(tmp, K1inv, F);
}

/* L3DView::unprojectSegment, view.cc:302-342: out = P1[3], P2[3], dir[3] (double) */
void l3do_unproject_segment(const double* RtKinv, const double* C, const float* seg, float depth_p1, float depth_p2,
                            double* out)
{
    double p1[3] = { seg[0], seg[1], 1.0 }, p2[3] = { seg[2], seg[3], 1.0 }, r1[3], r2[3];
    mat3_vec(RtKinv, p1, r1);
    mat3_vec(RtKinv, p2, r2);
    double n1 = sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
    double n2 = sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
    for (int i = 0; i < 3; ++i) { r1[i] /= n1; r2[i] /= n2; }
    for (int i = 0; i < 3; ++i) {
        out[i] = C[i] + r1[i] * (double)depth_p1;
        out[3 + i] = C[i] + r2[i] * (double)depth_p2;
    }
    double d[3] = { out[3] - out[0], out[4] - out[1], out[5] - out[2] };
    double dn = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (int i = 0; i < 3; ++i) out[6 + i] = d[i] / dn;
}

/* view.cc:353-377 */
static float lower_unc(float k_lower, float median, float depth) { return depth < median ? k_lower * depth : k_lower * median; }
static float upper_unc(float k_upper, float median, float depth) { return depth < median ? k_upper * depth : k_upper * median; }
static float sigma_squared(float k_lower, float k_upper, float median, float depth)
{
    float d1 = lower_unc(k_lower, median, depth);
    float d2 = upper_unc(k_upper, median, depth);
    return -(d2 - d1) * (d2 - d1) / (2.0f * logf(0.01f));
}

/* Line3D::distance_point2line_3D, line3D.cc:1684-1691.  The expression at :1689 is `P1 + (dir * ((X - P1).transpose()) * dir)`: by C++
 * precedence (dir * v^T) * dir -- the 3x3 OUTER PRODUCT M(i,j) = dir[i]*v[j] first (Eigen evaluates a nested product into a temporary),
 * then the matrix-vector product M*dir, each row accumulated over j in index order -- not dir * (v . dir).  The two differ by an ulp of a
 * double before the cast to float.  (What stays unpinned: the order in which Eigen itself adds the three terms of a fixed-size sum differs
 * between its versions -- 3.2 unrolls x0 + (x1 + x2), 3.3 with unaligned vectorisation (x0 + x1) + x2 -- and the reference pins no version.) */
static float distance_point2line_3D(const double* P1, const double* dir, const double* X)
{
    double v[3] = { X[0] - P1[0], X[1] - P1[1], X[2] - P1[2] };
    double pr[3];
    for (int i = 0; i < 3; ++i) {
        double m0 = dir[i] * v[0], m1 = dir[i] * v[1], m2 = dir[i] * v[2];         /* row i of dir * v^T */
        pr[i] = P1[i] + ((m0 * dir[0] + m1 * dir[1]) + m2 * dir[2]);
    }
    double d[3] = { pr[0] - X[0], pr[1] - X[1], pr[2] - X[2] };
    return (float)sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
}

/* Line3D::similarity_coll3D, line3D.cc:1600-1681.
 * seg = {P1[3], P2[3], dir[3]} doubles; depths = {depth_p1, depth_p2};
 * cam = {k_lower, k_upper, median_depth} of the segment's view. */
float l3do_similarity_coll3D(const double* seg1, const float* depths1, const float* cam1,
                             const double* seg2, const float* depths2, const float* cam2, float sigma_a)
{
    float d1 = distance_point2line_3D(seg2, seg2 + 6, seg1);
    float d2 = distance_point2line_3D(seg2, seg2 + 6, seg1 + 3);
    float min_d1 = lower_unc(cam1[0], cam1[2], depths1[0]);
    float min_d2 = lower_unc(cam1[0], cam1[2], depths1[1]);
    float sigma_sqr_d1 = sigma_squared(cam1[0], cam1[1], cam1[2], depths1[0]);
    float sigma_sqr_d2 = sigma_squared(cam1[0], cam1[1], cam1[2], depths1[1]);

    float sim1, sim2;
    if (d1 < min_d1) sim1 = 1.0f;
    else sim1 = l3do_expf(-(d1 - min_d1) * (d1 - min_d1) / (2.0f * sigma_sqr_d1));
    if (d2 < min_d2) sim2 = 1.0f;
    else sim2 = l3do_expf(-(d2 - min_d2) * (d2 - min_d2) / (2.0f * sigma_sqr_d2));
    float w_d12 = fminf(sim1, sim2);

    float d3 = distance_point2line_3D(seg1, seg1 + 6, seg2);
    float d4 = distance_point2line_3D(seg1, seg1 + 6, seg2 + 3);
    float min_d3 = lower_unc(cam2[0], cam2[2], depths2[0]);
    float min_d4 = lower_unc(cam2[0], cam2[2], depths2[1]);
    float sigma_sqr_d3 = sigma_squared(cam2[0], cam2[1], cam2[2], depths2[0]);
    float sigma_sqr_d4 = sigma_squared(cam2[0], cam2[1], cam2[2], depths2[1]);

    float sim3, sim4;
    if (d3 < min_d3) sim3 = 1.0f;
    else sim3 = l3do_expf(-(d3 - min_d3) * (d3 - min_d3) / (2.0f * sigma_sqr_d3));
    if (d4 < min_d4) sim4 = 1.0f;
    else sim4 = l3do_expf(-(d4 - min_d4) * (d4 - min_d4) / (2.0f * sigma_sqr_d4));
    float w_d34 = fminf(sim3, sim4);
    float w_d = fminf(w_d12, w_d34);

    double dd = seg1[6] * seg2[6] + seg1[7] * seg2[7] + seg1[8] * seg2[8];
    float angle = (float)(l3do_acos(fmax(fmin(dd, 1.0), -1.0)) / 3.14159265358979323846 /* M_PI */ * (double)180.0f);
    if (angle > 90.0f)
        angle = 180.0f - angle;
    float w_a = l3do_expf(-angle * angle / (2.0f * sigma_a * sigma_a));

    float sim = fminf(w_d, w_a);
    if (sim <= 0.01f)
        return 0.0f;
    return sim;
}

/* contract math exported for tests */
float l3do_test_expf(float x) { return l3do_expf(x); }
float l3do_test_acosf(float x) { return l3do_acosf(x); }
double l3do_test_acos(double x) { return l3do_acos(x); }

/* ---- doors in front of the restated device functions (same signatures as oracle/ref_devfn_door.cc, which fronts the
 * reference's own functions compiled from /root/reference): tests/test_oracle_pins.py fires the same inputs through both.
 * Points are xyz triples, n items. */
static inline f3 ld3(const float* p, int i) { return mk3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
static inline void st3(float* p, int i, f3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }
void l3do_devfn_distance_p2l_2D(int n, const float* line, const float* p, float* out)
{ for (int i = 0; i < n; ++i) out[i] = distance_p2l_2D(ld3(line, i), ld3(p, i)); }
void l3do_devfn_segment_length_2D(int n, const float* p1, const float* p2, float* out)
{ for (int i = 0; i < n; ++i) out[i] = segment_length_2D(ld3(p1, i), ld3(p2, i)); }
void l3do_devfn_angle_between_lines_deg_3D(int n, const float* P1, const float* P2, const float* Q1, const float* Q2, float* out)
{ for (int i = 0; i < n; ++i) out[i] = angle_between_lines_deg_3D(ld3(P1, i), ld3(P2, i), ld3(Q1, i), ld3(Q2, i)); }
void l3do_devfn_point_on_segment_2D(int n, const float* p1, const float* p2, const float* q, int* out)
{ for (int i = 0; i < n; ++i) out[i] = point_on_segment_2D(ld3(p1, i), ld3(p2, i), ld3(q, i)) ? 1 : 0; }
void l3do_devfn_segment_overlap_2D(int n, const float* sp1, const float* sp2, const float* q1, const float* q2, float* out)
{ for (int i = 0; i < n; ++i) out[i] = segment_overlap_2D(ld3(sp1, i), ld3(sp2, i), ld3(q1, i), ld3(q2, i)); }
void l3do_devfn_normalize_hom_coords_2D(int n, const float* p, float* out)
{ for (int i = 0; i < n; ++i) st3(out, i, normalize_hom_coords_2D(ld3(p, i))); }
/* the restatement keeps matrices dense (stride 3); a padded row stride is repacked here, the arithmetic is get_ray's */
void l3do_devfn_get_ray_src(int n, const float* p, const float* RtKinv, int stride, float* out)
{
    for (int i = 0; i < n; ++i) {
        float M[9];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M[r * 3 + c] = RtKinv[(size_t)i * 3 * stride + r * stride + c];
        st3(out, i, get_ray(ld3(p, i), M));
    }
}
void l3do_devfn_unproject_point_src(int n, const float* p, const float* C, const float* depth, const float* RtKinv, int stride, float* out)
{
    for (int i = 0; i < n; ++i) {
        float M[9];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M[r * 3 + c] = RtKinv[(size_t)i * 3 * stride + r * stride + c];
        st3(out, i, unproject_point_src(ld3(p, i), ld3(C, i), depth[i], M));
    }
}
void l3do_devfn_collinearity_pair(int n, const float* p1, const float* p2, const float* q1, const float* q2, const float* sigma_sqr, float* out)
{ for (int i = 0; i < n; ++i) out[i] = collinearity_pair(ld3(p1, i), ld3(p2, i), ld3(q1, i), ld3(q2, i), sigma_sqr[i]); }
void l3do_devfn_hypothesis_confidence(int n, const float* p1, const float* p2, const float* P1, const float* P2, const float* Q1, const float* Q2, const float* Cc,
                                      const float* tgt, const float* par, float* out)
{
    for (int i = 0; i < n; ++i)
        out[i] = hypothesis_confidence(ld3(p1, i), ld3(p2, i), ld3(P1, i), ld3(P2, i), ld3(Q1, i), ld3(Q2, i), ld3(Cc, i), tgt + 4 * i, par[3 * i], par[3 * i + 1], par[3 * i + 2]);
}
/* out: 13 floats per item -- 1/0, then l2_p1, l2_p2, l1_q1, l1_q2 (zeros when it is no potential match) */
void l3do_devfn_pairwise_overlap(int n, const float* p1, const float* p2, const float* q1, const float* q2, const float* e1, const float* e2,
                                 const float* e3, const float* e4, float* out)
{
    for (int i = 0; i < n; ++i) {
        f3 a = ld3(p1, i), b = ld3(p2, i), c = ld3(q1, i), d = ld3(q2, i), r[4];
        float* o = out + 13 * (size_t)i;
        for (int k = 0; k < 13; ++k) o[k] = 0.0f;
        if (pairwise_overlap(a, b, c, d, cross3(a, b), cross3(c, d), ld3(e1, i), ld3(e2, i), ld3(e3, i), ld3(e4, i), &r[0], &r[1], &r[2], &r[3])) {
            o[0] = 1.0f;
            for (int k = 0; k < 4; ++k) { o[1 + 3 * k] = r[k].x; o[2 + 3 * k] = r[k].y; o[3 + 3 * k] = r[k].z; }
        }
    }
}
void l3do_devfn_normalize3(int n, const float* v, float* out) { for (int i = 0; i < n; ++i) st3(out, i, normalize3(ld3(v, i))); }
void l3do_devfn_cross3(int n, const float* a, const float* b, float* out) { for (int i = 0; i < n; ++i) st3(out, i, cross3(ld3(a, i), ld3(b, i))); }
void l3do_devfn_length3(int n, const float* v, float* out) { for (int i = 0; i < n; ++i) out[i] = length3(ld3(v, i)); }
void l3do_devfn_dot3(int n, const float* a, const float* b, float* out) { for (int i = 0; i < n; ++i) out[i] = dot3(ld3(a, i), ld3(b, i)); }
float l3do_devfn_eps_g(void) { return EPS_G; }
float l3do_devfn_min_overlap_lower(void) { return MIN_OVERLAP_LOWER_T_G; }
float l3do_devfn_min_overlap_upper(void) { return MIN_OVERLAP_UPPER_T_G; }
float l3do_devfn_collin_aff_t(void) { return COLLIN_AFF_T_G; }
