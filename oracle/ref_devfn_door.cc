// oracle/ref_devfn_door.cc -- TEST INFRASTRUCTURE.  The extern "C" door in front of the reference's own texture-free
// device functions.  oracle/make_ref_devfn.py assembles, at build time and in memory, a translation unit from
//   <cuda_runtime.h>, <math_constants.h>      the genuine NVIDIA headers that ship inside this image's triton wheel
//   /root/reference/helper_math.h             unmodified, found through -I
//   /root/reference/cudawrapper.h:43-46       the four device constants
//   /root/reference/cudawrapper.cu:56-61,93-99,116-141,165-285,337-344   inside namespace L3D, as they are
// followed by this file, and pipes it to g++ (-O2 -ffp-contract=off) -> oracle/_ref/libdevfn_ref.so.  Nothing of the
// reference is written to disk and no stand-in header is involved.  The texture-reading functions (D_epipolar_line,
// D_get_ray_tgt, D_get_triangulation_depth, D_project_point_tgt, D_hypothesis_confidence) and the kernels cannot be
// built this way and stay pinned by restatement only.
//
// Every function takes n items; points are xyz triples.
#include <type_traits>

#include <list>

extern "C" {

void l3dref_set_launch(unsigned block_x, unsigned block_y, unsigned thread_x, unsigned thread_y, unsigned dim_x, unsigned dim_y);   // ref_devfn_launch.cc
static inline float3 ld3(const float* p, int i) { return make_float3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
static inline void st3(float* p, int i, float3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }

// cudawrapper.cu:58-61
void l3dref_distance_p2l_2D(int n, const float* line, const float* p, float* out)
{ for (int i = 0; i < n; ++i) out[i] = L3D::D_distance_p2l_2D_f3(ld3(line, i), ld3(p, i)); }
// cudawrapper.cu:95-99
void l3dref_segment_length_2D(int n, const float* p1, const float* p2, float* out)
{ for (int i = 0; i < n; ++i) out[i] = L3D::D_segment_length_2D_f3(ld3(p1, i), ld3(p2, i)); }
// cudawrapper.cu:118-130
void l3dref_angle_between_lines_deg_3D(int n, const float* P1, const float* P2, const float* Q1, const float* Q2, float* out)
{ for (int i = 0; i < n; ++i) out[i] = L3D::D_angle_between_lines_deg_3D_f3(ld3(P1, i), ld3(P2, i), ld3(Q1, i), ld3(Q2, i)); }
// cudawrapper.cu:135-141
void l3dref_point_on_segment_2D(int n, const float* p1, const float* p2, const float* q, int* out)
{ for (int i = 0; i < n; ++i) out[i] = L3D::D_point_on_segment_2D_f3(ld3(p1, i), ld3(p2, i), ld3(q, i)) ? 1 : 0; }
// cudawrapper.cu:166-252 (live body 209-251)
void l3dref_segment_overlap_2D(int n, const float* sp1, const float* sp2, const float* q1, const float* q2, float* out)
{ for (int i = 0; i < n; ++i) out[i] = L3D::D_segment_overlap_2D(ld3(sp1, i), ld3(sp2, i), ld3(q1, i), ld3(q2, i)); }
// cudawrapper.cu:255-267
void l3dref_normalize_hom_coords_2D(int n, const float* p, float* out)
{ for (int i = 0; i < n; ++i) st3(out, i, L3D::D_normalize_hom_coords_2D(ld3(p, i))); }
// cudawrapper.cu:270-285; RtKinv: one 3x3 per item, rows `stride` floats apart (the reference passes DataArray strides)
void l3dref_get_ray_src(int n, const float* p, const float* RtKinv, int stride, float* out)
{ for (int i = 0; i < n; ++i) st3(out, i, L3D::D_get_ray_src(ld3(p, i), RtKinv + (size_t)i * 3 * stride, stride)); }
// cudawrapper.cu:338-344
void l3dref_unproject_point_src(int n, const float* p, const float* C, const float* depth, const float* RtKinv, int stride, float* out)
{ for (int i = 0; i < n; ++i) st3(out, i, L3D::D_unproject_point_src(ld3(p, i), ld3(C, i), depth[i], RtKinv + (size_t)i * 3 * stride, stride)); }
// cudawrapper.cu:492-529: K_collinearity's body for one pair of segments (make_ref_devfn.py wraps the reference's lines into
// l3dref_collinearity_body; the points stand for the kernel's texture fetches)
void l3dref_collinearity_pair(int n, const float* p1, const float* p2, const float* q1, const float* q2, const float* sigma_sqr, float* out)
{ for (int i = 0; i < n; ++i) out[i] = L3D::l3dref_collinearity_body(ld3(p1, i), ld3(p2, i), ld3(q1, i), ld3(q2, i), sigma_sqr[i]); }
// cudawrapper.cu:380-427 without :407: D_hypothesis_confidence with the target segment handed in instead of fetched (tgt: 4 floats per item;
// par: sigma_p, sigma_a, spatial_k per item)
void l3dref_hypothesis_confidence(int n, const float* p1, const float* p2, const float* P1, const float* P2, const float* Q1, const float* Q2, const float* Cc,
                                  const float* tgt, const float* par, float* out)
{
    for (int i = 0; i < n; ++i)
        out[i] = L3D::l3dref_hypothesis_confidence_body(ld3(p1, i), ld3(p2, i), ld3(P1, i), ld3(P2, i), ld3(Q1, i), ld3(Q2, i), ld3(Cc, i),
                                                        make_float4(tgt[4 * i], tgt[4 * i + 1], tgt[4 * i + 2], tgt[4 * i + 3]), par[3 * i], par[3 * i + 1], par[3 * i + 2]);
}
// cudawrapper.cu:569-588 (K_pairwise_matches between the epipolar lines and the triangulation); out: 13 floats per item -- 1/0, then l2_p1, l2_p2,
// l1_q1, l1_q2 (zeros when it is no potential match)
void l3dref_pairwise_overlap(int n, const float* p1, const float* p2, const float* q1, const float* q2, const float* e1, const float* e2,
                             const float* e3, const float* e4, float* out)
{
    for (int i = 0; i < n; ++i)
        L3D::l3dref_pairwise_overlap_body(ld3(p1, i), ld3(p2, i), ld3(q1, i), ld3(q2, i), ld3(e1, i), ld3(e2, i), ld3(e3, i), ld3(e4, i), out + 13 * (size_t)i);
}
// cudawrapper.cu:614-714: K_verify_matches, one thread at a time over the R candidates (make_ref_devfn.py: the kernel's text except its texture
// fetch of the source segment; its two texture-reading callees are bound to tables).  matches_data: R x (srcID, camera, tgtID, confidence) as floats
// -- the confidences are written in place; matches_depths: R x 4; match_offsets: S x (start, count); camera_offsets: N x (start, count) into
// tgt_segs; P: N x 3 x 4 row-major; RtKinv: 3 rows of r_stride floats.
void l3dref_verify_matches(float* matches_data, const float* matches_depths, const int* match_offsets, const int* camera_offsets, int size,
                           const float* src_segs, const float* RtKinv, int r_stride, const float* C_src, const float* tgt_segs, const float* P,
                           float sigma_p, float sigma_a, float spatial_k)
{
    L3D::l3dref_tab_src = src_segs; L3D::l3dref_tab_tgt = tgt_segs; L3D::l3dref_tab_P = P;
    for (int y = 0; y < size; ++y) {
        l3dref_set_launch(0, (unsigned)(y / 256), 0, (unsigned)(y % 256), 1, 256);                 // dimBlock = (1, 16 * 16), cudawrapper.cu:1011
        L3D::K_verify_matches(reinterpret_cast<float4*>(matches_data), reinterpret_cast<float4*>(const_cast<float*>(matches_depths)),
                              reinterpret_cast<const int2*>(match_offsets), reinterpret_cast<const int2*>(camera_offsets), size, RtKinv,
                              make_float3(C_src[0], C_src[1], C_src[2]), sigma_p, sigma_a, spatial_k, r_stride);
    }
}
// cudawrapper.cu:538-611: K_pairwise_matches for one neighbour camera, one thread at a time over the height x width grid (make_ref_devfn.py: the
// kernel's text except its texture fetches; D_epipolar_line / D_get_ray_tgt bound to tables).  buffer: height x stride float4.
// [y_begin, y_end): the rows (source segments) to run -- the kernel's own bound is `height`
void l3dref_pairwise_matches(float* buffer, int width, int height, const float* RtKinv_src, int r_stride, int offset, int cID, const float* C_src, int stride,
                             const float* src_segs, const float* tgt_segs, const float* F, const float* RtKinv_tgt, const float* centers, int y_begin, int y_end)
{
    L3D::l3dref_tab_src = src_segs; L3D::l3dref_tab_tgt = tgt_segs; L3D::l3dref_tab_F = F; L3D::l3dref_tab_R = RtKinv_tgt; L3D::l3dref_tab_C = centers;
    for (int y = y_begin < 0 ? 0 : y_begin; y < (y_end < height ? y_end : height); ++y)
        for (int x = 0; x < width; ++x) {
            l3dref_set_launch((unsigned)(x / 16), (unsigned)(y / 16), (unsigned)(x % 16), (unsigned)(y % 16), 16, 16);     // dimBlock = (16, 16), cudawrapper.cu:900
            L3D::K_pairwise_matches(reinterpret_cast<float4*>(buffer), width, height, RtKinv_src, offset, cID, make_float3(C_src[0], C_src[1], C_src[2]), stride, r_stride);
        }
}
// cudawrapper.cu:476-535: K_collinearity over the size x size grid, one thread at a time (the kernel's text except its texture fetches);
// relation: size x stride floats
void l3dref_collinearity(float* relation, int size, float coll_sigma_sqr, int stride, const float* segs)
{
    L3D::l3dref_tab_src = segs;
    for (int y = 0; y < size; ++y)
        for (int x = 0; x < size; ++x) {
            l3dref_set_launch((unsigned)(x / 16), (unsigned)(y / 16), (unsigned)(x % 16), (unsigned)(y % 16), 16, 16);     // dimBlock = (16, 16), cudawrapper.cu:842
            L3D::K_collinearity(relation, size, coll_sigma_sqr, stride);
        }
}
// sparsematrix.h:68-85 on a std::list, as compute_pairwise_matches (cudawrapper.cu:951: matches.sort(sortMatchingPairs)) and
// L3DView::addMatches (view.cc:170: sortMatchingPairsByConf) use them.  perm[k] = input index of the k-th element after the sort.
void l3dref_sort_matching_pairs(int n, const unsigned* seg1, const unsigned* cam2, const unsigned* seg2, const float* conf, int by_conf, int* perm)
{
    std::list<L3D::L3DMatchingPair> lst;
    for (int i = 0; i < n; ++i) {
        L3D::L3DMatchingPair mp;
        mp.segID1_ = seg1[i]; mp.camID2_ = cam2[i]; mp.segID2_ = seg2[i]; mp.confidence_ = conf[i]; mp.active_ = true;
        mp.depths_ = make_float4((float)i, 0.0f, 0.0f, 0.0f);                      // (the input index rides along)
        lst.push_back(mp);
    }
    if (by_conf) lst.sort(L3D::sortMatchingPairsByConf);
    else lst.sort(L3D::sortMatchingPairs);
    int k = 0;
    for (std::list<L3D::L3DMatchingPair>::const_iterator it = lst.begin(); it != lst.end(); ++it) perm[k++] = (int)it->depths_.x;
}
// helper_math.h (host definitions): normalize / cross / length / dot of float3 as the functions above see them
void l3dref_normalize3(int n, const float* v, float* out) { for (int i = 0; i < n; ++i) st3(out, i, normalize(ld3(v, i))); }
void l3dref_cross3(int n, const float* a, const float* b, float* out) { for (int i = 0; i < n; ++i) st3(out, i, cross(ld3(a, i), ld3(b, i))); }
void l3dref_length3(int n, const float* v, float* out) { for (int i = 0; i < n; ++i) out[i] = length(ld3(v, i)); }
void l3dref_dot3(int n, const float* a, const float* b, float* out) { for (int i = 0; i < n; ++i) out[i] = dot(ld3(a, i), ld3(b, i)); }

// cudawrapper.cu:717-762 and :765-829 -- the two kernels of replicator_dynamics_diffusion (texture-free; every thread is independent:
// no shared memory, no barrier), run one "thread" at a time over the grid the reference launches (x = 0, y = row / entry; the launch
// variables: ref_devfn_launch.cc).  data / P / W / P_prime: float4 records (row, column, value, unused) as SparseMatrix keeps them.
void l3dref_sparse_row_normalization(float* data, const int* start_indices, int num_rows, int num_entries)
{
    for (int y = 0; y < num_rows; ++y) {
        l3dref_set_launch(0, (unsigned)(y / 256), 0, (unsigned)(y % 256), 1, 256);     // dimBlock = (1, 16 * 16), cudawrapper.cu:1139
        L3D::K_sparseMat_row_normalization(reinterpret_cast<float4*>(data), start_indices, num_rows, num_entries);
    }
}
void l3dref_sparse_diffusion_step(const float* P, const float* W, const int* P_rows, const int* W_cols, float* P_prime, const int* P_prime_rows, int num_entries)
{
    for (int y = 0; y < num_entries; ++y) {
        l3dref_set_launch(0, (unsigned)(y / 256), 0, (unsigned)(y % 256), 1, 256);
        L3D::K_sparseMat_diffusion_step(reinterpret_cast<const float4*>(P), reinterpret_cast<const float4*>(W), P_rows, W_cols,
                                        reinterpret_cast<float4*>(P_prime), P_prime_rows, num_entries);
    }
}

// Which overloads does a HOST compiler pick for `acos(fmax(fmin(float, 1.0f), -1.0f))` (cudawrapper.cu:124)?  4 = float
// (acosf: what nvcc's device code uses as well), 8 = double.  The numeric contract follows the float reading.
int l3dref_sizeof_angle_acos(void) { return (int)sizeof(decltype(acos(fmax(fmin(1.0f, 1.0f), -1.0f)))); }
float l3dref_eps_g(void) { return L3D::L3D_EPS_G; }
float l3dref_min_overlap_lower(void) { return L3D::L3D_MIN_OVERLAP_LOWER_T_G; }
float l3dref_min_overlap_upper(void) { return L3D::L3D_MIN_OVERLAP_UPPER_T_G; }
float l3dref_collin_aff_t(void) { return L3D::L3D_COLLIN_AFF_T_G; }

}  // extern "C"
