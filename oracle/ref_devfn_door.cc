// oracle/ref_devfn_door.cc -- TEST INFRASTRUCTURE.  The extern "C" door in front of the reference's own texture-free
// device functions.  oracle/make_ref_devfn.py assembles, at build time and in memory, a translation unit from
//   <cuda_runtime.h>, <math_constants.h>      the genuine NVIDIA headers that ship inside this image's triton wheel
//   /root/reference/helper_math.h             unmodified, found through -I
//   /root/reference/cudawrapper.h:43-46       the four device constants
//   /root/reference/cudawrapper.cu:56-61,93-99,116-141,165-285,337-344   inside namespace L3D, as they are
//   /root/reference/sparsematrix.h:36-49,67-85   L3DMatchingPair's fields and its two comparators (the boost members :51-65 left out)
// followed by this file, and pipes it to g++ (-O2 -ffp-contract=off) -> oracle/_ref/libdevfn_ref.so.  Nothing of the
// reference is written to disk and no stand-in header is involved: this library holds ONLY unmodified reference text behind
// this door.  The kernels and the texture-reading functions cannot be built this way; what can be said about them with
// builder-written splices is a separate library, oracle/_spliced/libkernels_spliced.so (ref_spliced_door.cc) -- corroboration,
// not the reference compiled here.
//
// Every function takes n items; points are xyz triples.
#include <type_traits>

#include <list>

extern "C" {

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

// Which overloads does a HOST compiler pick for `acos(fmax(fmin(float, 1.0f), -1.0f))` (cudawrapper.cu:124)?  4 = float
// (acosf: what nvcc's device code uses as well), 8 = double.  The numeric contract follows the float reading.
int l3dref_sizeof_angle_acos(void) { return (int)sizeof(decltype(acos(fmax(fmin(1.0f, 1.0f), -1.0f)))); }
float l3dref_eps_g(void) { return L3D::L3D_EPS_G; }
float l3dref_min_overlap_lower(void) { return L3D::L3D_MIN_OVERLAP_LOWER_T_G; }
float l3dref_min_overlap_upper(void) { return L3D::L3D_MIN_OVERLAP_UPPER_T_G; }
float l3dref_collin_aff_t(void) { return L3D::L3D_COLLIN_AFF_T_G; }

}  // extern "C"
