// oracle/ref_spliced_door.cc -- TEST INFRASTRUCTURE, CORROBORATION ONLY.  The extern "C" door in front of the reference's KERNELS as
// oracle/make_ref_devfn.py can build them here: the reference's own lines (line ranges of cudawrapper.cu, read where they lie), with
//   * every texture fetch replaced by a builder-written table read (make_ref_devfn.py lists each replaced line),
//   * the three texture-reading callees D_epipolar_line, D_get_ray_tgt, D_project_point_tgt RESTATED over tables,
//   * the launch variables threadIdx / blockIdx / blockDim (declared extern by the genuine <device_launch_parameters.h>) given
//     storage by ref_devfn_launch.cc, one "thread" at a time,
//   * three kernel bodies wrapped into functions whose parameters stand for the fetched values.
// That is NOT "the reference compiled here": it goes to oracle/_spliced/libkernels_spliced.so, apart from oracle/_ref/ (which holds
// only unmodified reference text behind an extern "C" door).  It corroborates the oracle's restatement of the kernels' control
// flow and arithmetic; the pins proper are the device functions of oracle/_ref/libdevfn_ref.so and clustering.cc.
extern "C" {

void l3dref_set_launch(unsigned block_x, unsigned block_y, unsigned thread_x, unsigned thread_y, unsigned dim_x, unsigned dim_y);   // ref_devfn_launch.cc

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


}  // extern "C"
