/*
 * line3d_amd.h -- C ABI of the MI355X-native Line3D matching / affinity hot path.
 *
 * Drop-in boundary: these entry points are what a Line3D build binds instead of the three
 * functions of the reference's device seam (cudawrapper.h:49-75) -- plain pointers and sizes,
 * no C++ containers, no torch types.  INTEGRATION.md shows the ~80-line cudawrapper
 * replacement that forwards the reference's DataArray/std::list arguments to these calls so
 * that line3D.cc / segments.h link unchanged.
 *
 * Conventions: every function returns L3D_OK (0) or an error code; l3d_last_error(ctx) gives
 * the message (the reference prints CUDA errors and carries on, dataArray.h:153-156; here they
 * are reported).  All matrices are row-major float32 unless stated.  Output arrays marked
 * "callee-allocated" are released with l3d_free().  A context owns one GPU, one HIP stream and
 * grow-only device arenas; it is single-caller like the reference (global texture references,
 * cudawrapper.cu:5-10) but several contexts may coexist.
 */
#ifndef LINE3D_AMD_H
#define LINE3D_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define L3D_OK 0
#define L3D_ERR_INVALID 1
#define L3D_ERR_HIP 2
#define L3D_ERR_NOMEM 3
#define L3D_ERR_NODEVICE 4
#define L3D_ERR_UNSUPPORTED 5

/* cudawrapper.h:35,43-46 / commons.h:42-66 */
#define L3D_RDD_MAX_ITER 10
#define L3D_DEF_COLLINEARITY_S 2.0f

typedef struct l3d_ctx l3d_ctx;

/* L3DMatchingPair, sparsematrix.h:37-65 (active_ is always true on this path and is dropped) */
typedef struct l3d_match {
    uint32_t segID1;   /* segment in the source view */
    uint32_t camID2;   /* neighbour: LOCAL index on input, GLOBAL view id on output (cudawrapper.cu:1103) */
    uint32_t segID2;   /* segment in the neighbour */
    float depths[4];   /* src p1, src p2, tgt q1, tgt q2 (cudawrapper.cu:594-601) */
    float confidence;
} l3d_match;

/* CLEdge, clustering.h:57-61 */
typedef struct l3d_edge {
    int32_t i, j;
    float w;
} l3d_edge;

/* L3DSegment3D (commons.h:69-78) + the three per-view scalars similarity_coll3D reads
 * (view.cc:353-377): one hypothesis handed to l3d_similarity_coll3D_batch */
typedef struct l3d_hypothesis {
    double P1[3], P2[3], dir[3];
    float depth_p1, depth_p2;
    float k_lower, k_upper, median_depth;
    uint32_t pad;
} l3d_hypothesis;

int l3d_ctx_create(int device, l3d_ctx** ctx);
void l3d_ctx_destroy(l3d_ctx* ctx);
const char* l3d_last_error(const l3d_ctx* ctx);
void l3d_free(void* p);

/* Replaces compute_collinearity (cudawrapper.h:49-51, K_collinearity cudawrapper.cu:476-535)
 * together with the host scan of the dense S x S relation in the L3DSegments constructor
 * (segments.h:73-98): returns the non-zero upper-triangle entries (i < j, ascending (i,j)),
 * i.e. exactly what the constructor inserts into segment2collinearities_ (both directions).
 * segments: n_segments x 4 (p1x,p1y,p2x,p2y).  Outputs callee-allocated. */
int l3d_compute_collinearity(l3d_ctx* ctx, const float* segments, int n_segments, float collin_s,
                             int32_t** out_i, int32_t** out_j, float** out_w, int* out_n);
/* The same for several segment sets at once (all views of a scene): no host round trip per set.  The triplets of set v are
 * entries [set_start[v], set_start[v+1]) of the concatenated outputs (callee-allocated; set_start: n_sets + 1, caller's). */
int l3d_compute_collinearity_batch(l3d_ctx* ctx, const float* const* segments, const int* n_segments, int n_sets, float collin_s,
                                   int32_t** out_i, int32_t** out_j, float** out_w, int* set_start);

/* Replaces compute_pairwise_matches (cudawrapper.h:54-70, cudawrapper.cu:858-1128):
 * K_pairwise_matches for every neighbour in to_be_matched, selection of pairs with four
 * positive depths, (segment, camera, target) ordering, K_verify_matches over the union with the
 * already existing (reverse) matches, best-hypothesis median depth, and the conf > 1 filter.
 *
 *   src_segs      S_src x 4            source view segments            (tex_segments)
 *   RtKinv_src    3 x 3, C_src 3       source camera                   (line3D.cc:787-803)
 *   tgt_segs      sum(S_n) x 4         all neighbours' segments        (tex_segments_f4, line3D.cc:770-784)
 *   offsets       N x (start,count)    into tgt_segs                   (line3D.cc:779)
 *   F, RtKinv     N x 3 x 3; centers N x 3; P N x 3 x 4              (line3D.cc:739-763)
 *   to_be_matched n_tbm local neighbour indices                        (line3D.cc:732-736)
 *   in_matches    existing matches, camID2 = LOCAL index               (view.cc:200-224)
 *   local2global  N global view ids                                    (line3D.cc:729)
 *   seg_begin/seg_end  source-segment range to process ([0,S_src) for the whole view); the
 *                 verification of a source segment only reads candidates of that segment, so a
 *                 range is exact -- this is what the multi-GPU sharding uses.
 * Outputs: *out_matches (callee-allocated, sorted (segID1, local camera, segID2), camID2 GLOBAL,
 * confidence already divided by 2, cudawrapper.cu:1089-1110), *out_n, *median_depth
 * (cudawrapper.cu:1066-1076; untouched when nothing is verified), and, if non-NULL,
 * *out_best_depths (callee-allocated 2*(*out_n_best) floats: the depth pairs entering the median,
 * in segment order) so that sharded callers can merge medians exactly.
 * With n_tbm == 0 the reference returns immediately (cudawrapper.cu:877-878): the output is the
 * input list unchanged (LOCAL camera ids, confidence 0) and *median_depth is left alone. */
int l3d_compute_pairwise_matches(l3d_ctx* ctx,
                                 const float* src_segs, int S_src, const float* RtKinv_src, const float* C_src,
                                 const float* tgt_segs, const int32_t* offsets, int N,
                                 const float* F, const float* RtKinv, const float* centers, const float* P,
                                 const int32_t* to_be_matched, int n_tbm,
                                 const l3d_match* in_matches, int n_in, const uint32_t* local2global,
                                 float uncertainty_k_upper, float uncertainty_k_lower,
                                 float sigma_p, float sigma_a, float spatial_k,
                                 int seg_begin, int seg_end,
                                 l3d_match** out_matches, int* out_n, float* median_depth,
                                 float** out_best_depths, int* out_n_best);

/* Replaces replicator_dynamics_diffusion (cudawrapper.h:73-74, cudawrapper.cu:1131-1191) plus the
 * SparseMatrix construction of performDiffusion (line3D.cc:1258, sparsematrix.cc:63-191):
 * A = the affinity edge list in list order, n = number of nodes; out (caller-allocated, nnz
 * entries) = the entries of the returned matrix (row-sorted) as (i,j,w).  The reference's
 * positional lock-step product (cudawrapper.cu:786-800) is reproduced, not "fixed". */
int l3d_replicator_dynamics_diffusion(l3d_ctx* ctx, const l3d_edge* A, int nnz, int n, int iters, l3d_edge* out);

/* Batched Line3D::similarity_coll3D (line3D.cc:1600-1681) for the affinity fill
 * (clusterSegments2D, line3D.cc:968-1221): sim[k] = similarity(hyp[pairs[2k]], hyp[pairs[2k+1]]). */
int l3d_similarity_coll3D_batch(l3d_ctx* ctx, const l3d_hypothesis* hyp, int n_hyp,
                                const int32_t* pairs, int n_pairs, float sigma_a, float* sim);

/* The affinity fill of Line3D::clusterSegments2D (line3D.cc:968-1221) on the device: candidate enumeration with the
 * reference's `used` bookkeeping (three edge families: potential correspondence / collinear with it / collinear with the
 * source), similarity_coll3D, the thresholds (L3D_MIN_AFFINITY 0.25 / 0.01), first-touch node numbering and the symmetric
 * edge list A -- from flat tables.  Segments are numbered densely view by view (views in ascending camera id):
 * dense id = seg_base[view] + segment.  Hypotheses (greedySelection, line3D.cc:899-965) are numbered in dense order.
 *   seg_base        n_views + 1
 *   view_hyp_begin  n_views + 1: the hypotheses of a view are [view_hyp_begin[v], view_hyp_begin[v+1])
 *   hyp, score, hyp_dense   n_hyp: 3-D hypothesis, min(confidence, 1), dense id of its segment
 *   best            per dense id: its hypothesis or -1
 *   pot_start/pot_tgt   potential_correspondences_ (line3D.cc:861-865) per dense id as CSR, targets as dense ids, ascending
 *                       (entries whose camera names no view, or whose segment does not exist, left out)
 *   coll_start/coll_other/coll_w   segment2collinearities_ (segments.h:89-93) per dense id as CSR, ascending dense ids
 * Outputs (callee-allocated, l3d_free): edges (n_edges = 2 x kept candidates, (a,b,w) then (b,a,w), in the reference's
 * order), node_hyp[node] = hypothesis (local2global), the number of enumerated candidate pairs. */
typedef struct l3d_affinity_input {
    int32_t n_views;
    const int32_t* seg_base;
    const int32_t* view_hyp_begin;
    int32_t n_hyp;
    const l3d_hypothesis* hyp;
    const float* score;
    const int32_t* hyp_dense;
    const int32_t* best;
    const int64_t* pot_start;
    const int32_t* pot_tgt;
    const int64_t* coll_start;
    const int32_t* coll_other;
    const float* coll_w;
    float sigma_a;
} l3d_affinity_input;
int l3d_affinity_fill(l3d_ctx* ctx, const l3d_affinity_input* in, l3d_edge** edges, int* n_edges, int32_t** node_hyp, int* n_nodes,
                      int* n_candidates);
/* The fill enumerates its candidates a block of source segments at a time (the reference walks them source by source, line3D.cc:996-1221), so
 * neither the transient arrays nor any count of the whole fill is bound to 31 bits; *n_candidates above saturates at INT_MAX.  The 64-bit
 * figures of the last fill on ctx: candidate pairs enumerated, candidates that passed their threshold (the list has two entries per passed
 * candidate and IS bound to 2^31 entries, like the clustering stages behind it: L3D_ERR_UNSUPPORTED beyond). */
int l3d_last_fill_counts(l3d_ctx* ctx, int64_t* n_candidates, int64_t* n_passed);

/* The edge list Line3D::performClustering walks (clustering.cc:14-40; the merge loop itself: l3d_perform_clustering_device below), prepared on
 * the device: optionally performDiffusion (line3D.cc:1255-1303: replicator_dynamics_diffusion, then A(i,j) = A(j,i) =
 * min(W(i,j), W(j,i)), rebuilt in (i,j) order), then the STABLE ascending order by weight of clustering.cc:14.
 * A == NULL: the list the last l3d_affinity_fill returned, still resident on the device (nnz must match; it is consumed).
 * L3D_ERR_UNSUPPORTED: the diffused list is not a symmetric pattern of unique entries (never the case for the list of
 * l3d_affinity_fill) -- take l3d_replicator_dynamics_diffusion and the reference's map arithmetic instead. */
int l3d_clustering_edges(l3d_ctx* ctx, const l3d_edge* A, int nnz, int n_nodes, int perform_diffusion, int iters, l3d_edge* sorted_out);
/* The same list grouped by CONNECTED COMPONENT of the (diffused) graph -- labels by hooking + pointer jumping on the device --, in
 * stable ascending weight order inside every group: the merge loop of performClustering never relates nodes of different
 * components, so the groups can be walked independently, in parallel, with the result of the sequential walk.
 * group_start (callee-allocated, l3d_free): n_groups + 1 offsets into sorted_out. */
int l3d_clustering_edges_grouped(l3d_ctx* ctx, const l3d_edge* A, int nnz, int n_nodes, int perform_diffusion, int iters, l3d_edge* sorted_out,
                                 int32_t** group_start, int* n_groups);
/* clusterSegments2D's tail in one call (line3D.cc:1239-1246 -> clustering.cc:6-47, universe.h:59-115): [performDiffusion,]
 * the grouped order above and the MERGE LOOP of performClustering on the device, one wave per connected component (the loop is
 * sequential only inside a component; each is walked in LDS with the unions, ranks and roots of the one sequential walk).
 * labels (n_nodes, caller-allocated; NULL: they only stay on the device, for l3d_fit_labelled_clusters): labels[k] =
 * CLUniverse::find(k), bit-equal to performClustering's.  c: the reference
 * passes 1.0 (line3D.cc:1245).  n_components (may be NULL): connected components that hold an edge.
 * A == NULL: the resident list of l3d_affinity_fill, as above.  L3D_ERR_UNSUPPORTED as for l3d_clustering_edges. */
int l3d_perform_clustering_device(l3d_ctx* ctx, const l3d_edge* A, int nnz, int n_nodes, int perform_diffusion, int iters, float c, int32_t* labels,
                                  int* n_components);

/* The line fit of Line3D::processClusteredSegments (line3D.cc:1306-1597: getLineEquation3D, projectToLine) for many clusters at
 * once, on the device.  A cluster = its members' hypothesis indices in key order (camera, segment):
 * group_start (n_groups + 1) into member_hyp.  hyp / hyp_cam: all hypotheses (3-D end points in the normalised scene) and their
 * camera ids; hyp == NULL: the table of the last l3d_affinity_fill, still on the device (n_hyp must match).
 * Rinv (3x3 row-major), scale_inv, tneg: Line3D::inverseTransform (line3D.cc:1782-1786), applied to every end point.
 * Outputs (callee-allocated, l3d_free): seg_count[g] = 3-D segments of cluster g (the stretches seen by at least three cameras,
 * in sweep order), segs = 6 doubles (start, end) per segment, the clusters back to back. */
int l3d_fit_clusters(l3d_ctx* ctx, const int32_t* group_start, int n_groups, const int32_t* member_hyp, const l3d_hypothesis* hyp,
                     const uint32_t* hyp_cam, int n_hyp, const double* Rinv, double scale_inv, const double* tneg,
                     int32_t** seg_count, double** segs, int* n_segs);
/* processClusteredSegments from the LABELS (line3D.cc:1306-1368): the grouping on the device too -- clusters in ascending label
 * order (the reference's std::map), members in key order, those with >= 4 members seen from >= 4 cameras are fitted as above.
 * labels / node_hyp (n_nodes each): cluster label and hypothesis index of every node of the affinity list; NULL: the arrays
 * l3d_perform_clustering_device / l3d_affinity_fill* left on the device.  Out (callee-allocated, l3d_free): the fitted
 * clusters -- group_start (n_groups + 1) into member_hyp -- and seg_count / segs as l3d_fit_clusters.
 * PRECONDITION: hyp_cam is non-decreasing in the hypothesis index (hypotheses numbered view by view, as greedySelection numbers
 * them, line3D.cc:899-965): the distinct cameras of a cluster are counted as camera changes between members in hypothesis
 * order.  Checked; L3D_ERR_INVALID otherwise. */
int l3d_fit_labelled_clusters(l3d_ctx* ctx, const int32_t* labels, const int32_t* node_hyp, int n_nodes, const l3d_hypothesis* hyp,
                              const uint32_t* hyp_cam, int n_hyp, const double* Rinv, double scale_inv, const double* tneg,
                              int32_t** group_start, int32_t** member_hyp, int* n_groups, int32_t** seg_count, double** segs, int* n_segs);


/* ---- Line3D::matchViews as one device-resident chain --------------------------------------------------------
 * The schedule of matchViews (line3D.cc:620-648) is static: which neighbours a view still has to match and
 * which earlier views hand it reverse matches follows from the neighbour graph and the processing order alone.
 * The caller describes every view of the chain (same arrays as l3d_compute_pairwise_matches) plus its sources:
 * for each already-matched local camera `source_cam[i]`, the chain index `source_index[i]` (< own index) of that
 * view, whose kept matches towards this view become the existing matches (line3D.cc:806,838-872) -- on the
 * device, without touching the host.  Nothing waits for the host: stage 1 of all views is enqueued first, then
 * the per-view verification; the callback is invoked once per view, in order, as its kept list arrives, while
 * the GPU works on later views.  Results are bit-identical to calling l3d_compute_pairwise_matches per view.
 * Views with n_tbm == 0 are not computed (cudawrapper.cu:877-878): callback with verified = 0.
 * Callback arguments: kept matches (sorted, GLOBAL camera ids, confidence/2), the depth pairs entering the
 * median (cudawrapper.cu:1058-1062) and the number of candidates verified (0: the reference returns before
 * touching median_depth, cudawrapper.cu:955-956).  The buffers are only valid during the call.  Return non-zero
 * from the callback to abort. */
typedef struct l3d_chain_view {
    uint32_t view_id;
    const float* src_segs; int32_t S_src;
    const float* RtKinv_src; const float* C_src;
    const float* tgt_segs; int32_t n_tgt;
    const int32_t* offsets; int32_t N;
    const float* F; const float* RtKinv; const float* centers; const float* P;
    const int32_t* to_be_matched; int32_t n_tbm;
    const uint32_t* local2global;
    const int32_t* source_cam; const int32_t* source_index; int32_t n_sources;
    float sigma_p, sigma_a, spatial_k;
} l3d_chain_view;
/* `kept` points into pinned host memory owned by the context and stays valid until the next chain (l3d_match_chain /
 * l3d_shard_chain_open) starts on it or the context is destroyed: a callback may keep the pointer instead of copying.
 * `best_depths` is only valid during the call. */
typedef int (*l3d_chain_callback)(void* user, int index, int verified, const l3d_match* kept, int n_kept,
                                  const float* best_depths, int n_best, int n_candidates);
int l3d_match_chain(l3d_ctx* ctx, const l3d_chain_view* views, int n_views, l3d_chain_callback cb, void* user);

/* ---- Line3D::matchViews with its products kept in HBM ------------------------------------------------------------
 * l3d_match_chain_resident runs the chain of l3d_match_chain WITHOUT handing any kept list to the host, and builds what
 * performMatching leaves behind (line3D.cc:834-884) on the device, from the kept arena that never left HBM:
 *   - potential_correspondences_ (line3D.cc:861-865): for every kept match both directions, as a CSR over DENSE segment
 *     ids -- sorted, duplicates dropped (the reference's nested std::map: set semantics, ascending iteration);
 *   - the match file of every view after its own only-best overwrite (line3D.cc:884, view.cc:165-183): per segment the
 *     first kept match with the highest confidence;
 *   - the median depth of every view (cudawrapper.cu:1058-1076).
 * Views with nothing left to match (cudawrapper.cu:877-878) hand back their existing list with LOCAL camera ids and
 * confidence 0; their entries are formed with those numbers read as view ids, as the reference does.
 * Dense ids: the segments of ALL views of the scene, view by view in ascending camera id: dense = seg_base[i] + segment.
 * summary (caller's, n_views entries): per chain view its kept / candidate counts and median depth (1.0 where the
 * reference leaves it untouched).  n_pot: entries of the CSR.  The products stay valid until the next chain runs on ctx. */
typedef struct l3d_dense_map {
    int32_t n_views;               /* all views of the scene */
    const uint32_t* view_ids;      /* ascending */
    const int32_t* seg_base;       /* n_views + 1 */
} l3d_dense_map;
typedef struct l3d_chain_summary {
    int32_t verified;              /* 0: nothing left to match (early return) */
    int32_t n_kept;                /* length of the view's list (early return: the localized existing list) */
    int64_t n_candidates;
    float median_depth;
    int32_t pad;
} l3d_chain_summary;
int l3d_match_chain_resident(l3d_ctx* ctx, const l3d_chain_view* views, int n_views, const l3d_dense_map* map,
                             l3d_chain_summary* summary, int64_t* n_pot);
/* copies of the resident products (inspection, tests, host fallbacks).  l3d_chain_kept_list: the kept list of chain view
 * `index` as l3d_match_chain's callback would have received it (callee-allocated, l3d_free; 0 records for a view that was
 * not verified).  l3d_chain_products_get: pot_start (n_dense + 1), pot_tgt (n_pot), best_match (n_dense records; segID1 ==
 * 0xffffffff where a segment has none; camera ids as in the kept lists: GLOBAL, LOCAL for early-return views) -- any
 * pointer may be NULL. */
int l3d_chain_kept_list(l3d_ctx* ctx, int index, l3d_match** out, int* n);
int l3d_chain_products_get(l3d_ctx* ctx, int64_t* pot_start, int32_t* pot_tgt, l3d_match* best_match);

/* Line3D::greedySelection (line3D.cc:899-965) on the resident products: one 3-D hypothesis per segment that has a best
 * match (L3DView::unprojectSegment, view.cc:302-342, in double), numbered in dense order, left on the device for
 * l3d_affinity_fill_resident and l3d_fit_clusters.  geometry: one entry per view of the dense map.
 * Outputs: view_hyp_begin (caller's, n_views + 1), hyp_dense (callee-allocated, l3d_free: dense id per hypothesis). */
typedef struct l3d_view_geometry {
    double RtKinv[9];              /* R^T K^-1, row-major */
    double C[3];
    float k_lower, k_upper;        /* view.cc:90-121 */
    float median_depth;
    int32_t n_segments;
    const float* segments;         /* host pointer of the view's segments (registered: l3d_register_segments) */
} l3d_view_geometry;
int l3d_products_hypotheses(l3d_ctx* ctx, const l3d_view_geometry* geometry, int n_views, int32_t* view_hyp_begin, int32_t** hyp_dense, int* n_hyp);
/* l3d_affinity_fill on the resident tables: hypotheses of l3d_products_hypotheses, potential correspondences and best
 * matches of l3d_match_chain_resident; only the collinearity CSR (segment2collinearities_, static per scene) comes from the
 * host -- uploaded when coll_changed != 0, otherwise the copy of the previous call is used.  Outputs as l3d_affinity_fill. */
int l3d_affinity_fill_resident(l3d_ctx* ctx, const int64_t* coll_start, const int32_t* coll_other, const float* coll_w, int coll_changed,
                               float sigma_a, l3d_edge** edges, int* n_edges, int32_t** node_hyp, int* n_nodes, int* n_candidates);
/* (edges == NULL: the list is not copied to the host -- it stays resident for l3d_perform_clustering_device / l3d_clustering_edges(A
 * = NULL) and can be fetched later, also after the clustering consumed it: ) */
int l3d_resident_edges_get(l3d_ctx* ctx, l3d_edge* out, int nnz);
/* the resident hypothesis table (n_hyp entries of l3d_products_hypotheses) copied to the host */
int l3d_products_hypotheses_get(l3d_ctx* ctx, l3d_hypothesis* hyp, float* score);

/* ---- the resident chain with every view's source segments sharded over the GPUs of one node ---------------------
 * One process per GPU.  Each rank opens the chain with (rank, world) and per view k: l3d_shard_chain_enqueue writes
 * this rank's kept records for its source-segment range [S*rank/world, S*(rank+1)/world) into `send_slot`
 * (*slot_bytes bytes, device memory); the CALLER all-gathers the ranks' slots of view k into
 * gathered_base + (k*world + r)*slot_bytes (e.g. torch.distributed.all_gather_into_tensor = RCCL over xGMI, enqueued on
 * l3d_ctx_stream(ctx)), then calls l3d_shard_chain_mark.  Later views pull their reverse matches out of the gathered
 * slots on the device.  l3d_shard_chain_fetch (optional per rank, any time after mark) blocks the HOST until view k is
 * complete and calls the callback with the ranks' kept lists concatenated in rank (= segment) order -- the sorted list
 * of the unsharded run.  gathered_base must be zero-initialised (n_views*world*slot_bytes bytes) and stay valid until
 * close; the `views` array must outlive the chain. */
typedef struct l3d_shard_chain l3d_shard_chain;
void* l3d_ctx_stream(l3d_ctx* ctx);                  /* the context's hipStream_t, for framework interop */
int l3d_shard_chain_open(l3d_ctx* ctx, const l3d_chain_view* views, int n_views, int rank, int world, int slot_records,
                         l3d_shard_chain** out, size_t* slot_bytes);
int l3d_shard_chain_enqueue(l3d_shard_chain* chain, int k, void* send_slot, const void* gathered_base);
int l3d_shard_chain_mark(l3d_shard_chain* chain, int k);
int l3d_shard_chain_fetch(l3d_shard_chain* chain, int k, l3d_chain_callback cb, void* user);
int l3d_shard_chain_close(l3d_shard_chain* chain);
/* The same protocol as ONE native call (send/gathered buffers owned by the context): the calling thread enqueues view
 * after view -- kernels, then `exchange` on the context's stream, then the completion event -- and a second host thread
 * trails behind with fetch -> cb (cb == NULL: this rank does no host bookkeeping).  `exchange` must enqueue, on `stream`,
 * the all-gather of the ranks' slots of one view: send_slot (slot_bytes) -> recv_block (world*slot_bytes, rank order);
 * non-zero return = failure.  Adapters: l3d_exchange_rccl (user = l3d_rccl_link: an RCCL communicator of the `world`
 * ranks and the address of ncclAllGather, both taken from the RCCL the process has already loaded -- this library does
 * not link against RCCL), l3d_exchange_local (world == 1), l3d_exchange_replay (user = device address of the gathered
 * blocks of a recorded run: measures one rank of a world-W job on a single GPU, scripts/emulate_rank.py). */
typedef int (*l3d_exchange_fn)(void* user, int view, const void* send_slot, void* recv_block, size_t slot_bytes, int world, void* stream);
typedef struct l3d_rccl_link { void* comm; void* all_gather; } l3d_rccl_link;
int l3d_shard_chain_run(l3d_shard_chain* chain, l3d_exchange_fn exchange, void* exchange_user, l3d_chain_callback cb, void* cb_user);
int l3d_exchange_rccl(void* user, int view, const void* send_slot, void* recv_block, size_t slot_bytes, int world, void* stream);
int l3d_exchange_local(void* user, int view, const void* send_slot, void* recv_block, size_t slot_bytes, int world, void* stream);
int l3d_exchange_replay(void* user, int view, const void* send_slot, void* recv_block, size_t slot_bytes, int world, void* stream);
const void* l3d_shard_chain_gathered(l3d_shard_chain* chain);   /* device address of the gathered blocks after l3d_shard_chain_run */
/* A rank that fails inside l3d_shard_chain_run (capacity, HIP error, failing callback) keeps calling `exchange` for every
 * remaining view with a slot that says "gave up": no rank is left waiting in a collective.  Afterwards every rank reads the
 * same verdict out of the gathered slot headers: L3D_ERR_NOMEM on ALL ranks when any slot overflowed.  l3d_shard_chain_info
 * (any out pointer may be NULL): this chain's candidate capacity and slot_records, and after a run the OR of the ranks'
 * overflow bits (1 candidate capacity, 2 slot_records, 4 a rank gave up, 8 the compact arena of the ring mode) and the largest candidate / kept count one rank
 * reported for one view -- what a caller needs to reopen with more room (l3d_set_chain_capacities sets the candidate capacity
 * of the next chain); l3d_line3d_shard_run does exactly that, up to three times. */
/* matchViews' products (as l3d_match_chain_resident leaves them: potential correspondences, best matches, medians; line3D.cc:834-884)
 * built on THIS rank's device from the gathered slots of a finished l3d_shard_chain_run -- every rank holds every view's kept records,
 * so no rank has to hand lists to the host (cb == NULL on all ranks).  The context then serves l3d_chain_kept_list,
 * l3d_chain_products_get, l3d_products_hypotheses, l3d_affinity_fill_resident as after the single-GPU chain. */
int l3d_shard_chain_products(l3d_shard_chain* chain, const l3d_dense_map* map, l3d_chain_summary* summary, int64_t* n_pot);
int l3d_shard_chain_info(l3d_shard_chain* chain, size_t* cand_cap, int* slot_records, int* overflow_bits, int* max_candidates, int* max_kept);
/* Ring mode of l3d_shard_chain_run (cb == NULL; automatic when the gathered blocks of all views would exceed 8 GB -- 2048 views x 8 ranks
 * x 3.8 MB = 63 GB per rank at 4000 segments x 24 neighbours --, L3D_SLOT_RING=1 / l3d_set_option forces it, 0 forbids it): the gathered
 * buffer holds only the views later views still read (the neighbour window of the schedule + a batch); older blocks are retired into the
 * compact kept arena, in the order of the unsharded run, before they are overwritten -- what the reference does with its per-view match files
 * (view.cc:150-224).  l3d_shard_chain_gathered then returns NULL.  Overflow bit 8 = the compact arena's first guess was too small;
 * l3d_shard_chain_arena_needed: the kept matches of the whole run (records of 32 bytes), for the retry (l3d_set_chain_capacities' second
 * argument; l3d_line3d_shard_run does it). */
long long l3d_shard_chain_arena_needed(l3d_shard_chain* chain);
/* A PARTITIONED job on the segment-sharded run (call between l3d_shard_chain_open and l3d_shard_chain_run, cb == NULL there): every rank sees
 * every view's gathered slots go by, so what it KEEPS is its choice -- the views its block [own_begin, own_end) of the chain needs (2 x reach
 * either side, reach = the largest chain distance between a view and a neighbour), the sources of the early-return views and the views their
 * local camera numbers name (cudawrapper.cu:877-878, line3D.cc:861-865).  The run retires only those into this rank's arena (ring mode, forced);
 * l3d_shard_chain_products then builds this rank's share of matchViews' products -- the state l3d_match_chain_partition leaves, without any
 * speculation: exact on every scene, the chain's work split 1/world per view, the kept records and the table split by view block.
 * l3d_products_hypotheses and l3d_affinity_fill_sharded follow as there.  (l3d_line3d_shard_run: commit = 3.) */
int l3d_shard_chain_partition(l3d_shard_chain* chain, int own_begin, int own_end);
/* the keep set of l3d_shard_chain_partition as a function of the schedule alone (host logic, no context): keep[k] = 1 for the chain views a rank that
 * owns [own_begin, own_end) retires; *reach (may be NULL) = the largest chain distance between a view and one of its neighbours */
int l3d_partition_keep_views(const l3d_chain_view* views, int n_views, int own_begin, int own_end, unsigned char* keep, int* reach);

/* ---- Line3D::matchViews sharded by BLOCKS OF VIEWS over the ranks, speculatively, with exact verification ----------------------
 * (line3D.cc:620-648 is a chain over views: the kept matches of a view become candidates of its later neighbours, :806,838-872.)
 * Rank r runs the ordinary single-GPU resident chain on views [B_r - warmup_views, B_{r+1}) only (B_r = n_views * r / world), started
 * cold: nothing is known about the views in front.  The chain's memory is short -- about three neighbour windows on the synthetic
 * scenes (scripts/speculate_blocks.py) -- so by view B_r the kept lists normally ARE the one chain's.  That is checked, not assumed:
 * every rank publishes a 64-bit digest of every kept list it computed (`exchange`, an all-gather: view = -1); rank r's block is exact iff
 * rank r-1's is and the `window` views in front of B_r came out of r's warm-up exactly as r-1 computed them (from B_r on every view then
 * has the one chain's inputs).  Every rank reads the same table and reaches the same verdict.  *verdict = 0: the ranks all-gathered
 * their blocks (view = -2) and THIS context now holds matchViews' products exactly as after l3d_match_chain_resident over all views
 * (arena, potential correspondences, best matches, medians; summary / n_pot as there).  A block whose speculation did NOT hold is repaired, not
 * abandoned (round 5): every rank that missed takes over its predecessor's last `window` views -- records, best depth pairs and positions, one
 * all-gather, view = -5 -- and re-runs its block warm from them, all of them at once; the digests are exchanged again, until nobody misses (rank j
 * is exact after round j at the latest; a warm-up shorter than the window simply makes a rank take that path).  *verdict = 1 only when a block is shorter than the window
 * (it cannot vouch for its successor's sources) or option block_recover = 0 (the round-4 behaviour): nothing was committed, run l3d_shard_chain_run.
 * No per-view collective: four data exchanges per pass (digests; blocks; view = -4 the pieces of the products table, of which every rank
 * builds the rows of its own block) and three 256-byte ones of status words (view = -3; the second carries the sizes of the pieces): every step
 * only one rank can fail in (an allocation, a launch) is followed by one, so either all ranks enter the big collective behind it or all
 * return an error -- nobody is left waiting.  window = the largest distance between a view and one of its sources. */
int l3d_match_chain_blocks(l3d_ctx* ctx, const l3d_chain_view* views, int n_views, const l3d_dense_map* map, l3d_chain_summary* summary, int64_t* n_pot,
                           int rank, int world, int warmup_views, int window, l3d_exchange_fn exchange, void* exchange_user, int* verdict);

/* ---- Line3D::matchViews sharded by blocks of views with NOTHING REPLICATED (the configs[4] job: 2048 views x 4000 segments x 24 neighbours keep
 * 6.4 G matches -- beyond a 32-bit record index and, with the table built from them, beyond one GPU's HBM; the reference streams a view at a time
 * and spills every view's matches to a file, view.cc:150-224, line3D.cc:626-648).  Same speculation and verification as l3d_match_chain_blocks;
 * but rank r runs its chain 2 x reach views PAST its block (reach = the largest distance, in chain positions, between a view and one of its
 * neighbours) and is checked max(window, 2 x reach) views in front of it -- so it holds, computed by itself, the exact kept records of every view
 * within 2 x reach of its block.  From those it builds, locally: the rows of potential_correspondences_ of the views within `reach` of its block
 * (complete: a row needs the records of the view and of its neighbours), the best matches and medians of every view it holds.  That is all the
 * affinity fill of ITS block's sources reads (l3d_affinity_fill_sharded).  No block is gathered, no piece of the table travels; the records that
 * point at an early-return view (cudawrapper.cu:877-878: their entries are filed under LOCAL camera numbers read as view ids and can name any view
 * of the scene) are all-gathered, each source's by the rank that owns it (view = -6).  *verdict = 0: the context holds this rank's share --
 * l3d_chain_kept_list serves the views it holds (empty lists for the others), l3d_chain_products_get its rows (empty rows elsewhere; *n_pot = its
 * entries), l3d_products_hypotheses numbers the hypotheses of the views it holds from 0 (l3d_affinity_fill_sharded makes the numbers global).
 * *verdict = 1: as l3d_match_chain_blocks (only when a block is re-run more than `world` times or the schedule holds more than 64 early returns:
 * a failed speculation is repaired by re-running that block warm, see there). */
int l3d_match_chain_partition(l3d_ctx* ctx, const l3d_chain_view* views, int n_views, const l3d_dense_map* map, l3d_chain_summary* summary, int64_t* n_pot,
                              int rank, int world, int warmup_views, int window, l3d_exchange_fn exchange, void* exchange_user, int* verdict);
/* what this rank's share covers after l3d_match_chain_partition, in views of the dense map: info = { rank, world, own block [begin, end), rows
 * [begin, end), held [begin, end), rounds of warm re-runs, blocks re-run in them (whole job; also set by l3d_match_chain_blocks) }; n_pot_all:
 * entries of the table over all ranks */
int l3d_partition_info(l3d_ctx* ctx, int info[10], int64_t* n_pot_all);
/* The fill SHARDED BY SOURCE KEY over the ranks of a job whose matchViews ran partitioned (l3d_match_chain_partition below): every
 * rank enumerates the candidates of the sources of ITS block of views -- from the rows, best matches and hypotheses it holds, nothing of another
 * rank is read -- and five small all-gathers (`exchange`, views -7 .. -10 and status words -3) make global what has to be: hypotheses per view
 * (global hypothesis numbers), the first-touch minima per hypothesis (64-bit positions: rank << 44 | position inside the rank's enumeration), the
 * candidates that passed their threshold (12 bytes each; concatenated in rank order = the reference's enumeration order, line3D.cc:996-1221) and
 * the hypothesis table (read by the line fit).  Node numbering and the edge list are then formed by every rank from the same data: afterwards the
 * context is in the state l3d_affinity_fill_resident(edges = NULL) leaves it in, with GLOBAL hypothesis numbers -- l3d_perform_clustering_device,
 * l3d_fit_labelled_clusters, l3d_resident_edges_get, l3d_products_hypotheses_get follow unchanged (replicas).  Outputs: n_edges / node_hyp / n_nodes
 * as l3d_affinity_fill_resident; n_candidates_all (may be NULL): candidate pairs enumerated by all ranks; view_hyp_begin_global (caller's,
 * n_views + 1 of the dense map), hyp_dense_global (callee-allocated, l3d_free: dense segment id per global hypothesis), n_hyp_global.
 * A rank that fails on its own still enters every collective with a mark, so all ranks return an error together. */
int l3d_affinity_fill_sharded(l3d_ctx* ctx, const int64_t* coll_start, const int32_t* coll_other, const float* coll_w, int coll_changed, float sigma_a,
                              l3d_exchange_fn exchange, void* exchange_user, int* n_edges, int32_t** node_hyp, int* n_nodes,
                              int64_t* n_candidates_all, int32_t* view_hyp_begin_global, int32_t** hyp_dense_global, int* n_hyp_global);

/* ---- residency: keep a view's segments in HBM across calls ------------------------------------
 * The reference re-uploads every neighbour's segments for every view (line3D.cc:793-800).  A
 * caller may instead register segment arrays once; l3d_compute_pairwise_matches recognises
 * host pointers that were registered (same pointer, same size) and skips the upload. */
int l3d_register_segments(l3d_ctx* ctx, const float* segments, int n_segments);
int l3d_unregister_segments(l3d_ctx* ctx, const float* segments);
/* the same for many arrays at once (all views of a scene): one device allocation, one wait */
int l3d_register_segments_batch(l3d_ctx* ctx, const float* const* arrays, const int* n_segments, int n_arrays);

/* ---- measurement ---------------------------------------------------------------------------
 * With profiling on, every kernel launch is bracketed by HIP events on the context's stream;
 * l3d_profile_get returns the launch count and summed duration for one kernel name
 * ("pair_mask", "pair_fill", "verify", ...; l3d_profile_names lists them, ';'-separated). */
/* stage-2 algorithm: 0 = depth-window search (default; segments that outgrow the kernel's LDS image use its global-scratch
 * variant; views with more than ~50 neighbours take the all-pairs kernel), 1 = all-pairs loop in the reference's formulation.
 * Results are bit-identical. */
int l3d_set_verify_mode(l3d_ctx* ctx, int mode);
/* stage-1 filters in front of the exact epipolar/overlap/triangulation sequence: bit 0 = wedge test, bit 1 = interval bounds of the two
 * overlap ratios (they reject -- and, in the chains, accept -- what they can decide; a pair with an intersection point on a segment end
 * point is always left to the exact test), bit 2 SET = the bounds do not accept; 3 = everything (default), 0 = the exact sequence
 * alone (A/B testing: results are bit-identical, candidate counts included) */
int l3d_set_pair_pretest(l3d_ctx* ctx, int mask);
/* Diagnostic / A-B switches (line3d_amd/csrc/l3d_options.hpp lists them with their meaning).  The environment (L3D_<NAME>) is
 * read ONCE, when the context is created; afterwards a switch changes only through this call.  name: "L3D_TIMING", "TIMING" or
 * "timing".  No switch selects a CPU path for the arithmetic.  L3D_ERR_INVALID for an unknown name. */
int l3d_set_option(l3d_ctx* ctx, const char* name, int value);
int l3d_get_option(l3d_ctx* ctx, const char* name, int* value);
/* testing: cap the LDS image of the depth-window kernel (bytes; 0 = device limit) so that segments take the
 * global-scratch variant; process-wide */
int l3d_set_verify_lds_budget(size_t bytes);
/* testing: initial candidate / kept-arena capacities (records) of the resident chain, 0 = the built-in estimate; small
 * values force the overflow -> grow -> restart-at-that-view path */
int l3d_set_chain_capacities(l3d_ctx* ctx, size_t cand_cap, size_t arena_cap);
/* loads the code objects of every kernel of the library now (in parallel) instead of at their first launch: takes the
 * ~10 x 3 ms of lazy loading out of the first matchViews / finish of a process */
int l3d_warm_up(l3d_ctx* ctx);
/* reserves the device arenas of the finishing stages (greedy selection, affinity fill, edge order, line fit) ahead of their
 * first use, from the size of the scene: a hint, the arenas still grow on demand */
int l3d_reserve_hint(l3d_ctx* ctx, int n_dense, int n_views, int n_neighbors);
int l3d_profile_enable(l3d_ctx* ctx, int on);
/* bracket only the named kernel with HIP events (NULL or "": all kernels) -- keeps a timed region nearly undisturbed */
int l3d_profile_only(l3d_ctx* ctx, const char* kernel);
int l3d_profile_reset(l3d_ctx* ctx);
int l3d_profile_get(l3d_ctx* ctx, const char* kernel, int64_t* launches, double* total_ms);
const char* l3d_profile_names(void);
/* counters of the last l3d_compute_pairwise_matches call:
 * [0] stage-1 pairs evaluated, [1] raw candidates (incl. existing), [2] verify inner iterations
 * (sum over segments of m^2), [3] kept matches */
int l3d_last_stats(l3d_ctx* ctx, double stats[4]);

/* contract math exported for tests (device evaluation of c_expf / c_acosf / c_acos) */
int l3d_test_contract_math(l3d_ctx* ctx, const float* x, int n, float* out_expf, float* out_acosf, double* out_acos);
/* the squared-distance gate threshold T(u) = largest float whose correctly rounded square root is <= u:
 * the ulp walk and the closed form the kernels use, evaluated on the device (tests) */
int l3d_test_sq_threshold(l3d_ctx* ctx, const float* u, int n, float* out_walk, float* out_closed);

/* =================================================================================================
 * Host pipeline behind the reference's public operator interface, class L3D::Line3D
 * (line3D.h:61-101): the same calls, argument meaning and error behaviour, with plain arrays in
 * place of cv::Mat / Eigen / std::list.  include/line3D_amd.hpp wraps these in a C++ class named
 * L3D::Line3D.  Images are replaced by their detected segments (the LSD front end is out of
 * scope): `segments` is what detectLineSegments would have produced (line3D.cc:1789-1871).
 * K, R (3x3 row-major) and t are doubles like the reference's Eigen arguments.
 * ================================================================================================= */
typedef struct l3d_line3d l3d_line3d;

/* Line3D::Line3D (line3D.cc:6-48); defaults commons.h:42-61.  Owns one l3d_ctx on `device`. */
int l3d_line3d_create(int device, int matching_neighbors, float uncertainty_t_upper_2D, float uncertainty_t_lower_2D,
                      float sigma_p, float sigma_a, float min_baseline, int use_collinearity, int verbose,
                      l3d_line3d** out);
void l3d_line3d_destroy(l3d_line3d* h);
const char* l3d_line3d_last_error(const l3d_line3d* h);
l3d_ctx* l3d_line3d_context(l3d_line3d* h);
int l3d_line3d_reset(l3d_line3d* h);                                             /* line3D.cc:62-92 */
int l3d_line3d_num_cameras(const l3d_line3d* h);                                 /* line3D.h:98 */
/* Line3D::addImage (line3D.cc:95-217) and addImage_fixed_sim (line3D.cc:220-342) */
int l3d_line3d_add_image(l3d_line3d* h, uint32_t image_id, unsigned width, unsigned height,
                         const float* segments, int n_segments, const double* K, const double* R, const double* t,
                         const uint32_t* worldpoint_ids, int n_worldpoints);
int l3d_line3d_add_image_fixed_sim(l3d_line3d* h, uint32_t image_id, unsigned width, unsigned height,
                                   const float* segments, int n_segments, const double* K, const double* R, const double* t,
                                   const uint32_t* sim_ids, const float* sims, int n_sims);
/* Line3D::compute3Dmodel (line3D.cc:345-374) = prepare + match_views + finish */
int l3d_line3d_compute3Dmodel(l3d_line3d* h, int perform_diffusion);
int l3d_line3d_prepare(l3d_line3d* h);          /* findVisualNeighbors + transformGeometry; inputs become HBM-resident */
int l3d_line3d_match_views(l3d_line3d* h);      /* Line3D::matchViews, line3D.cc:620-648 (re-runnable) */
int l3d_line3d_finish(l3d_line3d* h, int perform_diffusion);   /* optimizeLocalMatches + clusterSegments2D */
/* step-wise matchViews for view sharding: begin; for each view in match_order: compute a source-segment
 * range, (all-gather), commit the merged kept list; end. */
int l3d_line3d_match_begin(l3d_line3d* h, int* n_order);
int l3d_line3d_match_order(l3d_line3d* h, uint32_t* view_ids, int* n_segments);
int l3d_line3d_view_num_to_be_matched(l3d_line3d* h, uint32_t view_id);
int l3d_line3d_match_view_compute(l3d_line3d* h, uint32_t view_id, int seg_begin, int seg_end,
                                  l3d_match** out, int* n_out, float* median, float** best_depths, int* n_best);
int l3d_line3d_match_view_commit(l3d_line3d* h, uint32_t view_id, const l3d_match* matches, int n,
                                 const float* best_depths, int n_best, float median);
int l3d_line3d_match_end(l3d_line3d* h);
/* matchViews as the resident chain sharded over ranks: the facade builds the static schedule and forwards to
 * l3d_shard_chain_* (same protocol: enqueue -> caller's all-gather -> mark; fetch = this rank's host bookkeeping) */
int l3d_line3d_shard_open(l3d_line3d* h, int rank, int world, int slot_records, int* n_views, size_t* slot_bytes);
int l3d_line3d_shard_view_verified(l3d_line3d* h, int k);
int l3d_line3d_shard_enqueue(l3d_line3d* h, int k, void* send_slot, const void* gathered_base);
int l3d_line3d_shard_mark(l3d_line3d* h, int k);
int l3d_line3d_shard_fetch(l3d_line3d* h, int k);
int l3d_line3d_shard_close(l3d_line3d* h, int committed);
/* open -> run -> close in one call (l3d_shard_chain_run); commit: 0 = this rank only computes and exchanges; 1 = it also does the
 * host bookkeeping (kept lists handed to the host view by view: the round-2 protocol, one rank of the job); 2 = it builds matchViews'
 * products on its device from the gathered slots (l3d_shard_chain_products) -- any number of ranks, no list leaves the device, the
 * rest of compute3Dmodel runs on the resident tables as after the single-GPU chain; 3 = as 2, partitioned: the rank keeps the records and
 * builds the rows of ITS block of views only (l3d_shard_chain_partition), l3d_line3d_finish_sharded follows on every rank.
 * gathered_out (optional) receives the device address of the gathered blocks (valid until the next chain) */
int l3d_line3d_shard_run(l3d_line3d* h, int rank, int world, int slot_records, l3d_exchange_fn exchange, void* exchange_user, int commit,
                         const void** gathered_out, size_t* slot_bytes_out);
/* matchViews with the VIEWS sharded over the ranks in blocks, each block started cold a few neighbour windows early, the speculation verified
 * (l3d_match_chain_blocks above; warmup_views < 0: eight windows; a block whose speculation fails all the same is re-run warm, not the pass).  *verdict = 0: this rank holds matchViews' products as after the
 * single-GPU resident chain -- compute3Dmodel goes on from there (l3d_line3d_finish); *verdict = 1 (identical on every rank): the speculation
 * did not hold, nothing was committed, run l3d_line3d_shard_run. */
int l3d_line3d_block_run(l3d_line3d* h, int rank, int world, int warmup_views, l3d_exchange_fn exchange, void* exchange_user, int* verdict);
/* matchViews sharded by blocks of views with NOTHING replicated (l3d_match_chain_partition; warmup_views < 0: eight windows), and the rest of
 * compute3Dmodel as a COLLECTIVE of the job's ranks: greedy selection on the views a rank holds, the affinity fill sharded by source key
 * (l3d_affinity_fill_sharded), then -- every rank from the same affinity list -- diffusion, clustering, line fit: every rank ends with the whole
 * result (Line3D::getResult).  l3d_line3d_finish on such an object is the same call with the exchange of the run; l3d_line3d_view_matches serves the
 * views this rank holds.  exchange == NULL in finish_sharded: the one the run was given. */
int l3d_line3d_partition_run(l3d_line3d* h, int rank, int world, int warmup_views, l3d_exchange_fn exchange, void* exchange_user, int* verdict);
int l3d_line3d_finish_sharded(l3d_line3d* h, int perform_diffusion, l3d_exchange_fn exchange, void* exchange_user);
/* performClustering (clustering.h:125, clustering.cc:6-47) on the host (fallback and cross-check of l3d_perform_clustering_device): labels[k] = CLUniverse::find(k) */
int l3d_perform_clustering(const l3d_edge* edges, int n_edges, int num_nodes, float c, int32_t* labels);
/* Line3D::getResult (line3D.cc:377-381), flattened; Line3D::getSegment2D (line3D.cc:2004-2013) */
int l3d_line3d_result_sizes(const l3d_line3d* h, int* n_lines, int* n_seg3d, int* n_seg2d);
int l3d_line3d_get_result(const l3d_line3d* h, int* line_n3d, int* line_n2d, double* seg3d, uint32_t* seg2d);
int l3d_line3d_get_segment2D(const l3d_line3d* h, uint32_t cam, uint32_t seg, float out[4]);
/* Line3D::save3DLinesAsSTL / save3DLinesAsTXT (line3D.h:91-95, line3D.cc:384-473; TXT format README.txt:177-185) for
 * the current result, with the reference's number formatting ("%e" / stream default = "%g") */
#define L3D_FORMAT_STL 0
#define L3D_FORMAT_TXT 1
int l3d_line3d_save_result(const l3d_line3d* h, const char* filename, int format);
/* inspection */
/* 1: matchViews through one l3d_compute_pairwise_matches call per view (the reference's control flow);
 * 0 (default): the device-resident chain (l3d_match_chain).  Results are identical. */
int l3d_line3d_set_sync_matching(l3d_line3d* h, int on);
int l3d_line3d_keep_view_matches(l3d_line3d* h, int on);
/* which way the last l3d_line3d_match_views took: 0 = the resident chain (products on the device), 1 = the chain with host bookkeeping, 2 = per-view
 * seam calls on request (set_sync_matching), 3 = per-view seam calls because the schedule is not static (an early-return view's local camera
 * numbers name a view that still accepts reverse matches, line3D.cc:844-845); -1 = matchViews has not run */
int l3d_line3d_match_path(const l3d_line3d* h);
int l3d_line3d_view_matches(const l3d_line3d* h, uint32_t view_id, const l3d_match** m, int* n, float* median);
int l3d_line3d_affinity(const l3d_line3d* h, const l3d_edge** A, int* nnz, int* n_nodes);
/* the device-resident products of the last matchViews (l3d_match_chain_resident) and, after finish, the hypothesis table of
 * greedySelection, copied to the host for inspection: sizes first (0 views: no resident products -- sync / host-bookkeeping /
 * sharded matching), then the arrays (any pointer may be NULL): seg_base n_views + 1, pot_start n_dense + 1, pot_tgt n_pot,
 * best n_dense (segID1 == 0xffffffff: none), hyp / score n_hyp */
int l3d_line3d_products_sizes(const l3d_line3d* h, int* n_views, int* n_dense, int64_t* n_pot, int* n_hyp);
int l3d_line3d_products_get(l3d_line3d* h, int32_t* seg_base, int64_t* pot_start, int32_t* pot_tgt, l3d_match* best, l3d_hypothesis* hyp, float* score);
int l3d_line3d_stats(const l3d_line3d* h, double* stats12);

/* =================================================================================================
 * SfM front ends of the reference's drivers: VisualSfM NVM (main_vsfm.cpp:121-223) and bundler
 * bundle.rd.out (main_bundler.cpp:110-204), reduced to what feeds Line3D::addImage -- focal length, R, t, distortion
 * coefficients, observed world point ids per camera.  Image decoding / undistortion / LSD stay outside.
 * A failed read still returns a scene object carrying the message (l3d_sfm_last_error); free it with l3d_sfm_free.
 * ================================================================================================= */
typedef struct l3d_sfm_scene l3d_sfm_scene;
int l3d_sfm_read_nvm(const char* path, l3d_sfm_scene** out);
int l3d_sfm_read_bundler(const char* path, l3d_sfm_scene** out);
void l3d_sfm_free(l3d_sfm_scene* scene);
const char* l3d_sfm_last_error(const l3d_sfm_scene* scene);
int l3d_sfm_num_cameras(const l3d_sfm_scene* scene);
int l3d_sfm_num_points(const l3d_sfm_scene* scene);
int l3d_sfm_camera(const l3d_sfm_scene* scene, int i, double* focal, double dist[2], double R[9], double t[3], int* n_worldpoints);
const char* l3d_sfm_camera_name(const l3d_sfm_scene* scene, int i);
int l3d_sfm_camera_worldpoints(const l3d_sfm_scene* scene, int i, uint32_t* ids);
/* the drivers' K from a focal length and the image size (main_vsfm.cpp:232-241): [[f,0,w/2],[0,f,h/2],[0,0,1]] */
void l3d_sfm_intrinsics(double focal, unsigned int width, unsigned int height, double K[9]);

/* =================================================================================================
 * Segment cache of Line3D::addImage: "<data dir>/segments_<id>_<w>x<h>_coll<0|1>.bin" (line3D.cc:143-150),
 * a boost binary archive of one L3DSegments -- the collinearity map, then the DataArray<float>* of padded rows
 * (serialization.h:49-69, segments.h:124-131, dataArray.h:296-318) -- read and written without boost
 * (layout and its pin status: line3d_amd/csrc/l3d_segcache.cpp).  A failed read still returns an object carrying the
 * message (l3d_segment_cache_last_error); free it with l3d_segment_cache_free.
 * Collinearities are the DIRECTED entries of segment2collinearities_ (i -> j and j -> i), ascending (i, j).
 * ================================================================================================= */
typedef struct l3d_segment_cache l3d_segment_cache;
int l3d_segment_cache_filename(uint32_t image_id, unsigned int width, unsigned int height, int use_collinearity, char* out, size_t out_size);
int l3d_segment_cache_read(const char* path, l3d_segment_cache** out);
void l3d_segment_cache_free(l3d_segment_cache* cache);
const char* l3d_segment_cache_last_error(const l3d_segment_cache* cache);
int l3d_segment_cache_num_segments(const l3d_segment_cache* cache);
int l3d_segment_cache_num_collinearities(const l3d_segment_cache* cache);
int l3d_segment_cache_library_version(const l3d_segment_cache* cache);
int l3d_segment_cache_get(const l3d_segment_cache* cache, float* segments, int32_t* ci, int32_t* cj, float* cw);
int l3d_segment_cache_write(const char* path, const float* segments, int n_segments, const int32_t* ci, const int32_t* cj, const float* cw,
                            int n_collinearities, int library_version);
/* Line3D::addImage when the cache file exists (line3D.cc:160-168): the segments AND the collinearities of the file are
 * used as they are (nothing is recomputed); world points as in l3d_line3d_add_image */
int l3d_line3d_add_image_cached(l3d_line3d* h, uint32_t image_id, unsigned width, unsigned height, const l3d_segment_cache* cache,
                                const double* K, const double* R, const double* t, const uint32_t* worldpoint_ids, int n_worldpoints);

/* Line3D::addImage / addImage_fixed_sim with the reference's cache behaviour (line3D.cc:128-199, 253-324): the cache file of the
 * view is "<data_directory>/segments_<id>_<w'>x<h'>_coll<0|1>.bin" with (w', h') the image size after the max_img_width
 * down-scaling the detector would have worked at (line3D.cc:133-138; the view itself keeps the original size).
 *   file exists, load_and_store == 0   the file is removed (line3D.cc:153-156), `segments` are used
 *   file exists, load_and_store != 0   segments AND collinearities come from the file, `segments` are ignored (:159-168)
 *   otherwise                          `segments` are used; with load_and_store != 0 the cache is written (:180-182) -- at
 *                                      prepare(), when the collinearities of all new views have been computed in one batch
 * `segments` is what detectLineSegments (line3D.cc:1789-1871: LSD, length filter, longest 3000) would have produced.
 * links: observed world point ids (add_image_ex) or (view id, similarity) pairs (add_image_fixed_sim_ex). */
int l3d_line3d_add_image_ex(l3d_line3d* h, uint32_t image_id, unsigned width, unsigned height, const float* segments, int n_segments,
                            const double* K, const double* R, const double* t, const uint32_t* worldpoint_ids, int n_worldpoints,
                            const char* data_directory, int max_img_width, int load_and_store);
int l3d_line3d_add_image_fixed_sim_ex(l3d_line3d* h, uint32_t image_id, unsigned width, unsigned height, const float* segments, int n_segments,
                                      const double* K, const double* R, const double* t, const uint32_t* sim_ids, const float* sims, int n_sims,
                                      const char* data_directory, int max_img_width, int load_and_store);

#ifdef __cplusplus
}
#endif
#endif /* LINE3D_AMD_H */
