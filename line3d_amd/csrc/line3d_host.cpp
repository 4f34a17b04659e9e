// line3d_host.cpp -- the pipeline facade of the C ABI (include/line3d_amd.h: l3d_line3d_*): the reference's operator interface
// (addImage / addImage_fixed_sim / compute3Dmodel / getResult, line3D.h) over the host pipeline in line3d_host_views.cpp (views, seam path),
// line3d_host_chain.cpp (resident matchViews) and line3d_host_finish.cpp (selection, affinity, diffusion, clustering, fit).
#include "line3d_host_internal.hpp"


// =================================================================================================
extern "C" {

int l3d_line3d_create(int device, int matching_neighbors, float unc_upper, float unc_lower, float sigma_p, float sigma_a,
                      float min_baseline, int use_collinearity, int verbose, l3d_line3d** out)
{
    if (!out) return L3D_ERR_INVALID;
    *out = nullptr;
    l3d_ctx* ctx = nullptr;
    int rc = l3d_ctx_create(device, &ctx);
    if (rc) return rc;
    L* h = new L();
    h->ctx = ctx;
    h->verbose = verbose != 0;
    h->matching_neighbors = matching_neighbors;
    h->unc_upper = fabsf(unc_upper);                     // line3D.cc:18-28
    h->unc_lower = fabsf(unc_lower);
    if (h->unc_lower < 1.0f) h->unc_lower = 1.0f;
    if (h->unc_upper <= h->unc_lower) h->unc_upper = h->unc_lower + 1.0f;
    h->sigma_p = sigma_p; h->sigma_a = sigma_a; h->min_baseline = min_baseline;
    h->use_collinearity = use_collinearity != 0;
    h->force_sync = l3d::ctx_options(ctx).match_sync != 0;
    h->warm_thread = std::thread([ctx]() { (void)l3d_warm_up(ctx); });
    h->host_bookkeeping = l3d::ctx_options(ctx).host_bookkeeping != 0;
    *out = h;
    return L3D_OK;
}

void l3d_line3d_destroy(l3d_line3d* h)
{
    if (!h) return;
    if (h->warm_thread.joinable()) h->warm_thread.join();
    destroy_finalizer(h);                                   // joins the worker threads
    drop_plan(h);
    l3d_ctx_destroy(h->ctx);
    delete h;
}

const char* l3d_line3d_last_error(const l3d_line3d* h) { return h ? h->err.c_str() : "null handle"; }
l3d_ctx* l3d_line3d_context(l3d_line3d* h) { return h ? h->ctx : nullptr; }

// Line3D::reset, line3D.cc:62-92
int l3d_line3d_reset(l3d_line3d* h)
{
    if (!h) return L3D_ERR_INVALID;
    for (auto& kv : h->views) { l3d_unregister_segments(h->ctx, kv.second.segs.data()); l3d_unregister_segments(h->ctx, kv.second.nb_segs.data()); }
    h->views.clear(); h->vlist.clear(); h->view_similarities.clear(); h->num_wps.clear(); h->common_wps.clear();
    h->worldpoints2views.clear(); h->visual_neighbors.clear(); h->fundamentals.clear(); h->matched.clear();
    h->pot.clear(); h->pot_foreign.clear(); h->hyps.clear(); h->best_idx.clear(); h->A.clear(); h->n_edges = 0; h->A_on_host = true; h->local2global.clear(); h->result.clear();
    h->computation = false; h->prepared = false;
    drop_plan(h);
    return L3D_OK;
}

static int add_common(l3d_line3d* h, uint32_t id, unsigned width, unsigned height, const float* segs, int n,
                      const double* K, const double* R, const double* t, int n_links)
{
    if (!h) return L3D_ERR_INVALID;
    // the guards of addImage, line3D.cc:101-127 (print-and-return in the reference; a status here)
    if (h->computation) return h->fail(L3D_ERR_INVALID, "reconstruction already performed! cannot add more images (try reset first)");
    if (h->views.count(id)) return h->fail(L3D_ERR_INVALID, "imageID already in use!");
    if (n_links == 0) return h->fail(L3D_ERR_INVALID, "unlinked images cannot be added!");
    if (width == 0 || height == 0) return h->fail(L3D_ERR_INVALID, "image is empty!");
    if (n <= 0 || !segs || !K || !R || !t) return h->fail(L3D_ERR_INVALID, "no segments");   // detectLineSegments failed: no view, :186-190
    return make_view(h, id, width, height, segs, n, K, R, t);
}

// addImage_fixed_sim with precomputed segments (the detector is out of scope), line3D.cc:220-342
int l3d_line3d_add_image_fixed_sim(l3d_line3d* h, uint32_t id, unsigned width, unsigned height, const float* segs, int n,
                                   const double* K, const double* R, const double* t,
                                   const uint32_t* sim_ids, const float* sims, int n_sims)
{
    int rc = add_common(h, id, width, height, segs, n, K, R, t, n_sims);
    if (rc) return rc;
    for (int i = 0; i < n_sims; ++i)                       // setViewSimilarity, :1938-1946
        if (sims[i] > 0.01f) h->view_similarities[id][sim_ids[i]] = sims[i];
    return L3D_OK;
}

// addImage with precomputed segments, line3D.cc:95-217
int l3d_line3d_add_image(l3d_line3d* h, uint32_t id, unsigned width, unsigned height, const float* segs, int n,
                         const double* K, const double* R, const double* t, const uint32_t* worldpoints, int n_wps)
{
    int rc = add_common(h, id, width, height, segs, n, K, R, t, n_wps);
    if (rc) return rc;
    process_worldpoints(h, id, worldpoints, n_wps);
    return L3D_OK;
}

// addImage when the segment cache exists, line3D.cc:160-168: segments and collinearities come from the file
int l3d_line3d_add_image_cached(l3d_line3d* h, uint32_t id, unsigned width, unsigned height, const l3d_segment_cache* cache,
                                const double* K, const double* R, const double* t, const uint32_t* worldpoints, int n_wps)
{
    if (!h) return L3D_ERR_INVALID;
    if (!cache) return h->fail(L3D_ERR_INVALID, "null segment cache");
    const int n = l3d_segment_cache_num_segments(cache), nc = l3d_segment_cache_num_collinearities(cache);
    std::vector<float> segs((size_t)n * 4 + 1), cw((size_t)nc + 1);
    std::vector<int32_t> ci((size_t)nc + 1), cj((size_t)nc + 1);
    l3d_segment_cache_get(cache, segs.data(), ci.data(), cj.data(), cw.data());
    if (h->computation) return h->fail(L3D_ERR_INVALID, "reconstruction already performed! cannot add more images (try reset first)");
    if (h->views.count(id)) return h->fail(L3D_ERR_INVALID, "imageID already in use!");
    if (n_wps == 0) return h->fail(L3D_ERR_INVALID, "unlinked images cannot be added!");
    if (width == 0 || height == 0) return h->fail(L3D_ERR_INVALID, "image is empty!");
    if (n <= 0 || !K || !R || !t) return h->fail(L3D_ERR_INVALID, "no segments");
    int rc = make_view(h, id, width, height, segs.data(), n, K, R, t, ci.data(), cj.data(), cw.data(), nc);
    if (rc) return rc;
    process_worldpoints(h, id, worldpoints, n_wps);
    return L3D_OK;
}

// the cache decisions of addImage / addImage_fixed_sim, line3D.cc:128-199: returns 1 when the view was added from the cache file,
// 0 when the caller's segments are to be used (cache_path set when the file is to be written), < 0: error code negated
static int add_from_cache_or_plan_write(l3d_line3d* h, uint32_t id, unsigned width, unsigned height, const double* K, const double* R, const double* t,
                                        const char* data_directory, int max_img_width, int load_and_store, std::string& cache_path, std::string* expected = nullptr)
{
    unsigned new_w = width, new_h = height;
    if (max_img_width > 0 && (int)std::max(width, height) > max_img_width) {                 // :133-138
        const float scale = float(max_img_width) / fmaxf((float)height, (float)width);
        new_w = (unsigned)roundf(float(width) * scale);
        new_h = (unsigned)roundf(float(height) * scale);
    }
    char name[160];
    if (l3d_segment_cache_filename(id, new_w, new_h, h->use_collinearity ? 1 : 0, name, sizeof(name)) != L3D_OK) return -L3D_ERR_INVALID;
    const std::string file = std::string(data_directory ? data_directory : "") + name;
    if (expected) *expected = file;
    FILE* f = fopen(file.c_str(), "rb");
    const bool exists = f != nullptr;
    if (f) fclose(f);
    cache_path.clear();
    if (exists && !load_and_store) { remove(file.c_str()); return 0; }                        // :153-156
    if (exists) {                                                                            // :159-168
        l3d_segment_cache* cache = nullptr;
        int rc = l3d_segment_cache_read(file.c_str(), &cache);
        if (rc != L3D_OK) { h->fail(rc, l3d_segment_cache_last_error(cache)); l3d_segment_cache_free(cache); return -rc; }   // (the reference exits, serialization.h:63)
        const int n = l3d_segment_cache_num_segments(cache), nc = l3d_segment_cache_num_collinearities(cache);
        std::vector<float> segs((size_t)n * 4 + 1), cw((size_t)nc + 1);
        std::vector<int32_t> ci((size_t)nc + 1), cj((size_t)nc + 1);
        l3d_segment_cache_get(cache, segs.data(), ci.data(), cj.data(), cw.data());
        l3d_segment_cache_free(cache);
        if (n <= 0) return -h->fail(L3D_ERR_INVALID, "no segments");
        rc = make_view(h, id, width, height, segs.data(), n, K, R, t, ci.data(), cj.data(), cw.data(), nc);
        return rc ? -rc : 1;
    }
    if (load_and_store) cache_path = file;                                                   // :180-182
    return 0;
}

int l3d_line3d_add_image_ex(l3d_line3d* h, uint32_t id, unsigned width, unsigned height, const float* segs, int n, const double* K, const double* R,
                            const double* t, const uint32_t* worldpoints, int n_wps, const char* data_directory, int max_img_width, int load_and_store)
{
    if (!h) return L3D_ERR_INVALID;
    if (h->computation) return h->fail(L3D_ERR_INVALID, "reconstruction already performed! cannot add more images (try reset first)");
    if (h->views.count(id)) return h->fail(L3D_ERR_INVALID, "imageID already in use!");
    if (n_wps == 0) return h->fail(L3D_ERR_INVALID, "unlinked images cannot be added!");
    if (width == 0 || height == 0 || !K || !R || !t) return h->fail(L3D_ERR_INVALID, "image is empty!");
    std::string cache_path, expected;
    const int from_cache = add_from_cache_or_plan_write(h, id, width, height, K, R, t, data_directory, max_img_width, load_and_store, cache_path, &expected);
    if (from_cache < 0) return -from_cache;
    if (!from_cache) {
        // the image-typed overloads of the facade bring no segments: where the reference would detect them (line3D.cc:169-190) this library stops
        if (n <= 0 || !segs)
            return h->fail(L3D_ERR_INVALID, ("image [" + std::to_string(id) + "]: no segment cache " + expected + " and no segments given -- line segment "
                                             "detection is not part of this library (run the reference once with loadAndStoreSegments, or pass the segments)").c_str());
        const int rc = add_common(h, id, width, height, segs, n, K, R, t, n_wps);
        if (rc) return rc;
        h->views[id].cache_to_write = cache_path;
    }
    process_worldpoints(h, id, worldpoints, n_wps);
    return L3D_OK;
}

int l3d_line3d_add_image_fixed_sim_ex(l3d_line3d* h, uint32_t id, unsigned width, unsigned height, const float* segs, int n, const double* K, const double* R,
                                      const double* t, const uint32_t* sim_ids, const float* sims, int n_sims, const char* data_directory, int max_img_width,
                                      int load_and_store)
{
    if (!h) return L3D_ERR_INVALID;
    if (h->computation) return h->fail(L3D_ERR_INVALID, "reconstruction already performed! cannot add more images (try reset first)");
    if (h->views.count(id)) return h->fail(L3D_ERR_INVALID, "imageID already in use!");
    if (n_sims == 0) return h->fail(L3D_ERR_INVALID, "unlinked images cannot be added!");
    if (width == 0 || height == 0 || !K || !R || !t) return h->fail(L3D_ERR_INVALID, "image is empty!");
    std::string cache_path, expected;
    const int from_cache = add_from_cache_or_plan_write(h, id, width, height, K, R, t, data_directory, max_img_width, load_and_store, cache_path, &expected);
    if (from_cache < 0) return -from_cache;
    if (!from_cache) {
        // the image-typed overloads of the facade bring no segments: where the reference would detect them (line3D.cc:169-190) this library stops
        if (n <= 0 || !segs)
            return h->fail(L3D_ERR_INVALID, ("image [" + std::to_string(id) + "]: no segment cache " + expected + " and no segments given -- line segment "
                                             "detection is not part of this library (run the reference once with loadAndStoreSegments, or pass the segments)").c_str());
        const int rc = add_common(h, id, width, height, segs, n, K, R, t, n_sims);
        if (rc) return rc;
        h->views[id].cache_to_write = cache_path;
    }
    for (int i = 0; i < n_sims; ++i)                       // setViewSimilarity, :1938-1946
        if (sims[i] > 0.01f) h->view_similarities[id][sim_ids[i]] = sims[i];
    return L3D_OK;
}

int l3d_line3d_num_cameras(const l3d_line3d* h) { return h ? (int)h->views.size() : 0; }

// verbose: the counters compute_pairwise_matches prints per view (cudawrapper.cu:953,1114; line3D.cc:652), as totals of the pass -- the resident
// chain never hands a view's lists to the host
static int match_views_reported(l3d_line3d* h)
{
    const int rc = match_views(h);
    if (!rc && h->verbose)
        printf("[L3D] #raw_matches:          %.0f (all views)\n[L3D] #filtered_matches (2): %.0f (all views)\n[L3D] segment pairs tested:  %.0f\n",
               h->stat_raw, h->stat_kept, h->stat_pairs);
    return rc;
}

int l3d_line3d_prepare(l3d_line3d* h) { return h ? prepare(h) : L3D_ERR_INVALID; }
int l3d_line3d_match_views(l3d_line3d* h)
{
    if (!h || !h->prepared) return h ? h->fail(L3D_ERR_INVALID, "prepare first") : L3D_ERR_INVALID;
    return match_views_reported(h);
}
int l3d_line3d_finish(l3d_line3d* h, int perform_diffusion)
{
    if (!h || !h->prepared) return h ? h->fail(L3D_ERR_INVALID, "prepare first") : L3D_ERR_INVALID;
    const double t0 = now_s();
    if (h->partitioned && !h->part_exchange) return h->fail(L3D_ERR_INVALID, "finish: matchViews' products are partitioned over the ranks (l3d_line3d_finish_sharded)");
    if (h->resident_products) { const int rg = greedy_selection_resident(h); if (rg) return rg; }
    else greedy_selection(h);                              // optimizeLocalMatches, :888-896
    if (hopt(h).timing) fprintf(stderr, "[l3d finish] %-28s %8.2f ms\n", "greedy selection", (now_s() - t0) * 1e3);
    const int rc = cluster_segments_2D(h, perform_diffusion != 0);
    if (hopt(h).timing) fprintf(stderr, "[l3d finish] %-28s %8.2f ms\n", "total", (now_s() - t0) * 1e3);
    if (!rc && h->verbose)      // line3D.cc:959-961, 1226-1229, 1251
        printf("[L3D] #clusterable_segments:  %zu\n[L3D] A: #num_entries = %zu\n[L3D] A: #num_rows    = %zu\n[L3D] %zu 3D lines found!\n",
               h->hyps.size(), h->n_edges, h->local2global.size(), h->result.size());
    return rc;
}
// Line3D::compute3Dmodel, line3D.cc:345-374
int l3d_line3d_compute3Dmodel(l3d_line3d* h, int perform_diffusion)
{
    if (!h) return L3D_ERR_INVALID;
    int rc = prepare(h);
    if (!rc) rc = match_views_reported(h);
    if (!rc) rc = l3d_line3d_finish(h, perform_diffusion);
    return rc;
}

// ---- step-wise matching (multi-GPU: every rank computes a source-segment range of each view, the
// kept lists are all-gathered, every rank commits the same merged list) ---------------------------
int l3d_line3d_match_begin(l3d_line3d* h, int* n_order)
{
    if (!h || !h->prepared) return h ? h->fail(L3D_ERR_INVALID, "prepare first") : L3D_ERR_INVALID;
    match_begin(h);
    if (n_order) *n_order = (int)h->order.size();
    return L3D_OK;
}
int l3d_line3d_match_order(l3d_line3d* h, uint32_t* ids, int* n_segments)
{
    if (!h) return L3D_ERR_INVALID;
    for (size_t i = 0; i < h->order.size(); ++i) { if (ids) ids[i] = h->order[i]; if (n_segments) n_segments[i] = h->views[h->order[i]].S(); }
    return L3D_OK;
}
// number of neighbours still to be matched from this view (line3D.cc:732-736); 0 = the early-return case
int l3d_line3d_view_num_to_be_matched(l3d_line3d* h, uint32_t view_id)
{
    if (!h) return -1;
    auto it = h->visual_neighbors.find(view_id);
    if (it == h->visual_neighbors.end()) return -1;
    int n = 0;
    for (uint32_t nb : it->second) if (!h->matched.count(((uint64_t)view_id << 32) | nb)) ++n;
    return n;
}
int l3d_line3d_match_view_compute(l3d_line3d* h, uint32_t view_id, int seg_begin, int seg_end,
                                  l3d_match** out, int* n_out, float* median, float** best, int* n_best)
{
    if (!h) return L3D_ERR_INVALID;
    View* v = h->find_view(view_id);
    if (!v) return h->fail(L3D_ERR_INVALID, "unknown view");
    return compute_view(h, *v, seg_begin, seg_end, out, n_out, median, best, n_best);
}
// best_depths: the merged depth pairs of all ranges (2*n_best floats); pass n_best < 0 to use `median` as is
int l3d_line3d_match_view_commit(l3d_line3d* h, uint32_t view_id, const l3d_match* matches, int n,
                                 const float* best_depths, int n_best, float median)
{
    if (!h) return L3D_ERR_INVALID;
    View* v = h->find_view(view_id);
    if (!v) return h->fail(L3D_ERR_INVALID, "unknown view");
    if (n_best >= 0) {
        median = -1.0f;                                    // cudawrapper.cu:1066-1073
        if (n_best > 0) {
            std::vector<float> d(best_depths, best_depths + (size_t)n_best * 2);
            std::sort(d.begin(), d.end());
            median = d[d.size() / 2];
        }
    }
    h->stat_last_tbm = l3d_line3d_view_num_to_be_matched(h, view_id);
    commit_view(h, *v, matches, n, median);
    return L3D_OK;
}
// ---- matchViews as the resident chain sharded over ranks (one process per GPU; see include/line3d_amd.h) -----------
int l3d_line3d_shard_open(l3d_line3d* h, int rank, int world, int slot_records, int* n_views, size_t* slot_bytes)
{
    if (!h || !h->prepared) return h ? h->fail(L3D_ERR_INVALID, "prepare first") : L3D_ERR_INVALID;
    if (h->shard_plan_) return h->fail(L3D_ERR_INVALID, "a sharded chain is already open");
    match_begin(h);
    ChainPlan* P = get_plan(h);
    if (!P) return h->fail(L3D_ERR_INVALID, "schedule is not static (early-return quirk): use the per-view path");
    P->t0 = now_s();
    int rc = l3d_shard_chain_open(h->ctx, P->cv.data(), (int)P->n, rank, world, slot_records, &P->shard, slot_bytes);
    if (rc) return h->fail(rc, std::string("shard_chain_open: ") + l3d_last_error(h->ctx));
    start_finalizer(h, *P);
    h->shard_plan_ = P;
    if (n_views) *n_views = (int)P->n;
    return L3D_OK;
}
int l3d_line3d_shard_view_verified(l3d_line3d* h, int k)
{
    if (!h || !h->shard_plan_) return -1;
    ChainPlan* P = static_cast<ChainPlan*>(h->shard_plan_);
    if (k < 0 || (size_t)k >= P->n) return -1;
    return P->n_tbm[(size_t)k] > 0 ? 1 : 0;
}
int l3d_line3d_shard_enqueue(l3d_line3d* h, int k, void* send_slot, const void* gathered_base)
{
    if (!h || !h->shard_plan_) return L3D_ERR_INVALID;
    int rc = l3d_shard_chain_enqueue(static_cast<ChainPlan*>(h->shard_plan_)->shard, k, send_slot, gathered_base);
    return rc ? h->fail(rc, std::string("shard_chain_enqueue: ") + l3d_last_error(h->ctx)) : L3D_OK;
}
int l3d_line3d_shard_mark(l3d_line3d* h, int k)
{
    if (!h || !h->shard_plan_) return L3D_ERR_INVALID;
    return l3d_shard_chain_mark(static_cast<ChainPlan*>(h->shard_plan_)->shard, k);
}
// host bookkeeping of view k on this rank (optional per rank; views must be fetched in order)
int l3d_line3d_shard_fetch(l3d_line3d* h, int k)
{
    if (!h || !h->shard_plan_) return L3D_ERR_INVALID;
    ChainPlan* P = static_cast<ChainPlan*>(h->shard_plan_);
    int rc = l3d_shard_chain_fetch(P->shard, k, chain_callback, &P->user);
    return rc ? h->fail(rc, std::string("shard_chain_fetch: ") + l3d_last_error(h->ctx)) : L3D_OK;
}
// committed != 0: this rank fetched every view -> its host state is finalised (finish() may follow)
int l3d_line3d_shard_close(l3d_line3d* h, int committed)
{
    if (!h || !h->shard_plan_) return L3D_ERR_INVALID;
    ChainPlan* P = static_cast<ChainPlan*>(h->shard_plan_);
    int rc = l3d_shard_chain_close(P->shard);
    finish_chain_host(h, *P, committed != 0 && rc == L3D_OK);
    if (h->pot_check_failed && rc == L3D_OK) rc = h->fail(L3D_ERR_INVALID, "L3D_CHECK_POT: a potential-correspondence list is not in normal form");
    double st[4];
    l3d_last_stats(h->ctx, st);
    h->stat_pairs += st[0];
    h->stat_raw += st[1];
    h->t_match = now_s() - P->t0;
    P->shard = nullptr;
    h->shard_plan_ = nullptr;
    return rc;
}
int l3d_line3d_shard_run(l3d_line3d* h, int rank, int world, int slot_records, l3d_exchange_fn exchange, void* exchange_user, int commit,
                         const void** gathered_out, size_t* slot_bytes_out)
{
    if (!h) return L3D_ERR_INVALID;
    // A capacity failure is a verdict all ranks share (l3d_shard_chain_info): every rank reopens with the same, larger
    // capacities and runs again -- the bookkeeping of the failed attempt is dropped by the reopen (match_begin).
    size_t cand_cap_next = 0, arena_cap_next = 0;
    int rc = L3D_OK;
    // sizes a capacity verdict of an earlier pass of this job taught us (identical on every rank: the verdict is shared)
    if (h->shard_world_seen == world) { slot_records = std::max(slot_records, h->shard_slot_records_seen); cand_cap_next = h->shard_cand_cap_seen; }
    else { h->shard_world_seen = world; h->shard_slot_records_seen = 0; h->shard_cand_cap_seen = 0; }
    for (int attempt = 0; attempt < 4; ++attempt) {
        int n_views = 0;
        size_t slot_bytes = 0;
        const double t0 = now_s();
        if (cand_cap_next) l3d_set_chain_capacities(h->ctx, cand_cap_next, 0);
        rc = l3d_line3d_shard_open(h, rank, world, slot_records, &n_views, &slot_bytes);
        if (cand_cap_next) l3d_set_chain_capacities(h->ctx, 0, 0);
        if (rc) return rc;
        const double t1 = now_s();
        ChainPlan* P = static_cast<ChainPlan*>(h->shard_plan_);
        const bool host_commit = commit == 1;
        if (commit == 3) {       // partitioned: this rank keeps what its block of views needs (l3d_shard_chain_partition), nothing else
            // (options part_vrank / part_vworld at world 1: the block another job's rank would own -- one rank's share of a job too big for one GPU, on one GPU)
            const int vw = world == 1 && hopt(h).part_vworld > 0 ? hopt(h).part_vworld : world, vr = vw != world ? std::max(0, std::min(hopt(h).part_vrank, vw - 1)) : rank;
            rc = l3d_shard_chain_partition(P->shard, (int)(((long long)n_views * vr) / vw), (int)(((long long)n_views * (vr + 1)) / vw));
            if (rc) { const std::string m = std::string("shard_chain_partition: ") + l3d_last_error(h->ctx); l3d_line3d_shard_close(h, 0); return h->fail(rc, m); }
        }
        if (arena_cap_next) l3d_set_chain_capacities(h->ctx, 0, arena_cap_next);          // (the compact arena of the slot ring: read by the run)
        rc = l3d_shard_chain_run(P->shard, exchange, exchange_user, host_commit ? chain_callback : nullptr, host_commit ? &P->user : nullptr);
        if (arena_cap_next) l3d_set_chain_capacities(h->ctx, 0, 0);
        std::string msg = rc ? std::string("shard_chain_run: ") + l3d_last_error(h->ctx) : std::string();
        if (rc == L3D_OK && (commit == 2 || commit == 3)) {
            // commit on the device: this rank builds matchViews' products from the gathered slots (every rank may), no list goes to the host
            std::vector<uint32_t> ids; std::vector<int32_t> base;
            dense_map(h, ids, base);
            l3d_dense_map map;
            map.n_views = (int32_t)ids.size(); map.view_ids = ids.data(); map.seg_base = base.data();
            h->chain_summary.assign(P->n, l3d_chain_summary());
            rc = l3d_shard_chain_products(P->shard, &map, h->chain_summary.data(), &h->resident_n_pot);
            if (rc) msg = std::string("shard_chain_products: ") + l3d_last_error(h->ctx);
            else {
                if (commit == 3) { h->partitioned = true; h->part_exchange = exchange; h->part_user = exchange_user; }      // (a share: nothing to check against a host construction)
                rc = adopt_resident_products(h, *P);
                if (rc) { msg = h->err; h->partitioned = false; }
            }
        }
        size_t cand_cap = 0; int bits = 0, max_cand = 0, max_kept = 0, recs = slot_records;
        l3d_shard_chain_info(P->shard, &cand_cap, &recs, &bits, &max_cand, &max_kept);
        const long long P_shard_arena = l3d_shard_chain_arena_needed(P->shard);
        if (gathered_out) *gathered_out = l3d_shard_chain_gathered(P->shard);
        if (slot_bytes_out) *slot_bytes_out = slot_bytes;
        const double t2 = now_s();
        const int rc2 = l3d_line3d_shard_close(h, commit == 1 && rc == L3D_OK);
        if (hopt(h).timing) fprintf(stderr, "[l3d shard_run] open (schedule, tables, arenas) %.2f  run %.2f  close (finalise host state) %.2f ms\n",
                                          (t1 - t0) * 1e3, (t2 - t1) * 1e3, (now_s() - t2) * 1e3);
        if (rc == L3D_OK) return rc2;
        h->fail(rc, msg);
        if (hopt(h).timing) fprintf(stderr, "[l3d shard_run] attempt %d failed (%d): %s\n", attempt, rc, msg.c_str());
        if (rc != L3D_ERR_NOMEM || (bits & 4 && !(bits & 11)) || !(bits & 11)) return rc;    // not a capacity verdict: nothing a retry would change
        if (bits & 8) { const long long need = P_shard_arena; arena_cap_next = (size_t)need + (size_t)need / 4 + 65536; }
        if ((bits & 2) && exchange == l3d_exchange_replay) return rc;                        // recorded blocks have the recorded slot size: the caller records again with more room
        if (bits & 1) cand_cap_next = h->shard_cand_cap_seen = std::max(cand_cap * 2, (size_t)max_cand + (size_t)max_cand / 4 + 65536);
        if (bits & 2) slot_records = h->shard_slot_records_seen = std::max(slot_records * 2, max_kept + max_kept / 4 + 1024);
    }
    return rc;
}
// matchViews with the VIEWS sharded over the ranks in blocks, each block started cold a few windows early and the speculation verified
// (l3d_match_chain_blocks).  *verdict = 0: this rank holds matchViews' products as after the single-GPU resident chain; 1: the speculation
// did not hold on this scene, nothing was committed -- the caller runs l3d_line3d_shard_run (every rank gets the same verdict).
// warmup_views < 0: eight neighbour windows.
static int block_run_impl(l3d_line3d* h, int rank, int world, int warmup_views, l3d_exchange_fn exchange, void* exchange_user, int* verdict, bool partition)
{
    if (!h || !verdict) return L3D_ERR_INVALID;
    *verdict = 1;
    const double t0 = now_s();
    match_begin(h);
    h->partitioned = false;
    ChainPlan* Pp = get_plan(h);
    if (!Pp) return L3D_OK;                               // (a schedule the chain cannot express: the caller's other paths handle it)
    ChainPlan& P = *Pp;
    int window = 1;
    for (size_t k = 0; k < P.n; ++k) for (int si : P.src_idx[k]) window = std::max(window, (int)k - si);
    // (round 4: eight windows -- any miss cost the pass.  Round 5: a block whose speculation fails is re-run warm, all missed blocks at once, so a
    // miss costs one more block time, not the pass.  Still eight: the chain's memory was measured at 3-7 windows (profiles/r4_speculate_*.txt), and on
    // the 512-view scene at 8 ranks a warm-up of 4 windows re-runs 7 blocks, of 6 windows 3, of 8 windows none (profiles/r5_warmup_needed_512.txt) -- what a
    // shorter warm-up saves on every rank (2 windows = 12 views) a single miss gives back five times over (a block = 64 views))
    if (warmup_views < 0) warmup_views = 8 * window;
    std::vector<uint32_t> ids; std::vector<int32_t> base;
    dense_map(h, ids, base);
    l3d_dense_map map;
    map.n_views = (int32_t)ids.size(); map.view_ids = ids.data(); map.seg_base = base.data();
    h->chain_summary.assign(P.n, l3d_chain_summary());
    h->resident_products = false;
    const double t1 = now_s();
    int rc = partition ? l3d_match_chain_partition(h->ctx, P.cv.data(), (int)P.n, &map, h->chain_summary.data(), &h->resident_n_pot, rank, world, warmup_views, window,
                                                   exchange, exchange_user, verdict)
                       : l3d_match_chain_blocks(h->ctx, P.cv.data(), (int)P.n, &map, h->chain_summary.data(), &h->resident_n_pot, rank, world, warmup_views, window,
                                                exchange, exchange_user, verdict);
    h->t_gpu_call += now_s() - t1;
    if (rc) return h->fail(rc, std::string(partition ? "match_chain_partition: " : "match_chain_blocks: ") + l3d_last_error(h->ctx));
    if (*verdict != 0) return L3D_OK;
    h->partitioned = partition;
    h->part_exchange = exchange; h->part_user = exchange_user;
    rc = adopt_resident_products(h, P);
    if (rc) return rc;
    double st[4];
    l3d_last_stats(h->ctx, st);
    h->stat_pairs += st[0];
    h->stat_raw += st[1];
    h->t_match = now_s() - t0;
    return L3D_OK;
}
int l3d_line3d_block_run(l3d_line3d* h, int rank, int world, int warmup_views, l3d_exchange_fn exchange, void* exchange_user, int* verdict)
{
    return block_run_impl(h, rank, world, warmup_views, exchange, exchange_user, verdict, false);
}
// matchViews sharded by blocks of views with NOTHING replicated (l3d_match_chain_partition): this rank holds its block's share of the kept records
// and of matchViews' products; the rest of compute3Dmodel is collective -- l3d_line3d_finish_sharded on every rank
int l3d_line3d_partition_run(l3d_line3d* h, int rank, int world, int warmup_views, l3d_exchange_fn exchange, void* exchange_user, int* verdict)
{
    return block_run_impl(h, rank, world, warmup_views, exchange, exchange_user, verdict, true);
}
// Line3D::compute3Dmodel's tail after l3d_line3d_partition_run, on every rank of the job: greedy selection on the views this rank holds, the affinity
// fill sharded by source key (l3d_affinity_fill_sharded: five small all-gathers), then -- replicas, every rank from the same edge list -- diffusion,
// clustering, line fit.  Every rank ends with the whole result.
int l3d_line3d_finish_sharded(l3d_line3d* h, int perform_diffusion, l3d_exchange_fn exchange, void* exchange_user)
{
    if (!h || !h->prepared) return h ? h->fail(L3D_ERR_INVALID, "prepare first") : L3D_ERR_INVALID;
    if (!h->partitioned || !h->resident_products) return h->fail(L3D_ERR_INVALID, "finish_sharded: matchViews did not run partitioned (l3d_line3d_partition_run)");
    if (exchange) { h->part_exchange = exchange; h->part_user = exchange_user; }
    if (!h->part_exchange) return h->fail(L3D_ERR_INVALID, "finish_sharded: no exchange");
    return l3d_line3d_finish(h, perform_diffusion);
}
// how the last l3d_line3d_match_views ran: 0 = the resident chain with its products on the device, 1 = the chain with host bookkeeping, 2 = per-view
// seam calls because the caller asked (l3d_line3d_set_sync_matching), 3 = per-view seam calls because the schedule is not static (-1: not yet)
int l3d_line3d_match_path(const l3d_line3d* h) { return h ? h->last_match_path : -1; }
int l3d_line3d_match_end(l3d_line3d* h) { if (!h) return L3D_ERR_INVALID; finalize_matching(h); return L3D_OK; }

// performClustering (clustering.h:125, clustering.cc:6-47) as a host entry point: labels[k] = find(k)
int l3d_perform_clustering(const l3d_edge* edges, int n_edges, int num_nodes, float c, int32_t* labels)
{
    if (n_edges < 0 || num_nodes < 0 || (n_edges > 0 && !edges) || (num_nodes > 0 && !labels)) return L3D_ERR_INVALID;
    for (int k = 0; k < n_edges; ++k)
        if (edges[k].i < 0 || edges[k].i >= num_nodes || edges[k].j < 0 || edges[k].j >= num_nodes) return L3D_ERR_INVALID;
    std::vector<int> lab;
    perform_clustering(edges, (size_t)n_edges, num_nodes, c, lab);
    for (int k = 0; k < num_nodes; ++k) labels[k] = lab[(size_t)k];
    return L3D_OK;
}

// ---- results ---------------------------------------------------------------------------------
int l3d_line3d_result_sizes(const l3d_line3d* h, int* n_lines, int* n_seg3d, int* n_seg2d)
{
    if (!h) return L3D_ERR_INVALID;
    int a = 0, b = 0;
    for (auto& l : h->result) { a += (int)l.segs3D.size(); b += (int)l.segs2D.size(); }
    if (n_lines) *n_lines = (int)h->result.size();
    if (n_seg3d) *n_seg3d = a;
    if (n_seg2d) *n_seg2d = b;
    return L3D_OK;
}
// per line: counts; seg3d: 6 doubles each (P1,P2); seg2d: (camID, segID) pairs  -- Line3D::getResult
int l3d_line3d_get_result(const l3d_line3d* h, int* line_n3d, int* line_n2d, double* seg3d, uint32_t* seg2d)
{
    if (!h) return L3D_ERR_INVALID;
    size_t a = 0, b = 0, li = 0;
    for (auto& l : h->result) {
        line_n3d[li] = (int)l.segs3D.size(); line_n2d[li] = (int)l.segs2D.size(); ++li;
        for (auto& s : l.segs3D) { seg3d[a++] = s.first.x; seg3d[a++] = s.first.y; seg3d[a++] = s.first.z; seg3d[a++] = s.second.x; seg3d[a++] = s.second.y; seg3d[a++] = s.second.z; }
        for (Key k : l.segs2D) { seg2d[b++] = kcam(k); seg2d[b++] = kseg(k); }
    }
    return L3D_OK;
}
// Line3D::getSegment2D, line3D.cc:2004-2013
// Line3D::save3DLinesAsSTL / save3DLinesAsTXT (line3D.cc:384-473; TXT format: README.txt:177-185) for the current result.
// Numbers are formatted the way the reference formats them: "%e" in the STL file, the stream default (6 significant
// digits, "%g") in the TXT file; lines without 3-D segments are skipped in the TXT file only.
int l3d_line3d_save_result(const l3d_line3d* h, const char* filename, int format)
{
    if (!h || !filename || (format != L3D_FORMAT_STL && format != L3D_FORMAT_TXT)) return L3D_ERR_INVALID;
    FILE* f = fopen(filename, "w");
    if (!f) return L3D_ERR_INVALID;
    if (format == L3D_FORMAT_STL) {
        fprintf(f, "solid lineModel\n");
        for (auto& l : h->result)
            for (auto& sg : l.segs3D) {
                fprintf(f, " facet normal 1.0e+000 0.0e+000 0.0e+000\n  outer loop\n");
                fprintf(f, "   vertex %e %e %e\n", sg.first.x, sg.first.y, sg.first.z);
                fprintf(f, "   vertex %e %e %e\n", sg.second.x, sg.second.y, sg.second.z);
                fprintf(f, "   vertex %e %e %e\n", sg.first.x, sg.first.y, sg.first.z);
                fprintf(f, "  endloop\n endfacet\n");
            }
        fprintf(f, "endsolid lineModel\n");
    } else {
        for (auto& l : h->result) {
            if (l.segs3D.empty()) continue;
            fprintf(f, "%zu ", l.segs3D.size());
            for (auto& sg : l.segs3D) fprintf(f, "%g %g %g %g %g %g ", sg.first.x, sg.first.y, sg.first.z, sg.second.x, sg.second.y, sg.second.z);
            fprintf(f, "%zu ", l.segs2D.size());
            for (Key k : l.segs2D) {
                float c[4] = { 0, 0, 0, 0 };
                (void)l3d_line3d_get_segment2D(h, kcam(k), kseg(k), c);
                fprintf(f, "%u %u %g %g %g %g ", kcam(k), kseg(k), (double)c[0], (double)c[1], (double)c[2], (double)c[3]);
            }
            fprintf(f, "\n");
        }
    }
    return fclose(f) == 0 ? L3D_OK : L3D_ERR_INVALID;
}

int l3d_line3d_get_segment2D(const l3d_line3d* h, uint32_t cam, uint32_t seg, float out[4])
{
    if (out) out[0] = out[1] = out[2] = out[3] = 0.0f;        // zeroed before any early return
    if (!h || !out) return L3D_ERR_INVALID;
    auto it = h->views.find(cam);
    if (it == h->views.end() || seg >= (uint32_t)it->second.S()) return L3D_ERR_INVALID;
    memcpy(out, &it->second.segs[(size_t)seg * 4], 16);
    return L3D_OK;
}

// ---- inspection for tests / bench ----------------------------------------------------------------
int l3d_line3d_set_sync_matching(l3d_line3d* h, int on) { if (!h) return L3D_ERR_INVALID; h->force_sync = on != 0; return L3D_OK; }
int l3d_line3d_keep_view_matches(l3d_line3d* h, int on) { if (!h) return L3D_ERR_INVALID; h->keep_view_matches = on != 0; return L3D_OK; }
int l3d_line3d_view_matches(const l3d_line3d* h, uint32_t view_id, const l3d_match** m, int* n, float* median)
{
    if (!h) return L3D_ERR_INVALID;
    auto it = h->view_matches.find(view_id);
    if (m) *m = it == h->view_matches.end() ? nullptr : it->second.data();
    if (n) *n = it == h->view_matches.end() ? 0 : (int)it->second.size();
    auto vt = h->views.find(view_id);
    if (median && vt != h->views.end()) *median = vt->second.median_depth;
    return L3D_OK;
}
int l3d_line3d_affinity(const l3d_line3d* h, const l3d_edge** A, int* nnz, int* n_nodes)
{
    if (!h) return L3D_ERR_INVALID;
    if (A) { if (int rc = ensure_edges(const_cast<l3d_line3d*>(h))) return rc; *A = h->A.data(); }
    if (nnz) *nnz = (int)h->n_edges;
    if (n_nodes) *n_nodes = (int)h->local2global.size();
    return L3D_OK;
}
int l3d_line3d_products_sizes(const l3d_line3d* h, int* n_views, int* n_dense, int64_t* n_pot, int* n_hyp)
{
    if (!h) return L3D_ERR_INVALID;
    const bool on = h->resident_products;
    int nd = 0;
    for (const View* v : h->vlist) nd += v->S();
    if (n_views) *n_views = on ? (int)h->vlist.size() : 0;
    if (n_dense) *n_dense = on ? nd : 0;
    if (n_pot) *n_pot = on ? h->resident_n_pot : 0;
    if (n_hyp) *n_hyp = on ? (int)h->hyps.size() : 0;
    return L3D_OK;
}
int l3d_line3d_products_get(l3d_line3d* h, int32_t* seg_base, int64_t* pot_start, int32_t* pot_tgt, l3d_match* best, l3d_hypothesis* hyp, float* score)
{
    if (!h) return L3D_ERR_INVALID;
    if (!h->resident_products) return h->fail(L3D_ERR_INVALID, "no resident products");
    if (seg_base) { int b = 0; size_t i = 0; for (const View* v : h->vlist) { seg_base[i++] = b; b += v->S(); } seg_base[i] = b; }
    int rc = l3d_chain_products_get(h->ctx, pot_start, pot_tgt, best);
    if (!rc && (hyp || score) && !h->hyps.empty()) rc = l3d_products_hypotheses_get(h->ctx, hyp, score);
    return rc ? h->fail(rc, std::string("products_get: ") + l3d_last_error(h->ctx)) : L3D_OK;
}
/* stats[12]: pairs, raw candidates, kept, #hypotheses, t_match, t_gpu_call, t_commit, t_finalize, t_affinity, t_cluster, #edges, #lines */
int l3d_line3d_stats(const l3d_line3d* h, double* s)
{
    if (!h || !s) return L3D_ERR_INVALID;
    s[0] = h->stat_pairs; s[1] = h->stat_raw; s[2] = h->stat_kept; s[3] = (double)h->hyps.size();
    s[4] = h->t_match; s[5] = h->t_gpu_call; s[6] = h->t_commit; s[7] = h->t_finalize; s[8] = h->t_affinity; s[9] = h->t_cluster;
    s[10] = (double)h->n_edges; s[11] = (double)h->result.size();
    return L3D_OK;
}

}  // extern "C"
