// l3d_capi.hip -- the C ABI (include/line3d_amd.h): context, device arenas, launch sequencing.
// Host orchestration of the reference seam functions (cudawrapper.cu:833-1191) re-designed for a
// device-resident flow: no dense S x S buffer, no per-neighbour download, no host list sort --
// candidates are produced on the device already in (segment, camera, target) order.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include <thread>

#include "l3d_ctx.hpp"
#include "l3d_hostsort.hpp"

using namespace l3d;

static_assert(sizeof(l3d_match) == sizeof(Match), "l3d_match layout");
static_assert(sizeof(l3d_match) == 32, "l3d_match is 32 bytes");
static_assert(sizeof(l3d_hypothesis) == sizeof(Hypothesis), "l3d_hypothesis layout");

namespace {
const char* kProfNames = "pair_mask;row_count;scan;pair_fill;cand_move;exist;verify;verify_window;seg_post;kept_write;collinearity;collinearity_fill;rownorm;diffusion_step;similarity;tgt_rays;prod_keys;prod_sort;prod_rows;hypotheses;uf_components";
}  // namespace

namespace l3d {
// The ONE place the library reads the environment (called by l3d_ctx_create).
Options options_from_env()
{
    Options o;
#define X(field, env, def, doc) if (kCrossChecks || !option_is_crosscheck(env)) if (const char* e = getenv(env)) o.field = (*e == 0) ? 1 : atoi(e);
    L3D_OPTION_TABLE(X)
#undef X
    return o;
}
const Options& ctx_options(const l3d_ctx* c) { return c->opt; }
}  // namespace l3d

extern "C" {

int l3d_ctx_create(int device, l3d_ctx** out)
{
    if (!out) return L3D_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return L3D_ERR_NODEVICE;   // fail loudly: no CPU fallback
    if (device < 0 || device >= n) return L3D_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return L3D_ERR_HIP;
    l3d_ctx* c = new l3d_ctx();
    c->device = device;
    c->opt = options_from_env();                  // the one place the environment is read
    c->ch_pin_res.flags = hipHostMallocCoherent;  // result records: written by kernels, read by the host behind unfenced events (l3d_ctx.hpp: get_local_event)
    publish_tunables(c->opt);
    c->chain_ring = c->opt.chain_ring != 0;
    c->wedge_pretest = c->opt.pretest & 7;         // diagnostic: stage-1 filter mask
    {   // L3D_STREAM_PRIO=1: the chain's stream (per-view critical path) at the highest priority.  Measured on config 2: no
        // difference to plain streams (21.5 ms either way), so plain streams are the default.
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const bool prio = c->opt.stream_prio == 1;
        if (!prio || hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest) != hipSuccess) {
            (void)hipGetLastError();
            if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return L3D_ERR_HIP; }
        }
    }
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return L3D_ERR_HIP; }
    if (hipStreamCreateWithFlags(&c->stage1_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipStreamDestroy(c->copy_stream); (void)hipStreamDestroy(c->stream); delete c; return L3D_ERR_HIP;
    }
    if (c->opt.pair_stats && hipMalloc(reinterpret_cast<void**>(&c->pair_dbg), 4096) == hipSuccess) (void)hipMemset(c->pair_dbg, 0, 4096);
    *out = c;
    return L3D_OK;
}

int l3d_set_option(l3d_ctx* c, const char* name, int value)
{
    if (!c) return L3D_ERR_INVALID;
    int* f = option_field(c->opt, name);
    if (!f) return fail(c, L3D_ERR_INVALID, std::string("l3d_set_option: unknown option ") + (name ? name : "(null)"));
    *f = value;
    publish_tunables(c->opt);
    c->chain_ring = c->opt.chain_ring != 0;
    c->wedge_pretest = c->opt.pretest & 7;        // (l3d_set_pair_pretest writes opt.pretest too: setting another option does not reset its mask)
    return L3D_OK;
}

int l3d_get_option(l3d_ctx* c, const char* name, int* value)
{
    if (!c || !value) return L3D_ERR_INVALID;
    if (name && strcmp(name, "crosschecks") == 0) { *value = kCrossChecks ? 1 : 0; return L3D_OK; }     // (which build this is)
    if (name && strcmp(name, "shard_graph_launches") == 0) { *value = (int)std::min<long long>(c->shard_graph_launches, 0x7fffffff); return L3D_OK; }   // (a counter, not a switch)
    int* f = option_field(c->opt, name);
    if (!f) return fail(c, L3D_ERR_INVALID, std::string("l3d_get_option: unknown option ") + (name ? name : "(null)"));
    *value = *f;
    return L3D_OK;
}

void l3d_ctx_destroy(l3d_ctx* c)
{
    if (!c) return;
    if (c->stamps.p) {
        unsigned long long h[16];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, c->stamps.p, 128, hipMemcpyDeviceToHost);
        const double tot = (double)(h[0] + h[1] + h[2] + h[3] + h[4]);
        fprintf(stderr, "[l3d verify_window wave-cycles] build %.1f%%  setup %.1f%%  scan %.1f%%  drain %.1f%%  final %.1f%%  (waves %llu, avg %.0f cycles)\n",
                100 * h[0] / tot, 100 * h[1] / tot, 100 * h[2] / tot, 100 * h[3] / tot, 100 * h[4] / tot, h[5], tot / (double)(h[5] ? h[5] : 1));
        fprintf(stderr, "[l3d verify_window balance] the longest wave of all launches: %llu cycles = %.1f x the average wave\n", h[15], (double)h[15] / (tot / (double)(h[5] ? h[5] : 1)));
        fprintf(stderr, "[l3d verify_window walk] hypotheses %llu: entries walked per hypothesis %.2f, inside the d1 window %.2f, inside both windows %.2f, of another camera %.2f\n", h[14],
                h[10] / (double)(h[14] ? h[14] : 1), h[11] / (double)(h[14] ? h[14] : 1), h[12] / (double)(h[14] ? h[14] : 1), h[13] / (double)(h[14] ? h[14] : 1));
        fprintf(stderr, "[l3d verify_window pairs] evaluated %llu  pass the 3-D gate %.1f%%  confidence > 0 %.1f%%  confidence > 0.5 %.1f%%\n", h[6], 100.0 * h[7] / (double)(h[6] ? h[6] : 1),
                100.0 * h[8] / (double)(h[6] ? h[6] : 1), 100.0 * h[9] / (double)(h[6] ? h[6] : 1));
    }
    if (c->pair_dbg) {
        unsigned long long h[8];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, c->pair_dbg, 64, hipMemcpyDeviceToHost);
        fprintf(stderr, "[l3d pair_mask] pairs %llu  after wedge test %.3f%%  after overlap-bound test %.3f%%  candidates %.3f%%  (sector test off: source side %.2f%%, target side %.2f%% of the pairs)\n",
                h[0], 100.0 * h[1] / (double)h[0], 100.0 * h[2] / (double)h[0], 100.0 * h[3] / (double)h[0], 100.0 * h[4] / (double)h[0], 100.0 * h[5] / (double)h[0]);
        if (h[6]) {           // (diagnostic build -DL3D_BOUND_CHECK: pairs level 2 decided AGAINST the exact test)
            unsigned long long rec[8 * 48];
            (void)hipMemcpy(rec, c->pair_dbg + 8, sizeof(rec), hipMemcpyDeviceToHost);
            fprintf(stderr, "[l3d pair_mask] %llu pairs were decided by level 2 AGAINST the exact test; the first ones:\n", h[6]);
            for (unsigned long long i = 0; i < h[6] && i < 48; ++i) {
                float f[8]; memcpy(f, &rec[i * 8 + 4], 32);
                fprintf(stderr, "   view %llu src %llu cam %llu tgt %llu | bounds %.7g %.7g  exact overlaps %.7g %.7g | t-intervals src [%.7g, %.7g] tgt [%.7g, %.7g]\n",
                        rec[i * 8] >> 32, rec[i * 8] & 0xffffffffull, rec[i * 8 + 1] >> 32, rec[i * 8 + 1] & 0xffffffffull, f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
            }
        }
        (void)hipFree(c->pair_dbg);
    }
    if (c->opt.timing)
        fprintf(stderr, "[l3d timing] tables+stage1-launch %.1f  exist-sort %.1f  launch1b %.1f  sync1 %.1f  launch2 %.1f  sync2 %.1f  d2h-kept %.1f  median %.1f ms  (max candidates per segment %d)\n",
                c->tacc[0] * 1e3, c->tacc[1] * 1e3, c->tacc[2] * 1e3, c->tacc[3] * 1e3, c->tacc[4] * 1e3, c->tacc[5] * 1e3, c->tacc[6] * 1e3, c->tacc[7] * 1e3, c->mmax_seen);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    prof_resolve(c);
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (auto e : c->prof_event_pool) (void)hipEventDestroy(e);
    for (auto e : c->local_event_pool) (void)hipEventDestroy(e);
    DevBuf* bufs[] = { &c->src_segs, &c->tgt_segs, &c->tables, &c->tbm, &c->l2g, &c->exist, &c->mask, &c->rowcnt, &c->row_start,
                       &c->cand_meta, &c->cand_depths, &c->cand_conf, &c->kept_cnt, &c->kept_start, &c->best, &c->kept, &c->scal, &c->stamps, &c->vw_scratch, &c->vw_bstart, &c->vw_segstate, &c->ch_tables, &c->ch_mask, &c->ch_rowcnt, &c->ch_cursor, &c->ch_best, &c->ch_kept, &c->ch_keptcam, &c->ch_rt, &c->ch_rtinfo, &c->ch_rtjobs, &c->ch_existpart, &c->ch_res, &c->ch_flags, &c->ch_send, &c->ch_gathered, &c->ch_stage, &c->ch_rowA, &c->ch_ringA_meta, &c->ch_ringA_depths, &c->ch_segorder,
                       &c->ch_rays, &c->aff_hyp, &c->aff_first, &c->aff_pass_pairs, &c->aff_pass_w, &c->aff_l2g, &c->edges_keep, &c->g0, &c->g1, &c->g2, &c->g3, &c->g4, &c->g5, &c->g6, &c->g7 };
    for (auto* b : bufs) b->release();
    c->ch_bestpos.release(); c->ch_hdr.release();
    for (auto& g : c->shard_graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    c->shard_graphs.clear();
    c->products.release();
    c->pin_tab.release(); c->pin_ex.release(); c->pin_scal.release(); c->pin_best.release(); c->pin_kept.release();
    c->ch_pin_tables.release(); c->ch_pin_res.release(); c->ch_pin_kept.release(); c->ch_pin_best.release(); c->pin_arena.release();
    for (auto& kv : c->resident) if (!c->resident_arena_of.count(kv.first)) (void)hipFree(kv.second.first);
    for (auto& a : c->resident_arenas) if (a.first) (void)hipFree(a.first);
    if (c->mask_stream) (void)hipStreamDestroy(c->mask_stream);
    if (c->prod_stream) (void)hipStreamDestroy(c->prod_stream);
    (void)hipStreamDestroy(c->stage1_stream);
    (void)hipStreamDestroy(c->copy_stream);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* l3d_last_error(const l3d_ctx* c) { return c ? c->err.c_str() : "null context"; }
void l3d_free(void* p) { free(p); }

int l3d_register_segments(l3d_ctx* c, const float* segments, int n_segments)
{
    if (!c || !segments || n_segments < 0) return fail(c, L3D_ERR_INVALID, "l3d_register_segments: bad argument");
    (void)hipSetDevice(c->device);
    l3d_unregister_segments(c, segments);
    const size_t bytes = (size_t)n_segments * 16;
    void* d = nullptr;
    HIPCHK(c, hipMalloc(&d, bytes ? bytes : 16));
    if (bytes) HIPCHK(c, hipMemcpy(d, segments, bytes, hipMemcpyHostToDevice));
    c->resident[segments] = { d, bytes };
    return L3D_OK;
}

// Many arrays at once (all views of a scene): ONE device allocation, the copies queued back to back, one wait -- 128 separate
// hipMalloc + synchronous hipMemcpy pairs were 20 ms of a 24 ms prepare() at config 2.
int l3d_register_segments_batch(l3d_ctx* c, const float* const* arrays, const int* counts, int n)
{
    if (!c || n < 0 || (n > 0 && (!arrays || !counts))) return fail(c, L3D_ERR_INVALID, "l3d_register_segments_batch: bad argument");
    if (n == 0) return L3D_OK;
    (void)hipSetDevice(c->device);
    size_t total = 0;
    for (int i = 0; i < n; ++i) {
        if (!arrays[i] || counts[i] < 0) return fail(c, L3D_ERR_INVALID, "l3d_register_segments_batch: bad argument");
        total += ((size_t)counts[i] * 16 + 255) & ~(size_t)255;
    }
    for (int i = 0; i < n; ++i) l3d_unregister_segments(c, arrays[i]);
    char* base = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&base), total ? total : 256));
    const int arena = (int)c->resident_arenas.size();
    c->resident_arenas.push_back({ base, 0 });
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        if (c->resident.count(arrays[i])) continue;                          // (the same array twice in one batch)
        const size_t bytes = (size_t)counts[i] * 16;
        if (bytes) {
            hipError_t e = hipMemcpyAsync(base + off, arrays[i], bytes, hipMemcpyHostToDevice, c->stream);
            if (e != hipSuccess) { (void)hipStreamSynchronize(c->stream); return fail(c, L3D_ERR_HIP, std::string("l3d_register_segments_batch: ") + hipGetErrorString(e)); }
        }
        c->resident[arrays[i]] = { base + off, bytes };
        c->resident_arena_of[arrays[i]] = arena;
        c->resident_arenas[(size_t)arena].second += 1;
        off += (bytes + 255) & ~(size_t)255;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->resident_arenas[(size_t)arena].second == 0) { (void)hipFree(base); c->resident_arenas[(size_t)arena].first = nullptr; }
    return L3D_OK;
}

int l3d_unregister_segments(l3d_ctx* c, const float* segments)
{
    if (!c) return L3D_ERR_INVALID;
    auto it = c->resident.find(segments);
    if (it == c->resident.end()) return L3D_OK;
    auto ia = c->resident_arena_of.find(segments);
    if (ia == c->resident_arena_of.end()) (void)hipFree(it->second.first);
    else {                                                                    // a slice of a batch allocation: freed with its last slice
        auto& a = c->resident_arenas[(size_t)ia->second];
        if (--a.second == 0 && a.first) { (void)hipFree(a.first); a.first = nullptr; }
        c->resident_arena_of.erase(ia);
    }
    c->resident.erase(it);
    return L3D_OK;
}

int l3d_set_chain_capacities(l3d_ctx* c, size_t cand_cap, size_t arena_cap) { if (!c) return L3D_ERR_INVALID; c->test_cand_cap = cand_cap; c->test_arena_cap = arena_cap; return L3D_OK; }
int l3d_set_verify_lds_budget(size_t bytes) { verify_window_set_lds_budget(bytes); return L3D_OK; }
int l3d_set_pair_pretest(l3d_ctx* c, int mask) { if (!c || mask < 0 || mask > 7) return L3D_ERR_INVALID; c->wedge_pretest = mask; c->opt.pretest = mask; return L3D_OK; }
int l3d_set_verify_mode(l3d_ctx* c, int mode) { if (!c || mode < 0 || mode > 1) return L3D_ERR_INVALID; c->verify_mode = mode; return L3D_OK; }
int l3d_profile_enable(l3d_ctx* c, int on) { if (!c) return L3D_ERR_INVALID; c->prof_on = on != 0; return L3D_OK; }
int l3d_profile_only(l3d_ctx* c, const char* kernel) { if (!c) return L3D_ERR_INVALID; c->prof_only = kernel ? kernel : ""; return L3D_OK; }
int l3d_profile_reset(l3d_ctx* c)
{
    if (!c) return L3D_ERR_INVALID;
    prof_resolve(c);
    c->prof.clear();
    return L3D_OK;
}
int l3d_profile_get(l3d_ctx* c, const char* kernel, int64_t* launches, double* total_ms)
{
    if (!c || !kernel) return L3D_ERR_INVALID;
    prof_resolve(c);
    auto it = c->prof.find(kernel);
    if (launches) *launches = it == c->prof.end() ? 0 : it->second.launches;
    if (total_ms) *total_ms = it == c->prof.end() ? 0.0 : it->second.ms;
    return L3D_OK;
}
const char* l3d_profile_names(void) { return kProfNames; }
// Device arenas of the finishing stages (greedy selection, affinity fill, edge order, line fit) reserved ahead of their first use from
// the scene's size: n_dense segments in all views, about n_neighbors per view.  Grow-only arenas make this a hint: a stage that
// needs more still gets it.  (The first finish of a process spent most of its time in hipMalloc.)
int l3d_reserve_hint(l3d_ctx* c, int n_dense, int n_views, int n_neighbors)
{
    if (!c || n_dense < 0 || n_views < 0 || n_neighbors < 0) return L3D_ERR_INVALID;
    (void)hipSetDevice(c->device);
    const size_t nd = (size_t)n_dense, N = (size_t)std::max(1, n_neighbors);
    const size_t n_pot = 3 * N * nd + 1024, n_items = (3 * N * nd) / 2 + 1024, n_edges = N * nd + 1024, slots = 2 * n_pot;
    const size_t bslots = std::min(slots, (size_t)(c->opt.prod_block_keys > 0 ? c->opt.prod_block_keys : (1 << 28)) + 64);   // the products are built in blocks of key slots
    Products& P = c->products;
    // (transposed build of the products, round 6: 4 bytes per forward target and per backward entry, six ints per row, the table sized from its count)
    const bool tr = c->opt.prod_transpose != 0;
    const size_t tslots = std::min(n_pot, (size_t)(c->opt.prod_block_keys > 0 ? c->opt.prod_block_keys : (1 << 30)));
    struct R { DevBuf* b; size_t bytes; } rs[] = {
        { &P.keys, tr ? tslots * 4 : bslots * 8 }, { &P.keys2, tr ? tslots * 4 : bslots * 8 }, { &P.flag, tr ? 6 * (nd + 128) * 4 : bslots * 4 }, { &P.pos, tr ? 0 : bslots * 4 },
        { &P.tmp, tr ? (nd / 256 + 1024) * 8 + (64u << 10) : bslots * 4 + (64u << 10) }, { &P.pot_start, (nd + 2) * 8 },
        { &P.pot_tgt, (tr ? n_pot : slots) * 4 }, { &P.best_ref, nd * 8 }, { &P.hyp_of, (nd + 2) * 8 }, { &P.score, nd * 4 }, { &P.hyp_dense, nd * 4 }, { &P.best_hyp, nd * 4 },
        { &P.aux, nd * 4 + n_pot + 1024 }, { &c->aff_hyp, nd * sizeof(Hypothesis) },
        { &c->g1, (nd + 2) * 16 + (size_t)n_views * 4 + 1024 }, { &c->g2, n_items * 8 }, { &c->g3, n_items * 8 }, { &c->g4, n_items * 4 },
        { &c->g5, (nd * 2 + n_items * 6 + 8) * 4 }, { &c->g6, n_edges * 2 * sizeof(l3d_edge) + nd * 4 }, { &c->g7, n_edges * 16 + (1u << 20) }, { &c->g0, n_edges * 2 * sizeof(l3d_edge) * 2 },
    };
    bool grows = false;
    for (const R& r : rs) grows = grows || r.b->cap < r.bytes + 256;
    if (grows) {        // a growing arena is reallocated: whatever an earlier stage left resident in these buffers is gone
        c->resident_edges = 0; c->resident_nodes = 0; c->resident_labels = 0; c->resident_hyp = 0;
        P.valid = false; P.hyp_valid = false;
    }
    for (const R& r : rs) HIPCHK(c, r.b->reserve(r.bytes + 256));
    return L3D_OK;
}

// The first launch out of a translation unit loads its code object (a few milliseconds each, and the first matchViews / finish of a
// process touch all of them): loaded here, in parallel, so that a caller can do it while it still reads its input.
int l3d_warm_up(l3d_ctx* c)
{
    if (!c) return L3D_ERR_INVALID;
    (void)hipSetDevice(c->device);
    const double t0 = now_s();
    // One after the other, in the order the first compute3Dmodel needs them: collinearity and stage 1, the chain, verification, the
    // products and their sort, then the finishing stages.  The runtime loads modules under one lock: eight threads loading at once took
    // 21 ms where this loop takes 11 (measured: profiles/README.md, r4), and a kernel launch of the caller waits for that lock too -- a
    // caller that does not wait for this function (the facade does not) finds each module loaded by the time it gets there, or waits for
    // that one only.  (Rounds 2-3 also pushed a 16 MB copy through the runtime here to build its pageable-copy staging ahead of the first
    // read-back of an edge list: 17 ms that nothing needs since the edge list stays on the device.)
    struct M { void (*f)(); const char* name; } mods[] = {
        { warm_kernels, "kernels" }, { warm_chain, "chain" }, { warm_verify_window, "verify_window" }, { warm_products, "products" }, { warm_sort, "sort" },
        { warm_affinity, "affinity" }, { warm_rdd, "rdd" }, { warm_linefit, "linefit" }, { warm_chain_sharded, "chain_sharded" } };
    for (const M& m : mods) {
        const double a0 = now_s();
        m.f();
        if (c->opt.timing >= 2) fprintf(stderr, "[l3d warm_up]   module %-16s %6.2f ms\n", m.name, (now_s() - a0) * 1e3);
    }
    (void)hipGetLastError();
    if (c->opt.timing) fprintf(stderr, "[l3d warm_up] all modules loaded after %.2f ms\n", (now_s() - t0) * 1e3);
    return L3D_OK;
}
int l3d_last_stats(l3d_ctx* c, double stats[4])
{
    if (!c || !stats) return L3D_ERR_INVALID;
    memcpy(stats, c->stats, sizeof(c->stats));
    return L3D_OK;
}

// -------------------------------------------------------------------------------------------------
int l3d_compute_pairwise_matches(l3d_ctx* c,
                                 const float* src_segs, int S_src, const float* RtKinv_src, const float* C_src,
                                 const float* tgt_segs, const int32_t* offsets, int N,
                                 const float* F, const float* RtKinv, const float* centers, const float* P,
                                 const int32_t* to_be_matched, int n_tbm,
                                 const l3d_match* in_matches, int n_in, const uint32_t* local2global,
                                 float k_upper, float k_lower, float sigma_p, float sigma_a, float spatial_k,
                                 int seg_begin, int seg_end,
                                 l3d_match** out_matches, int* out_n, float* median_depth,
                                 float** out_best_depths, int* out_n_best)
{
    (void)k_upper; (void)k_lower;   // only printed by the reference (cudawrapper.cu:1074-1082)
    if (!c) return L3D_ERR_INVALID;
    if (!out_matches || !out_n || !median_depth) return fail(c, L3D_ERR_INVALID, "null output pointer");
    *out_matches = nullptr; *out_n = 0;
    if (out_best_depths) *out_best_depths = nullptr;
    if (out_n_best) *out_n_best = 0;
    memset(c->stats, 0, sizeof(c->stats));
    if (S_src < 0 || N < 0 || n_tbm < 0 || n_in < 0 || n_tbm > N) return fail(c, L3D_ERR_INVALID, "negative or inconsistent size");
    if (n_in > 0 && !in_matches) return fail(c, L3D_ERR_INVALID, "in_matches is null");

    // cudawrapper.cu:877-878: nothing to match -> the list comes back untouched
    if (n_tbm == 0) {
        l3d_match* o = (l3d_match*)malloc(sizeof(l3d_match) * (size_t)(n_in > 0 ? n_in : 1));
        if (!o) return fail(c, L3D_ERR_NOMEM, "malloc");
        if (n_in) memcpy(o, in_matches, sizeof(l3d_match) * (size_t)n_in);
        *out_matches = o; *out_n = n_in;
        return L3D_OK;
    }
    if (!src_segs || !RtKinv_src || !C_src || !tgt_segs || !offsets || !F || !RtKinv || !centers || !P || !to_be_matched || !local2global)
        return fail(c, L3D_ERR_INVALID, "null input pointer");
    if (seg_begin < 0) seg_begin = 0;
    if (seg_end > S_src) seg_end = S_src;
    if (seg_end < seg_begin) seg_end = seg_begin;

    int total_tgt = 0, maxW = 0;
    for (int i = 0; i < N; ++i) {
        if (offsets[2 * i] < 0 || offsets[2 * i + 1] < 0) return fail(c, L3D_ERR_INVALID, "negative offset");
        total_tgt = std::max(total_tgt, offsets[2 * i] + offsets[2 * i + 1]);
    }
    double pairs = 0;
    for (int j = 0; j < n_tbm; ++j) {
        if (to_be_matched[j] < 0 || to_be_matched[j] >= N) return fail(c, L3D_ERR_INVALID, "to_be_matched out of range");
        maxW = std::max(maxW, offsets[2 * to_be_matched[j] + 1]);
        pairs += (double)(seg_end - seg_begin) * offsets[2 * to_be_matched[j] + 1];
    }
    const int W64 = 4 * ((maxW + 255) / 256);
    if (W64 > kMaxW64) return fail(c, L3D_ERR_INVALID, "a neighbour has more than 16384 segments");
    c->stats[0] = pairs;

    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    double tp = now_s();
#define TPHASE(k) do { const double t_ = now_s(); c->tacc[k] += t_ - tp; tp = t_; } while (0)

    // ---- small tables: offsets | F | RtKinv | centers | P | RtKinv_src | C_src | tbm | l2g -> one pinned block, one H2D
    const float4 *d_src = nullptr, *d_tgt = nullptr;
    int rc;
    if ((rc = to_device(c, c->src_segs, src_segs, (size_t)S_src * 16, &d_src))) return rc;
    if ((rc = to_device(c, c->tgt_segs, tgt_segs, (size_t)total_tgt * 16, &d_tgt))) return rc;
    const size_t o_off = 0, o_F = o_off + (size_t)N * 8, o_R = o_F + (size_t)N * 36, o_C = o_R + (size_t)N * 36,
                 o_P = o_C + (size_t)N * 12, o_Rs = o_P + (size_t)N * 48, o_Cs = o_Rs + 36, o_tbm = o_Cs + 12,
                 o_l2g = o_tbm + (size_t)n_tbm * 4, t_bytes = o_l2g + (size_t)N * 4;
    HIPCHK(c, c->pin_tab.reserve(t_bytes));
    unsigned char* tab = c->pin_tab.as<unsigned char>();
    memcpy(tab + o_off, offsets, (size_t)N * 8);
    memcpy(tab + o_F, F, (size_t)N * 36);
    memcpy(tab + o_R, RtKinv, (size_t)N * 36);
    memcpy(tab + o_C, centers, (size_t)N * 12);
    memcpy(tab + o_P, P, (size_t)N * 48);
    memcpy(tab + o_Rs, RtKinv_src, 36);
    memcpy(tab + o_Cs, C_src, 12);
    memcpy(tab + o_tbm, to_be_matched, (size_t)n_tbm * 4);
    memcpy(tab + o_l2g, local2global, (size_t)N * 4);
    HIPCHK(c, c->tables.reserve(t_bytes));
    HIPCHK(c, hipMemcpyAsync(c->tables.p, tab, t_bytes, hipMemcpyHostToDevice, st));
    const unsigned char* tb = c->tables.as<unsigned char>();

    const size_t nrow = (size_t)S_src * N;
    HIPCHK(c, c->mask.reserve((size_t)n_tbm * S_src * W64 * 8));
    HIPCHK(c, c->rowcnt.reserve(nrow * 4));
    HIPCHK(c, c->row_start.reserve((nrow + 1) * 4));
    HIPCHK(c, c->kept_cnt.reserve((size_t)S_src * 4 + 4));
    HIPCHK(c, c->kept_start.reserve((size_t)S_src * 4 + 8));
    HIPCHK(c, c->best.reserve((size_t)S_src * 8 + 8));
    HIPCHK(c, c->scal.reserve(64));
    HIPCHK(c, c->pin_scal.reserve(64));
    HIPCHK(c, hipMemsetAsync(c->rowcnt.p, 0, nrow * 4, st));
    HIPCHK(c, hipMemsetAsync(c->kept_cnt.p, 0, (size_t)S_src * 4, st));
    HIPCHK(c, hipMemsetAsync(c->scal.p, 0, 4, st));

    PairArgs pa;
    pa.src_segs = d_src; pa.tgt_segs = d_tgt;
    pa.offsets = reinterpret_cast<const int2*>(tb + o_off);
    pa.F = reinterpret_cast<const float*>(tb + o_F);
    pa.RtKinv = reinterpret_cast<const float*>(tb + o_R);
    pa.centers = reinterpret_cast<const float*>(tb + o_C);
    pa.RtKinv_src = reinterpret_cast<const float*>(tb + o_Rs);
    pa.C_src = reinterpret_cast<const float*>(tb + o_Cs);
    pa.tbm = reinterpret_cast<const int*>(tb + o_tbm);
    pa.mask = c->mask.as<unsigned long long>();
    pa.S_src = S_src; pa.N = N; pa.n_tbm = n_tbm; pa.W64 = W64;
    pa.seg_begin = seg_begin; pa.seg_end = seg_end; pa.cand_cap = 0; pa.wedge_pretest = c->wedge_pretest; pa.dbg = c->pair_dbg; pa.dbg_view = -1; pa.rowcnt = nullptr;
    const unsigned* d_l2g = reinterpret_cast<const unsigned*>(tb + o_l2g);

    // stage 1 starts now; the host orders the existing matches meanwhile
    if (seg_end > seg_begin) {
        { ProfScope p(c, "pair_mask"); launch_pair_mask(pa, maxW, st, c->opt.pair_spb); }
        { ProfScope p(c, "row_count"); launch_row_count(pa, c->rowcnt.as<int>(), st); }
    }
    TPHASE(0);

    // ---- existing matches: localized by the caller; keep those of the processed range, order them
    // (segment, camera, target) and rank them inside their (segment, camera) run
    HIPCHK(c, c->pin_ex.reserve((size_t)n_in * sizeof(ExistRec) + 16));
    ExistRec* ex = c->pin_ex.as<ExistRec>();
    int n_ex = 0;
    {
        // Callers that append the reverse matches view by view (ascending camera) hand them over sorted by
        // (camera, origin order); a stable counting sort on the segment then already yields the final order --
        // verified in one pass, general sort otherwise.
        std::vector<int>& cnt = c->h_cnt;
        cnt.assign((size_t)S_src + 1, 0);
        for (int i = 0; i < n_in; ++i) {
            const l3d_match& m = in_matches[i];
            if ((int)m.segID1 < seg_begin || (int)m.segID1 >= seg_end) continue;
            if ((int)m.camID2 >= N) return fail(c, L3D_ERR_INVALID, "in_matches camera index out of range");
            cnt[(size_t)m.segID1 + 1]++;
        }
        for (int i = 0; i < S_src; ++i) cnt[(size_t)i + 1] += cnt[(size_t)i];
        n_ex = cnt[(size_t)S_src];
        for (int i = 0; i < n_in; ++i) {
            const l3d_match& m = in_matches[i];
            if ((int)m.segID1 < seg_begin || (int)m.segID1 >= seg_end) continue;
            ExistRec& r = ex[cnt[(size_t)m.segID1]++];
            r.seg = (int)m.segID1; r.cam = (int)m.camID2; r.tgt = m.segID2; r.rank = 0;
            memcpy(r.d, m.depths, 16);
        }
        auto less = [](const ExistRec& a, const ExistRec& b) {
            if (a.seg != b.seg) return a.seg < b.seg;
            if (a.cam != b.cam) return a.cam < b.cam;
            return a.tgt < b.tgt;
        };
        bool sorted = true;
        for (int i = 1; i < n_ex && sorted; ++i) sorted = !less(ex[i], ex[i - 1]);
        if (!sorted) std::stable_sort(ex, ex + n_ex, less);
        for (int i = 1; i < n_ex; ++i)
            if (ex[i].seg == ex[i - 1].seg && ex[i].cam == ex[i - 1].cam) ex[i].rank = ex[i - 1].rank + 1;
    }
    TPHASE(1);
    HIPCHK(c, c->exist.reserve((size_t)n_ex * sizeof(ExistRec) + 16));
    if (n_ex) HIPCHK(c, hipMemcpyAsync(c->exist.p, ex, (size_t)n_ex * sizeof(ExistRec), hipMemcpyHostToDevice, st));
    { ProfScope p(c, "exist"); launch_exist_hist(c->exist.as<ExistRec>(), n_ex, N, c->rowcnt.as<int>(), st); }
    { ProfScope p(c, "scan"); launch_scan(c->rowcnt.as<int>(), c->row_start.as<int>(), (int)nrow, nullptr, st); }
    launch_seg_mmax(c->row_start.as<int>(), N, seg_begin, seg_end, c->scal.as<int>(), st);
    int* hs = c->pin_scal.as<int>();
    HIPCHK(c, hipMemcpyAsync(hs, c->row_start.as<int>() + nrow, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(hs + 1, c->scal.p, 4, hipMemcpyDeviceToHost, st));
    TPHASE(2);
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    TPHASE(3);
    const int R = hs[0], mmax = hs[1];
    if (mmax > c->mmax_seen) c->mmax_seen = mmax;
    c->stats[1] = R;
    if (R == 0) {            // cudawrapper.cu:955-956: matches stays empty, median_depth untouched
        *out_matches = (l3d_match*)malloc(sizeof(l3d_match));
        return L3D_OK;
    }

    HIPCHK(c, c->cand_meta.reserve((size_t)R * 8));
    HIPCHK(c, c->cand_depths.reserve((size_t)R * 16));
    HIPCHK(c, c->cand_conf.reserve((size_t)R * 4));
    HIPCHK(c, c->kept.reserve((size_t)R * sizeof(Match)));

    if (seg_end > seg_begin) {
        ProfScope p(c, "pair_fill");
        launch_pair_fill(pa, c->row_start.as<int>(), c->cand_meta.as<uint2>(), c->cand_depths.as<float4>(), st);
    }
    { ProfScope p(c, "exist"); launch_exist_place(c->exist.as<ExistRec>(), n_ex, N, c->row_start.as<int>(), c->cand_meta.as<uint2>(), c->cand_depths.as<float4>(), st); }

    VerifyArgs va;
    va.exist_cams = nullptr; va.n_exist_cams = 0;
    va.src_segs = d_src; va.tgt_segs = d_tgt; va.offsets = pa.offsets;
    va.P = reinterpret_cast<const float*>(tb + o_P);
    va.RtKinv_src = pa.RtKinv_src; va.C_src = pa.C_src;
    va.row_start = c->row_start.as<int>();
    va.cand_meta = c->cand_meta.as<uint2>(); va.cand_depths = c->cand_depths.as<float4>(); va.cand_conf = c->cand_conf.as<float>();
    va.N = N; va.seg_begin = seg_begin; va.seg_end = seg_end; va.nrow_total = (int)nrow;
    va.sigma_p = sigma_p; va.sigma_a = sigma_a; va.spatial_k = spatial_k;
    va.mmax = mmax; va.only_above = -1; va.skip_above = 0; va.cand_cap = 0; va.res = nullptr; va.bstart_g = nullptr;
    va.big = 0; va.scratch = nullptr; va.scratch_stride = 0; va.kept_cnt = nullptr; va.best_depths = nullptr; va.seg_order = nullptr;
    va.debug = c->opt.vw_debug;
    va.stamps = nullptr;
    if (c->opt.vw_stamps) {
        if (!c->stamps.p) { HIPCHK(c, c->stamps.reserve(128)); HIPCHK(c, hipMemsetAsync(c->stamps.p, 0, 128, st)); }
        va.stamps = c->stamps.as<unsigned long long>();
    }
    if (c->verify_mode == 0 && verify_window_supported(N)) {
        // segments that fit the LDS image in one launch, the (few) bigger ones in a second launch on a global scratch
        int mfit = mmax;
        while (mfit > 64 && verify_window_lds_bytes(mfit, N) > verify_window_max_lds(c->opt.vw_lds)) mfit = mfit * 3 / 4;
        va.mmax = mfit; va.skip_above = 1;
        { ProfScope p(c, "verify_window"); launch_verify_window(va, st, c->opt.vw_wide_max); }
        if (mfit < mmax) {
            HIPCHK(c, c->vw_scratch.reserve(((size_t)R + kVWSlack) * 16));
            va.big = 1; va.scratch = c->vw_scratch.as<float>(); va.scratch_stride = (long long)R + kVWSlack;
            ProfScope p(c, "verify_window"); launch_verify_window(va, st, c->opt.vw_wide_max);
        }
    } else { ProfScope p(c, "verify"); launch_verify(va, st); }
    { ProfScope p(c, "seg_post"); launch_seg_post(va, c->kept_cnt.as<int>(), c->best.as<float2>(), st); }
    { ProfScope p(c, "scan"); launch_scan(c->kept_cnt.as<int>(), c->kept_start.as<int>(), S_src, nullptr, st); }
    { ProfScope p(c, "kept_write"); launch_kept_write(va, c->kept_start.as<int>(), d_l2g, c->kept.as<Match>(), st); }
    TPHASE(4);

    const size_t nb = (size_t)(seg_end - seg_begin) * 2;
    HIPCHK(c, c->pin_best.reserve(nb * 4 + 16));
    float* best = c->pin_best.as<float>();
    HIPCHK(c, hipMemcpyAsync(hs + 2, c->kept_start.as<int>() + S_src, 4, hipMemcpyDeviceToHost, st));
    if (nb) HIPCHK(c, hipMemcpyAsync(best, c->best.as<float2>() + seg_begin, nb * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    TPHASE(5);
    const int n_kept = hs[2];
    l3d_match* o = (l3d_match*)malloc(sizeof(l3d_match) * (size_t)(n_kept > 0 ? n_kept : 1));
    if (!o) return fail(c, L3D_ERR_NOMEM, "malloc");
    if (n_kept) {
        HIPCHK(c, c->pin_kept.reserve(sizeof(l3d_match) * (size_t)n_kept));
        HIPCHK(c, hipMemcpyAsync(c->pin_kept.p, c->kept.p, sizeof(l3d_match) * (size_t)n_kept, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        memcpy(o, c->pin_kept.p, sizeof(l3d_match) * (size_t)n_kept);
    }
    *out_matches = o; *out_n = n_kept;
    c->stats[3] = n_kept;
    TPHASE(6);

    // median of the best hypotheses' depths, cudawrapper.cu:1066-1076
    std::vector<float> depths;
    depths.reserve(nb);
    for (size_t i = 0; i + 1 < nb; i += 2)
        if (best[i] != -1.0f) { depths.push_back(best[i]); depths.push_back(best[i + 1]); }
    if (out_best_depths && out_n_best) {
        float* bd = (float*)malloc(sizeof(float) * (depths.size() ? depths.size() : 1));
        if (!bd) return fail(c, L3D_ERR_NOMEM, "malloc");
        if (!depths.empty()) memcpy(bd, depths.data(), depths.size() * 4);
        *out_best_depths = bd; *out_n_best = (int)(depths.size() / 2);
    }
    *median_depth = -1.0f;
    if (!depths.empty()) {
        std::nth_element(depths.begin(), depths.begin() + (long)(depths.size() / 2), depths.end());   // = sorted[size/2]
        *median_depth = depths[depths.size() / 2];
    }
    TPHASE(7);
#undef TPHASE
    return L3D_OK;
}

// -------------------------------------------------------------------------------------------------
int l3d_compute_collinearity(l3d_ctx* c, const float* segments, int S, float collin_s,
                             int32_t** out_i, int32_t** out_j, float** out_w, int* out_n)
{
    if (!c) return L3D_ERR_INVALID;
    if (!out_i || !out_j || !out_w || !out_n || S < 0 || (S > 0 && !segments)) return fail(c, L3D_ERR_INVALID, "bad argument");
    *out_i = nullptr; *out_j = nullptr; *out_w = nullptr; *out_n = 0;
    if (S < 2) return L3D_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int W64 = 4 * ((S + 255) / 256);
    const float4* d_segs = nullptr;
    int rc;
    if ((rc = to_device(c, c->g0, segments, (size_t)S * 16, &d_segs))) return rc;
    HIPCHK(c, c->g1.reserve((size_t)S * W64 * 8));
    HIPCHK(c, c->g2.reserve((size_t)S * 4));
    HIPCHK(c, c->g3.reserve((size_t)(S + 1) * 4));
    HIPCHK(c, hipMemsetAsync(c->g1.p, 0, (size_t)S * W64 * 8, st));
    HIPCHK(c, hipMemsetAsync(c->g2.p, 0, (size_t)S * 4, st));
    const float sigma_sqr = collin_s * collin_s;   // cudawrapper.cu:850
    { ProfScope p(c, "collinearity"); launch_collinearity(d_segs, S, sigma_sqr, c->g1.as<unsigned long long>(), W64, c->g2.as<int>(), st); }
    { ProfScope p(c, "scan"); launch_scan(c->g2.as<int>(), c->g3.as<int>(), S, nullptr, st); }
    int n = 0;
    HIPCHK(c, hipMemcpyAsync(&n, c->g3.as<int>() + S, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    if (n == 0) return L3D_OK;
    HIPCHK(c, c->g4.reserve((size_t)n * 4));
    HIPCHK(c, c->g5.reserve((size_t)n * 4));
    c->resident_edges = 0; c->resident_nodes = 0; c->resident_labels = 0;   // (g4 and g6 are reused)
    HIPCHK(c, c->g6.reserve((size_t)n * 4));
    { ProfScope p(c, "collinearity_fill");
      launch_collinearity_fill(d_segs, S, sigma_sqr, c->g1.as<unsigned long long>(), W64, c->g3.as<int>(), c->g4.as<int>(), c->g5.as<int>(), c->g6.as<float>(), st); }
    int32_t* oi = (int32_t*)malloc((size_t)n * 4);
    int32_t* oj = (int32_t*)malloc((size_t)n * 4);
    float* ow = (float*)malloc((size_t)n * 4);
    if (!oi || !oj || !ow) { free(oi); free(oj); free(ow); return fail(c, L3D_ERR_NOMEM, "malloc"); }
    HIPCHK(c, hipMemcpyAsync(oi, c->g4.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(oj, c->g5.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(ow, c->g6.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    *out_i = oi; *out_j = oj; *out_w = ow; *out_n = n;
    return L3D_OK;
}

namespace l3d {
__global__ void k_gather_ints(const int* __restrict__ base, const long long* __restrict__ idx, int n, int* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = base[idx[i]];
}
}  // namespace l3d

// The collinearity relations of several segment sets (the views of a scene) without a host round trip per set: all bit-row and
// counting launches, ONE read-back of the totals, all fill launches, ONE read-back of the triplets.
int l3d_compute_collinearity_batch(l3d_ctx* c, const float* const* segments, const int* n_segments, int n_sets, float collin_s,
                                   int32_t** out_i, int32_t** out_j, float** out_w, int* set_start)
{
    if (!c) return L3D_ERR_INVALID;
    if (!out_i || !out_j || !out_w || !set_start || n_sets < 0 || (n_sets > 0 && (!segments || !n_segments))) return fail(c, L3D_ERR_INVALID, "bad argument");
    *out_i = nullptr; *out_j = nullptr; *out_w = nullptr;
    for (int v = 0; v <= n_sets; ++v) set_start[v] = 0;
    if (n_sets == 0) return L3D_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    struct Set { const float4* d; int S, W64; size_t o_mask, o_cnt, o_start; };
    std::vector<Set> sets((size_t)n_sets);
    size_t up_bytes = 0, mask_bytes = 0, cnt_ints = 0, start_ints = 0;
    for (int v = 0; v < n_sets; ++v) {
        Set& q = sets[(size_t)v];
        q.S = n_segments[v];
        if (q.S < 0 || (q.S > 0 && !segments[v])) return fail(c, L3D_ERR_INVALID, "bad argument");
        q.W64 = 4 * ((q.S + 255) / 256);
        q.d = q.S >= 2 ? static_cast<const float4*>(resident_ptr(c, segments[v], (size_t)q.S * 16)) : nullptr;
        if (q.S >= 2 && !q.d) up_bytes += (size_t)q.S * 16;
        q.o_mask = mask_bytes; q.o_cnt = cnt_ints; q.o_start = start_ints;
        if (q.S >= 2) { mask_bytes += (size_t)q.S * q.W64 * 8; cnt_ints += (size_t)q.S; start_ints += (size_t)q.S + 1; }
    }
    if (start_ints == 0) return L3D_OK;
    HIPCHK(c, c->g0.reserve(up_bytes + 16));
    HIPCHK(c, c->g1.reserve(mask_bytes + 16));
    HIPCHK(c, c->g2.reserve(cnt_ints * 4 + 16));
    HIPCHK(c, c->g3.reserve(start_ints * 4 + (size_t)n_sets * 12 + 64));
    HIPCHK(c, hipMemsetAsync(c->g1.p, 0, mask_bytes, st));
    HIPCHK(c, hipMemsetAsync(c->g2.p, 0, cnt_ints * 4, st));
    const float sigma_sqr = collin_s * collin_s;   // cudawrapper.cu:850
    {
        size_t uo = 0;
        for (int v = 0; v < n_sets; ++v) {
            Set& q = sets[(size_t)v];
            if (q.S < 2) continue;
            if (!q.d) {
                HIPCHK(c, hipMemcpyAsync(c->g0.as<unsigned char>() + uo, segments[v], (size_t)q.S * 16, hipMemcpyHostToDevice, st));
                q.d = reinterpret_cast<const float4*>(c->g0.as<unsigned char>() + uo);
                uo += (size_t)q.S * 16;
            }
            unsigned long long* mask = reinterpret_cast<unsigned long long*>(c->g1.as<unsigned char>() + q.o_mask);
            { ProfScope p(c, "collinearity"); launch_collinearity(q.d, q.S, sigma_sqr, mask, q.W64, c->g2.as<int>() + q.o_cnt, st); }
            { ProfScope p(c, "scan"); launch_scan(c->g2.as<int>() + q.o_cnt, c->g3.as<int>() + q.o_start, q.S, nullptr, st); }
        }
    }
    // totals of all sets in one read-back
    std::vector<long long> idx((size_t)n_sets, 0);
    std::vector<int> totals((size_t)n_sets, 0);
    int n_live = 0;
    for (int v = 0; v < n_sets; ++v) if (sets[(size_t)v].S >= 2) idx[(size_t)n_live++] = (long long)(sets[(size_t)v].o_start + (size_t)sets[(size_t)v].S);
    long long* d_idx = reinterpret_cast<long long*>(c->g3.as<unsigned char>() + ((start_ints * 4 + 15) & ~(size_t)15));
    int* d_tot = reinterpret_cast<int*>(d_idx + n_sets);
    HIPCHK(c, hipMemcpyAsync(d_idx, idx.data(), (size_t)n_live * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_gather_ints, dim3((n_live + 255) / 256), dim3(256), 0, st, c->g3.as<int>(), d_idx, n_live, d_tot);
    HIPCHK(c, hipMemcpyAsync(totals.data(), d_tot, (size_t)n_live * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    long long total = 0;
    for (int v = 0, k = 0; v < n_sets; ++v) {
        set_start[v] = (int)total;
        if (sets[(size_t)v].S >= 2) total += totals[(size_t)k++];
        if (total > 0x7fffffffll) return fail(c, L3D_ERR_NOMEM, "more than 2^31 collinear pairs");
    }
    set_start[n_sets] = (int)total;
    if (total == 0) return L3D_OK;
    const size_t n = (size_t)total;
    HIPCHK(c, c->g4.reserve(n * 4));
    HIPCHK(c, c->g5.reserve(n * 4));
    c->resident_edges = 0; c->resident_nodes = 0; c->resident_labels = 0;   // (g4 and g6 are reused)
    HIPCHK(c, c->g6.reserve(n * 4));
    for (int v = 0; v < n_sets; ++v) {
        const Set& q = sets[(size_t)v];
        if (q.S < 2 || set_start[v + 1] == set_start[v]) continue;
        const unsigned long long* mask = reinterpret_cast<const unsigned long long*>(c->g1.as<unsigned char>() + q.o_mask);
        ProfScope p(c, "collinearity_fill");
        launch_collinearity_fill(q.d, q.S, sigma_sqr, mask, q.W64, c->g3.as<int>() + q.o_start, c->g4.as<int>() + set_start[v], c->g5.as<int>() + set_start[v],
                                 c->g6.as<float>() + set_start[v], st);
    }
    int32_t* oi = (int32_t*)malloc(n * 4);
    int32_t* oj = (int32_t*)malloc(n * 4);
    float* ow = (float*)malloc(n * 4);
    if (!oi || !oj || !ow) { free(oi); free(oj); free(ow); return fail(c, L3D_ERR_NOMEM, "malloc"); }
    hipError_t e1 = hipMemcpyAsync(oi, c->g4.p, n * 4, hipMemcpyDeviceToHost, st);
    hipError_t e2 = hipMemcpyAsync(oj, c->g5.p, n * 4, hipMemcpyDeviceToHost, st);
    hipError_t e3 = hipMemcpyAsync(ow, c->g6.p, n * 4, hipMemcpyDeviceToHost, st);
    hipError_t e4 = hipStreamSynchronize(st);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) { free(oi); free(oj); free(ow); return fail(c, L3D_ERR_HIP, "collinearity batch: read-back failed"); }
    *out_i = oi; *out_j = oj; *out_w = ow;
    return L3D_OK;
}

}  // extern "C"

extern "C" {

// -------------------------------------------------------------------------------------------------
int l3d_similarity_coll3D_batch(l3d_ctx* c, const l3d_hypothesis* hyp, int n_hyp, const int32_t* pairs, int n_pairs,
                                float sigma_a, float* sim)
{
    if (!c) return L3D_ERR_INVALID;
    if (n_hyp < 0 || n_pairs < 0 || (n_pairs > 0 && (!hyp || !pairs || !sim))) return fail(c, L3D_ERR_INVALID, "bad argument");
    if (n_pairs == 0) return L3D_OK;
    for (int k = 0; k < 2 * n_pairs; ++k)
        if (pairs[k] < 0 || pairs[k] >= n_hyp) return fail(c, L3D_ERR_INVALID, "pair index out of range");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    HIPCHK(c, c->g0.reserve((size_t)n_hyp * sizeof(Hypothesis)));
    HIPCHK(c, c->g1.reserve((size_t)n_pairs * 8));
    HIPCHK(c, c->g2.reserve((size_t)n_pairs * 4));
    HIPCHK(c, hipMemcpyAsync(c->g0.p, hyp, (size_t)n_hyp * sizeof(Hypothesis), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->g1.p, pairs, (size_t)n_pairs * 8, hipMemcpyHostToDevice, st));
    const float two_log = 2.0f * logf(0.01f);      // view.cc:376
    { ProfScope p(c, "similarity"); launch_similarity(c->g0.as<Hypothesis>(), c->g1.as<int2>(), n_pairs, sigma_a, two_log, c->g2.as<float>(), st); }
    HIPCHK(c, hipMemcpyAsync(sim, c->g2.p, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    return L3D_OK;
}

int l3d_test_sq_threshold(l3d_ctx* c, const float* u, int n, float* out_walk, float* out_closed)
{
    if (!c || n < 0 || (n > 0 && (!u || !out_walk || !out_closed))) return L3D_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    HIPCHK(c, c->g0.reserve((size_t)n * 4 + 16)); HIPCHK(c, c->g1.reserve((size_t)n * 4 + 16)); HIPCHK(c, c->g2.reserve((size_t)n * 4 + 16));
    HIPCHK(c, hipMemcpyAsync(c->g0.p, u, (size_t)n * 4, hipMemcpyHostToDevice, st));
    launch_test_sqthr(c->g0.as<float>(), n, c->g1.as<float>(), c->g2.as<float>(), st);
    HIPCHK(c, hipMemcpyAsync(out_walk, c->g1.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(out_closed, c->g2.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return L3D_OK;
}

int l3d_test_contract_math(l3d_ctx* c, const float* x, int n, float* e, float* ac, double* acd)
{
    if (!c || n < 0 || (n > 0 && (!x || !e || !ac || !acd))) return L3D_ERR_INVALID;
    if (n == 0) return L3D_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    HIPCHK(c, c->g0.reserve((size_t)n * 4)); HIPCHK(c, c->g1.reserve((size_t)n * 4));
    HIPCHK(c, c->g2.reserve((size_t)n * 4)); HIPCHK(c, c->g3.reserve((size_t)n * 8));
    HIPCHK(c, hipMemcpyAsync(c->g0.p, x, (size_t)n * 4, hipMemcpyHostToDevice, st));
    launch_test_math(c->g0.as<float>(), n, c->g1.as<float>(), c->g2.as<float>(), c->g3.as<double>(), st);
    HIPCHK(c, hipMemcpyAsync(e, c->g1.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(ac, c->g2.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(acd, c->g3.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    return L3D_OK;
}

}  // extern "C"
