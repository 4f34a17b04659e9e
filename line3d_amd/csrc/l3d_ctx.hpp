// l3d_ctx.hpp -- the context behind the C ABI (shared by l3d_capi.hip and l3d_chain.hip): device arenas,
// pinned staging, per-kernel HIP-event profiling.
#pragma once

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/line3d_amd.h"
#include "l3d_kernels.hpp"
#include "l3d_options.hpp"

namespace l3d {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 4 + 256;      // grow-only arena with slack
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    // a buffer whose size is a plan, not a guess (rings of slots, an arena with its own slack): no 25 % on top
    hipError_t reserve_exact(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        hipError_t e = hipMalloc(&p, bytes + 256);
        if (e == hipSuccess) cap = bytes + 256;
        return e;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct PinBuf {                     // pinned host staging (async copies that really are async)
    void* p = nullptr;
    size_t cap = 0;
    unsigned flags = hipHostMallocDefault;      // hipHostMallocCoherent for buffers KERNELS write and the host reads behind an event (result records):
                                                // explicit, so that HIP_HOST_COHERENT=0 in the environment cannot make them non-coherent
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 2 + 4096;
        hipError_t e = hipHostMalloc(&p, want, flags);
        if (e == hipSuccess) cap = want;
        return e;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// Pinned host memory handed out in slices that stay valid until the next reset: the kept lists of a chain are copied
// (SDMA) straight to where the caller's bookkeeping will read them -- no staging buffer, no second host copy.
struct PinArena {
    static constexpr size_t kChunk = 64u << 20;
    std::vector<PinBuf> chunks;
    size_t cur = 0, used = 0;
    void reset() { cur = 0; used = 0; }
    void* alloc(size_t bytes, hipError_t* err)
    {
        *err = hipSuccess;
        bytes = (bytes + 255) & ~(size_t)255;
        for (;;) {
            if (cur < chunks.size()) {
                if (used + bytes <= chunks[cur].cap) { void* r = static_cast<char*>(chunks[cur].p) + used; used += bytes; return r; }
                ++cur; used = 0;
                continue;
            }
            chunks.emplace_back();
            *err = chunks.back().reserve(std::max(bytes, kChunk));
            if (*err != hipSuccess) { chunks.pop_back(); return nullptr; }
        }
    }
    void release() { for (auto& b : chunks) b.release(); chunks.clear(); reset(); }
};

// matchViews' products PARTITIONED over the ranks of a job (l3d_match_chain_partition): what this rank's share covers, in views of the dense map
struct ProductsPart {
    bool active = false;
    int rank = 0, world = 1;
    int own_dv0 = 0, own_dv1 = 0;       // the rank's block: its sources in the affinity fill
    int row_dv0 = 0, row_dv1 = 0;       // views whose rows of the potential-correspondence table are complete here (the block and `reach` views either side)
    int held_dv0 = 0, held_dv1 = 0;     // views whose kept records, best matches and medians are here (2 x reach either side): they get hypotheses
    long long n_pot_all = 0;            // entries of the table over all ranks
    int recovery_rounds = 0;            // rounds of warm re-runs after failed speculations (all missed blocks of a round run at once)
    int blocks_rerun = 0;               // blocks re-run in those rounds (whole job)
};

// Device-resident products of the last resident chain (l3d_products.hip) and the hypothesis table built from them
struct Products {
    bool valid = false, hyp_valid = false;
    ProductsPart part;
    int n_dense = 0, n_views_all = 0, n_chain = 0, n_hyp = 0;
    long long n_pot = 0, total_kept = 0;
    DevBuf keys, keys2, flag, pos, tmp, pot_start, pot_tgt, best_ref, median, tables, ttab, rowstage, tstage;
    DevBuf e_cnt, e_poff, e_boff, e_E, e_T, e_tab;     // the chain's early pair transposes (l3d_chain.hip: per (view, camera) counts / entry offsets, column starts, entries, staging, per-view tables)
    std::vector<int> e_boff_off;                        // host copy: where a (chain view, camera)'s column starts lie in e_boff
    DevBuf geo, hyp_of, score, hyp_dense, best_hyp, coll, aux;       // greedy selection / affinity fill on the resident tables
    long long coll_n = 0;           // entries of the collinearity CSR resident in `coll` (with n_dense + 1 starts in front)
    unsigned long long coll_sig = 0; // checksum of what that copy was uploaded from (sizes, row starts, dense map)
    std::vector<int> seg_base;      // host copies: the dense map, the chain's result records, what the early-return views need
    std::vector<unsigned> view_ids;
    std::vector<l3d::ChainResult> res;
    std::vector<int> chain_view;    // per chain index: index of its view in the dense map
    std::vector<char> chain_verified;
    std::vector<int> view_hyp_begin; // per view of the dense map: first hypothesis (l3d_products_hypotheses)
    std::vector<std::vector<int>> early_src_index, early_src_cam;    // per chain index (early-return views only): its sources
    std::vector<unsigned> chain_view_id;
    void release()
    {
        DevBuf* b[] = { &keys, &keys2, &flag, &pos, &tmp, &pot_start, &pot_tgt, &best_ref, &median, &tables, &ttab, &rowstage, &tstage, &e_cnt, &e_poff, &e_boff, &e_E, &e_T, &e_tab, &geo, &hyp_of, &score, &hyp_dense, &best_hyp, &coll, &aux };
        for (DevBuf* x : b) x->release();
        valid = hyp_valid = false; coll_n = 0; coll_sig = 0;
    }
};

// one view's launch sequence of the sharded chain as an instantiated graph (l3d_chain_sharded.hip): sig = checksum of everything the
// launches depend on, seen = passes in a row with that checksum
struct ShardGraph { unsigned long long sig = 0; int seen = 0; hipGraphExec_t exec = nullptr; };

// a rank's part of an affinity fill sharded by source key (l3d_affinity.hip: affinity_fill_core): the sources it enumerates, where its
// candidates stand in the whole enumeration (ranks own ascending source ranges: rank << 44 orders them without knowing the other ranks' counts)
// and what turns its local hypothesis indices into global ones
// assume_symmetric: the table was built by l3d_products.hip, which files every potential correspondence under both of its segments at once (the
// two keys of k_prod_keys): "is the source among the target's targets" is true by construction and is not looked up -- a rank holds the rows of
// its own sources' targets only in part
struct FillPart { int h0 = 0, h1 = 0; unsigned long long pos_base = 0; const int* loc2glob = nullptr; int assume_symmetric = 0; };     // loc2glob: device, global number of every local hypothesis

struct ProfEntry {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    int64_t launches = 0;
    double ms = 0.0;
};

}  // namespace l3d

struct l3d_ctx {
    int device = 0;
    l3d::Options opt;                        // every L3D_* switch: the environment read once by l3d_ctx_create, then l3d_set_option
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;       // bulk D2H of the resident chain, concurrent with kernels
    hipStream_t prod_stream = nullptr;       // option prod_early: a view's pair transposes, behind its kept writer (created on first use)
    hipStream_t mask_stream = nullptr;       // option mask_stream: k_pair_mask alone, ahead of the rest of stage 1 (created on first use)
    hipStream_t stage1_stream = nullptr;     // stage 1 of the resident chain (independent of the chain state) runs ahead here
    std::string err;                         // written under err_mu: the chains report from several host threads
    std::mutex err_mu;
    // arenas of the matching path
    l3d::DevBuf src_segs, tgt_segs, tables, tbm, l2g, exist, mask, rowcnt, row_start, cand_meta, cand_depths, cand_conf;
    l3d::DevBuf kept_cnt, kept_start, best, kept, scal, stamps, vw_scratch;
    l3d::PinBuf pin_tab, pin_ex, pin_scal, pin_best, pin_kept;
    // arenas of the resident chain (l3d_chain.hip)
    l3d::DevBuf ch_tables, ch_mask, ch_rowcnt, ch_cursor, ch_best, ch_kept, ch_keptcam, ch_res, ch_flags, ch_send, ch_gathered, ch_stage, ch_rowA, ch_ringA_meta, ch_ringA_depths, ch_segorder, ch_rays, ch_rt, ch_rtinfo, ch_rtjobs, ch_existpart;
    l3d::PinBuf ch_pin_tables, ch_pin_res, ch_pin_kept, ch_pin_best;
    long long shard_graph_launches = 0;          // views enqueued as one graph launch so far (l3d_get_option "shard_graph_launches")
    std::vector<l3d::ShardGraph> shard_graphs;   // per view of the sharded chain (repeated passes replay them)
    l3d::DevBuf ch_hdr;                      // sharded run, ring mode: per-view arena offsets, header table, flags
    l3d::DevBuf ch_bestpos;                  // per segment of every view: position of its best kept match in the view's slice (resident runs)
    l3d::Products products;
    std::vector<l3d::RayJob> ray_jobs;       // job list of k_tgt_rays of the running chain
    l3d::PinArena pin_arena;                 // kept lists of the running / last chain (valid until the next chain starts)
    std::vector<int> h_cnt;
    int mmax_seen = 0;
    int chain_ring = 1;             // single-GPU chain: 1 = stage-1 candidate ring + k_place (default), 0 = triangulation on the chain stream (L3D_CHAIN_RING=0, A/B)
    l3d::DevBuf vw_bstart, vw_segstate;             // split verification (k_vw_walk): bucket starts of the built images | per-segment state + unit table
    size_t part_arena_seen = 0;                     // records a partitioned segment-sharded run kept on this rank (sizes the next pass's arena)
    size_t test_cand_cap = 0, test_arena_cap = 0;   // tests: initial capacities of the resident chain (0 = estimate)
    int chain_seen_views = 0; double chain_seen_pairs = 0; size_t chain_seen_cand_cap = 0, chain_seen_arena_cap = 0; double chain_seen_kept = 0;   // what the last chain over this scene needed (and kept)
    unsigned long long* pair_dbg = nullptr;   // L3D_PAIR_STATS=1: device counters of k_pair_mask's levels (printed at destroy)
    int wedge_pretest = 3;          // stage-1 filters: bit 0 wedge test, bit 1 overlap-bound test (cleared only for A/B testing), bit 2 SET: level 2 does not accept
    int verify_mode = 0;            // 0: depth-window search (all-pairs kernel only beyond ~50 neighbours), 1: all-pairs
    // other paths
    l3d::DevBuf g0, g1, g2, g3, g4, g5, g6, g7;
    l3d::DevBuf aff_hyp;            // hypothesis table of the last l3d_affinity_fill (kept for l3d_fit_clusters)
    l3d::DevBuf aff_l2g;            // sharded fill: global number of every local hypothesis
    l3d::DevBuf aff_first, aff_pass_pairs, aff_pass_w;   // affinity fill in blocks of sources: first-touch minima per hypothesis, the candidates that passed (pairs, weights)
    long long fill_items = 0, fill_passed = 0;           // candidates enumerated / passed by the last fill (64-bit: l3d_last_fill_counts)
    int resident_hyp = 0;           // its number of hypotheses (0: none)
    int resident_edges = 0;         // entries of the edge list l3d_affinity_fill left in g6 (0: none); consumed by l3d_clustering_edges
    int resident_nodes = 0;         // nodes of that list; resident_nodes_p: their hypothesis indices (behind the edges in g6)
    const int* resident_nodes_p = nullptr;
    int resident_labels = 0;        // labels l3d_perform_clustering_device left on the device (in g4), for l3d_fit_labelled_clusters
    const int* resident_labels_p = nullptr;
    l3d::DevBuf edges_keep;         // copy of that list taken when the clustering consumes it (l3d_resident_edges_get)
    int kept_edges = 0;
    std::unordered_map<const void*, std::pair<void*, size_t>> resident;
    std::vector<std::pair<char*, int>> resident_arenas;                  // batch registrations: (one allocation, slices still registered)
    std::unordered_map<const void*, int> resident_arena_of;             // host pointer -> its batch allocation
    bool prof_on = false;
    std::string prof_only;          // bracket only this kernel (keeps the timed region of bench.py nearly undisturbed)
    std::map<std::string, l3d::ProfEntry> prof;
    std::vector<hipEvent_t> event_pool;
    std::vector<hipEvent_t> local_event_pool;  // events without system-scope fences (get_local_event: the single-GPU chain)
    std::vector<l3d::RtInfo> rtinfo_host;     // host copy of the chain's per-view run-table descriptors (uploaded asynchronously)
    std::vector<hipEvent_t> prof_event_pool;   // the brackets of ProfScope: events WITHOUT the system-scope fences (a default event flushes the caches at every record)
    int prof_tick = 0;                         // (option prof_stride: with prof_only set, every n-th launch of that kernel is bracketed)
    std::mutex event_mu;
    double stats[4] = { 0, 0, 0, 0 };
    double tacc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };   // host-side phase timers of l3d_compute_pairwise_matches (L3D_TIMING=1)
};

namespace l3d {

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

inline int fail(l3d_ctx* c, int code, const std::string& msg)
{
    if (c) { std::lock_guard<std::mutex> lk(c->err_mu); c->err = msg; }
    return code;
}

#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return l3d::fail(ctx, L3D_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Synchronisation events with HIP's default system-scope fences: what the SHARDED chain orders includes slots that other GPUs wrote into this
// one's memory (the RCCL all-gather), which a consumer kernel must not read through stale cache lines.
inline hipEvent_t get_event(l3d_ctx* c)
{
    std::lock_guard<std::mutex> lk(c->event_mu);         // (the sharded run enqueues from two threads)
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
// The single-GPU chain's events (view done, stage 1 done, ring slot free): created WITHOUT the system-scope acquire / release fences a default HIP
// event carries -- 126 records per config-2 pass, each a cache write-back and invalidate the next kernels pay for (12.27 -> 12.0 ms per pass).  What they
// order is (a) kernels on two streams of the SAME device (agent scope: the end-of-kernel release and the next kernel's acquire do that) and (b) the host
// reading result records that the kernels write straight into pinned HOST memory allocated hipHostMallocCoherent (ch_pin_res: uncached on the device,
// nothing for a fence to write back; the event's own signal is a later posted write on the same path).  That holds for the RESIDENT chain only: with a
// delivery callback (l3d_match_chain) the host, having seen a view's event, starts D2H copies of DEVICE memory (the kept slice, the best depth pairs) on
// another stream -- nothing but the event orders those copies behind the kernels, so run_chain takes default (fenced) events for its per-view events
// whenever a callback is given.  Nothing another device wrote is read behind these events.  (Option event_fence = 1: default events everywhere, A/B.)
inline hipEvent_t get_local_event(l3d_ctx* c)
{
    if (c->opt.event_fence != 0) return get_event(c);
    std::lock_guard<std::mutex> lk(c->event_mu);
    if (!c->local_event_pool.empty()) { hipEvent_t e = c->local_event_pool.back(); c->local_event_pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) { (void)hipGetLastError(); (void)hipEventCreate(&e); }
    return e;
}
inline void put_local_event(l3d_ctx* c, hipEvent_t e)
{
    if (!e) return;
    std::lock_guard<std::mutex> lk(c->event_mu);
    (c->opt.event_fence != 0 ? c->event_pool : c->local_event_pool).push_back(e);
}
inline void put_event(l3d_ctx* c, hipEvent_t e) { if (e) { std::lock_guard<std::mutex> lk(c->event_mu); c->event_pool.push_back(e); } }

inline hipEvent_t get_prof_event(l3d_ctx* c)
{
    std::lock_guard<std::mutex> lk(c->event_mu);
    if (!c->prof_event_pool.empty()) { hipEvent_t e = c->prof_event_pool.back(); c->prof_event_pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) { (void)hipGetLastError(); (void)hipEventCreate(&e); }
    return e;
}

struct ProfScope {
    l3d_ctx* c; const char* name; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    bool on;
    // events go on the stream the kernel is launched on; prof_only (when set) restricts the brackets to one kernel name
    ProfScope(l3d_ctx* c_, const char* n, hipStream_t s = nullptr) : c(c_), name(n), st(s ? s : c_->stream)
    {
        on = c->prof_on && (c->prof_only.empty() || c->prof_only == n);
        if (on && !c->prof_only.empty() && c->opt.prof_stride > 1) on = (c->prof_tick++ % c->opt.prof_stride) == 0;
        if (on) { a = get_prof_event(c); b = get_prof_event(c); (void)hipEventRecord(a, st); }
    }
    ~ProfScope()
    {
        if (on) { (void)hipEventRecord(b, st); std::lock_guard<std::mutex> lk(c->event_mu); c->prof[name].pending.emplace_back(a, b); }
    }
};

inline void prof_resolve(l3d_ctx* c)
{
    (void)hipStreamSynchronize(c->stream);
    if (c->stage1_stream) (void)hipStreamSynchronize(c->stage1_stream);
    if (c->mask_stream) (void)hipStreamSynchronize(c->mask_stream);
    if (c->prod_stream) (void)hipStreamSynchronize(c->prod_stream);
    std::lock_guard<std::mutex> lk(c->event_mu);         // (the pools and the pending lists: ProfScope of another enqueue thread takes it too)
    for (auto& kv : c->prof) {
        for (auto& pr : kv.second.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { kv.second.ms += ms; kv.second.launches += 1; }
            c->prof_event_pool.push_back(pr.first);
            c->prof_event_pool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}

// host pointer -> device pointer, uploading unless the array is registered as resident
template <class T>
int to_device(l3d_ctx* c, DevBuf& buf, const void* host, size_t bytes, const T** out)
{
    auto it = c->resident.find(host);
    if (it != c->resident.end() && it->second.second == bytes) { *out = reinterpret_cast<const T*>(it->second.first); return L3D_OK; }
    HIPCHK(c, buf.reserve(bytes ? bytes : 16));
    if (bytes) HIPCHK(c, hipMemcpyAsync(buf.p, host, bytes, hipMemcpyHostToDevice, c->stream));
    *out = buf.as<T>();
    return L3D_OK;
}

// device pointer of an array registered with l3d_register_segments, or null
inline const void* resident_ptr(l3d_ctx* c, const void* host, size_t bytes)
{
    auto it = c->resident.find(host);
    return (it != c->resident.end() && it->second.second == bytes) ? it->second.first : nullptr;
}

}  // namespace l3d
