// line3d_host_internal.hpp -- what the translation units of the host pipeline share (line3d_host*.cpp): the flat-array forms of the reference's
// view / hypothesis / result objects, the handle behind the C ABI's l3d_line3d, the static schedule of matchViews, and the functions one unit
// calls in another.  Cited line numbers refer to the reference files under /root/reference.
#pragma once
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/line3d_amd.h"
#include "l3d_linalg.hpp"
#include "l3d_linefit.hpp"
#include "l3d_unproject.hpp"
#include "l3d_hostsort.hpp"
#include "l3d_options.hpp"

using namespace l3d::la;

namespace l3dh {

typedef uint64_t Key;   // (camID << 32) | segID : orders like L3DSegment2D::operator< (commons.h:92-94)
inline Key mk(uint32_t cam, uint32_t seg) { return ((Key)cam << 32) | seg; }
inline uint32_t kcam(Key k) { return (uint32_t)(k >> 32); }
inline uint32_t kseg(Key k) { return (uint32_t)k; }

inline double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct View {                                   // L3DView, view.h:40-153
    uint32_t id = 0;
    int index = 0;                              // dense index in ascending id order (set in prepare)
    M3 K, R, Kinv, Rt, RtKinv;
    V3 t, C;
    double P[12];
    unsigned width = 0, height = 0;
    double pp[2];
    float unc_upper_px = 0, unc_lower_px = 0, k_upper = 0, k_lower = 0, median_depth = 1.0f;
    std::vector<float> segs;                    // S x 4
    bool coll_pending = false;                  // the relation is still to be computed (prepare: all views in one batch)
    std::string cache_to_write;                 // addImage with loadAndStoreSegments: the segment cache to write once the relation is there (line3D.cc:180-182)
    std::vector<int> coll_start;                // CSR of segment2collinearities_ (segments.h:84-97)
    std::vector<int> coll_other;
    std::vector<float> coll_w;
    bool store_exists = false;                  // the "_raw.bin" match file
    std::vector<l3d_match> store;
    std::vector<float> nb_segs;                 // concatenated neighbour segments (resident on the GPU)
    int S() const { return (int)(segs.size() / 4); }

    void derive()                               // view.cc:24-34 / :243-257
    {
        Kinv = inverse(K);
        Rt = transpose(R);
        RtKinv = mul(Rt, Kinv);
        C = mul(Rt, V3{ -1.0 * t.x, -1.0 * t.y, -1.0 * t.z });
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
                double s = 0.0;
                const double col[3] = { c < 3 ? R(0, c) : t.x, c < 3 ? R(1, c) : t.y, c < 3 ? R(2, c) : t.z };
                for (int k = 0; k < 3; ++k) s += K(r, k) * col[k];
                P[r * 4 + c] = s;
            }
        k_upper = (float)specific_k(unc_upper_px);     // defineSpatialUncertainty, view.cc:90-121
        k_lower = (float)specific_k(unc_lower_px);
    }
    double specific_k(double dist_px) const        // view.cc:124-147
    {
        V3 n = mul(RtKinv, V3{ pp[0], pp[1], 1.0 });
        n = n / norm(n);
        const V3 Pl = C + n;
        V3 d = mul(RtKinv, V3{ pp[0] + dist_px, pp[1], 1.0 });
        d = d / norm(d);
        const double tt = (dot(Pl, n) - dot(n, C)) / dot(n, d);
        const V3 Q = C + tt * d;
        return norm(Pl - Q);
    }
    void transform(const double* Qinv, double scale)   // view.cc:227-261
    {
        t = t * scale;
        double Rt34[12], out[12];
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) Rt34[r * 4 + c] = R(r, c); }
        Rt34[3] = t.x; Rt34[7] = t.y; Rt34[11] = t.z;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
                double s = 0.0;
                for (int k = 0; k < 4; ++k) s += Rt34[r * 4 + k] * Qinv[k * 4 + c];
                out[r * 4 + c] = s;
            }
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R(r, c) = out[r * 4 + c];
        t = { out[3], out[7], out[11] };
        derive();
    }
};

struct Hyp {                                    // L3DCorrespondenceRRW + L3DSegment3D, commons.h:69-160
    Key src;
    float score;
    V3 P1, P2, dir;
    float depth_p1, depth_p2;
};

// A processed view's kept list in chain mode: a slice of the context's pinned arena (valid until the next chain starts,
// include/line3d_amd.h) -- or an own vector for the lists the host builds itself (early-return views).
struct KeptList {
    const l3d_match* p = nullptr;
    size_t n = 0;
    std::vector<l3d_match> own;
    const l3d_match* begin() const { return p; }
    const l3d_match* end() const { return p + n; }
    const l3d_match* data() const { return p; }
    size_t size() const { return n; }
    void reset() { p = nullptr; n = 0; own.clear(); }
    void use_own() { p = own.data(); n = own.size(); }
};

// std::vector whose resize() leaves trivially-constructible elements uninitialised (large edge lists are written in full by
// worker threads right after the resize; a zero fill by the calling thread would cost more than the write)
template <class T>
struct NoInitAlloc {
    typedef T value_type;
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
    T* allocate(size_t n) { return static_cast<T*>(::operator new(n * sizeof(T))); }
    void deallocate(T* p, size_t) { ::operator delete(p); }
    template <class U> void construct(U* p) { ::new ((void*)p) U; }
    template <class U, class... Args> void construct(U* p, Args&&... a) { ::new ((void*)p) U(std::forward<Args>(a)...); }
    template <class U> bool operator==(const NoInitAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const NoInitAlloc<U>&) const { return false; }
};
typedef std::vector<l3d_edge, NoInitAlloc<l3d_edge>> EdgeVec;

// matched_ (line3D.h: map of maps of bool) as a sorted vector of (a << 32 | b): a few thousand keys, looked up far more often than inserted, and
// copied whole once per matchViews (the state a pass ends in is the schedule's, ChainPlan::matched_final)
struct KeySet {
    std::vector<uint64_t> v;
    size_t count(uint64_t k) const { return std::binary_search(v.begin(), v.end(), k) ? 1 : 0; }
    void insert(uint64_t k) { auto it = std::lower_bound(v.begin(), v.end(), k); if (it == v.end() || *it != k) v.insert(it, k); }
    void clear() { v.clear(); }
};

struct FinalLine {
    std::vector<Key> segs2D;
    std::vector<std::pair<V3, V3>> segs3D;
};

}  // namespace l3dh
using namespace l3dh;


struct l3d_line3d {
    l3d_ctx* ctx = nullptr;
    std::string err;
    bool verbose = false;
    // parameters, line3D.cc:6-31
    int matching_neighbors = 10;
    float unc_upper = 5.0f, unc_lower = 1.0f, sigma_p = 3.5f, sigma_a = 10.0f, min_baseline = 0.25f;
    bool use_collinearity = true;
    bool computation = false;
    bool prepared = false;

    std::map<uint32_t, View> views;
    std::vector<View*> vlist;                                  // ascending id
    std::map<uint32_t, std::map<uint32_t, float>> view_similarities;
    std::map<uint32_t, unsigned> num_wps;
    std::map<uint32_t, std::map<uint32_t, unsigned>> common_wps;
    std::unordered_map<uint32_t, std::vector<uint32_t>> worldpoints2views;   // ascending by construction? no: sorted on use
    std::map<uint32_t, std::vector<uint32_t>> visual_neighbors; // ascending ids
    std::map<uint64_t, M3> fundamentals;                       // (a<<32|b)
    KeySet matched;                                            // (a<<32|b): matched_[a][b]

    // geometry transformation
    double transf_scale_inv = 1.0;
    M3 transf_Rinv = identity3();
    V3 transf_tneg;

    // matching products
    std::vector<uint32_t> order;                               // views with >=1 neighbour, ascending
    std::vector<std::vector<std::pair<uint32_t, Key>>> pot;    // per view index: (seg, other key), potential_correspondences_
    std::vector<std::pair<Key, Key>> pot_foreign;              // keys whose camera is not a view (early-return quirk)
    std::map<uint32_t, std::vector<l3d_match>> view_matches;   // kept matches per view (for inspection)
    std::vector<std::vector<std::pair<size_t, std::array<std::vector<std::pair<uint32_t, Key>>, 4>>>> fin_buckets;   // finaliser scratch, reused across passes
    std::vector<std::array<std::vector<std::pair<uint32_t, Key>>, 4>> fin_parts;
    std::vector<KeptList> saved;                               // chain mode: performMatching's `matches` per processed view
    bool keep_view_matches = false;
    bool pot_check_failed = false;                             // L3D_CHECK_POT=1 (tests)
    void* finalizer = nullptr;                                 // ChainFinalizer with its worker threads (created on first use)
    void* plan_cache = nullptr;                                // ChainPlan of the current set of views (the schedule is static)
    void* shard_plan_ = nullptr;                               // open sharded chain (ChainPlan*, l3d_line3d_shard_*)
    bool force_sync = false;                                   // matchViews through the per-view seam call (A/B, L3D_MATCH_SYNC=1)
    std::thread warm_thread;                                   // l3d_warm_up, started with the object: the code objects load while the caller adds its images
    bool host_bookkeeping = false;                             // chain with per-view delivery + host lists (L3D_HOST_BOOKKEEPING=1: A/B, cross-check of the device products)
    bool resident_products = false;                            // the last matchViews left its products on the device: no host lists exist
    int last_match_path = -1;                                  // how the last matchViews ran (l3d_line3d_match_path)
    bool partitioned = false;                                  // ... PARTITIONED over the ranks of a job (l3d_line3d_partition_run): finish is collective (l3d_line3d_finish_sharded)
    l3d_exchange_fn part_exchange = nullptr;                   // the job's all-gather, for the collective finish
    void* part_user = nullptr;
    std::vector<l3d_chain_summary> chain_summary;
    int64_t resident_n_pot = 0;
    int shard_world_seen = 0, shard_slot_records_seen = 0;     // sharded native run: slot / candidate sizes a capacity verdict made necessary
    size_t shard_cand_cap_seen = 0;

    // final hypotheses
    std::vector<Hyp> hyps;                                     // best_match_ in key order
    std::vector<std::vector<int>> best_idx;                    // per view index: seg -> hyp index or -1
    EdgeVec A;                     // the affinity list on the host -- filled on demand (ensure_edges) when it was left on the device
    size_t n_edges = 0;
    bool A_on_host = true;
    std::vector<Key> local2global;
    std::vector<FinalLine> result;
    std::vector<size_t> hyp_begin;                             // per view index: first hypothesis (greedy_selection)

    // flat tables of the device affinity fill (l3d_affinity_input): kept between calls (no allocation, no page faults); the
    // collinearity part only changes with the set of views (drop_plan)
    struct AffTables {
        std::vector<l3d_hypothesis, NoInitAlloc<l3d_hypothesis>> hyp;
        std::vector<float, NoInitAlloc<float>> score, coll_w;
        std::vector<int32_t, NoInitAlloc<int32_t>> hyp_dense, best, pot_tgt, coll_other;
        std::vector<int64_t, NoInitAlloc<int64_t>> pot_start, coll_start;
        std::vector<uint32_t, NoInitAlloc<uint32_t>> hyp_cam;                  // camera id per hypothesis (device line fit)
        bool coll_valid = false;
    } aff;
    std::vector<int32_t> node_hyp;                             // hypothesis of every node of the affinity graph (device fill)
    std::vector<std::vector<int32_t>> aff_vt;                  // per view: its targets as dense ids (scratch of the table flattening)

    // statistics
    double stat_pairs = 0, stat_raw = 0, stat_kept = 0;
    int stat_last_tbm = -1;                                    // to-be-matched count of the view being committed
    double t_match = 0, t_gpu_call = 0, t_commit = 0, t_finalize = 0, t_affinity = 0, t_cluster = 0;

    int fail(int code, const std::string& m) { err = m; return code; }
    View* find_view(uint32_t id) { auto it = views.find(id); return it == views.end() ? nullptr : &it->second; }
    bool vn_has(uint32_t a, uint32_t b) const
    {
        auto it = visual_neighbors.find(a);
        return it != visual_neighbors.end() && std::binary_search(it->second.begin(), it->second.end(), b);
    }
};

// the switches of the handle's context (read once at l3d_ctx_create; l3d_set_option changes them)
static inline const l3d::Options& hopt(const l3d_line3d* h) { return l3d::ctx_options(h->ctx); }


typedef l3d_line3d L;

namespace l3dh {

// what a view hands to the device seam (l3d_match_view / l3d_chain_view), marshalled once per set of views
struct Marshal {
    std::vector<float> F, RtKinv, P, centers;
    std::vector<int32_t> offsets, tbm;
    std::vector<uint32_t> l2g;
    float RtKinv_src[9], C_src[3];
    float spatial_k;
};

struct ChainFinalizer;                             // (line3d_host_chain.cpp)

struct ChainUser {
    L* h;
    const std::vector<uint32_t>* order;
    const std::vector<int>* n_tbm;
    const std::vector<std::vector<int32_t>>* src_idx;
    struct ChainFinalizer* fin;
};

// Line3D::matchViews as one device-resident chain (l3d_match_chain): the schedule is simulated first (it does not
// depend on data), then the GPU runs ahead while the callback does the bookkeeping of each finished view.
// The static schedule of matchViews + the host-side finaliser, shared by the single-GPU chain and the sharded chain.
struct ChainPlan {
    size_t n = 0;
    std::vector<Marshal> ms;
    std::vector<std::vector<int32_t>> src_cam, src_idx;
    std::vector<l3d_chain_view> cv;
    std::vector<int> n_tbm;
    KeySet matched_final;                           // matched_ after the last view of the schedule
    ChainFinalizer* fin = nullptr;                  // the pipeline object's persistent finaliser
    // static tables of the finaliser (ChainFinalizer)
    bool fin_tables = false;
    std::vector<int> own_index, pending0;
    std::vector<std::vector<std::pair<uint32_t, size_t>>> targets;
    std::vector<std::vector<int>> contributors;
    ChainUser user;
    l3d_shard_chain* shard = nullptr;
    double t0 = 0;
};

// ---- line3d_host_views.cpp / line3d_host_chain.cpp / line3d_host_finish.cpp: what they call in one another and what the facade calls
void mark_matched(L* h, const View& v);
int match_views_sync(L* h);
void marshal_view(L* h, View& v, Marshal& m);
void match_begin(L* h);
void finalize_pot_range(std::vector<std::pair<uint32_t, Key>>& p, uint32_t lo, uint32_t hi);
void chain_notify(ChainFinalizer* f, int k);
void add_matches(View& v, const l3d_match* m, size_t n, bool remove_old, bool only_best);
V3 inverse_transform(const L* h, V3 P);
void process_worldpoints(L* h, uint32_t viewID, const uint32_t* wps, int n);
int make_view(L* h, uint32_t id, unsigned width, unsigned height, const float* segs, int n,
              const double* K, const double* R, const double* t,
              const int32_t* coll_i = nullptr, const int32_t* coll_j = nullptr, const float* coll_w = nullptr, int n_coll = -1);
int prepare(L* h);
int match_views(L* h);
ChainPlan* get_plan(L* h);
void drop_plan(L* h);
void dense_map(L* h, std::vector<uint32_t>& ids, std::vector<int32_t>& base);
int chain_callback(void* user, int index, int verified, const l3d_match* kept, int n_kept, const float* best, int n_best, int n_cand);
int adopt_resident_products(L* h, ChainPlan& P);
void start_finalizer(L* h, ChainPlan& P);
void perform_clustering(const l3d_edge* edges_in, size_t n_edges, int numNodes, float c, std::vector<int>& labels, bool presorted = false);
int greedy_selection_resident(L* h);
void greedy_selection(L* h);
void finish_chain_host(L* h, ChainPlan& P, bool ok);
void finalize_matching(L* h);
int ensure_edges(L* h);
int compute_view(L* h, View& v, int s0, int s1, l3d_match** out, int* n_out, float* median, float** best, int* n_best);
void commit_view(L* h, View& v, const l3d_match* matches, int n, float median_depth);
int cluster_segments_2D(L* h, bool perform_diff);
bool plan_chain(L* h, ChainPlan& P);
const M3& fundamental(L* h, uint32_t a, uint32_t b);
void find_visual_neighbors(L* h);
int transform_geometry(L* h);
void localized_existing(L* h, View& v, std::vector<l3d_match>& out);
void write_pending_caches(L* h);
int compute_pending_collinearities(L* h);
void set_collinearities(View& v, const int32_t* ci, const int32_t* cj, const float* cw, int cn);
void finalize_view_pot(std::vector<std::pair<uint32_t, Key>>& p, size_t S);
int check_resident_products(L* h, ChainPlan& P);
int match_views_resident(L* h, ChainPlan& P, double t0);
void unproject_segment(const View& v, uint32_t id, float d1, float d2, Hyp& o);
int best_of(const L* h, Key k);
void perform_clustering_grouped(const l3d_edge* sorted, const int32_t* group_start, int n_groups, int numNodes, float c, std::vector<int>& labels);
int perform_diffusion(L* h, const EdgeVec& A, int n, EdgeVec& out);
void align_cluster(const std::vector<std::pair<Key, std::pair<V3, V3>>>& t3, std::vector<std::pair<V3, V3>>& aligned);
void pack_collinearities(L* h, const std::vector<size_t>& voff);
int fill_affinity_resident(L* h);
void destroy_finalizer(L* h);                       // joins the finaliser's worker threads

}  // namespace l3dh
