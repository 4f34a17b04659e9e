// line3d_host.cpp -- host side of the hot path above the C ABI: the L3D::Line3D pipeline
// (line3D.cc, view.cc) re-built on flat arrays.  It mirrors the reference's operator interface
// (addImage / addImage_fixed_sim / compute3Dmodel / getResult) and its order-dependent semantics
// (toBeMatched, reverse-match propagation, only-best overwrite, first-touch node numbering), and
// calls the HIP path exclusively through include/line3d_amd.h.  Cited line numbers refer to the
// reference files under /root/reference.
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

namespace {

typedef uint64_t Key;   // (camID << 32) | segID : orders like L3DSegment2D::operator< (commons.h:92-94)
inline Key mk(uint32_t cam, uint32_t seg) { return ((Key)cam << 32) | seg; }
inline uint32_t kcam(Key k) { return (uint32_t)(k >> 32); }
inline uint32_t kseg(Key k) { return (uint32_t)k; }

double now_s()
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

struct FinalLine {
    std::vector<Key> segs2D;
    std::vector<std::pair<V3, V3>> segs3D;
};

}  // namespace

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
    std::set<uint64_t> matched;                                // (a<<32|b): matched_[a][b]

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

namespace {

typedef l3d_line3d L;

// ------------------------------------------------------------------------------------------------
// segment2collinearities_ of a view from the relation's upper-triangle triplets (i < j, ascending (i, j)): both directions
// (segments.h:89-93), per segment in ascending order of the other segment
void set_collinearities(View& v, const int32_t* ci, const int32_t* cj, const float* cw, int cn)
{
    const int n = v.S();
    v.coll_start.assign((size_t)n + 1, 0);
    std::vector<int> cnt((size_t)n, 0);
    for (int k = 0; k < cn; ++k) { cnt[ci[k]]++; cnt[cj[k]]++; }
    for (int s = 0; s < n; ++s) v.coll_start[s + 1] = v.coll_start[s] + cnt[s];
    v.coll_other.resize((size_t)v.coll_start[n]);
    v.coll_w.resize((size_t)v.coll_start[n]);
    std::vector<int> cur(v.coll_start.begin(), v.coll_start.end() - 1);
    // triplets come sorted by (i,j), i<j: for a segment s its partners j>s arrive ascending, and its
    // partners i<s arrive ascending (ascending i) and before them in index order -> fill lower part first
    for (int k = 0; k < cn; ++k) { const int s = cj[k]; v.coll_other[cur[s]] = ci[k]; v.coll_w[cur[s]] = cw[k]; cur[s]++; }
    for (int k = 0; k < cn; ++k) { const int s = ci[k]; v.coll_other[cur[s]] = cj[k]; v.coll_w[cur[s]] = cw[k]; cur[s]++; }
}

// The collinearity relations the L3DSegments constructor computes per image (segments.h:73-101, one kernel launch and one dense
// S x S download each) for all views added since the last call, in one batch (l3d_compute_collinearity_batch)
int compute_pending_collinearities(L* h)
{
    std::vector<View*> pend;
    for (auto& kv : h->views) if (kv.second.coll_pending) pend.push_back(&kv.second);
    if (pend.empty()) return L3D_OK;
    std::vector<const float*> segs(pend.size());
    std::vector<int> ns(pend.size()), start(pend.size() + 1, 0);
    for (size_t i = 0; i < pend.size(); ++i) { segs[i] = pend[i]->segs.data(); ns[i] = pend[i]->S(); }
    int32_t *ci = nullptr, *cj = nullptr; float* cw = nullptr;
    int rc = l3d_compute_collinearity_batch(h->ctx, segs.data(), ns.data(), (int)pend.size(), L3D_DEF_COLLINEARITY_S, &ci, &cj, &cw, start.data());
    if (rc) return h->fail(rc, std::string("collinearity: ") + l3d_last_error(h->ctx));
    std::atomic<size_t> next{ 0 };
    l3d::on_threads((unsigned)std::max<size_t>(1, std::min<size_t>(l3d::host_threads(), pend.size())), [&](unsigned) {
        for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= pend.size()) break;
            set_collinearities(*pend[i], ci + start[i], cj + start[i], cw + start[i], start[i + 1] - start[i]);
            pend[i]->coll_pending = false;
        }
    });
    l3d_free(ci); l3d_free(cj); l3d_free(cw);
    h->aff.coll_valid = false;
    return L3D_OK;
}

// coll_i/coll_j/coll_w (optional): the directed entries of a cached segment2collinearities_ map, ascending (i, j) -- used as
// they are instead of computing the relation (Line3D::addImage with an existing segment cache, line3D.cc:160-168)
int make_view(L* h, uint32_t id, unsigned width, unsigned height, const float* segs, int n,
              const double* K, const double* R, const double* t,
              const int32_t* coll_i = nullptr, const int32_t* coll_j = nullptr, const float* coll_w = nullptr, int n_coll = -1)
{
    View v;
    v.id = id;
    memcpy(v.K.m, K, 72);
    memcpy(v.R.m, R, 72);
    v.t = { t[0], t[1], t[2] };
    v.width = width; v.height = height;
    v.pp[0] = (double)((float)width / 2.0f);        // view.cc:20-21
    v.pp[1] = (double)((float)height / 2.0f);
    v.unc_upper_px = h->unc_upper; v.unc_lower_px = h->unc_lower;
    v.segs.assign(segs, segs + (size_t)n * 4);
    v.coll_start.assign((size_t)n + 1, 0);
    if (h->use_collinearity && n_coll >= 0) {       // the map of the cache file: iteration order of the nested std::map = ascending (i, j)
        for (int k = 0; k < n_coll; ++k) {
            if (coll_i[k] < 0 || coll_i[k] >= n || coll_j[k] < 0 || coll_j[k] >= n) return h->fail(L3D_ERR_INVALID, "cached collinearity names a segment that does not exist");
            if (k && (coll_i[k] < coll_i[k - 1] || (coll_i[k] == coll_i[k - 1] && coll_j[k] <= coll_j[k - 1]))) return h->fail(L3D_ERR_INVALID, "cached collinearities are not in ascending (i, j) order");
            if (coll_i[k] == coll_j[k]) return h->fail(L3D_ERR_INVALID, "cached collinearity of a segment with itself");
            v.coll_start[(size_t)coll_i[k] + 1]++;
        }
        for (int s = 0; s < n; ++s) v.coll_start[(size_t)s + 1] += v.coll_start[(size_t)s];
        v.coll_other.assign(coll_j, coll_j + n_coll);
        v.coll_w.assign(coll_w, coll_w + n_coll);
    } else if (h->use_collinearity && n > 1) {      // L3DSegments ctor, segments.h:73-101: computed for all new views together, in prepare()
        v.coll_pending = true;
    }
    v.derive();
    h->views[id] = std::move(v);
    h->aff.coll_valid = false;
    return L3D_OK;
}

// Line3D::processWorldpointList, line3D.cc:1874-1935
void process_worldpoints(L* h, uint32_t viewID, const uint32_t* wps, int n)
{
    h->num_wps[viewID] = 0;
    for (int i = 0; i < n; ++i) {
        std::vector<uint32_t>& w2v = h->worldpoints2views[wps[i]];
        std::sort(w2v.begin(), w2v.end());
        if (w2v.size() == 2) {
            const uint32_t v1 = w2v[0], v2 = w2v[1];
            h->common_wps[v1][v2] += 1;
            h->common_wps[v2][v1] += 1;
            ++h->num_wps[v1];
            ++h->num_wps[v2];
        }
        if (w2v.size() >= 2) {
            for (uint32_t v : w2v) {
                h->common_wps[v][viewID] += 1;
                h->common_wps[viewID][v] += 1;
            }
            ++h->num_wps[viewID];
        }
        if (std::find(w2v.begin(), w2v.end(), viewID) == w2v.end()) w2v.push_back(viewID);
    }
}

// Line3D::findVisualNeighbors, line3D.cc:476-549
void find_visual_neighbors(L* h)
{
    h->visual_neighbors.clear();
    for (auto& it : h->common_wps) {
        if (h->view_similarities.count(it.first)) continue;
        for (auto& n : it.second) {
            const float sim = 2.0f * float(n.second) / float(h->num_wps[it.first] + h->num_wps[n.first]);
            if (sim > 1e-12) h->view_similarities[it.first][n.first] = sim;
        }
    }
    struct VN { uint32_t cam; float sim; };
    for (auto& sit : h->view_similarities) {
        View* self = h->find_view(sit.first);
        std::vector<VN> vn;
        if (self) {
            for (auto& n : sit.second) {
                View* o = h->find_view(n.first);
                if (!o || !((float)norm(self->C - o->C) > h->min_baseline)) continue;
                bool ok = true;
                for (const VN& e : vn)
                    if ((float)norm(h->find_view(e.cam)->C - o->C) <= h->min_baseline) { ok = false; break; }
                if (ok) vn.push_back({ n.first, n.second });
            }
        }
        std::stable_sort(vn.begin(), vn.end(), [](const VN& a, const VN& b) { return a.sim > b.sim; });
        if (h->matching_neighbors > 0 && (int)vn.size() > h->matching_neighbors) vn.resize((size_t)h->matching_neighbors);
        std::vector<uint32_t>& out = h->visual_neighbors[sit.first];
        for (const VN& e : vn) out.push_back(e.cam);
        std::sort(out.begin(), out.end());
    }
}

// Line3D::transformGeometry + findSimilarityTransform + euclideanTransformation + applyTransformation,
// line3D.cc:552-617, 1694-1779
int transform_geometry(L* h)
{
    h->fundamentals.clear();
    const double size = (double)h->views.size();
    std::vector<V3> in_pts;
    V3 m;
    for (auto& kv : h->views) { m = m + kv.second.C; in_pts.push_back(kv.second.C); }
    m = m / size;
    double q = 0.0;
    for (auto& p : in_pts) q += norm(p - m);
    q /= size;
    q = (double)sqrtf(2.0f) / q;
    std::vector<V3> out_pts;
    V3 cog_out;
    for (auto& p : in_pts) {
        const V3 t3 = { q * p.x + (-q * m.x), q * p.y + (-q * m.y), q * p.z + (-q * m.z) };
        cog_out = cog_out + t3;
        out_pts.push_back(t3);
    }
    cog_out = cog_out / size;
    const size_t n = in_pts.size();
    double scales_sum = 0.0;
    for (size_t i = 0; i < n; ++i) scales_sum += norm(out_pts[i] - cog_out) / norm(in_pts[i] - m);
    const double scale = scales_sum / double(n);
    const V3 cog_in = m * scale;
    M3 H;
    for (size_t i = 0; i < n; ++i) {
        const V3 a = in_pts[i] * scale - cog_in, b = out_pts[i] - cog_out;
        const double bv[3] = { b.x, b.y, b.z }, av[3] = { a.x, a.y, a.z };
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H(r, c) += bv[r] * av[c];
    }
    M3 U, V; double s[3];
    l3d::la::svd3(H, U, s, V);
    M3 Vt = transpose(V);
    M3 Rm = mul(U, Vt);
    if (det(Rm) < 0) { for (int c = 0; c < 3; ++c) Vt(2, c) *= -1; Rm = mul(U, Vt); }
    V3 tt = cog_out - mul(Rm, cog_in);
    tt = tt / scale;
    double Q[16] = { Rm(0, 0), Rm(0, 1), Rm(0, 2), tt.x * scale, Rm(1, 0), Rm(1, 1), Rm(1, 2), tt.y * scale,
                     Rm(2, 0), Rm(2, 1), Rm(2, 2), tt.z * scale, 0, 0, 0, 1 };
    double Qinv[16];
    if (!inverse4(Q, Qinv)) return h->fail(L3D_ERR_INVALID, "transformGeometry: singular similarity transform");
    h->transf_scale_inv = 1.0 / scale;
    h->transf_Rinv = transpose(Rm);
    h->transf_tneg = tt * -1.0;
    for (auto& kv : h->views) kv.second.transform(Qinv, scale);
    return L3D_OK;
}

V3 inverse_transform(const L* h, V3 P) { return mul(h->transf_Rinv, P * h->transf_scale_inv + h->transf_tneg); }   // :1782-1786

// Line3D::fundamental, line3D.cc:1968-1993 (+ cache both ways, :1949-1965)
const M3& fundamental(L* h, uint32_t a, uint32_t b)
{
    const uint64_t key = ((uint64_t)a << 32) | b;
    auto it = h->fundamentals.find(key);
    if (it != h->fundamentals.end()) return it->second;
    const View& v1 = h->views[a];
    const View& v2 = h->views[b];
    const M3 R = mul(v2.R, transpose(v1.R));
    const V3 t = v2.t - mul(R, v1.t);
    M3 T;
    T(0, 0) = 0.0;  T(0, 1) = -t.z; T(0, 2) = t.y;
    T(1, 0) = t.z;  T(1, 1) = 0.0;  T(1, 2) = -t.x;
    T(2, 0) = -t.y; T(2, 1) = t.x;  T(2, 2) = 0.0;
    const M3 E = mul(T, R);
    const M3 F = mul(mul(inverse(transpose(v2.K)), E), inverse(v1.K));
    h->fundamentals[((uint64_t)b << 32) | a] = transpose(F);
    return h->fundamentals[key] = F;
}

// The float tables of performMatching, line3D.cc:716-803
struct Marshal {
    std::vector<float> F, RtKinv, P, centers;
    std::vector<int32_t> offsets, tbm;
    std::vector<uint32_t> l2g;
    float RtKinv_src[9], C_src[3];
    float spatial_k;
};

void marshal_view(L* h, View& v, Marshal& m)
{
    const std::vector<uint32_t>& nbs = h->visual_neighbors[v.id];
    const size_t N = nbs.size();
    m.F.resize(N * 9); m.RtKinv.resize(N * 9); m.P.resize(N * 12); m.centers.resize(N * 3);
    m.offsets.resize(N * 2); m.l2g.resize(N); m.tbm.clear();
    int total = 0;
    for (size_t loc = 0; loc < N; ++loc) {
        const uint32_t nb = nbs[loc];
        const View& o = h->views[nb];
        m.l2g[loc] = nb;
        if (!h->matched.count(((uint64_t)v.id << 32) | nb)) m.tbm.push_back((int32_t)loc);
        const M3& F = fundamental(h, v.id, nb);
        for (int k = 0; k < 9; ++k) { m.F[loc * 9 + k] = (float)F.m[k]; m.RtKinv[loc * 9 + k] = (float)o.RtKinv.m[k]; }
        for (int k = 0; k < 12; ++k) m.P[loc * 12 + k] = (float)o.P[k];
        m.centers[loc * 3 + 0] = (float)o.C.x; m.centers[loc * 3 + 1] = (float)o.C.y; m.centers[loc * 3 + 2] = (float)o.C.z;
        m.offsets[loc * 2] = total; m.offsets[loc * 2 + 1] = o.S();
        total += o.S();
    }
    for (int k = 0; k < 9; ++k) m.RtKinv_src[k] = (float)v.RtKinv.m[k];
    m.C_src[0] = (float)v.C.x; m.C_src[1] = (float)v.C.y; m.C_src[2] = (float)v.C.z;
    m.spatial_k = (float)v.specific_k((double)(2.0f * h->sigma_p));     // line3D.cc:820
}

// loadAndLocalizeExistingMatches, view.cc:200-224
void localized_existing(L* h, View& v, std::vector<l3d_match>& out)
{
    out.clear();
    if (!v.store_exists) return;
    const std::vector<uint32_t>& nbs = h->visual_neighbors[v.id];
    for (const l3d_match& mm : v.store) {
        auto it = std::lower_bound(nbs.begin(), nbs.end(), mm.camID2);
        if (it != nbs.end() && *it == mm.camID2) {
            l3d_match x = mm;
            x.camID2 = (uint32_t)(it - nbs.begin());
            out.push_back(x);
        }
    }
}

// L3DView::addMatches(matches, remove_old, only_best), view.cc:162-197
void add_matches(View& v, const l3d_match* m, size_t n, bool remove_old, bool only_best)
{
    std::vector<l3d_match> tmp;
    if (only_best) {
        // per segID1 (ascending): first match with the highest confidence in list order (stable sort, front)
        bool grouped = true;
        for (size_t i = 1; i < n && grouped; ++i) grouped = m[i - 1].segID1 <= m[i].segID1;
        if (grouped) {
            for (size_t i = 0; i < n;) {
                size_t b = i, j = i + 1;
                for (; j < n && m[j].segID1 == m[i].segID1; ++j) if (m[j].confidence > m[b].confidence) b = j;
                tmp.push_back(m[b]);
                i = j;
            }
        } else {
            uint32_t mx = 0;
            for (size_t i = 0; i < n; ++i) mx = std::max(mx, m[i].segID1);
            if ((size_t)mx <= 16 * n + 1024) {                 // dense segment ids: one table instead of a map
                std::vector<size_t> best((size_t)mx + 1, (size_t)-1);
                for (size_t i = 0; i < n; ++i) {
                    size_t& b = best[m[i].segID1];
                    if (b == (size_t)-1 || m[i].confidence > m[b].confidence) b = i;
                }
                for (size_t b : best) if (b != (size_t)-1) tmp.push_back(m[b]);
            } else {
                std::map<uint32_t, size_t> best;
                for (size_t i = 0; i < n; ++i) {
                    auto it = best.find(m[i].segID1);
                    if (it == best.end()) best[m[i].segID1] = i;
                    else if (m[i].confidence > m[it->second].confidence) it->second = i;
                }
                for (auto& kv : best) tmp.push_back(m[kv.second]);
            }
        }
        m = tmp.data(); n = tmp.size();
    }
    if (v.store_exists && !remove_old) v.store.insert(v.store.end(), m, m + n);
    else v.store.assign(m, m + n);
    v.store_exists = true;
}

// The host bookkeeping of performMatching after compute_pairwise_matches, line3D.cc:834-884
void commit_view(L* h, View& v, const l3d_match* matches, int n, float median_depth)
{
    const double t0 = now_s();
    v.median_depth = median_depth;                                       // :835
    // per distinct camera id seen in the list: target view, "push the reversed match" (:844-845), and whether
    // the match is a re-verified existing one (camera already matched before this view ran): its two
    // potential_correspondences_ entries were recorded when that camera kept it (set semantics, :864-865)
    struct CamInfo { uint32_t cam; View* o; bool push; bool known; std::vector<l3d_match> rev; };
    std::vector<CamInfo> cams;
    const bool early_return = h->stat_last_tbm == 0;                    // local camera ids: never "known"
    auto info = [&](uint32_t cam) -> CamInfo& {
        for (CamInfo& c : cams) if (c.cam == cam) return c;
        CamInfo c;
        c.cam = cam;
        c.o = h->find_view(cam);
        c.push = h->vn_has(cam, v.id) && !h->matched.count(((uint64_t)cam << 32) | v.id);
        c.known = !early_return && h->matched.count(((uint64_t)v.id << 32) | cam) != 0;
        cams.push_back(std::move(c));
        return cams.back();
    };
    std::vector<std::pair<uint32_t, Key>>& mine = h->pot[(size_t)v.index];
    CamInfo* last = nullptr;
    for (int i = 0; i < n; ++i) {                                        // :838-866
        const l3d_match& mp = matches[i];
        if (!last || last->cam != mp.camID2) last = &info(mp.camID2);
        CamInfo& ci = *last;
        if (ci.push) {
            l3d_match r;
            r.segID1 = mp.segID2; r.camID2 = v.id; r.segID2 = mp.segID1; r.confidence = 0.0f;
            r.depths[0] = mp.depths[2]; r.depths[1] = mp.depths[3]; r.depths[2] = mp.depths[0]; r.depths[3] = mp.depths[1];
            ci.rev.push_back(r);
        }
        if (ci.known) continue;
        mine.emplace_back(mp.segID1, mk(ci.cam, mp.segID2));
        if (ci.o) h->pot[(size_t)ci.o->index].emplace_back(mp.segID2, mk(v.id, mp.segID1));
        else h->pot_foreign.emplace_back(mk(ci.cam, mp.segID2), mk(v.id, mp.segID1));
    }
    std::sort(cams.begin(), cams.end(), [](const CamInfo& a, const CamInfo& b) { return a.cam < b.cam; });
    for (CamInfo& c : cams)                                              // :868-872 (ascending camera id)
        if (!c.rev.empty()) add_matches(h->views[c.cam], c.rev.data(), c.rev.size(), false, false);
    for (uint32_t nb : h->visual_neighbors[v.id]) {                      // :875-881
        h->matched.insert(((uint64_t)v.id << 32) | nb);
        if (h->vn_has(nb, v.id)) h->matched.insert(((uint64_t)nb << 32) | v.id);
    }
    add_matches(v, matches, (size_t)n, true, true);                      // :884
    if (h->keep_view_matches) h->view_matches[v.id].assign(matches, matches + n);
    h->stat_kept += n;
    h->t_commit += now_s() - t0;
}

int compute_view(L* h, View& v, int s0, int s1, l3d_match** out, int* n_out, float* median, float** best, int* n_best)
{
    Marshal m;
    marshal_view(h, v, m);
    std::vector<l3d_match> existing;
    localized_existing(h, v, existing);
    if (s1 < 0) s1 = v.S();
    h->stat_last_tbm = (int)m.tbm.size();
    *median = 1.0f;                                                      // line3D.cc:811
    const double t0 = now_s();
    int rc = l3d_compute_pairwise_matches(h->ctx, v.segs.data(), v.S(), m.RtKinv_src, m.C_src,
                                          v.nb_segs.data(), m.offsets.data(), (int)m.l2g.size(),
                                          m.F.data(), m.RtKinv.data(), m.centers.data(), m.P.data(),
                                          m.tbm.data(), (int)m.tbm.size(), existing.data(), (int)existing.size(), m.l2g.data(),
                                          v.k_upper, v.k_lower, h->sigma_p, h->sigma_a, m.spatial_k, s0, s1,
                                          out, n_out, median, best, n_best);
    h->t_gpu_call += now_s() - t0;
    if (rc) return h->fail(rc, std::string("compute_pairwise_matches: ") + l3d_last_error(h->ctx));
    double st[4];
    l3d_last_stats(h->ctx, st);
    h->stat_pairs += st[0];
    h->stat_raw += st[1];
    return L3D_OK;
}

// One segment's entries tmp[b, e) -> sorted by key, duplicates dropped, appended at out[w...]; returns the new w.
// The list is the view's own forward entries (ascending key) followed by the reverse entries of the views that matched it
// (ascending view, ascending segment = ascending key): two sorted runs, merged linearly; anything else (more runs) falls
// back to an insertion sort.
inline size_t emit_sorted_unique(std::pair<uint32_t, Key>* tmp, size_t b, size_t e, std::pair<uint32_t, Key>* out, size_t w)
{
    if (b >= e) return w;
    size_t cut = e, descents = 0;
    for (size_t i = b + 1; i < e; ++i) if (tmp[i].second < tmp[i - 1].second) { if (!descents) cut = i; ++descents; }
    const size_t w0 = w;
    auto put = [&](const std::pair<uint32_t, Key>& x) { if (w == w0 || out[w - 1].second != x.second) out[w++] = x; };
    if (descents <= 1) {
        size_t i = b, j = cut;
        while (i < cut && j < e) { if (tmp[j].second < tmp[i].second) put(tmp[j++]); else put(tmp[i++]); }
        while (i < cut) put(tmp[i++]);
        while (j < e) put(tmp[j++]);
        return w;
    }
    for (size_t i = b + 1; i < e; ++i) {
        auto x = tmp[i];
        size_t j = i;
        for (; j > b && tmp[j - 1].second > x.second; --j) tmp[j] = tmp[j - 1];
        tmp[j] = x;
    }
    for (size_t i = b; i < e; ++i) put(tmp[i]);
    return w;
}

// potential_correspondences_ becomes a sorted, de-duplicated adjacency per view (it is a std::map of
// std::maps in the reference: set semantics, ascending iteration)
void finalize_view_pot(std::vector<std::pair<uint32_t, Key>>& p, size_t S)
{
    if (p.empty()) return;
    bool in_range = true;
    for (auto& e : p) if (e.first >= S) { in_range = false; break; }
    if (!in_range) { std::sort(p.begin(), p.end()); p.erase(std::unique(p.begin(), p.end()), p.end()); return; }
    // stable counting sort on the segment, then the (short, nearly sorted) per-segment key lists
    static thread_local std::vector<uint32_t> cnt;                      // scratch reused by the worker thread
    static thread_local std::vector<std::pair<uint32_t, Key>> tmp;
    cnt.assign(S + 1, 0);
    for (auto& e : p) cnt[e.first + 1]++;
    for (size_t i = 0; i < S; ++i) cnt[i + 1] += cnt[i];
    if (tmp.size() < p.size()) tmp.resize(p.size());
    for (auto& e : p) tmp[cnt[e.first]++] = e;
    size_t b = 0, w = 0;
    for (size_t s = 0; s < S; ++s) {
        const size_t e = cnt[s];
        w = emit_sorted_unique(tmp.data(), b, e, p.data(), w);
        b = e;
    }
    p.resize(w);
}

// the same normal form for the entries of one segment range [lo, hi) (one of the parallel parts of a view's merge)
void finalize_pot_range(std::vector<std::pair<uint32_t, Key>>& p, uint32_t lo, uint32_t hi)
{
    if (p.empty()) return;
    bool in_range = true;
    for (auto& e : p) if (e.first < lo || e.first >= hi) { in_range = false; break; }
    if (!in_range || hi - lo > (1u << 24)) { std::sort(p.begin(), p.end()); p.erase(std::unique(p.begin(), p.end()), p.end()); return; }
    static thread_local std::vector<uint32_t> cnt;
    static thread_local std::vector<std::pair<uint32_t, Key>> tmp;
    const size_t n = hi - lo;
    cnt.assign(n + 1, 0);
    for (auto& e : p) cnt[e.first - lo + 1]++;
    for (size_t i = 0; i < n; ++i) cnt[i + 1] += cnt[i];
    if (tmp.size() < p.size()) tmp.resize(p.size());
    for (auto& e : p) tmp[cnt[e.first - lo]++] = e;
    size_t b = 0, w = 0;
    for (size_t s = 0; s < n; ++s) {
        const size_t e = cnt[s];
        w = emit_sorted_unique(tmp.data(), b, e, p.data(), w);
        b = e;
    }
    p.resize(w);
}

void finalize_matching(L* h)
{
    const double t0 = now_s();
    // views are independent here: a few host threads
    const size_t nv = h->pot.size();
    const unsigned nt = std::max(1u, std::min(8u, std::min((unsigned)nv, l3d::usable_cpus())));
    std::atomic<size_t> next(0);
    auto work = [&]() {
        for (size_t vi = next.fetch_add(1); vi < nv; vi = next.fetch_add(1))
            finalize_view_pot(h->pot[vi], (size_t)h->vlist[vi]->S());
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    std::sort(h->pot_foreign.begin(), h->pot_foreign.end());
    h->pot_foreign.erase(std::unique(h->pot_foreign.begin(), h->pot_foreign.end()), h->pot_foreign.end());
    h->t_finalize += now_s() - t0;
}

// serializeToFile of addImage with loadAndStoreSegments (line3D.cc:180-182), deferred to prepare(): the collinearity map of
// segment2collinearities_ as directed entries, ascending (i, j)
void write_pending_caches(L* h)
{
    for (auto& kv : h->views) {
        View& v = kv.second;
        if (v.cache_to_write.empty()) continue;
        std::vector<int32_t> ci, cj;
        std::vector<float> cw;
        for (int s = 0; s < v.S(); ++s)
            for (int q = v.coll_start[(size_t)s]; q < v.coll_start[(size_t)s + 1]; ++q) { ci.push_back(s); cj.push_back(v.coll_other[(size_t)q]); cw.push_back(v.coll_w[(size_t)q]); }
        const int rc = l3d_segment_cache_write(v.cache_to_write.c_str(), v.segs.data(), v.S(), ci.data(), cj.data(), cw.data(), (int)ci.size(), 17);
        if (rc && h->verbose) fprintf(stderr, "[L3D] could not write %s\n", v.cache_to_write.c_str());      // (the reference's ofstream fails silently)
        v.cache_to_write.clear();
    }
}

void drop_plan(L* h);           // the cached matchViews schedule depends on the set of views and their neighbours

int prepare(L* h)
{
    drop_plan(h);
    if (h->views.size() < 4) return h->fail(L3D_ERR_INVALID, "not enough images! can't compute 3D model...");   // line3D.cc:347-351
    const bool timing = hopt(h).timing != 0;
    double tl = now_s();
    auto lap = [&](const char* what) { if (timing) { const double t = now_s(); fprintf(stderr, "[l3d prepare] %-28s %8.2f ms\n", what, (t - tl) * 1e3); tl = t; } };
    h->computation = true;
    find_visual_neighbors(h);
    int rc = transform_geometry(h);
    if (rc) return rc;
    lap("neighbours + normalisation");
    h->vlist.clear();
    int idx = 0;
    for (auto& kv : h->views) { kv.second.index = idx++; h->vlist.push_back(&kv.second); }
    // residency: every view's neighbour tile (concatenated neighbour segments) and its own segments stay
    // in HBM for the whole run (the reference re-uploads them per view, line3D.cc:793-800)
    {
        int nd_all = 0;
        for (View* v : h->vlist) nd_all += v->S();
        const int nv_all = (int)h->vlist.size(), nn = h->matching_neighbors;
        // the finishing stages' arenas are reserved while the tiles are built and copied (a hint: a stage that needs more still gets it).
        // Nobody waits for the code objects here: they load in the background since the object was created (l3d_warm_up), the modules this
        // function and matchViews launch from first
        int reserve_rc = L3D_OK;
        double t_reserve = 0;
        std::thread warm([h, nd_all, nv_all, nn, &reserve_rc, &t_reserve]() { const double a0 = now_s(); reserve_rc = l3d_reserve_hint(h->ctx, nd_all, nv_all, nn); t_reserve = now_s() - a0; });
        std::atomic<size_t> next{ 0 };
        l3d::on_threads((unsigned)std::max<size_t>(1, std::min<size_t>(l3d::host_threads(), h->vlist.size())), [&](unsigned) {
            for (;;) {
                const size_t i = next.fetch_add(1, std::memory_order_relaxed);
                if (i >= h->vlist.size()) break;
                View* v = h->vlist[i];
                v->nb_segs.clear();
                auto it = h->visual_neighbors.find(v->id);
                if (it != h->visual_neighbors.end())
                    for (uint32_t nb : it->second) { const View& o = h->views.find(nb)->second; v->nb_segs.insert(v->nb_segs.end(), o.segs.begin(), o.segs.end()); }
                if (v->nb_segs.empty()) v->nb_segs.resize(4, 0.0f);
            }
        });
        std::vector<const float*> arrs;
        std::vector<int> cnts;
        for (View* v : h->vlist) { arrs.push_back(v->segs.data()); cnts.push_back(v->S()); arrs.push_back(v->nb_segs.data()); cnts.push_back((int)(v->nb_segs.size() / 4)); }
        rc = l3d_register_segments_batch(h->ctx, arrs.data(), cnts.data(), (int)arrs.size());
        const double t_join0 = now_s();
        warm.join();
        if (timing) fprintf(stderr, "[l3d prepare]   (arenas of the finishing stages reserved in %.2f ms on their own thread; waited %.2f ms for it)\n", t_reserve * 1e3, (now_s() - t_join0) * 1e3);
        if (reserve_rc && (h->verbose || timing)) fprintf(stderr, "[l3d prepare] reserving the finishing stages' arenas ahead failed (%d): they are allocated when first needed\n", reserve_rc);
        if (rc) return h->fail(rc, std::string("register_segments: ") + l3d_last_error(h->ctx));
    }
    lap("neighbour tiles + residency");
    rc = compute_pending_collinearities(h);             // (the segments are resident now: nothing is uploaded again)
    if (rc) return rc;
    write_pending_caches(h);
    lap("collinearity (all views)");
    h->prepared = true;
    return L3D_OK;
}

// reset of everything matchViews produces (line3D.cc:355-358 + the views' match files)
void match_begin(L* h)
{
    h->resident_products = false;
    h->matched.clear();
    h->pot.resize(h->vlist.size());                     // (capacities survive from an earlier pass)
    for (auto& pv : h->pot) pv.clear();
    h->pot_foreign.clear();
    h->view_matches.clear();
    h->order.clear();
    for (View* v : h->vlist) { v->store.clear(); v->store_exists = false; v->median_depth = 1.0f; }
    for (auto& kv : h->visual_neighbors)
        if (!kv.second.empty() && h->views.count(kv.first)) h->order.push_back(kv.first);    // line3D.cc:626-632
    h->stat_pairs = h->stat_raw = h->stat_kept = 0;
    h->t_match = h->t_gpu_call = h->t_commit = h->t_finalize = 0;
}

// Line3D::matchViews, line3D.cc:620-648 -- one view after the other through the per-view seam call
int match_views_sync(L* h)
{
    const double t0 = now_s();
    match_begin(h);
    for (uint32_t id : h->order) {
        View& v = h->views[id];
        l3d_match* m = nullptr; int n = 0; float med = 1.0f;
        int rc = compute_view(h, v, 0, -1, &m, &n, &med, nullptr, nullptr);
        if (rc) return rc;
        commit_view(h, v, m, n, med);
        l3d_free(m);
    }
    finalize_matching(h);
    h->t_match = now_s() - t0;
    return L3D_OK;
}

// the static part of commit_view: matched_ after view v has been processed (line3D.cc:875-881)
void mark_matched(L* h, const View& v)
{
    for (uint32_t nb : h->visual_neighbors[v.id]) {
        h->matched.insert(((uint64_t)v.id << 32) | nb);
        if (h->vn_has(nb, v.id)) h->matched.insert(((uint64_t)nb << 32) | v.id);
    }
}

struct ChainFinalizer;
void chain_notify(ChainFinalizer* f, int k);

struct ChainUser {
    L* h;
    const std::vector<uint32_t>* order;
    const std::vector<int>* n_tbm;
    const std::vector<std::vector<int32_t>>* src_idx;
    struct ChainFinalizer* fin;
};

// In the chain the reverse matches travel on the device, so the host bookkeeping of a view shrinks to keeping its
// list (performMatching's `matches`, line3D.cc:822-884); potential_correspondences_ and the only-best stores are
// built from the kept lists afterwards, in parallel (finalize_chain).
int chain_callback(void* user, int index, int verified, const l3d_match* kept, int n_kept, const float* best, int n_best, int n_cand)
{
    ChainUser* u = static_cast<ChainUser*>(user);
    L* h = u->h;
    const double t0 = now_s();
    View& v = h->views[(*u->order)[(size_t)index]];
    KeptList& mine = h->saved[(size_t)index];
    if (!verified) {
        // cudawrapper.cu:877-878: the localized existing list comes back untouched (LOCAL camera ids, confidence 0).
        // It is what the earlier views pushed (line3D.cc:838-872), in push order: sources ascending, list order.
        mine.reset();
        const std::vector<uint32_t>& nbs = h->visual_neighbors[v.id];
        for (int a : (*u->src_idx)[(size_t)index])
            for (const l3d_match& mp : h->saved[(size_t)a]) {
                if (mp.camID2 != v.id) continue;
                l3d_match r;
                r.segID1 = mp.segID2; r.segID2 = mp.segID1; r.confidence = 0.0f;
                r.camID2 = (uint32_t)(std::lower_bound(nbs.begin(), nbs.end(), h->views[(*u->order)[(size_t)a]].id) - nbs.begin());
                r.depths[0] = mp.depths[2]; r.depths[1] = mp.depths[3]; r.depths[2] = mp.depths[0]; r.depths[3] = mp.depths[1];
                mine.own.push_back(r);
            }
        mine.use_own();
        v.median_depth = 1.0f;                          // line3D.cc:811,835
    } else {
        float median = 1.0f;                            // untouched when nothing was verified (cudawrapper.cu:955-956)
        if (n_cand > 0) {
            median = -1.0f;                             // cudawrapper.cu:1066-1073
            if (n_best > 0) {
                std::vector<float> d(best, best + (size_t)n_best * 2);
                std::nth_element(d.begin(), d.begin() + (long)(d.size() / 2), d.end());
                median = d[d.size() / 2];
            }
        }
        v.median_depth = median;
        mine.reset();
        mine.p = kept; mine.n = (size_t)n_kept;         // no copy: the list lives in the context's pinned arena
    }
    mark_matched(h, v);                                 // line3D.cc:875-881
    h->stat_kept += (double)mine.size();
    h->t_commit += now_s() - t0;
    chain_notify(u->fin, index);
    return 0;
}

// potential_correspondences_ (line3D.cc:861-865) and the only-best match files (line3D.cc:884, view.cc:165-183) from
// the kept lists, on a few host threads while the GPU is still busy with later views.  Two kinds of task:
//   split(k)    when the list of processed view k arrives: its entries are bucketed by the camera they point to
//               (reverse direction) and its own forward entries / only-best store are produced;
//   merge(view) when all lists that can mention a view are split: as kParts independent segment ranges (the split has
//               pre-sorted its entries into them) -- gather, counting sort by segment, linear merge of each segment's
//               two sorted runs, de-duplicate; the part that finishes last concatenates the ranges.
struct ChainFinalizer {
    L* h;
    // static tables of the schedule (owned by the cached ChainPlan):
    const std::vector<int>* own_index_ = nullptr;   // per view index: its position in the processing order or -1
    const std::vector<std::vector<std::pair<uint32_t, size_t>>>* targets_ = nullptr;   // per order index: (camera id, view index) receiving reverse entries, ascending id
    const std::vector<std::vector<int>>* contributors_ = nullptr;     // per view index: order indices of the views that list it as neighbour
    std::vector<char> own_sorted;                   // per view index: its own forward entries ascend by segment
    std::vector<std::atomic<int>> split_left;       // per order index: halves of the split still running (reverse entries / own entries)
    std::vector<std::vector<std::pair<size_t, std::array<std::vector<std::pair<uint32_t, Key>>, 4>>>>* buckets = nullptr;   // per order index: (target view index, entries); storage owned by the pipeline object
    std::vector<std::atomic<int>> pending;          // per view index: splits still missing
    static constexpr int kParts = 4;                // a view's merge runs as kParts independent segment ranges
    std::vector<std::atomic<int>> parts_left;       // per view index
    std::vector<std::array<std::vector<std::pair<uint32_t, Key>>, 4>>* parts = nullptr;   // storage owned by the pipeline object
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::pair<int, size_t>> queue;      // (0 / 1 = the two halves of a split, order index) or (2 = merge part, view index * kParts + part)
    bool done = false;
    std::vector<std::thread> workers;
    bool timing = false, trace = false;             // (set with h)
    struct LogRec { int kind, id; double t0, t1; };
    std::vector<LogRec> log;
    double t_split = 0, t_merge = 0, t_last_done = 0;
    int n_split = 0, n_merge = 0;

    int active = 0;                                 // jobs being executed (under mu)
    std::condition_variable cv_idle;
    ChainFinalizer() {}
    ~ChainFinalizer()
    {
        { std::lock_guard<std::mutex> lk(mu); done = true; }
        cv.notify_all();
        for (auto& t : workers) t.join();
    }
    // the worker threads live as long as the pipeline object; a pass only re-arms the counters (threads are idle here)
    void begin_pass(size_t nviews, size_t norder, const std::vector<int>& pending0)
    {
        if (pending.size() != nviews) { pending = std::vector<std::atomic<int>>(nviews); parts_left = std::vector<std::atomic<int>>(nviews); }
        if (split_left.size() != norder) split_left = std::vector<std::atomic<int>>(norder);
        for (size_t i = 0; i < nviews; ++i) pending[i] = pending0[i];
        for (auto& p : parts_left) p = 0;
        for (auto& p : split_left) p = 2;
        t_split = t_merge = 0; n_split = n_merge = 0; log.clear();
    }
    void push_merge(size_t vi) { parts_left[vi] = kParts; for (int r = 0; r < kParts; ++r) push(2, vi * kParts + (size_t)r); }

    void push(int kind, size_t id)
    {
        { std::lock_guard<std::mutex> lk(mu); queue.emplace_back(kind, id); }
        cv.notify_one();
    }
    // A finished view's kept list is split in two independent halves (two jobs, so that the LAST view's split -- the tail of
    // matchViews -- takes half as long): (0) the reverse entries, pre-sorted per target view and merge range; (1) its own
    // forward entries and the only-best store.  Whoever finishes second releases the merges that waited for this view.
    void split_reverse(size_t k)
    {
        const View& v = h->views[h->order[k]];
        const KeptList& lst = h->saved[k];
        auto& bk = (*buckets)[k];
        // cameras whose views receive the reverse entry of a kept match: the neighbours -- or, for an early-return view
        // (cudawrapper.cu:877-878: LOCAL camera ids come back), whatever views those numbers happen to name
        // (line3D.cc:861-865); ascending camera id, slot = position
        const std::vector<std::pair<uint32_t, size_t>>& tg = (*targets_)[k];
        if (bk.size() != tg.size()) bk.assign(tg.size(), {});        // (otherwise keep the entry vectors' capacity)
        std::vector<uint32_t> S_of(tg.size(), 1);       // segment count of each target: entries are pre-sorted into its merge ranges
        for (size_t i = 0; i < tg.size(); ++i) { bk[i].first = tg[i].second; for (auto& q : bk[i].second) q.clear(); S_of[i] = (uint32_t)std::max(1, h->vlist[tg[i].second]->S()); }
        size_t sl = (size_t)-1; uint32_t last_cam = 0xffffffffu;
        for (const l3d_match& m : lst) {
            if (m.camID2 != last_cam) {
                last_cam = m.camID2;
                auto it = std::lower_bound(tg.begin(), tg.end(), std::make_pair(last_cam, (size_t)0));
                sl = (it != tg.end() && it->first == last_cam) ? (size_t)(it - tg.begin()) : (size_t)-1;
            }
            if (sl != (size_t)-1) {
                const uint32_t part = m.segID2 >= S_of[sl] ? (uint32_t)(kParts - 1) : (uint32_t)((uint64_t)m.segID2 * kParts / S_of[sl]);
                bk[sl].second[part].emplace_back(m.segID2, mk(v.id, m.segID1));
            }
        }
        split_done(k);
    }
    void split_own(size_t k)
    {
        const View& v = h->views[h->order[k]];
        const KeptList& lst = h->saved[k];
        // own forward entries (already grouped by segment) and the only-best store do not depend on other lists
        std::vector<std::pair<uint32_t, Key>>& p = h->pot[(size_t)v.index];
        p.clear();
        p.reserve(lst.size() * 2);
        bool sorted = true;                             // (an early-return view's list is grouped by source view instead)
        for (const l3d_match& m : lst) { if (!p.empty() && m.segID1 < p.back().first) sorted = false; p.emplace_back(m.segID1, mk(m.camID2, m.segID2)); }
        own_sorted[(size_t)v.index] = sorted ? 1 : 0;
        add_matches(h->views[h->order[k]], lst.data(), lst.size(), true, true);
        split_done(k);
    }
    void split_done(size_t k)
    {
        if (--split_left[k] != 0) return;
        const View& v = h->views[h->order[k]];
        for (auto& e : (*buckets)[k]) if (--pending[e.first] == 0) push_merge(e.first);
        if (--pending[(size_t)v.index] == 0) push_merge((size_t)v.index);
    }
    // one segment range of a view's merge: gather (own forward entries are grouped by segment, the contributions are
    // not), normal form; the part that finishes last concatenates the ranges
    void merge_part(size_t vi, int r)
    {
        View& v = *h->vlist[vi];
        const uint32_t S = (uint32_t)v.S();
        // range r = segments s with floor(s * kParts / S) == r (the split has pre-sorted the contributions accordingly)
        const uint32_t lo = (uint32_t)(((uint64_t)S * (uint32_t)r + kParts - 1) / kParts), hi = r == kParts - 1 ? 0xffffffffu : (uint32_t)(((uint64_t)S * (uint32_t)(r + 1) + kParts - 1) / kParts);
        std::vector<std::pair<uint32_t, Key>>& p = h->pot[vi];
        std::vector<std::pair<uint32_t, Key>>& out = (*parts)[vi][(size_t)r];
        out.clear();
        const std::vector<int>& own_index = *own_index_;
        if (own_index[vi] >= 0 && own_sorted[vi]) {     // split(own) has put the forward entries there, ascending segment
            auto first = [](const std::pair<uint32_t, Key>& e, uint32_t x) { return e.first < x; };
            auto b = std::lower_bound(p.begin(), p.end(), lo, first);
            auto e = hi == 0xffffffffu ? p.end() : std::lower_bound(b, p.end(), hi, first);
            out.insert(out.end(), b, e);
        } else if (own_index[vi] >= 0) {
            for (auto& x : p) if (x.first >= lo && x.first < hi) out.push_back(x);
        }
        for (int k : (*contributors_)[vi])
            for (auto& e : (*buckets)[(size_t)k])
                if (e.first == vi) out.insert(out.end(), e.second[(size_t)r].begin(), e.second[(size_t)r].end());
        finalize_pot_range(out, lo, hi == 0xffffffffu ? std::max(S, lo) : hi);
        if (--parts_left[vi] == 0) {
            p.clear();
            for (int q = 0; q < kParts; ++q) p.insert(p.end(), (*parts)[vi][(size_t)q].begin(), (*parts)[vi][(size_t)q].end());
        }
    }
    void start(unsigned nthreads)
    {
        for (unsigned t = (unsigned)workers.size(); t < nthreads; ++t)
            workers.emplace_back([this]() {
                for (;;) {
                    std::pair<int, size_t> job;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [this]() { return done || !queue.empty(); });
                        if (queue.empty()) return;      // (done)
                        job = queue.back();
                        queue.pop_back();
                        ++active;
                    }
                    const double tj0 = now_s();
                    if (job.first == 0) split_reverse(job.second); else if (job.first == 1) split_own(job.second); else merge_part(job.second / kParts, (int)(job.second % kParts));
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        --active;
                        if (timing) { const double dt = now_s() - tj0; (job.first < 2 ? t_split : t_merge) += dt; (job.first < 2 ? n_split : n_merge) += 1; t_last_done = now_s();
                                      if (trace) log.push_back({ job.first, (int)job.second, tj0, t_last_done }); }
                    }
                    cv_idle.notify_all();
                }
            });
    }
    void notify(int k)
    {
        { std::lock_guard<std::mutex> lk(mu); queue.emplace_back(0, (size_t)k); queue.emplace_back(1, (size_t)k); }
        cv.notify_all();
    }
    void finish(bool drain)
    {
        // wait until every job has run (splits spawn merges while they run, so "no job queued or running" is final); without
        // `drain` (a failed chain) whatever was queued is dropped first
        const double td0 = now_s();
        {
            std::unique_lock<std::mutex> lk(mu);
            if (!drain) queue.clear();
            cv_idle.wait(lk, [this]() { return queue.empty() && active == 0; });
        }
        if (trace) for (size_t i = log.size() > 48 ? log.size() - 48 : 0; i < log.size(); ++i)
            fprintf(stderr, "[l3d finaliser job] kind %d id %d: start %+.3f end %+.3f ms (relative to the drain start)\n", log[i].kind, log[i].id, (log[i].t0 - td0) * 1e3, (log[i].t1 - td0) * 1e3);
        if (timing) fprintf(stderr, "[l3d finaliser] drain %.2f ms; %d split halves %.2f ms (avg %.3f), %d merges %.2f ms (avg %.3f)\n", (now_s() - td0) * 1e3, n_split, t_split * 1e3,
                            n_split ? t_split * 1e3 / n_split : 0.0, n_merge, t_merge * 1e3, n_merge ? t_merge * 1e3 / n_merge : 0.0);
    }
};

void chain_notify(ChainFinalizer* f, int k) { if (f) f->notify(k); }

// Line3D::matchViews as one device-resident chain (l3d_match_chain): the schedule is simulated first (it does not
// depend on data), then the GPU runs ahead while the callback does the bookkeeping of each finished view.
// The static schedule of matchViews + the host-side finaliser, shared by the single-GPU chain and the sharded chain.
struct ChainPlan {
    size_t n = 0;
    std::vector<Marshal> ms;
    std::vector<std::vector<int32_t>> src_cam, src_idx;
    std::vector<l3d_chain_view> cv;
    std::vector<int> n_tbm;
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

// simulate the schedule (it does not depend on data); false: fall back to the per-view path
bool plan_chain(L* h, ChainPlan& P)
{
    const size_t n = h->order.size();
    P.n = n;
    P.ms.assign(n, Marshal()); P.src_cam.assign(n, {}); P.src_idx.assign(n, {}); P.cv.assign(n, l3d_chain_view()); P.n_tbm.assign(n, 0);
    std::map<uint32_t, int> index_of;
    bool chain_ok = true;
    for (size_t k = 0; k < n && chain_ok; ++k) {
        View& v = h->views[h->order[k]];
        index_of[v.id] = (int)k;
        Marshal& m = P.ms[k];
        marshal_view(h, v, m);                          // toBeMatched from the simulated matched_ state
        P.n_tbm[k] = (int)m.tbm.size();
        std::vector<char> is_tbm(m.l2g.size(), 0);
        for (int32_t c : m.tbm) is_tbm[(size_t)c] = 1;
        for (size_t c = 0; c < m.l2g.size(); ++c) {
            if (is_tbm[c]) continue;
            auto it = index_of.find(m.l2g[c]);
            if (it == index_of.end() || it->second >= (int)k) { chain_ok = false; break; }   // cannot happen: matched => processed earlier
            P.src_cam[k].push_back((int32_t)c);
            P.src_idx[k].push_back(it->second);
        }
        if (m.tbm.empty()) {
            // the early return hands back LOCAL camera ids (cudawrapper.cu:877-878); if one of those numbers happens
            // to be a view that would accept reverse matches (line3D.cc:844-845) the data flow is no longer the
            // static one -> take the per-view path
            for (uint32_t c = 0; c < (uint32_t)m.l2g.size(); ++c)
                if (h->vn_has(c, v.id) && !h->matched.count(((uint64_t)c << 32) | v.id)) chain_ok = false;
        }
        l3d_chain_view& o = P.cv[k];
        o.view_id = v.id;
        o.src_segs = v.segs.data(); o.S_src = v.S();
        o.RtKinv_src = m.RtKinv_src; o.C_src = m.C_src;
        o.tgt_segs = v.nb_segs.data(); o.n_tgt = (int32_t)(v.nb_segs.size() / 4);
        o.offsets = m.offsets.data(); o.N = (int32_t)m.l2g.size();
        o.F = m.F.data(); o.RtKinv = m.RtKinv.data(); o.centers = m.centers.data(); o.P = m.P.data();
        o.to_be_matched = m.tbm.data(); o.n_tbm = (int32_t)m.tbm.size();
        o.local2global = m.l2g.data();
        o.source_cam = P.src_cam[k].data(); o.source_index = P.src_idx[k].data(); o.n_sources = (int32_t)P.src_cam[k].size();
        o.sigma_p = h->sigma_p; o.sigma_a = h->sigma_a; o.spatial_k = m.spatial_k;
        mark_matched(h, v);
    }
    h->matched.clear();                                 // back to the state matchViews starts from
    return chain_ok;
}

// the schedule is static: build it once per set of views (prepare() drops it)
ChainPlan* get_plan(L* h)
{
    if (h->plan_cache) return static_cast<ChainPlan*>(h->plan_cache);
    ChainPlan* P = new ChainPlan();
    if (!plan_chain(h, *P)) { delete P; return nullptr; }
    h->plan_cache = P;
    return P;
}
void drop_plan(L* h)
{
    h->aff.coll_valid = false;
    delete static_cast<ChainPlan*>(h->plan_cache);
    h->plan_cache = nullptr;
}

void start_finalizer(L* h, ChainPlan& P)
{
    const size_t n = P.n, nvl = h->vlist.size();
    h->saved.resize(n);                                 // (capacity of the per-view lists survives from an earlier pass)
    for (auto& lst : h->saved) lst.reset();
    if (!h->finalizer) h->finalizer = new ChainFinalizer();
    P.fin = static_cast<ChainFinalizer*>(h->finalizer);
    ChainFinalizer& fin = *P.fin;
    fin.h = h;
    fin.timing = hopt(h).timing != 0; fin.trace = hopt(h).timing >= 2;
    if (!P.fin_tables) {                                // who sends reverse entries to whom: part of the (static) schedule
        P.own_index.assign(nvl, -1); P.pending0.assign(nvl, 0); P.contributors.assign(nvl, {}); P.targets.assign(n, {});
        for (size_t k = 0; k < n; ++k) {
            const View& v = h->views[h->order[k]];
            P.own_index[(size_t)v.index] = (int)k;
            P.pending0[(size_t)v.index] += 1;           // its own list
            auto& tg = P.targets[k];
            if (P.n_tbm[k] != 0) {
                for (uint32_t nb : h->visual_neighbors[v.id]) { const View* o = h->find_view(nb); if (o) tg.emplace_back(nb, (size_t)o->index); }
            } else {                                    // early return: local camera ids 0..N-1 read as view ids
                const uint32_t N = (uint32_t)h->visual_neighbors[v.id].size();
                for (uint32_t c = 0; c < N; ++c) { const View* o = h->find_view(c); if (o) tg.emplace_back(c, (size_t)o->index); }
            }
            std::sort(tg.begin(), tg.end());
            for (auto& t : tg) {
                P.contributors[t.second].push_back((int)k);
                P.pending0[t.second] += 1;
            }
        }
        P.fin_tables = true;
    }
    fin.own_index_ = &P.own_index; fin.targets_ = &P.targets; fin.contributors_ = &P.contributors;
    fin.begin_pass(nvl, n, P.pending0);
    fin.own_sorted.assign(nvl, 1);
    fin.buckets = &h->fin_buckets;                      // (capacities survive from an earlier pass)
    fin.parts = &h->fin_parts;
    if (h->fin_parts.size() != nvl) h->fin_parts.assign(nvl, {});
    if (h->fin_buckets.size() != n) h->fin_buckets.assign(n, {});
    fin.start(std::max(1u, std::min(16u, l3d::usable_cpus())));
    P.user = ChainUser{ h, &h->order, &P.n_tbm, &P.src_idx, P.fin };
}

// after the last callback: wait for the workers, the LOCAL-id entries of early-return views, inspection copies
void finish_chain_host(L* h, ChainPlan& P, bool ok)
{
    const double t2 = now_s();
    P.fin->finish(ok);
    if (!ok) return;
    const size_t n = P.n;
    // early-return views (cudawrapper.cu:877-878) hand back LOCAL camera ids; where such a number names a view, the
    // reference records the pair under that view as well (line3D.cc:861-865): append and re-normalise (rare, tiny)
    h->pot_foreign.clear();
    for (size_t k = 0; k < n; ++k) {
        if (P.n_tbm[k] != 0) continue;
        const uint32_t vid = h->order[k];
        uint32_t last_cam = 0xffffffffu; bool foreign = false;
        for (const l3d_match& m : h->saved[k]) {        // (numbers that name a view went through the finaliser like any reverse entry)
            if (m.camID2 != last_cam) { last_cam = m.camID2; foreign = h->find_view(last_cam) == nullptr; }
            if (foreign) h->pot_foreign.emplace_back(mk(m.camID2, m.segID2), mk(vid, m.segID1));
        }
    }
    std::sort(h->pot_foreign.begin(), h->pot_foreign.end());
    h->pot_foreign.erase(std::unique(h->pot_foreign.begin(), h->pot_foreign.end()), h->pot_foreign.end());
    if (h->keep_view_matches) for (size_t k = 0; k < n; ++k) h->view_matches[h->order[k]].assign(h->saved[k].begin(), h->saved[k].end());
    h->t_finalize += now_s() - t2;
    if (hopt(h).check_pot) {
        // self-check (tests): every per-view list must be the plain normal form (sort + unique) of all its entries,
        // rebuilt here from the kept lists the slow way
        std::vector<std::vector<std::pair<uint32_t, Key>>> ref(h->pot.size());
        for (size_t k = 0; k < n; ++k) {
            const View& v = h->views[h->order[k]];
            for (const l3d_match& m : h->saved[k]) {
                ref[(size_t)v.index].emplace_back(m.segID1, mk(m.camID2, m.segID2));
                View* o = h->find_view(m.camID2);
                if (o) ref[(size_t)o->index].emplace_back(m.segID2, mk(v.id, m.segID1));
            }
        }
        for (size_t vi = 0; vi < ref.size(); ++vi) {
            std::sort(ref[vi].begin(), ref[vi].end());
            ref[vi].erase(std::unique(ref[vi].begin(), ref[vi].end()), ref[vi].end());
            if (ref[vi] != h->pot[vi]) { h->pot_check_failed = true; fprintf(stderr, "[l3d] potential-correspondence list of view index %zu differs from its normal form (%zu vs %zu entries)\n", vi, h->pot[vi].size(), ref[vi].size()); }
        }
    }
}

// the dense numbering of all segments: views in ascending id, dense id = base + segment
void dense_map(L* h, std::vector<uint32_t>& ids, std::vector<int32_t>& base)
{
    const size_t nv = h->vlist.size();
    ids.resize(nv); base.assign(nv + 1, 0);
    for (size_t i = 0; i < nv; ++i) { ids[i] = h->vlist[i]->id; base[i + 1] = base[i] + (int32_t)h->vlist[i]->S(); }
}

// L3D_CHECK_POT (tests): the device products against the plain host construction from the kept lists -- potential
// correspondences as the normal form (sort + unique) of all entries (line3D.cc:861-865), the only-best store of every view
// (view.cc:165-183: first match of the highest confidence per segment)
int check_resident_products(L* h, ChainPlan& P)
{
    std::vector<uint32_t> ids; std::vector<int32_t> base;
    dense_map(h, ids, base);
    const size_t nd = (size_t)base.back();
    std::vector<int64_t> pot_start(nd + 1);
    std::vector<int32_t> pot_tgt((size_t)h->resident_n_pot + 1);
    std::vector<l3d_match> best(nd + 1);
    int rc = l3d_chain_products_get(h->ctx, pot_start.data(), pot_tgt.data(), best.data());
    if (rc) return h->fail(rc, std::string("products_get: ") + l3d_last_error(h->ctx));
    std::vector<std::vector<std::pair<uint32_t, Key>>> ref(h->vlist.size());
    std::vector<std::vector<l3d_match>> lists(P.n);
    for (size_t k = 0; k < P.n; ++k) {
        l3d_match* m = nullptr; int n = 0;
        rc = l3d_chain_kept_list(h->ctx, (int)k, &m, &n);
        if (rc) return h->fail(rc, std::string("kept_list: ") + l3d_last_error(h->ctx));
        lists[k].assign(m, m + n);
        l3d_free(m);
        const View& v = h->views[h->order[k]];
        for (const l3d_match& mm : lists[k]) {
            ref[(size_t)v.index].emplace_back(mm.segID1, mk(mm.camID2, mm.segID2));
            View* o = h->find_view(mm.camID2);
            if (o) ref[(size_t)o->index].emplace_back(mm.segID2, mk(v.id, mm.segID1));
        }
    }
    bool ok = true;
    for (size_t vi = 0; vi < ref.size() && ok; ++vi) {
        std::sort(ref[vi].begin(), ref[vi].end());
        ref[vi].erase(std::unique(ref[vi].begin(), ref[vi].end()), ref[vi].end());
        const size_t S = (size_t)h->vlist[vi]->S();
        std::vector<std::vector<int32_t>> exp(S);
        for (auto& e : ref[vi]) {
            View* o = h->find_view(kcam(e.second));
            if (!o || e.first >= S || kseg(e.second) >= (uint32_t)o->S()) continue;       // (takes no part in the fill)
            exp[e.first].push_back(base[(size_t)o->index] + (int32_t)kseg(e.second));
        }
        for (size_t sg = 0; sg < S && ok; ++sg) {
            const size_t d = (size_t)base[vi] + sg;
            std::sort(exp[sg].begin(), exp[sg].end());
            const int64_t b = pot_start[d], e = pot_start[d + 1];
            if (e - b != (int64_t)exp[sg].size() || b < 0 || e > h->resident_n_pot || !std::equal(exp[sg].begin(), exp[sg].end(), pot_tgt.begin() + b)) {
                ok = false;
                fprintf(stderr, "[l3d] device potential correspondences of view index %zu segment %zu differ from the host construction (%lld vs %zu entries)\n", vi, sg, (long long)(e - b), exp[sg].size());
            }
        }
    }
    for (size_t k = 0; k < P.n && ok; ++k) {
        const View& v = h->views[h->order[k]];
        std::vector<int> bi((size_t)v.S(), -1);
        for (size_t i = 0; i < lists[k].size(); ++i) {
            const uint32_t sg = lists[k][i].segID1;
            if (sg >= (uint32_t)v.S()) continue;
            if (bi[sg] < 0 || lists[k][i].confidence > lists[k][(size_t)bi[sg]].confidence) bi[sg] = (int)i;
        }
        for (int sg = 0; sg < v.S() && ok; ++sg) {
            const l3d_match& got = best[(size_t)base[(size_t)v.index] + (size_t)sg];
            if (bi[(size_t)sg] < 0) { if (got.segID1 != 0xffffffffu) ok = false; }
            else if (memcmp(&got, &lists[k][(size_t)bi[(size_t)sg]], sizeof(l3d_match)) != 0) ok = false;
            if (!ok) fprintf(stderr, "[l3d] device best match of view %u segment %d differs from the host rule\n", v.id, sg);
        }
    }
    if (!ok) { h->pot_check_failed = true; return h->fail(L3D_ERR_INVALID, "L3D_CHECK_POT: the device products differ from the host construction"); }
    return L3D_OK;
}

// Line3D::matchViews with nothing but a few scalars per view coming back: the chain runs resident, the products of
// performMatching (potential_correspondences_, only-best stores, medians) are built on the device (l3d_products.hip)
// the facade's side of products that were built on the device (h->chain_summary filled by the builder): medians, matched marks,
// optional copies of the kept lists, the self-check of the tests
int adopt_resident_products(L* h, ChainPlan& P)
{
    for (size_t k = 0; k < P.n; ++k) {
        View& v = h->views[h->order[k]];
        v.median_depth = h->chain_summary[k].median_depth;      // line3D.cc:835
        h->stat_kept += h->chain_summary[k].n_kept;
        mark_matched(h, v);                                     // line3D.cc:875-881
    }
    h->resident_products = true;
    if (h->keep_view_matches) {
        for (size_t k = 0; k < P.n; ++k) {
            l3d_match* m = nullptr; int n = 0;
            int rc = l3d_chain_kept_list(h->ctx, (int)k, &m, &n);
            if (rc) return h->fail(rc, std::string("kept_list: ") + l3d_last_error(h->ctx));
            h->view_matches[h->order[k]].assign(m, m + n);
            l3d_free(m);
        }
    }
    if (hopt(h).check_pot) { int rc = check_resident_products(h, P); if (rc) return rc; }
    return L3D_OK;
}

int match_views_resident(L* h, ChainPlan& P, double t0)
{
    std::vector<uint32_t> ids; std::vector<int32_t> base;
    dense_map(h, ids, base);
    l3d_dense_map map;
    map.n_views = (int32_t)ids.size(); map.view_ids = ids.data(); map.seg_base = base.data();
    h->chain_summary.assign(P.n, l3d_chain_summary());
    h->resident_products = false;
    const double t1 = now_s();
    int rc = l3d_match_chain_resident(h->ctx, P.cv.data(), (int)P.n, &map, h->chain_summary.data(), &h->resident_n_pot);
    h->t_gpu_call += now_s() - t1;
    if (rc == L3D_ERR_UNSUPPORTED) return rc;
    if (rc) return h->fail(rc, std::string("match_chain_resident: ") + l3d_last_error(h->ctx));
    rc = adopt_resident_products(h, P);
    if (rc) return rc;
    double st[4];
    l3d_last_stats(h->ctx, st);
    h->stat_pairs += st[0];
    h->stat_raw += st[1];
    h->t_match = now_s() - t0;
    if (hopt(h).timing) fprintf(stderr, "[l3d match_views] resident chain + device products %.2f ms\n", (now_s() - t1) * 1e3);
    return L3D_OK;
}

int match_views(L* h)
{
    if (h->force_sync) return match_views_sync(h);
    const double t0 = now_s();
    match_begin(h);
    const double ta = now_s();
    ChainPlan* Pp = get_plan(h);
    if (!Pp) return match_views_sync(h);
    ChainPlan& P = *Pp;
    const double tb = now_s();
    if (!(h->host_bookkeeping || hopt(h).host_bookkeeping)) {
        const int rr = match_views_resident(h, P, t0);
        if (rr != L3D_ERR_UNSUPPORTED) return rr;           // (more kept matches than the device builder takes: host lists)
    }
    start_finalizer(h, P);
    const double t1 = now_s();
    int rc = l3d_match_chain(h->ctx, P.cv.data(), (int)P.n, chain_callback, &P.user);
    h->t_gpu_call += now_s() - t1 - h->t_commit;
    const double t2 = now_s();
    finish_chain_host(h, P, rc == L3D_OK);
    if (h->pot_check_failed) return h->fail(L3D_ERR_INVALID, "L3D_CHECK_POT: a potential-correspondence list is not in normal form");
    if (hopt(h).timing) fprintf(stderr, "[l3d match_views] begin %.2f  schedule %.2f  finaliser start %.2f  chain %.2f  finish %.2f ms\n",
                                      (ta - t0) * 1e3, (tb - ta) * 1e3, (t1 - tb) * 1e3, (t2 - t1) * 1e3, (now_s() - t2) * 1e3);
    if (rc) return h->fail(rc, std::string("match_chain: ") + l3d_last_error(h->ctx));
    double st[4];
    l3d_last_stats(h->ctx, st);
    h->stat_pairs += st[0];
    h->stat_raw += st[1];
    h->t_match = now_s() - t0;
    return L3D_OK;
}

// L3DView::unprojectSegment, view.cc:302-342 (the arithmetic is shared with the device: l3d_unproject.hpp)
void unproject_segment(const View& v, uint32_t id, float d1, float d2, Hyp& o)
{
    const float* s = &v.segs[(size_t)id * 4];
    l3d::unproject_segment_f64(v.RtKinv, v.C, s[0], s[1], s[2], s[3], d1, d2, o.P1, o.P2, o.dir);
    o.depth_p1 = d1; o.depth_p2 = d2;
}

// ---- small thread helpers of the finishing stages (greedy selection .. line fit run alone on the host) ----------
unsigned finish_threads() { return l3d::host_threads(); }

// fn(begin, end, thread) over [0, n) in contiguous slices
template <class F>
void parallel_slices(size_t n, unsigned nt, F fn)
{
    nt = (unsigned)std::max<size_t>(1, std::min<size_t>(nt, n));
    if (nt == 1) { fn((size_t)0, n, 0u); return; }
    l3d::on_threads(nt, [&](unsigned t) { fn(n * t / nt, n * (t + 1) / nt, t); });
}

// Line3D::greedySelection, line3D.cc:899-965: the stored list holds one (best) match per segment.  The views are independent:
// worker threads pick the best match of every segment of their views, the hypotheses are numbered view by view afterwards
// (a prefix over the views' counts) and unprojected in parallel.
void greedy_selection(L* h)
{
    const size_t nv = h->vlist.size();
    h->best_idx.resize(nv);
    std::vector<std::vector<int>> best(nv);                  // per view: index into the store of the segment's best match, or -1
    std::vector<size_t> count(nv + 1, 0);
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(finish_threads(), nv));
    auto for_views = [&](auto fn) {
        std::atomic<size_t> next{ 0 };
        auto worker = [&]() { for (;;) { const size_t vi = next.fetch_add(1, std::memory_order_relaxed); if (vi >= nv) break; fn(vi); } };
        l3d::on_threads(nt, [&](unsigned) { worker(); });
    };
    for_views([&](size_t vi) {
        View* v = h->vlist[vi];
        const uint32_t S = (uint32_t)v->S();
        std::vector<int>& b = best[vi];
        b.assign((size_t)S, -1);
        h->best_idx[(size_t)v->index].assign((size_t)S, -1);
        if (!v->store_exists) return;
        // group by segment (ascending), first of the highest confidence
        for (size_t i = 0; i < v->store.size(); ++i) {
            const uint32_t sg = v->store[i].segID1;
            if (sg >= S) continue;
            if (b[sg] < 0 || v->store[i].confidence > v->store[(size_t)b[sg]].confidence) b[sg] = (int)i;
        }
        size_t n = 0;
        for (uint32_t sg = 0; sg < S; ++sg) n += b[sg] >= 0;
        count[vi + 1] = n;
    });
    for (size_t vi = 0; vi < nv; ++vi) count[vi + 1] += count[vi];
    h->hyps.resize(count[nv]);
    h->hyp_begin = count;                                    // the hypotheses of view index vi are [hyp_begin[vi], hyp_begin[vi + 1])
    // (the flat copies the device affinity fill takes -- hypothesis, score, dense segment id -- are written in the same pass)
    std::vector<size_t> voff(nv + 1, 0);
    for (size_t vi = 0; vi < nv; ++vi) voff[vi + 1] = voff[vi] + (size_t)h->vlist[vi]->S();
    h->aff.hyp.resize(count[nv]); h->aff.score.resize(count[nv]); h->aff.hyp_dense.resize(count[nv]); h->aff.hyp_cam.resize(count[nv]);
    for_views([&](size_t vi) {
        View* v = h->vlist[vi];
        std::vector<int>& bi = h->best_idx[(size_t)v->index];
        const std::vector<int>& b = best[vi];
        size_t k = count[vi];
        for (uint32_t sg = 0; sg < (uint32_t)b.size(); ++sg) {
            if (b[sg] < 0) continue;
            const l3d_match& mp = v->store[(size_t)b[sg]];
            Hyp hy;
            hy.src = mk(v->id, sg);
            hy.score = fminf(mp.confidence, 1.0f);
            unproject_segment(*v, sg, mp.depths[0], mp.depths[1], hy);
            bi[sg] = (int)k;
            l3d_hypothesis& o = h->aff.hyp[k];
            o.P1[0] = hy.P1.x; o.P1[1] = hy.P1.y; o.P1[2] = hy.P1.z;
            o.P2[0] = hy.P2.x; o.P2[1] = hy.P2.y; o.P2[2] = hy.P2.z;
            o.dir[0] = hy.dir.x; o.dir[1] = hy.dir.y; o.dir[2] = hy.dir.z;
            o.depth_p1 = hy.depth_p1; o.depth_p2 = hy.depth_p2;
            o.k_lower = v->k_lower; o.k_upper = v->k_upper; o.median_depth = v->median_depth; o.pad = 0;
            h->aff.score[k] = hy.score;
            h->aff.hyp_dense[k] = (int32_t)(voff[vi] + sg);
            h->aff.hyp_cam[k] = v->id;
            h->hyps[k++] = hy;
        }
    });
}

int best_of(const L* h, Key k)
{
    auto it = h->views.find(kcam(k));
    if (it == h->views.end()) return -1;
    const std::vector<int>& bi = h->best_idx[(size_t)it->second.index];
    return kseg(k) < bi.size() ? bi[kseg(k)] : -1;
}

// Felzenszwalb-Huttenlocher segmentation, clustering.cc:6-47 + universe.h:59-115, on the host: the fallback for lists the device path
// refuses and the cross-check of l3d_perform_clustering_device (L3D_HOST_CLUSTERING=1)
// presorted: edges_in already is in the stable ascending weight order (l3d_clustering_edges)
void perform_clustering(const l3d_edge* edges_in, size_t n_edges, int numNodes, float c, std::vector<int>& labels, bool presorted = false)
{
    // stable ascending order of the weights (clustering.cc:14: std::stable_sort over CLEdge::operator<)
    std::unique_ptr<l3d_edge[]> gathered;
    const l3d_edge* sorted = edges_in;
    if (!presorted) {
        std::vector<uint32_t> order;
        {
            const l3d_edge* e = edges_in;
            l3d::parallel_stable_order(n_edges, (size_t)65536, (size_t)65536, [e](size_t i) { return l3d::float_order_key(e[i].w) >> 16; },
                                       [e](size_t i) { return l3d::float_order_key(e[i].w) & 0xffffu; }, finish_threads(), order);
        }
        // the edges in that order, gathered by the worker threads (the merge loop below then reads them sequentially)
        gathered.reset(new l3d_edge[n_edges + 1]);
        l3d_edge* g = gathered.get();
        parallel_slices(n_edges, finish_threads(), [&](size_t k0, size_t k1, unsigned) { for (size_t k = k0; k < k1; ++k) g[k] = edges_in[order[k]]; });
        sorted = g;
    }
    std::vector<int> rank((size_t)numNodes, 0), cid((size_t)numNodes), size((size_t)numNodes, 1);
    std::vector<float> thr((size_t)numNodes, c);
    for (int i = 0; i < numNodes; ++i) cid[i] = i;
    // universe.h:81-89 compresses only the queried node's link; halving every link on the way finds the same root (unions
    // go by rank, which no compression touches) with shorter chains afterwards
    auto find = [&](int node) { int y = node; while (y != cid[y]) { cid[y] = cid[cid[y]]; y = cid[y]; } return y; };
    for (size_t q = 0; q < n_edges; ++q) {
        const l3d_edge& ed = sorted[q];
        int a = find(ed.i), b = find(ed.j);
        if (a != b && ed.w <= thr[a] && ed.w <= thr[b]) {
            if (rank[a] > rank[b]) { cid[b] = a; size[a] += size[b]; }
            else { cid[a] = b; size[b] += size[a]; if (rank[a] == rank[b]) rank[b]++; }
            a = find(a);
            thr[a] = ed.w + c / (float)size[a];
        }
        // the affinity list holds every edge in both directions, and the stable order keeps the two together: whatever the
        // first one did (merged its components, found them merged, or failed a threshold), the reversed twin right behind it
        // meets the very same state and changes nothing
        if (q + 1 < n_edges && sorted[q + 1].i == ed.j && sorted[q + 1].j == ed.i && sorted[q + 1].w == ed.w) ++q;
    }
    labels.resize((size_t)numNodes);
    for (int k = 0; k < numNodes; ++k) labels[k] = find(k);
}

// The same segmentation from the edge list grouped by connected component (l3d_clustering_edges_grouped): the merge loop never
// relates nodes of different components, so every group is walked on its own -- same unions, same ranks, same roots as the one
// sequential walk over the whole sorted list -- by the worker threads (config 2: 3240 components, the largest 1326 edges).
void perform_clustering_grouped(const l3d_edge* sorted, const int32_t* group_start, int n_groups, int numNodes, float c, std::vector<int>& labels)
{
    std::unique_ptr<int[]> rank(new int[(size_t)numNodes + 1]), cid(new int[(size_t)numNodes + 1]), size(new int[(size_t)numNodes + 1]);
    std::unique_ptr<float[]> thr(new float[(size_t)numNodes + 1]);
    labels.resize((size_t)numNodes);
    const unsigned nt = finish_threads();
    parallel_slices((size_t)numNodes, nt, [&](size_t k0, size_t k1, unsigned) { for (size_t k = k0; k < k1; ++k) { rank[k] = 0; cid[k] = (int)k; size[k] = 1; thr[k] = c; } });
    int *cidp = cid.get(), *rankp = rank.get(), *sizep = size.get();
    float* thrp = thr.get();
    auto find = [cidp](int node) { int y = node; while (y != cidp[y]) { cidp[y] = cidp[cidp[y]]; y = cidp[y]; } return y; };
    std::atomic<int> next{ 0 };
    l3d::on_threads((unsigned)std::max(1, std::min<int>((int)nt, n_groups / 16 + 1)), [&](unsigned) {
        for (;;) {
            const int g0 = next.fetch_add(32, std::memory_order_relaxed);
            if (g0 >= n_groups) break;
            for (int g = g0; g < std::min(n_groups, g0 + 32); ++g)
                for (int q = group_start[g]; q < group_start[g + 1]; ++q) {
                    const l3d_edge& ed = sorted[q];
                    int a = find(ed.i), b = find(ed.j);
                    if (a != b && ed.w <= thrp[a] && ed.w <= thrp[b]) {
                        if (rankp[a] > rankp[b]) { cidp[b] = a; sizep[a] += sizep[b]; }
                        else { cidp[a] = b; sizep[b] += sizep[a]; if (rankp[a] == rankp[b]) rankp[b]++; }
                        a = find(a);
                        thrp[a] = ed.w + c / (float)sizep[a];
                    }
                    if (q + 1 < group_start[g + 1] && sorted[q + 1].i == ed.j && sorted[q + 1].j == ed.i && sorted[q + 1].w == ed.w) ++q;   // (reversed twin)
                }
        }
    });
    // (read-only walks: several threads may look up nodes of one component)
    parallel_slices((size_t)numNodes, nt, [&](size_t k0, size_t k1, unsigned) { for (size_t k = k0; k < k1; ++k) { int y = (int)k; while (y != cidp[y]) y = cidp[y]; labels[k] = y; } });
}

// Line3D::performDiffusion, line3D.cc:1255-1303: A (read) -> diffused, symmetrised list sorted by (i,j) in `out`
int perform_diffusion(L* h, const EdgeVec& A, int n, EdgeVec& out)
{
    EdgeVec W;
    W.resize(A.size());
    int rc = l3d_replicator_dynamics_diffusion(h->ctx, A.data(), (int)A.size(), n, L3D_RDD_MAX_ITER, W.data());
    if (rc) return h->fail(rc, std::string("rdd: ") + l3d_last_error(h->ctx));
    const double t_sym = now_s();
    // symmetrise by the minimum and rebuild A sorted by (i,j) (:1275-1301).  The diffused entries come back sorted by
    // (row, column); when they are unique and the pattern is symmetric -- always the case for the affinity list built by
    // clusterSegments2D -- the reference's map arithmetic reduces to A(i,j) = A(j,i) = min(W(i,j), W(j,i)) in that same order.
    const unsigned nt = finish_threads();
    std::atomic<int> unsorted{ 0 };
    parallel_slices(W.size(), nt, [&](size_t k0, size_t k1, unsigned) {
        for (size_t k = std::max<size_t>(k0, 1); k < k1; ++k)
            if (!(W[k - 1].i < W[k].i || (W[k - 1].i == W[k].i && W[k - 1].j < W[k].j))) { unsorted.store(1, std::memory_order_relaxed); break; }
    });
    if (!unsorted.load()) {
        std::vector<int> row((size_t)n + 1, 0);
        for (const l3d_edge& e : W) ++row[(size_t)e.i + 1];
        for (int r = 0; r < n; ++r) row[(size_t)r + 1] += row[(size_t)r];
        out.resize(W.size());
        std::atomic<int> missing{ 0 };
        parallel_slices(W.size(), nt, [&](size_t k0, size_t k1, unsigned) {
            for (size_t k = k0; k < k1; ++k) {
                const l3d_edge& e = W[k];
                const l3d_edge* lo = W.data() + row[(size_t)e.j];
                const l3d_edge* hi = W.data() + row[(size_t)e.j + 1];
                const l3d_edge* t = std::lower_bound(lo, hi, e.i, [](const l3d_edge& x, int col) { return x.j < col; });
                if (t == hi || t->j != e.i) { missing.fetch_add(1, std::memory_order_relaxed); break; }
                out[k] = { e.i, e.j, e.i <= e.j ? fminf(t->w, e.w) : fminf(e.w, t->w) };     // the visit of the later entry decides
            }
        });
        if (missing.load() == 0) {
            if (hopt(h).timing) fprintf(stderr, "[l3d rdd] %-24s %8.2f ms\n", "symmetrise", (now_s() - t_sym) * 1e3);
            return L3D_OK;
        }
    }
    std::map<std::pair<int, int>, float> entries;
    for (const l3d_edge& e : W) {
        const float w12 = e.w;
        float w21 = w12;
        auto it = entries.find({ e.j, e.i });
        if (it != entries.end()) w21 = it->second;
        const float w = fminf(w12, w21);
        entries[{ e.i, e.j }] = w;
        entries[{ e.j, e.i }] = w;
    }
    out.clear();
    for (auto& kv : entries) out.push_back({ kv.first.first, kv.first.second, kv.second });
    return L3D_OK;
}

// getLineEquation3D + projectToLine, line3D.cc:1392-1597 (the arithmetic lives in l3d_linefit.hpp, shared with the device kernel)
void align_cluster(const std::vector<std::pair<Key, std::pair<V3, V3>>>& t3, std::vector<std::pair<V3, V3>>& aligned)
{
    aligned.clear();
    if (t3.empty()) return;
    const int n2 = (int)t3.size() * 2;
    auto get = [&](int i) { return (i & 1) ? t3[(size_t)(i >> 1)].second.second : t3[(size_t)(i >> 1)].second.first; };
    V3 Pc, dir, min_point;
    l3d::fit::line_of_points(get, n2, Pc, dir, min_point);
    static thread_local std::vector<float> dist;             // (per-thread buffers: a fit is a few microseconds, allocations were a third of it)
    static thread_local std::vector<int> order;
    static thread_local std::vector<unsigned char> line_open;
    static thread_local std::vector<unsigned> cam_ids, cam_cnt;
    dist.resize((size_t)n2); order.resize((size_t)n2); line_open.resize(t3.size()); cam_ids.resize(t3.size()); cam_cnt.resize(t3.size());
    for (int i = 0; i < n2; ++i) { dist[(size_t)i] = l3d::fit::point_dist(get(i), min_point); order[(size_t)i] = i; }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return dist[(size_t)a] < dist[(size_t)b]; });
    l3d::fit::sweep_line(order.data(), n2, get, [&](int member) { return kcam(t3[(size_t)member].first); }, line_open.data(), cam_ids.data(), cam_cnt.data(),
                         [&](V3 s0, V3 e0) { aligned.emplace_back(s0, e0); });
}

// segment2collinearities_ of all views as one CSR over dense ids (static per scene: kept between calls)
void pack_collinearities(L* h, const std::vector<size_t>& voff)
{
    L::AffTables& T = h->aff;
    const size_t nv = h->vlist.size(), ndense = voff.back();
    if (T.coll_valid && T.coll_start.size() == ndense + 1) return;
    T.coll_start.resize(ndense + 1);
    int64_t* coll_start = T.coll_start.data();
    coll_start[0] = 0;
    for (size_t vi = 0; vi < nv; ++vi) {
        const View& sv = *h->vlist[vi];
        for (size_t sg = 0; sg < (size_t)sv.S(); ++sg) coll_start[voff[vi] + sg + 1] = coll_start[voff[vi] + sg] + (sv.coll_start[sg + 1] - sv.coll_start[sg]);
    }
    const size_t n_coll = (size_t)coll_start[ndense];
    T.coll_other.resize(n_coll + 1); T.coll_w.resize(n_coll + 1);
    int32_t* coll_other = T.coll_other.data();
    float* coll_w = T.coll_w.data();
    std::atomic<size_t> next{ 0 };
    l3d::on_threads(std::min<unsigned>(finish_threads(), (unsigned)std::max<size_t>(1, nv)), [&](unsigned) {
        for (;;) {
            const size_t vi = next.fetch_add(1, std::memory_order_relaxed);
            if (vi >= nv) break;
            const View& sv = *h->vlist[vi];
            const size_t cb = (size_t)coll_start[voff[vi]], cn = sv.coll_other.size();
            for (size_t q = 0; q < cn; ++q) { coll_other[cb + q] = (int32_t)(voff[vi] + (size_t)sv.coll_other[q]); coll_w[cb + q] = sv.coll_w[q]; }
        }
    });
    T.coll_valid = false;           // (the caller uploads, then marks it valid)
}

// Line3D::greedySelection (line3D.cc:899-965) on the device-resident products of matchViews (l3d_products_hypotheses): the host
// keeps only what the result needs -- which 2-D segment every hypothesis belongs to
int greedy_selection_resident(L* h)
{
    const size_t nv = h->vlist.size();
    std::vector<l3d_view_geometry> geo(nv);
    for (size_t i = 0; i < nv; ++i) {
        const View& v = *h->vlist[i];
        l3d_view_geometry& g = geo[i];
        memcpy(g.RtKinv, v.RtKinv.m, 72);
        g.C[0] = v.C.x; g.C[1] = v.C.y; g.C[2] = v.C.z;
        g.k_lower = v.k_lower; g.k_upper = v.k_upper; g.median_depth = v.median_depth;
        g.n_segments = v.S(); g.segments = v.segs.data();
    }
    std::vector<int32_t> vhb(nv + 1, 0);
    int32_t* hyp_dense = nullptr; int nh = 0;
    int rc = l3d_products_hypotheses(h->ctx, geo.data(), (int)nv, vhb.data(), &hyp_dense, &nh);
    if (rc) return h->fail(rc, std::string("hypotheses: ") + l3d_last_error(h->ctx));
    h->hyp_begin.assign(nv + 1, 0);
    for (size_t i = 0; i <= nv; ++i) h->hyp_begin[i] = (size_t)vhb[i];
    h->hyps.resize((size_t)nh);
    h->aff.hyp_cam.resize((size_t)nh);
    h->best_idx.clear();
    std::vector<size_t> voff(nv + 1, 0);
    for (size_t i = 0; i < nv; ++i) voff[i + 1] = voff[i] + (size_t)h->vlist[i]->S();
    parallel_slices(nv, finish_threads(), [&](size_t v0, size_t v1, unsigned) {
        for (size_t vi = v0; vi < v1; ++vi)
            for (size_t k = h->hyp_begin[vi]; k < h->hyp_begin[vi + 1]; ++k) {
                h->hyps[k].src = mk(h->vlist[vi]->id, (uint32_t)((size_t)hyp_dense[k] - voff[vi]));
                h->aff.hyp_cam[k] = h->vlist[vi]->id;
            }
    });
    l3d_free(hyp_dense);
    return L3D_OK;
}

// the affinity list on the host (the resident fill leaves it on the device: fetched on first use)
int ensure_edges(L* h)
{
    if (h->A_on_host) return L3D_OK;
    h->A.resize(h->n_edges);
    int rc = l3d_resident_edges_get(h->ctx, h->A.data(), (int)h->n_edges);
    if (rc) { h->A.clear(); return h->fail(rc, std::string("affinity list: ") + l3d_last_error(h->ctx)); }
    h->A_on_host = true;
    return L3D_OK;
}

// the affinity fill on the resident tables (l3d_affinity_fill_resident): only the collinearity CSR comes from the host, once per scene
int fill_affinity_resident(L* h)
{
    const size_t nv = h->vlist.size();
    std::vector<size_t> voff(nv + 1, 0);
    for (size_t i = 0; i < nv; ++i) voff[i + 1] = voff[i] + (size_t)h->vlist[i]->S();
    L::AffTables& T = h->aff;
    const bool timing = hopt(h).timing != 0;
    double tl = now_s();
    auto lap = [&](const char* what) { if (timing) { const double t = now_s(); fprintf(stderr, "[l3d finish]   fill: %-22s %8.2f ms\n", what, (t - tl) * 1e3); tl = t; } };
    const bool changed = !T.coll_valid || T.coll_start.size() != voff.back() + 1;
    if (changed) pack_collinearities(h, voff);
    lap("collinearity tables");
    l3d_edge* edges = nullptr; int32_t* node_hyp = nullptr; int n_edges = 0, n_nodes = 0, n_cand = 0;
    // (the list itself stays on the device, where the clustering walks it; l3d_line3d_affinity fetches it when somebody asks)
    int rc = l3d_affinity_fill_resident(h->ctx, T.coll_start.data(), T.coll_other.data(), T.coll_w.data(), changed ? 1 : 0, h->sigma_a, nullptr, &n_edges, &node_hyp, &n_nodes, &n_cand);
    if (rc) return h->fail(rc, std::string("affinity fill: ") + l3d_last_error(h->ctx));
    lap("device");
    T.coll_valid = true;
    h->A.clear(); h->n_edges = (size_t)n_edges; h->A_on_host = n_edges == 0;
    h->local2global.resize((size_t)n_nodes);
    h->node_hyp.resize((size_t)n_nodes);
    parallel_slices((size_t)n_nodes, finish_threads(), [&](size_t k0, size_t k1, unsigned) {     // (a gather over the hypothesis table: 0.5 M nodes at 512 views)
        for (size_t k = k0; k < k1; ++k) { h->node_hyp[k] = node_hyp[k]; h->local2global[k] = h->hyps[(size_t)node_hyp[k]].src; }
    });
    l3d_free(edges); l3d_free(node_hyp);
    lap("node table");
    if (timing) fprintf(stderr, "[l3d finish] %zu hypotheses, %d candidate pairs, %zu edges (resident tables)\n", h->hyps.size(), n_cand, h->n_edges);
    return L3D_OK;
}

// Line3D::clusterSegments2D, line3D.cc:968-1252: the affinity fill and the edge list of the clustering on the device
// (l3d_affinity_fill / l3d_affinity_fill_resident, l3d_perform_clustering_device, l3d_fit_labelled_clusters); host union-find, symmetrisation and
// edge order remain for edge lists the device path refuses.  (The literal `used` enumeration of round 1 lives on as a test helper:
// tests/cpp/literal_used_rule.c.)
int cluster_segments_2D(L* h, bool perform_diff)
{
    const double t0 = now_s();
    const bool timing = hopt(h).timing != 0;
    double tm_last = t0;
    auto lap = [&](const char* what) { if (timing) { const double t = now_s(); fprintf(stderr, "[l3d finish] %-28s %8.2f ms\n", what, (t - tm_last) * 1e3); tm_last = t; } };
    h->A.clear(); h->n_edges = 0; h->A_on_host = true; h->local2global.clear(); h->result.clear();
    const size_t nh = h->hyps.size();
    if (nh == 0) return L3D_OK;

    // dense index of every 2-D segment of every view (for the `used` bookkeeping)
    const size_t nv = h->vlist.size();
    std::vector<size_t> voff(nv + 1, 0);
    for (size_t i = 0; i < nv; ++i) voff[i + 1] = voff[i] + (size_t)h->vlist[i]->S();
    // camera id -> view index (ascending ids; ids are small in practice, else binary search)
    std::vector<uint32_t> cam_ids(nv);
    for (size_t i = 0; i < nv; ++i) cam_ids[i] = h->vlist[i]->id;
    std::vector<int> cam_direct;
    if (nv && cam_ids.back() < (1u << 22)) { cam_direct.assign((size_t)cam_ids.back() + 1, -1); for (size_t i = 0; i < nv; ++i) cam_direct[cam_ids[i]] = (int)i; }
    auto view_of = [&](uint32_t cam) -> int {
        if (!cam_direct.empty()) return cam < cam_direct.size() ? cam_direct[cam] : -1;
        auto it = std::lower_bound(cam_ids.begin(), cam_ids.end(), cam);
        return it != cam_ids.end() && *it == cam ? (int)(it - cam_ids.begin()) : -1;
    };
    // hypotheses are in (view, segment) order: the range of each view (greedy_selection)
    const std::vector<size_t>& hyp_begin = h->hyp_begin;
    if (hyp_begin.size() != nv + 1 || hyp_begin[nv] != nh) return h->fail(L3D_ERR_INVALID, "hypothesis ranges do not match the views");

    bool resident_list = false;                     // the affinity list is still on the device (l3d_affinity_fill ran last)
    if (h->resident_products) {
        const int rc = fill_affinity_resident(h);
        if (rc) return rc;
        resident_list = true;
        lap("affinity fill (resident tables)");
    } else {
        resident_list = true;
        // ---- the whole fill on the device (l3d_affinity.hip): flat tables in, edge list and node numbering out
        const unsigned nt = finish_threads();
        if (voff.back() > 0x7fffffffu || nh > 0x3fffffffu) return h->fail(L3D_ERR_INVALID, "affinity fill: too many segments");
        const size_t ndense = voff.back();
        std::vector<int32_t> seg_base(nv + 1), vhb(nv + 1);
        for (size_t i = 0; i <= nv; ++i) { seg_base[i] = (int32_t)voff[i]; vhb[i] = (int32_t)hyp_begin[i]; }
        vhb[nv] = (int32_t)nh;
        L::AffTables& T = h->aff;
        if (T.hyp.size() != nh || T.score.size() != nh || T.hyp_dense.size() != nh) return h->fail(L3D_ERR_INVALID, "hypothesis tables do not match the hypotheses");
        T.best.resize(ndense + 1);                      // (hypothesis, score, dense id: written by greedy_selection)
        l3d_hypothesis* hy = T.hyp.data();
        float* score = T.score.data();
        int32_t *hyp_dense = T.hyp_dense.data(), *best = T.best.data();
        lap("  pack: hypotheses");
        // potential correspondences and collinearities as CSR over dense ids (a view's rows are written by one thread)
        T.pot_start.resize(ndense + 1);
        int64_t* pot_start = T.pot_start.data();
        const bool pack_coll = !T.coll_valid || T.coll_start.size() != ndense + 1;
        if (pack_coll) T.coll_start.resize(ndense + 1);
        int64_t* coll_start = T.coll_start.data();
        std::vector<std::vector<int32_t>>& vt = h->aff_vt;
        vt.resize(nv);
        pot_start[0] = 0;
        if (pack_coll) coll_start[0] = 0;
        {
            std::atomic<size_t> next{ 0 };
            auto worker = [&]() {
                for (;;) {
                    const size_t vi = next.fetch_add(1, std::memory_order_relaxed);
                    if (vi >= nv) break;
                    const View& sv = *h->vlist[vi];
                    const size_t S = (size_t)sv.S();
                    for (size_t sg = 0; sg < S; ++sg) { pot_start[voff[vi] + sg + 1] = 0; best[voff[vi] + sg] = h->best_idx[vi][sg]; }
                    if (pack_coll) for (size_t sg = 0; sg < S; ++sg) coll_start[voff[vi] + sg + 1] = sv.coll_start[sg + 1] - sv.coll_start[sg];
                    std::vector<int32_t>& out = vt[vi];
                    out.clear();
                    out.reserve(h->pot[vi].size());
                    for (const auto& e : h->pot[vi]) {
                        // keys whose camera is not a view (early-return quirk) or whose segment does not exist never have a
                        // hypothesis or collinear segments: they take no part in the fill
                        const int tvi = view_of(kcam(e.second));
                        if (tvi < 0 || e.first >= S) continue;
                        const uint32_t tseg = kseg(e.second);
                        if (tseg >= (uint32_t)(voff[(size_t)tvi + 1] - voff[(size_t)tvi])) continue;
                        out.push_back((int32_t)(voff[(size_t)tvi] + tseg));
                        ++pot_start[voff[vi] + e.first + 1];
                    }
                }
            };
            l3d::on_threads(std::min<unsigned>(nt, (unsigned)nv), [&](unsigned) { worker(); });
        }
        lap("  pack: count + targets per view");
        for (size_t dd = 0; dd < ndense; ++dd) pot_start[dd + 1] += pot_start[dd];
        if (pack_coll) for (size_t dd = 0; dd < ndense; ++dd) coll_start[dd + 1] += coll_start[dd];
        const size_t n_pot = (size_t)pot_start[ndense], n_coll = (size_t)coll_start[ndense];
        T.pot_tgt.resize(n_pot + 1);
        if (pack_coll) { T.coll_other.resize(n_coll + 1); T.coll_w.resize(n_coll + 1); }
        int32_t *pot_tgt = T.pot_tgt.data(), *coll_other = T.coll_other.data();
        float* coll_w = T.coll_w.data();
        {
            std::atomic<size_t> next{ 0 };
            auto worker = [&]() {
                for (;;) {
                    const size_t vi = next.fetch_add(1, std::memory_order_relaxed);
                    if (vi >= nv) break;
                    const View& sv = *h->vlist[vi];
                    if (!vt[vi].empty()) memcpy(pot_tgt + pot_start[voff[vi]], vt[vi].data(), vt[vi].size() * 4);
                    if (!pack_coll) continue;
                    const size_t cb = (size_t)coll_start[voff[vi]], cn = sv.coll_other.size();
                    for (size_t q = 0; q < cn; ++q) { coll_other[cb + q] = (int32_t)(voff[vi] + (size_t)sv.coll_other[q]); coll_w[cb + q] = sv.coll_w[q]; }
                }
            };
            l3d::on_threads(std::min<unsigned>(nt, (unsigned)nv), [&](unsigned) { worker(); });
        }
        T.coll_valid = true;
        lap("pack tables");
        l3d_affinity_input in;
        in.n_views = (int32_t)nv; in.seg_base = seg_base.data(); in.view_hyp_begin = vhb.data();
        in.n_hyp = (int32_t)nh; in.hyp = hy; in.score = score; in.hyp_dense = hyp_dense; in.best = best;
        in.pot_start = pot_start; in.pot_tgt = pot_tgt;
        in.coll_start = coll_start; in.coll_other = coll_other; in.coll_w = coll_w;
        in.sigma_a = h->sigma_a;
        l3d_edge* edges = nullptr; int32_t* node_hyp = nullptr; int n_edges = 0, n_nodes = 0, n_cand = 0;
        int rc = l3d_affinity_fill(h->ctx, &in, &edges, &n_edges, &node_hyp, &n_nodes, &n_cand);
        if (rc) return h->fail(rc, std::string("affinity fill: ") + l3d_last_error(h->ctx));
        lap("affinity fill (device)");
        h->A.resize((size_t)n_edges);
        parallel_slices((size_t)n_edges, nt, [&](size_t k0, size_t k1, unsigned) { if (k1 > k0) memcpy(&h->A[k0], edges + k0, (k1 - k0) * sizeof(l3d_edge)); });
        h->local2global.resize((size_t)n_nodes);
        for (int k = 0; k < n_nodes; ++k) h->local2global[(size_t)k] = h->hyps[(size_t)node_hyp[k]].src;
        h->node_hyp.assign(node_hyp, node_hyp + n_nodes);
        l3d_free(edges); l3d_free(node_hyp);
        h->n_edges = h->A.size(); h->A_on_host = true;
        if (timing) fprintf(stderr, "[l3d finish] %zu hypotheses, %d candidate pairs, %zu edges, %u threads\n", nh, n_cand, h->A.size(), nt);
        lap("edge list to host");
    }
    h->t_affinity = now_s() - t0;
    if (h->n_edges == 0) return L3D_OK;                                         // :1232-1233

    const double t1 = now_s();
    const int n_nodes = (int)h->local2global.size();
    std::vector<int> labels;
    bool labels_on_device = false;
    {
        // the list clustering walks -- diffused and symmetrised when asked for, in stable ascending weight order -- comes from
        // the device, where the affinity list still is (l3d_clustering_edges); a list the device path does not take
        // (L3D_ERR_UNSUPPORTED) goes through the reference's map arithmetic on the host
        // ... and so does the merge loop itself, one wave per connected component (l3d_perform_clustering_device): only the labels
        // come back.  L3D_HOST_CLUSTERING=1 keeps the merge loop on the worker threads (the seam tests compare the two).
        const bool host_loop = hopt(h).host_clustering != 0;
        const int nnz = (int)h->n_edges, diff = perform_diff ? 1 : 0;
        int rc = L3D_ERR_UNSUPPORTED;
        if (resident_list && !host_loop) {
            // (the labels stay on the device as well: the grouping and the fits follow there, l3d_fit_labelled_clusters)
            rc = l3d_perform_clustering_device(h->ctx, nullptr, nnz, n_nodes, diff, L3D_RDD_MAX_ITER, 1.0f, nullptr, nullptr);   // :1245
            if (rc == L3D_OK) { labels_on_device = true; lap(perform_diff ? "diffusion + clustering (device)" : "clustering (device)"); }
        } else if (resident_list) {
            std::unique_ptr<l3d_edge[]> sorted(new l3d_edge[h->n_edges + 1]);
            int32_t* group_start = nullptr;
            int n_groups = 0;
            rc = l3d_clustering_edges_grouped(h->ctx, nullptr, nnz, n_nodes, diff, L3D_RDD_MAX_ITER, sorted.get(), &group_start, &n_groups);
            if (rc == L3D_OK) {
                lap(perform_diff ? "diffusion + grouped edge order (device)" : "grouped edge order (device)");
                perform_clustering_grouped(sorted.get(), group_start, n_groups, n_nodes, 1.0f, labels);   // :1245
                l3d_free(group_start);
            }
        }
        if (rc == L3D_ERR_UNSUPPORTED) {
            if (int e = ensure_edges(h)) return e;
            EdgeVec diffused;
            if (perform_diff) { rc = perform_diffusion(h, h->A, n_nodes, diffused); if (rc) return rc; lap("diffusion"); }
            const EdgeVec& edges = perform_diff ? diffused : h->A;
            perform_clustering(edges.data(), edges.size(), n_nodes, 1.0f, labels);
        } else if (rc != L3D_OK) return h->fail(rc, std::string("clustering: ") + l3d_last_error(h->ctx));
    }
    lap("clustering");

    if (labels_on_device) {
        // processClusteredSegments, line3D.cc:1306-1368, from the labels on the device: grouping (ascending label, members in key order,
        // >= 4 cameras) and the fits in one call; the host turns the answer into the result list
        int32_t *gstart = nullptr, *memb = nullptr, *cnt = nullptr; double* segs = nullptr; int n_groups = 0, n_segs = 0;
        const double tneg[3] = { h->transf_tneg.x, h->transf_tneg.y, h->transf_tneg.z };
        const int rc = l3d_fit_labelled_clusters(h->ctx, nullptr, nullptr, n_nodes, nullptr, h->aff.hyp_cam.data(), (int)h->hyps.size(), h->transf_Rinv.m, h->transf_scale_inv,
                                                 tneg, &gstart, &memb, &n_groups, &cnt, &segs, &n_segs);
        if (rc) return h->fail(rc, std::string("line fit: ") + l3d_last_error(h->ctx));
        lap("  fit: grouping + fits (device)");
        std::vector<size_t> soff((size_t)n_groups + 1, 0);
        for (int v = 0; v < n_groups; ++v) soff[(size_t)v + 1] = soff[(size_t)v] + (size_t)cnt[v];
        std::vector<FinalLine> fitted((size_t)n_groups);
        parallel_slices((size_t)n_groups, finish_threads(), [&](size_t v0, size_t v1, unsigned) {
            for (size_t v = v0; v < v1; ++v) {
                if (cnt[v] == 0) continue;
                FinalLine& fl = fitted[v];
                for (size_t k = soff[v]; k < soff[v + 1]; ++k) {
                    const double* q = segs + 6 * k;
                    fl.segs3D.emplace_back(V3{ q[0], q[1], q[2] }, V3{ q[3], q[4], q[5] });
                }
                for (int32_t i = gstart[v]; i < gstart[v + 1]; ++i) fl.segs2D.push_back(h->hyps[(size_t)memb[(size_t)i]].src);
            }
        });
        size_t n_lines = 0;
        for (int v = 0; v < n_groups; ++v) n_lines += cnt[v] != 0;
        l3d_free(gstart); l3d_free(memb); l3d_free(cnt); l3d_free(segs);
        h->result.reserve(n_lines);
        for (FinalLine& fl : fitted) if (!fl.segs3D.empty()) h->result.push_back(std::move(fl));
        lap("line fit");
        h->t_cluster = now_s() - t1;
        return L3D_OK;
    }

    // processClusteredSegments, line3D.cc:1306-1368: clusters in ascending label order (the reference's std::map), their
    // segments in key order; clusters seen from >= 4 cameras are fitted, independently of each other, by the worker threads
    std::vector<int> lstart((size_t)n_nodes + 1, 0), lnodes((size_t)n_nodes);
    for (int lid = 0; lid < n_nodes; ++lid) ++lstart[(size_t)labels[(size_t)lid] + 1];
    for (int l = 0; l < n_nodes; ++l) lstart[(size_t)l + 1] += lstart[(size_t)l];
    {
        std::vector<int> cur(lstart.begin(), lstart.end() - 1);
        for (int lid = 0; lid < n_nodes; ++lid) lnodes[(size_t)cur[(size_t)labels[(size_t)lid]]++] = lid;
    }
    std::vector<int> groups;                                                    // labels with >= 4 members (>= 4 cameras needs that)
    for (int l = 0; l < n_nodes; ++l) if (lstart[(size_t)l + 1] - lstart[(size_t)l] >= 4) groups.push_back(l);
    std::vector<FinalLine> fitted(groups.size());
    lap("  fit: clusters by label");
    if (resident_list && (int)h->node_hyp.size() == n_nodes) {
        // ---- the fits on the device (l3d_fit_clusters): members as hypothesis indices in key order (= ascending index)
        std::vector<int32_t> memb_tmp((size_t)n_nodes);
        std::vector<char> valid(groups.size(), 0);
        parallel_slices(groups.size(), finish_threads(), [&](size_t g0, size_t g1, unsigned) {
            for (size_t g = g0; g < g1; ++g) {
                const int l = groups[g];
                int32_t* mb = memb_tmp.data() + lstart[(size_t)l];
                const int n = lstart[(size_t)l + 1] - lstart[(size_t)l];
                for (int q = 0; q < n; ++q) mb[q] = h->node_hyp[(size_t)lnodes[(size_t)(lstart[(size_t)l] + q)]];
                std::sort(mb, mb + n);
                int ncam = 1;
                for (int q = 1; q < n; ++q) ncam += h->aff.hyp_cam[(size_t)mb[q]] != h->aff.hyp_cam[(size_t)mb[q - 1]];
                valid[g] = ncam >= 4;
            }
        });
        std::vector<int32_t> gstart(1, 0), memb;
        std::vector<size_t> gof;                                                // fitted[] slot of every cluster handed to the device
        memb.reserve((size_t)n_nodes);
        for (size_t g = 0; g < groups.size(); ++g) {
            if (!valid[g]) continue;
            const int l = groups[g];
            memb.insert(memb.end(), memb_tmp.begin() + lstart[(size_t)l], memb_tmp.begin() + lstart[(size_t)l + 1]);
            gstart.push_back((int32_t)memb.size());
            gof.push_back(g);
        }
        lap("  fit: member lists");
        int32_t* cnt = nullptr; double* segs = nullptr; int n_segs = 0;
        const double tneg[3] = { h->transf_tneg.x, h->transf_tneg.y, h->transf_tneg.z };
        // (hyp = null: the table l3d_affinity_fill uploaded in this finish is still on the device)
        const int rc = l3d_fit_clusters(h->ctx, gstart.data(), (int)gof.size(), memb.data(), nullptr, h->aff.hyp_cam.data(), (int)h->hyps.size(),
                                        h->transf_Rinv.m, h->transf_scale_inv, tneg, &cnt, &segs, &n_segs);
        if (rc) return h->fail(rc, std::string("line fit: ") + l3d_last_error(h->ctx));
        lap("  fit: device");
        std::vector<size_t> soff(gof.size() + 1, 0);
        for (size_t v = 0; v < gof.size(); ++v) soff[v + 1] = soff[v] + (size_t)cnt[v];
        parallel_slices(gof.size(), finish_threads(), [&](size_t v0, size_t v1, unsigned) {
            for (size_t v = v0; v < v1; ++v) {
                if (cnt[v] == 0) continue;
                FinalLine& fl = fitted[gof[v]];
                for (size_t k = soff[v]; k < soff[v + 1]; ++k) {
                    const double* q = segs + 6 * k;
                    fl.segs3D.emplace_back(V3{ q[0], q[1], q[2] }, V3{ q[3], q[4], q[5] });
                }
                for (int32_t i = gstart[v]; i < gstart[v + 1]; ++i) fl.segs2D.push_back(h->hyps[(size_t)memb[(size_t)i]].src);
            }
        });
        l3d_free(cnt); l3d_free(segs);
    } else {
        std::atomic<size_t> next{ 0 };
        auto worker = [&]() {
            std::vector<Key> keys;
            std::vector<std::pair<Key, std::pair<V3, V3>>> t3;
            for (;;) {
                const size_t g0 = next.fetch_add(16, std::memory_order_relaxed);
                if (g0 >= groups.size()) break;
                for (size_t g = g0; g < std::min(groups.size(), g0 + 16); ++g) {
                    const int l = groups[g];
                    keys.clear();
                    for (int q = lstart[(size_t)l]; q < lstart[(size_t)l + 1]; ++q) keys.push_back(h->local2global[(size_t)lnodes[(size_t)q]]);
                    std::sort(keys.begin(), keys.end());
                    int ncam = 1;
                    for (size_t q = 1; q < keys.size(); ++q) ncam += kcam(keys[q]) != kcam(keys[q - 1]);
                    if (ncam < 4) continue;
                    t3.clear();
                    for (Key k : keys) {
                        const int bb = best_of(h, k);
                        if (bb < 0) continue;
                        t3.push_back({ k, { inverse_transform(h, h->hyps[(size_t)bb].P1), inverse_transform(h, h->hyps[(size_t)bb].P2) } });
                    }
                    FinalLine& fl = fitted[g];
                    align_cluster(t3, fl.segs3D);
                    if (fl.segs3D.empty()) continue;
                    for (auto& e : t3) fl.segs2D.push_back(e.first);
                }
            }
        };
        const unsigned ntf = (unsigned)std::max<size_t>(1, std::min<size_t>(finish_threads(), groups.size() / 64 + 1));
        l3d::on_threads(ntf, [&](unsigned) { worker(); });
    }
    for (FinalLine& fl : fitted) if (!fl.segs3D.empty()) h->result.push_back(std::move(fl));
    lap("line fit");
    h->t_cluster = now_s() - t1;
    return L3D_OK;
}

}  // namespace

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
    delete static_cast<ChainFinalizer*>(h->finalizer);      // joins the worker threads
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
                                        const char* data_directory, int max_img_width, int load_and_store, std::string& cache_path)
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
    std::string cache_path;
    const int from_cache = add_from_cache_or_plan_write(h, id, width, height, K, R, t, data_directory, max_img_width, load_and_store, cache_path);
    if (from_cache < 0) return -from_cache;
    if (!from_cache) {
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
    std::string cache_path;
    const int from_cache = add_from_cache_or_plan_write(h, id, width, height, K, R, t, data_directory, max_img_width, load_and_store, cache_path);
    if (from_cache < 0) return -from_cache;
    if (!from_cache) {
        const int rc = add_common(h, id, width, height, segs, n, K, R, t, n_sims);
        if (rc) return rc;
        h->views[id].cache_to_write = cache_path;
    }
    for (int i = 0; i < n_sims; ++i)                       // setViewSimilarity, :1938-1946
        if (sims[i] > 0.01f) h->view_similarities[id][sim_ids[i]] = sims[i];
    return L3D_OK;
}

int l3d_line3d_num_cameras(const l3d_line3d* h) { return h ? (int)h->views.size() : 0; }

int l3d_line3d_prepare(l3d_line3d* h) { return h ? prepare(h) : L3D_ERR_INVALID; }
int l3d_line3d_match_views(l3d_line3d* h)
{
    if (!h || !h->prepared) return h ? h->fail(L3D_ERR_INVALID, "prepare first") : L3D_ERR_INVALID;
    return match_views(h);
}
int l3d_line3d_finish(l3d_line3d* h, int perform_diffusion)
{
    if (!h || !h->prepared) return h ? h->fail(L3D_ERR_INVALID, "prepare first") : L3D_ERR_INVALID;
    const double t0 = now_s();
    if (h->resident_products) { const int rg = greedy_selection_resident(h); if (rg) return rg; }
    else greedy_selection(h);                              // optimizeLocalMatches, :888-896
    if (hopt(h).timing) fprintf(stderr, "[l3d finish] %-28s %8.2f ms\n", "greedy selection", (now_s() - t0) * 1e3);
    const int rc = cluster_segments_2D(h, perform_diffusion != 0);
    if (hopt(h).timing) fprintf(stderr, "[l3d finish] %-28s %8.2f ms\n", "total", (now_s() - t0) * 1e3);
    return rc;
}
// Line3D::compute3Dmodel, line3D.cc:345-374
int l3d_line3d_compute3Dmodel(l3d_line3d* h, int perform_diffusion)
{
    if (!h) return L3D_ERR_INVALID;
    int rc = prepare(h);
    if (!rc) rc = match_views(h);
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
        if (arena_cap_next) l3d_set_chain_capacities(h->ctx, 0, arena_cap_next);          // (the compact arena of the slot ring: read by the run)
        rc = l3d_shard_chain_run(P->shard, exchange, exchange_user, host_commit ? chain_callback : nullptr, host_commit ? &P->user : nullptr);
        if (arena_cap_next) l3d_set_chain_capacities(h->ctx, 0, 0);
        std::string msg = rc ? std::string("shard_chain_run: ") + l3d_last_error(h->ctx) : std::string();
        if (rc == L3D_OK && commit == 2) {
            // commit on the device: this rank builds matchViews' products from the gathered slots (every rank may), no list goes to the host
            std::vector<uint32_t> ids; std::vector<int32_t> base;
            dense_map(h, ids, base);
            l3d_dense_map map;
            map.n_views = (int32_t)ids.size(); map.view_ids = ids.data(); map.seg_base = base.data();
            h->chain_summary.assign(P->n, l3d_chain_summary());
            rc = l3d_shard_chain_products(P->shard, &map, h->chain_summary.data(), &h->resident_n_pot);
            if (rc) msg = std::string("shard_chain_products: ") + l3d_last_error(h->ctx);
            else { rc = adopt_resident_products(h, *P); if (rc) msg = h->err; }
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
int l3d_line3d_block_run(l3d_line3d* h, int rank, int world, int warmup_views, l3d_exchange_fn exchange, void* exchange_user, int* verdict)
{
    if (!h || !verdict) return L3D_ERR_INVALID;
    *verdict = 1;
    const double t0 = now_s();
    match_begin(h);
    ChainPlan* Pp = get_plan(h);
    if (!Pp) return L3D_OK;                               // (a schedule the chain cannot express: the caller's other paths handle it)
    ChainPlan& P = *Pp;
    int window = 1;
    for (size_t k = 0; k < P.n; ++k) for (int si : P.src_idx[k]) window = std::max(window, (int)k - si);
    if (warmup_views < 0) warmup_views = 8 * window;       // (the chain forgets a cold start after 3-7 windows on the synthetic scenes, the denser the later: scripts/speculate_blocks.py)
    std::vector<uint32_t> ids; std::vector<int32_t> base;
    dense_map(h, ids, base);
    l3d_dense_map map;
    map.n_views = (int32_t)ids.size(); map.view_ids = ids.data(); map.seg_base = base.data();
    h->chain_summary.assign(P.n, l3d_chain_summary());
    h->resident_products = false;
    const double t1 = now_s();
    int rc = l3d_match_chain_blocks(h->ctx, P.cv.data(), (int)P.n, &map, h->chain_summary.data(), &h->resident_n_pot, rank, world, warmup_views, window,
                                    exchange, exchange_user, verdict);
    h->t_gpu_call += now_s() - t1;
    if (rc) return h->fail(rc, std::string("match_chain_blocks: ") + l3d_last_error(h->ctx));
    if (*verdict != 0) return L3D_OK;
    rc = adopt_resident_products(h, P);
    if (rc) return rc;
    double st[4];
    l3d_last_stats(h->ctx, st);
    h->stat_pairs += st[0];
    h->stat_raw += st[1];
    h->t_match = now_s() - t0;
    return L3D_OK;
}
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
