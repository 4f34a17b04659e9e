// line3d_host_chain.cpp -- matchViews on the resident chain: schedule, finaliser threads, products on the device (line3D.cc:780-896)
// (one translation unit of the host pipeline; shared declarations: line3d_host_internal.hpp)
#include "line3d_host_internal.hpp"

namespace l3dh {


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
    P.matched_final = h->matched;
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
    }
    h->matched = P.matched_final;                               // line3D.cc:875-881 for every view of the schedule (plan_chain simulated exactly that)
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
    // (partitioned products: this rank holds a share of the lists -- the self-check needs all of them; the tests compare the shares with the one chain's)
    if (hopt(h).check_pot && !h->partitioned) { int rc = check_resident_products(h, P); if (rc) return rc; }
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
    const double t2 = now_s();
    rc = adopt_resident_products(h, P);
    if (rc) return rc;
    if (hopt(h).timing) fprintf(stderr, "[l3d match_views] the call %.3f ms, adopting its products on the host %.3f ms\n", (t2 - t1) * 1e3, (now_s() - t2) * 1e3);
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
    if (h->force_sync) { h->last_match_path = 2; return match_views_sync(h); }
    const double t0 = now_s();
    match_begin(h);
    const double ta = now_s();
    ChainPlan* Pp = get_plan(h);
    if (!Pp) { h->last_match_path = 3; return match_views_sync(h); }     // (the schedule is not static: an early return's local camera numbers name a view that still takes reverse matches)
    ChainPlan& P = *Pp;
    const double tb = now_s();
    if (!(h->host_bookkeeping || hopt(h).host_bookkeeping)) {
        const int rr = match_views_resident(h, P, t0);
        h->last_match_path = 0;
        if (rr != L3D_ERR_UNSUPPORTED) return rr;           // (more kept matches than the device builder takes: host lists)
    }
    h->last_match_path = 1;
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

void destroy_finalizer(L* h) { delete static_cast<ChainFinalizer*>(h->finalizer); h->finalizer = nullptr; }

}  // namespace l3dh
