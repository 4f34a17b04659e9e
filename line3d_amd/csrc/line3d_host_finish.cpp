// line3d_host_finish.cpp -- what follows matchViews in compute3Dmodel: greedy selection, affinity fill, diffusion, clustering, line fit (line3D.cc:899-1597, clustering.cc)
// (one translation unit of the host pipeline; shared declarations: line3d_host_internal.hpp)
#include "line3d_host_internal.hpp"

namespace l3dh {

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
void perform_clustering(const l3d_edge* edges_in, size_t n_edges, int numNodes, float c, std::vector<int>& labels, bool presorted)
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

// the affinity fill of a job whose matchViews ran partitioned (l3d_affinity_fill_sharded): this rank's block of sources, then the global tables --
// hypothesis numbers, which segment every hypothesis belongs to -- replace the local ones of greedy_selection_resident
int fill_affinity_sharded(L* h)
{
    const size_t nv = h->vlist.size();
    std::vector<size_t> voff(nv + 1, 0);
    for (size_t i = 0; i < nv; ++i) voff[i + 1] = voff[i] + (size_t)h->vlist[i]->S();
    L::AffTables& T = h->aff;
    const bool changed = !T.coll_valid || T.coll_start.size() != voff.back() + 1;
    if (changed) pack_collinearities(h, voff);
    std::vector<int32_t> vhb(nv + 1, 0);
    int32_t *node_hyp = nullptr, *hyp_dense = nullptr;
    int n_edges = 0, n_nodes = 0, nh_all = 0;
    int64_t cand_all = 0;
    const double t0 = now_s();
    int rc = l3d_affinity_fill_sharded(h->ctx, T.coll_start.data(), T.coll_other.data(), T.coll_w.data(), changed ? 1 : 0, h->sigma_a, h->part_exchange, h->part_user,
                                       &n_edges, &node_hyp, &n_nodes, &cand_all, vhb.data(), &hyp_dense, &nh_all);
    if (rc) return h->fail(rc, std::string("affinity fill (sharded): ") + l3d_last_error(h->ctx));
    T.coll_valid = true;
    h->hyp_begin.assign(nv + 1, 0);
    for (size_t i = 0; i <= nv; ++i) h->hyp_begin[i] = (size_t)vhb[i];
    h->hyps.assign((size_t)nh_all, Hyp());
    h->aff.hyp_cam.assign((size_t)nh_all, 0u);
    parallel_slices(nv, finish_threads(), [&](size_t v0, size_t v1, unsigned) {
        for (size_t vi = v0; vi < v1; ++vi)
            for (size_t k = h->hyp_begin[vi]; k < h->hyp_begin[vi + 1]; ++k) {
                h->hyps[k].src = mk(h->vlist[vi]->id, (uint32_t)((size_t)hyp_dense[k] - voff[vi]));
                h->aff.hyp_cam[k] = h->vlist[vi]->id;
            }
    });
    h->A.clear(); h->n_edges = (size_t)n_edges; h->A_on_host = n_edges == 0;
    h->local2global.resize((size_t)n_nodes);
    h->node_hyp.resize((size_t)n_nodes);
    parallel_slices((size_t)n_nodes, finish_threads(), [&](size_t k0, size_t k1, unsigned) {
        for (size_t k = k0; k < k1; ++k) { h->node_hyp[k] = node_hyp[k]; h->local2global[k] = h->hyps[(size_t)node_hyp[k]].src; }
    });
    l3d_free(node_hyp); l3d_free(hyp_dense);
    if (hopt(h).timing) fprintf(stderr, "[l3d finish] sharded fill: %d hypotheses in the job, %lld candidate pairs, %zu edges, %.2f ms\n", nh_all, (long long)cand_all, h->n_edges, (now_s() - t0) * 1e3);
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
    size_t nh = h->hyps.size();
    if (nh == 0 && !h->partitioned) return L3D_OK;          // (partitioned: a rank without hypotheses still takes part in the collective fill)

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
    if (h->partitioned) {
        const int rc = fill_affinity_sharded(h);
        if (rc) return rc;
        resident_list = true;
        nh = h->hyps.size();                        // (global from here on)
        lap("affinity fill (sharded by source key)");
    } else if (h->resident_products) {
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


}  // namespace l3dh
