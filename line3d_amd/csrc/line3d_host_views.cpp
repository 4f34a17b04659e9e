// line3d_host_views.cpp -- views, neighbours, scene normalisation, the per-view seam path and the host bookkeeping of performMatching (line3D.cc:166-344, 838-884; view.cc)
// (one translation unit of the host pipeline; shared declarations: line3d_host_internal.hpp)
#include "line3d_host_internal.hpp"

namespace l3dh {


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
              const int32_t* coll_i, const int32_t* coll_j, const float* coll_w, int n_coll)
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
        const bool hint = hopt(h).reserve_hint != 0;       // (0: a memory-bound job -- a rank of a partitioned run -- lets every stage take what it turns out to need)
        std::thread warm([h, nd_all, nv_all, nn, hint, &reserve_rc, &t_reserve]() { const double a0 = now_s(); if (hint) reserve_rc = l3d_reserve_hint(h->ctx, nd_all, nv_all, nn); t_reserve = now_s() - a0; });
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
    h->partitioned = false;
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


}  // namespace l3dh
