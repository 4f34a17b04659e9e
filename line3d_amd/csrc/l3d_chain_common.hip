// l3d_chain_common.hip -- see l3d_chain_common.hpp
#include <algorithm>
#include <string>

#include "l3d_chain_common.hpp"
#include "l3d_runtable.hpp"

namespace l3d {

int chain_plan_views(l3d_ctx* c, const l3d_chain_view* views, int n_views, int rank, int world, std::vector<ChainViewDev>& vd, ChainLayout& L, const char* what)
{
    const std::string w(what);
    vd.assign((size_t)n_views, ChainViewDev());
    L = ChainLayout();
    for (int k = 0; k < n_views; ++k) {
        const l3d_chain_view& v = views[k];
        if (v.S_src < 0 || v.N < 0 || v.n_tbm < 0 || v.n_tbm > v.N || v.n_sources < 0 || v.n_tgt < 0) return fail(c, L3D_ERR_INVALID, w + ": inconsistent sizes");
        ChainViewDev& d = vd[(size_t)k];
        d.verified = v.n_tbm > 0;
        d.s0 = (int)(((long long)v.S_src * rank) / world);
        d.s1 = (int)(((long long)v.S_src * (rank + 1)) / world);
        L.maxS = std::max(L.maxS, v.S_src); L.maxN = std::max(L.maxN, v.N);
        if (!d.verified) continue;
        if (!v.src_segs || !v.tgt_segs || !v.offsets || !v.F || !v.RtKinv || !v.centers || !v.P || !v.RtKinv_src || !v.C_src ||
            !v.to_be_matched || !v.local2global || (v.n_sources && (!v.source_cam || !v.source_index)))
            return fail(c, L3D_ERR_INVALID, w + ": null input pointer");
        if (v.N > 255) return fail(c, L3D_ERR_INVALID, w + ": more than 255 neighbours");
        for (int s = 0; s < v.n_sources; ++s)
            if (v.source_index[s] < 0 || v.source_index[s] >= k || v.source_cam[s] < 0 || v.source_cam[s] >= v.N)
                return fail(c, L3D_ERR_INVALID, w + ": a source must be an earlier view of the chain");
        int maxW = 0;
        double p = 0;
        for (int j = 0; j < v.n_tbm; ++j) {
            const int cam = v.to_be_matched[j];
            if (cam < 0 || cam >= v.N) return fail(c, L3D_ERR_INVALID, w + ": to_be_matched out of range");
            maxW = std::max(maxW, v.offsets[2 * cam + 1]);
            p += (double)(d.s1 - d.s0) * v.offsets[2 * cam + 1];
        }
        for (int i = 0; i < v.N; ++i)
            if (v.offsets[2 * i] < 0 || v.offsets[2 * i + 1] < 0 || v.offsets[2 * i] + v.offsets[2 * i + 1] > v.n_tgt)
                return fail(c, L3D_ERR_INVALID, w + ": offsets outside the target tile");
        L.pairs += p; L.max_pairs = std::max(L.max_pairs, p);
        d.maxW = maxW;
        d.W64 = 4 * ((maxW + 255) / 256);
        if (d.W64 > kMaxW64) return fail(c, L3D_ERR_INVALID, "a neighbour has more than 16384 segments");
        // residency: segments stay in HBM; arrays not registered yet are registered now
        if (!resident_ptr(c, v.src_segs, (size_t)v.S_src * 16)) { int rc = l3d_register_segments(c, v.src_segs, v.S_src); if (rc) return rc; }
        if (!resident_ptr(c, v.tgt_segs, (size_t)v.n_tgt * 16)) { int rc = l3d_register_segments(c, v.tgt_segs, v.n_tgt); if (rc) return rc; }
        d.src = reinterpret_cast<const float4*>(resident_ptr(c, v.src_segs, (size_t)v.S_src * 16));
        d.tgt = reinterpret_cast<const float4*>(resident_ptr(c, v.tgt_segs, (size_t)v.n_tgt * 16));
        const size_t N = (size_t)v.N;
        size_t o = L.tab_bytes;
        d.o_off = o; o += N * 8; d.o_F = o; o += N * 36; d.o_R = o; o += N * 36; d.o_C = o; o += N * 12; d.o_P = o; o += N * 48;
        d.o_Rs = o; o += 36; d.o_Cs = o; o += 12; d.o_tbm = o; o += (size_t)v.n_tbm * 4; d.o_l2g = o; o += N * 4;
        d.o_sc = o; o += (size_t)v.n_sources * 4; d.o_si = o; o += (size_t)v.n_sources * 4; d.o_ss = o; o += (size_t)v.n_sources * 4;
        L.tab_bytes = chain_align16(o);
        L.mask_bytes += chain_align16((size_t)v.n_tbm * v.S_src * d.W64 * 8);
        L.max_mask_bytes = std::max(L.max_mask_bytes, chain_align16((size_t)v.n_tbm * v.S_src * d.W64 * 8));
        L.rowcnt_ints += (size_t)v.S_src * v.N;
        L.best_elems += (size_t)v.S_src;
    }
    return L3D_OK;
}


// ---- run tables rebuilt from records (l3d_runtable.hpp)
__global__ __launch_bounds__(256) void k_qt_from_records(const RtJob* __restrict__ jobs, int* __restrict__ err)
{
    const RtJob j = jobs[blockIdx.y];
    const int lane = threadIdx.x & 63;
    // (whole waves iterate together: the order check takes the previous record's key from the neighbouring lane)
    for (int i0 = blockIdx.x * 256 + (threadIdx.x & ~63); i0 < j.n; i0 += gridDim.x * 256) {
        const int i = i0 + lane;
        unsigned key = 0xffffffffu;
        if (i < j.n) {
            const Match r = j.recs[i];
            int lo = 0, hi = j.N;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (j.ids[mid] < r.camID2) lo = mid + 1; else hi = mid; }
            const bool known = lo < j.N && j.ids[lo] == r.camID2 && (int)r.segID1 < j.S && r.segID2 < 65536u;
            const unsigned q = known ? (unsigned)j.qs[lo] : 0xffu;
            j.qt[i] = ((known ? q : 0xffffu) << 16) | (r.segID2 & 0xffffu);
            key = known ? (r.segID1 << 8) | q : 0xffffffffu;
            j.skey[i] = key;
            if (!known) atomicAdd(err, 1);
        }
        const unsigned prev = __shfl_up(key, 1);                       // (lane 0 has no neighbour: k_qt_order_seams checks every 64th boundary)
        if (lane > 0 && i < j.n && key != 0xffffffffu && prev != 0xffffffffu && prev > key) atomicAdd(err, 1);
    }
}
// the wave boundaries of the order check (lane 0 of every wave of k_qt_from_records has no neighbour): one more thin pass over every 64th key
__global__ __launch_bounds__(256) void k_qt_order_seams(const RtJob* __restrict__ jobs, int* __restrict__ err)
{
    const RtJob j = jobs[blockIdx.y];
    for (int i = (blockIdx.x * 256 + threadIdx.x + 1) * 64; i < j.n; i += gridDim.x * 256 * 64) {
        const unsigned a = j.skey[i - 1], b = j.skey[i];
        if (a != 0xffffffffu && b != 0xffffffffu && a > b) atomicAdd(err, 1);
    }
}
__global__ __launch_bounds__(256) void k_rt_from_qt(const RtJob* __restrict__ jobs)
{
    const RtJob j = jobs[blockIdx.y];
    const int cells = (j.N + 1) * j.S;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < cells; c += gridDim.x * 256) {
        const int q = c / j.S, s = c - q * j.S;
        // first record whose (segment, camera) is >= (s, q); row N: (s + 1, 0)
        const unsigned want = q < j.N ? ((unsigned)s << 8) | (unsigned)q : ((unsigned)(s + 1) << 8);
        int lo = 0, hi = j.n;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (j.skey[mid] < want) lo = mid + 1; else hi = mid; }
        j.rt[c] = lo;
    }
}
void launch_qt_from_records(const RtJob* jobs_dev, int n_jobs, int max_n, int* err, hipStream_t st)
{
    if (n_jobs > 0 && max_n > 0) {
        hipLaunchKernelGGL(k_qt_from_records, dim3((unsigned)std::max(1, std::min(512, (max_n + 1023) / 1024)), (unsigned)n_jobs), dim3(256), 0, st, jobs_dev, err);
        hipLaunchKernelGGL(k_qt_order_seams, dim3((unsigned)std::max(1, std::min(64, (max_n / 64 + 255) / 256)), (unsigned)n_jobs), dim3(256), 0, st, jobs_dev, err);
    }
}
void launch_rt_from_qt(const RtJob* jobs_dev, int n_jobs, int max_cells, hipStream_t st)
{
    if (n_jobs > 0 && max_cells > 0) hipLaunchKernelGGL(k_rt_from_qt, dim3((unsigned)std::max(1, std::min(256, (max_cells + 255) / 256)), (unsigned)n_jobs), dim3(256), 0, st, jobs_dev);
}

int chain_upload_tables(l3d_ctx* c, const l3d_chain_view* views, int n_views, std::vector<ChainViewDev>& vd, ChainLayout& L, bool with_rays, hipStream_t st)
{
    HIPCHK(c, c->ch_pin_tables.reserve(L.tab_bytes + 16));
    HIPCHK(c, c->ch_tables.reserve(L.tab_bytes + 16));
    unsigned char* tab = c->ch_pin_tables.as<unsigned char>();
    for (int k = 0; k < n_views; ++k) {
        const l3d_chain_view& v = views[k];
        const ChainViewDev& d = vd[(size_t)k];
        if (!d.verified) continue;
        const size_t N = (size_t)v.N;
        memcpy(tab + d.o_off, v.offsets, N * 8); memcpy(tab + d.o_F, v.F, N * 36); memcpy(tab + d.o_R, v.RtKinv, N * 36);
        memcpy(tab + d.o_C, v.centers, N * 12); memcpy(tab + d.o_P, v.P, N * 48); memcpy(tab + d.o_Rs, v.RtKinv_src, 36);
        memcpy(tab + d.o_Cs, v.C_src, 12); memcpy(tab + d.o_tbm, v.to_be_matched, (size_t)v.n_tbm * 4);
        memcpy(tab + d.o_l2g, v.local2global, N * 4);
        if (v.n_sources) { memcpy(tab + d.o_sc, v.source_cam, (size_t)v.n_sources * 4); memcpy(tab + d.o_si, v.source_index, (size_t)v.n_sources * 4); }
        // the LOCAL camera number this view has in each source's neighbour list (run tables: which runs of the source's list point here); -1: not listed
        for (int q = 0; q < v.n_sources; ++q) {
            const l3d_chain_view& w = views[v.source_index[q]];
            int slot = -1;
            for (int j = 0; j < w.N && w.local2global; ++j) if (w.local2global[j] == v.view_id) { slot = j; break; }
            memcpy(tab + d.o_ss + (size_t)q * 4, &slot, 4);
        }
    }
    if (L.tab_bytes) HIPCHK(c, hipMemcpyAsync(c->ch_tables.p, tab, L.tab_bytes, hipMemcpyHostToDevice, st));
    L.dtab = c->ch_tables.as<unsigned char>();
    // the viewing rays of every view's target endpoints, once per chain (they only depend on the neighbour's camera and segment)
    size_t n_ray = 0;
    int max_n_tgt = 0;
    std::vector<RayJob>& jobs = c->ray_jobs;             // (lives in the context: the upload below is asynchronous)
    jobs.clear();
    for (int k = 0; k < n_views; ++k) {
        vd[(size_t)k].rays = nullptr; vd[(size_t)k].src_rays = nullptr;
        if (with_rays && vd[(size_t)k].verified && views[k].n_tbm != 0) n_ray += (size_t)views[k].n_tgt + (size_t)views[k].S_src;
    }
    HIPCHK(c, c->ch_rays.reserve(n_ray * 32 + 2 * (size_t)n_views * sizeof(RayJob) + 512));
    float4* rbase = c->ch_rays.as<float4>();
    RayJob* djobs = reinterpret_cast<RayJob*>(c->ch_rays.as<unsigned char>() + ((n_ray * 32 + 255) & ~(size_t)255));
    size_t ro = 0;
    for (int k = 0; k < n_views; ++k) {
        ChainViewDev& d = vd[(size_t)k];
        if (!with_rays || !d.verified || views[k].n_tbm == 0) continue;
        d.rays = rbase + 2 * ro; ro += (size_t)views[k].n_tgt;
        jobs.push_back(RayJob{ d.tgt, reinterpret_cast<const int2*>(L.dtab + d.o_off), reinterpret_cast<const float*>(L.dtab + d.o_R), d.rays, views[k].n_tgt, views[k].N });
        // (the view's own segments under its own camera: what k_pair_fill needs once per (segment, camera) row)
        d.src_rays = rbase + 2 * ro; ro += (size_t)views[k].S_src;
        jobs.push_back(RayJob{ d.src, nullptr, reinterpret_cast<const float*>(L.dtab + d.o_Rs), d.src_rays, views[k].S_src, 1 });
        max_n_tgt = std::max(max_n_tgt, std::max(views[k].n_tgt, views[k].S_src));
    }
    if (!jobs.empty()) {
        HIPCHK(c, hipMemcpyAsync(djobs, jobs.data(), jobs.size() * sizeof(RayJob), hipMemcpyHostToDevice, st));
        ProfScope p(c, "tgt_rays", st);
        launch_tgt_rays(djobs, (int)jobs.size(), max_n_tgt, st);
    }
    return L3D_OK;
}

int chain_assign_arenas(l3d_ctx* c, const l3d_chain_view* views, int n_views, std::vector<ChainViewDev>& vd, ChainLayout& L, bool fused_rows, bool best_positions, int mask_ring, hipStream_t st, bool run_tables)
{
    const size_t nv = (size_t)n_views;
    HIPCHK(c, c->ch_mask.reserve((mask_ring > 0 ? std::min(L.mask_bytes, (size_t)mask_ring * L.max_mask_bytes) : L.mask_bytes) + 16));
    HIPCHK(c, c->ch_rowcnt.reserve((L.rowcnt_ints + 2 * nv) * 4 + 16));
    // (row starts | upper-bound counts | their block sums: the last two zeroed, k_pair_mask adds into them)
    L.rowA_ints = L.rowcnt_ints + 4 * nv;
    L.rowub_ints = fused_rows ? L.rowcnt_ints + 4 * nv : 0;
    L.rowblk_ints = fused_rows ? L.rowcnt_ints / 256 + 8 * nv : 0;
    HIPCHK(c, c->ch_rowA.reserve((L.rowA_ints + L.rowub_ints + L.rowblk_ints) * 4 + 64));
    if (fused_rows) HIPCHK(c, hipMemsetAsync(c->ch_rowA.as<int>() + L.rowA_ints, 0, (L.rowub_ints + L.rowblk_ints) * 4, st));
    HIPCHK(c, c->ch_best.reserve(L.best_elems * 8 + 16));
    if (best_positions) HIPCHK(c, c->ch_bestpos.reserve(L.best_elems * 4 + 16));
    HIPCHK(c, hipMemsetAsync(c->ch_rowcnt.p, 0, (L.rowcnt_ints + 2 * nv) * 4, st));
    if (run_tables) HIPCHK(c, c->ch_rt.reserve((L.rowcnt_ints + L.best_elems + 4 * nv) * 4 + 64));      // (N + 1) x S ints per verified view
    size_t mo = 0, ro = 0, bo = 0, ao = 0, ko = 0, to = 0;
    int* stats_base = c->ch_rowcnt.as<int>() + L.rowcnt_ints;
    for (int k = 0; k < n_views; ++k) {
        ChainViewDev& d = vd[(size_t)k];
        d.stats = stats_base + 2 * k;
        if (!d.verified) continue;
        const l3d_chain_view& v = views[k];
        if (mask_ring > 0 && (size_t)mask_ring * L.max_mask_bytes < L.mask_bytes) {
            d.mask = reinterpret_cast<unsigned long long*>(c->ch_mask.as<unsigned char>() + (mo % (size_t)mask_ring) * L.max_mask_bytes);
            mo += 1;
        } else {
            d.mask = reinterpret_cast<unsigned long long*>(c->ch_mask.as<unsigned char>() + mo);
            mo += chain_align16((size_t)v.n_tbm * v.S_src * d.W64 * 8);
        }
        d.rowcnt = c->ch_rowcnt.as<int>() + ro; ro += (size_t)v.S_src * v.N;
        d.rowA = c->ch_rowA.as<int>() + ao;
        if (fused_rows) {
            d.rowub = c->ch_rowA.as<int>() + L.rowA_ints + ao;
            d.rowblk = c->ch_rowA.as<int>() + L.rowA_ints + L.rowub_ints + ko;
        }
        ao += ((size_t)v.S_src * v.N + 4) & ~(size_t)3;     // 16-byte aligned slices
        ko += (((size_t)v.S_src * v.N + 255) / 256 + 4) & ~(size_t)3;
        d.best = c->ch_best.as<float2>() + bo;
        d.bestpos = best_positions ? c->ch_bestpos.as<int>() + bo : nullptr;
        bo += (size_t)v.S_src;
        d.rt = run_tables ? c->ch_rt.as<int>() + to : nullptr;
        to += (((size_t)v.N + 1) * v.S_src + 3) & ~(size_t)3;
    }
    const size_t nrow_max = (size_t)L.maxS * L.maxN;
    HIPCHK(c, c->row_start.reserve((nrow_max + 1) * 4));
    HIPCHK(c, c->ch_cursor.reserve(nrow_max * 4 + 16));
    HIPCHK(c, c->kept_cnt.reserve((size_t)L.maxS * 4 + 4));
    HIPCHK(c, c->ch_segorder.reserve((size_t)L.maxS * 4 + 16));
    HIPCHK(c, c->kept_start.reserve((size_t)L.maxS * 4 + 8));
    return L3D_OK;
}

int chain_reserve_candidates(l3d_ctx* c, const ChainLayout& L, size_t cand_cap, int ring)
{
    if (c->opt.vw_split != 0 || L.maxN > 16) {     // bucket starts in global memory (more than 16 neighbours) | split verification (k_vw_walk): per segment the bucket starts of its built image, header, 64-bit best, counters | unit table
        const size_t S = (size_t)std::max(L.maxS, 1);
        HIPCHK(c, c->vw_bstart.reserve(S * (kVWBuckets + 1) * 4 + 64));
        HIPCHK(c, c->vw_segstate.reserve(S * 36 + 256));
    }
    HIPCHK(c, c->cand_meta.reserve(cand_cap * 8));
    HIPCHK(c, c->cand_depths.reserve(cand_cap * 16));
    HIPCHK(c, c->cand_conf.reserve(cand_cap * 4));
    HIPCHK(c, c->vw_scratch.reserve((cand_cap + kVWSlack) * 16));
    if (ring > 0) {     // ring of stage-1 candidate buffers: stage 1 (incl. the triangulation of its candidates) runs views ahead of the chain
        HIPCHK(c, c->ch_ringA_meta.reserve((size_t)ring * cand_cap * 8));
        HIPCHK(c, c->ch_ringA_depths.reserve((size_t)ring * cand_cap * 16));
    }
    return L3D_OK;
}

PairArgs chain_pair_args(const l3d_ctx* c, const l3d_chain_view& v, const ChainViewDev& d, const unsigned char* dtab)
{
    PairArgs pa;
    pa.src_segs = d.src; pa.tgt_segs = d.tgt;
    pa.offsets = reinterpret_cast<const int2*>(dtab + d.o_off);
    pa.F = reinterpret_cast<const float*>(dtab + d.o_F);
    pa.RtKinv = reinterpret_cast<const float*>(dtab + d.o_R);
    pa.centers = reinterpret_cast<const float*>(dtab + d.o_C);
    pa.RtKinv_src = reinterpret_cast<const float*>(dtab + d.o_Rs);
    pa.C_src = reinterpret_cast<const float*>(dtab + d.o_Cs);
    pa.tbm = reinterpret_cast<const int*>(dtab + d.o_tbm);
    pa.mask = d.mask;
    pa.S_src = v.S_src; pa.N = v.N; pa.n_tbm = v.n_tbm; pa.W64 = d.W64;
    pa.seg_begin = d.s0; pa.seg_end = d.s1; pa.cand_cap = 0; pa.wedge_pretest = c->wedge_pretest; pa.dbg = c->pair_dbg; pa.dbg_view = (int)v.view_id; pa.rowcnt = nullptr;
    pa.depth_in_fill = 1;               // the four depths of a stage-1 pair are triangulated once, by k_pair_fill
    const bool src_rays_env = c->opt.src_rays != 0;   // (0: A/B, k_pair_fill normalises per row)
    pa.tgt_rays = d.rays; pa.src_rays = src_rays_env ? d.src_rays : nullptr;
    return pa;
}

VerifyArgs chain_verify_args(l3d_ctx* c, const l3d_chain_view& v, const ChainViewDev& d, const unsigned char* dtab, size_t cand_cap)
{
    VerifyArgs va;
    va.exist_cams = nullptr; va.n_exist_cams = 0;
    va.src_segs = d.src; va.tgt_segs = d.tgt;
    va.offsets = reinterpret_cast<const int2*>(dtab + d.o_off);
    va.P = reinterpret_cast<const float*>(dtab + d.o_P);
    va.RtKinv_src = reinterpret_cast<const float*>(dtab + d.o_Rs);
    va.C_src = reinterpret_cast<const float*>(dtab + d.o_Cs);
    va.row_start = c->row_start.as<int>();
    va.cand_meta = c->cand_meta.as<uint2>(); va.cand_depths = c->cand_depths.as<float4>(); va.cand_conf = c->cand_conf.as<float>();
    va.N = v.N; va.seg_begin = d.s0; va.seg_end = d.s1; va.nrow_total = v.S_src * v.N;
    va.sigma_p = v.sigma_p; va.sigma_a = v.sigma_a; va.spatial_k = v.spatial_k;
    va.debug = 0; va.stamps = nullptr; va.cand_cap = (int)cand_cap; va.res = nullptr; va.bstart_g = nullptr;
    va.seg_order = c->ch_segorder.as<int>();
    va.mmax = 0; va.only_above = -1; va.skip_above = 0; va.big = 0;
    va.kept_cnt = nullptr; va.best_depths = nullptr; va.scratch = nullptr; va.scratch_stride = 0;
    return va;
}

void chain_launch_verify(l3d_ctx* c, VerifyArgs& va, const ChainViewDev& d, const int* exist_cams, int n_exist_cams, int raw_max_per_segment, size_t cand_cap, hipStream_t st)
{
    // LDS budget from the raw statistics (+ room for reverse matches); bigger segments take the global-scratch blocks
    // (raw_max_per_segment < 0: no statistics -- the largest image the budget allows; the budget, not the image, sets the occupancy)
    int mmax = raw_max_per_segment < 0 ? 16384 : raw_max_per_segment + raw_max_per_segment / 4 + 64;
    while (mmax > 64 && verify_window_lds_bytes(mmax, va.N) > verify_window_max_lds(c->opt.vw_lds)) mmax = mmax * 3 / 4;
    va.mmax = mmax;
    if (va.seg_end <= va.seg_begin) return;
    if (c->verify_mode == 0 && verify_window_supported(va.N)) {
        // one launch: segments that fit the LDS image, the ones that outgrow it (reverse matches are not in the estimate) on a global
        // scratch, and the per-segment epilogue (best hypothesis, kept count)
        va.skip_above = 1; va.only_above = -1; va.big = 2;
        va.scratch = c->vw_scratch.as<float>(); va.scratch_stride = (long long)cand_cap + kVWSlack;
        va.kept_cnt = c->kept_cnt.as<int>(); va.best_depths = d.best;
        va.exist_cams = exist_cams; va.n_exist_cams = n_exist_cams;            // reverse-match runs are ordered by the segment's workgroup
        // split verification (option vw_split: 1 always, 0 never, -1 = a launch of few segments -- one rank's slice of a view, up to vw_wide_max -- whose
        // candidate capacity allows vw_split_avg and more per segment: a dense scene): long segments are only BUILT by this launch, their hypotheses are
        // verified in units by the next (k_vw_walk).  A launch that fills the chip gains nothing (measured: NOTEBOOK section 11.y), a rank's slice does
        const int unit = std::max(256, (c->opt.vw_unit / 256) * 256);
        const size_t S = (size_t)(va.seg_end - va.seg_begin);
        const bool want = c->opt.vw_split > 0 || (c->opt.vw_split < 0 && (int)S <= c->opt.vw_wide_max && cand_cap / std::max<size_t>(S, 1) >= (size_t)c->opt.vw_split_avg);
        if (va.N > 16 && c->opt.vw_gb != 0 && !va.stamps && c->vw_bstart.cap >= S * (kVWBuckets + 1) * 4) va.bstart_g = c->vw_bstart.as<int>();
        VWSplitArgs sp;
        if (want && !va.stamps && c->vw_bstart.cap >= S * (kVWBuckets + 1) * 4 && c->vw_segstate.cap >= S * 36 + 16) {
            sp.split_unit = unit;
            sp.units_max = (int)std::min<size_t>(S + cand_cap / (size_t)unit + 1, 0x3fffffff);
            unsigned char* ss = c->vw_segstate.as<unsigned char>();
            sp.bstart_g = c->vw_bstart.as<int>();
            sp.seg_hdr = reinterpret_cast<int4*>(ss);                                  // 16 B per segment
            sp.best64 = reinterpret_cast<unsigned long long*>(ss + S * 16);            // 8 B
            sp.done = reinterpret_cast<int*>(ss + S * 24);                             // 4 B
            sp.unit_start = reinterpret_cast<int*>(ss + S * 28);                       // 4 B x (segments + 1)
        }
        ProfScope p(c, "verify_window", st);
        launch_verify_window(va, st, c->opt.vw_wide_max, &sp);
    } else {
        va.skip_above = 0; va.only_above = -1; va.big = 0; va.scratch = nullptr; va.scratch_stride = 0;
        va.kept_cnt = nullptr; va.best_depths = nullptr;
        { ProfScope p(c, "verify", st); launch_verify(va, st); }
        { ProfScope p(c, "seg_post", st); launch_seg_post(va, c->kept_cnt.as<int>(), d.best, st); }
    }
}

}  // namespace l3d
