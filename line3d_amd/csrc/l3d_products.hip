// l3d_products.hip -- what Line3D::performMatching leaves behind for the rest of compute3Dmodel, built ON THE DEVICE from the kept
// arena of the resident chain (which never leaves HBM):
//
//   potential_correspondences_   line3D.cc:861-865: for every kept match (view, segment) <-> (camera, target) both directions, in a
//                                map of maps: set semantics, ascending iteration.  Here: one 64-bit key per direction over DENSE
//                                segment ids, ONE stable radix sort (hipCUB), duplicates dropped, CSR -- the reference's host loop
//                                over std::map inserts (and rounds 1-2's finaliser threads on the host) disappear.
//   the match files              line3D.cc:884, view.cc:165-183: a view's own list overwrites whatever earlier views pushed, reduced
//                                to the best match per segment (first of the highest confidence): found by the kept writer while
//                                the confidences are in registers (l3d_kept.hpp); here only turned into arena references.
//   median depths                cudawrapper.cu:1058-1076: std::sort + middle element on the host there; a radix select per view here.
//   greedy selection             line3D.cc:899-965 + L3DView::unprojectSegment view.cc:302-342: one 3-D hypothesis per segment with a
//                                best match, numbered in dense order (prefix sum), unprojected in double with the host's operations.
//
// Views with nothing left to match return early (cudawrapper.cu:877-878): their list is the localized existing list -- what the
// earlier views pushed, in push order, LOCAL camera ids, confidence 0 -- and the reference then files its entries under those
// local numbers read as view ids (line3D.cc:838-866).  Reproduced: such a view's entries are formed from its sources' records.
#include "l3d_sort.hpp"

#include <algorithm>
#include <vector>

#include "l3d_products.hpp"
#include "l3d_linalg.hpp"
#include "l3d_unproject.hpp"

using namespace l3d;

namespace l3d {

struct ProdView {                   // per chain index
    long long kept_base;
    int n_kept, R;
    int dense_base, S;
    int early, pad;
    const int* bestpos;
    const float2* best;
};
// The keys are formed, sorted and reduced to CSR rows in BLOCKS of consecutive dense views (ascending: the blocks' row ranges
// concatenate to the table), so that the transient memory -- two 64-bit key arrays, flags, positions: 24 bytes per key slot -- is
// bounded by the block size and not by the scene (the reference spills every view's matches to disk, view.cc:150-224).
// A block holds the keys whose SOURCE segment lies in its dense range [d0, d1): the forward key of a record of one of its views, the
// backward key of a record that points into it.  Every chain view / early-return pair that can contribute gets a region of
// 2 x (its records) slots in the block's key array (out_off; -1: it cannot contribute); a key outside the range is written invalid.
struct ProdBlock {
    int d0, d1;                     // dense range
    long long slots;                // key slots of the block (without the sentinel)
    size_t off_view, off_src;       // this block's out_off tables inside the uploaded table buffer (long long per chain view / per ProdSrc)
};
struct ProdSrc {                    // one (early-return view, source) pair
    int view, src;                  // chain indices
    int alias_base, alias_S;        // dense range of the view the source's LOCAL camera number names (-1: no such view)
    long long out_off;              // first key slot of this pair
    int rank, pad;                  // position of the source in the view's list (push order)
};

constexpr unsigned long long kInvalidKey = ~0ull;

__device__ __forceinline__ int find_view(const unsigned* __restrict__ ids, int n, unsigned id)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (ids[mid] < id) lo = mid + 1; else hi = mid; }
    return lo < n && ids[lo] == id ? lo : -1;
}

// two keys per kept record of the verified views: (dense source << nb | dense target) and the reverse (line3D.cc:864-865)
__global__ __launch_bounds__(256) void k_prod_keys(const Match* __restrict__ arena, const ProdView* __restrict__ pv, const long long* __restrict__ out_off,
                                                   const unsigned* __restrict__ ids, const int* __restrict__ seg_base, int n_all, int nb, int d0, int d1,
                                                   unsigned long long* __restrict__ keys)
{
    const ProdView v = pv[blockIdx.y];
    const long long off = out_off[blockIdx.y];
    if (v.early || off < 0) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < v.n_kept; i += gridDim.x * 256) {
        const Match r = arena[v.kept_base + i];
        const int t = find_view(ids, n_all, r.camID2);
        unsigned long long f = kInvalidKey, b = kInvalidKey;
        if (t >= 0 && (int)r.segID1 < v.S) {
            const int tb = seg_base[t], tS = seg_base[t + 1] - tb;
            if ((int)r.segID2 < tS) {
                const int ai = v.dense_base + (int)r.segID1, di = tb + (int)r.segID2;
                const unsigned long long a = (unsigned long long)ai, d = (unsigned long long)di;
                if (ai >= d0 && ai < d1) f = (a << nb) | d;
                if (di >= d0 && di < d1) b = (d << nb) | a;
            }
        }
        keys[off + 2 * (long long)i] = f;
        keys[off + 2 * (long long)i + 1] = b;
    }
}

// early-return views: the records of source `src` that point at the view, read reversed; the camera of such an entry is the
// source's LOCAL number in the view's neighbour list (alias = the view that number happens to name).  Also the view's best
// match per segment: the first entry of the list in push order (all confidences are 0): atomicMin of (source rank, record).
// out_off == nullptr: the pass that only finds the best matches and the list lengths (once per build); otherwise the keys of one block.
__global__ __launch_bounds__(256) void k_prod_keys_early(const Match* __restrict__ arena, const ProdView* __restrict__ pv, const ProdSrc* __restrict__ ps,
                                                         const long long* __restrict__ out_off, const unsigned* __restrict__ chain_view_id, int nb, int d0, int d1,
                                                         unsigned long long* __restrict__ keys, unsigned long long* __restrict__ best_ref, int* __restrict__ list_len)
{
    const ProdSrc e = ps[blockIdx.y];
    const long long off = out_off ? out_off[blockIdx.y] : 0;
    if (off < 0) return;
    const ProdView v = pv[e.view], s = pv[e.src];
    const unsigned vid = chain_view_id[e.view];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < s.n_kept; i += gridDim.x * 256) {
        const Match r = arena[s.kept_base + i];
        unsigned long long f = kInvalidKey, b = kInvalidKey;
        if (r.camID2 == vid) {
            if (!out_off) atomicAdd(&list_len[e.view], 1);                      // (length of the view's list: statistics)
            // the list entry: segID1 = r.segID2 (the view's segment), camID2 = local number, segID2 = r.segID1
            if ((int)r.segID2 < v.S) {
                if (!out_off) { if (!e.pad) atomicMin(&best_ref[v.dense_base + (int)r.segID2], ((unsigned long long)e.rank << 40) | (unsigned long long)(s.kept_base + i)); }
                else if (e.alias_base >= 0 && (int)r.segID1 < e.alias_S) {
                    const int ai = v.dense_base + (int)r.segID2, di = e.alias_base + (int)r.segID1;
                    const unsigned long long a = (unsigned long long)ai, d = (unsigned long long)di;
                    if (ai >= d0 && ai < d1) f = (a << nb) | d;
                    if (di >= d0 && di < d1) b = (d << nb) | a;
                }
            }
        }
        if (out_off) { keys[off + 2 * (long long)i] = f; keys[off + 2 * (long long)i + 1] = b; }
    }
}

// best match per segment as a reference into the arena: verified views from the kept writer's positions, early-return views
// from the atomicMin keys (-> record index, "read reversed")
__global__ __launch_bounds__(256) void k_prod_best(const ProdView* __restrict__ pv, long long* __restrict__ best_ref)
{
    const ProdView v = pv[blockIdx.y];
    for (int s = blockIdx.x * 256 + threadIdx.x; s < v.S; s += gridDim.x * 256) {
        long long* o = best_ref + v.dense_base + s;
        if (v.early) { const long long k = *o; if (k != -1) *o = (k & kBestIndexMask) | kBestReversed; }
        else if (v.bestpos) { const int p = v.bestpos[s]; *o = p < 0 || v.n_kept == 0 ? -1 : v.kept_base + p; }
    }
}

// the median of the depths of every segment's best hypothesis (cudawrapper.cu:1058-1076: both depths of every segment that has
// one, sorted, element size/2): radix select over order-preserving keys, one workgroup per view
__global__ __launch_bounds__(256) void k_prod_median(const ProdView* __restrict__ pv, float* __restrict__ median)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned s_sel[3];       // prefix, rank, count
    const ProdView v = pv[blockIdx.x];
    const int tid = threadIdx.x;
    if (v.early || !v.best || v.R == 0) { if (tid == 0) median[blockIdx.x] = 1.0f; return; }     // untouched (line3D.cc:811, cudawrapper.cu:955-956)
    auto key_of = [](float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    if (tid == 0) s_sel[2] = 0;
    __syncthreads();
    unsigned cnt = 0;
    for (int s = tid; s < v.S; s += 256) cnt += v.best[s].x != -1.0f ? 2u : 0u;
    atomicAdd(&s_sel[2], cnt);
    __syncthreads();
    const unsigned n2 = s_sel[2];
    if (n2 == 0) { if (tid == 0) median[blockIdx.x] = -1.0f; return; }                           // cudawrapper.cu:1066
    if (tid == 0) { s_sel[0] = 0; s_sel[1] = n2 / 2; }
    for (int pass = 3; pass >= 0; --pass) {
        hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = s_sel[0];
        const int shift = pass * 8;
        for (int s = tid; s < v.S; s += 256) {
            const float2 b = v.best[s];
            if (b.x == -1.0f) continue;
            const unsigned k0 = key_of(b.x), k1 = key_of(b.y);
            if (pass == 3 || (k0 >> (shift + 8)) == prefix) atomicAdd(&hist[(k0 >> shift) & 255u], 1u);
            if (pass == 3 || (k1 >> (shift + 8)) == prefix) atomicAdd(&hist[(k1 >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned rank = s_sel[1], cum = 0;
            int bkt = 0;
            for (; bkt < 256; ++bkt) { if (cum + hist[bkt] > rank) break; cum += hist[bkt]; }
            s_sel[0] = (prefix << 8) | (unsigned)bkt;
            s_sel[1] = rank - cum;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const unsigned k = s_sel[0];
        median[blockIdx.x] = __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
    }
}

// sorted keys -> first-occurrence flags (the sentinel behind the last key is invalid by construction)
__global__ __launch_bounds__(256) void k_prod_flags(const unsigned long long* __restrict__ keys, long long n, int* __restrict__ flag)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    flag[i] = (k != kInvalidKey && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
}

// unique keys -> CSR.  The thread of a first occurrence writes its target; where the source changes it also writes the row starts
// of every source in between (segments without entries), and the first invalid key closes the table.
// One block of sources [d0, d1): its rows start at `base` (the entries of the blocks in front of it); the closing (first invalid) key
// writes the start of row d1 = the end of this block's rows = the base of the next block (or of the table when d1 = n_dense).
__global__ __launch_bounds__(256) void k_prod_csr(const unsigned long long* __restrict__ keys, const int* __restrict__ flag, const int* __restrict__ pos,
                                                  long long n, int nb, int d0, int d1, long long base, long long* __restrict__ pot_start, int* __restrict__ pot_tgt)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    const unsigned long long mask = (1ull << nb) - 1ull;
    const bool invalid = k == kInvalidKey;
    if (invalid && i > 0 && keys[i - 1] == kInvalidKey) return;
    const long long p = base + pos[i];
    const long long src = invalid ? (long long)d1 : (long long)(k >> nb);
    const long long prev = i == 0 ? (long long)d0 - 1 : (long long)(keys[i - 1] >> nb);
    if (!invalid && flag[i]) pot_tgt[p] = (int)(k & mask);
    if (i == 0 || prev != src) for (long long d = prev + 1; d <= src; ++d) pot_start[d] = p;
}

// partitioned products: the rows this rank does not hold are empty -- in front of its rows they start at 0, behind them at its last entry
__global__ __launch_bounds__(256) void k_prod_fill_rows(long long* __restrict__ pot_start, long long lo, long long hi, long long value)
{
    const long long i = lo + (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < hi) pot_start[i] = value;
}

// a rank's piece of the table (l3d_match_chain_blocks): its row starts, numbered from 0, shifted to where its entries sit in the whole table
__global__ __launch_bounds__(256) void k_prod_shift_rows(const long long* __restrict__ piece, long long n_rows, long long base, long long* __restrict__ pot_start)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n_rows) pot_start[i] = piece[i] + base;
}

}  // namespace l3d

namespace {

int bits_for(int n) { int b = 1; while ((1ll << b) <= (long long)n) ++b; return b; }

}  // namespace

void l3d::launch_prod_shift_rows(const long long* piece, long long n_rows, long long base, long long* pot_start_at, hipStream_t st)
{
    if (n_rows > 0) hipLaunchKernelGGL(k_prod_shift_rows, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, piece, n_rows, base, pot_start_at);
}

int l3d::build_products(l3d_ctx* c, const l3d_chain_view* views, int n_views, const ProdChainView* pvh, const ChainResult* hres,
                        const l3d_dense_map* map, l3d_chain_summary* summary, int64_t* n_pot_out, int dv0, int dv1, const char* held)
{
    Products& P = c->products;
    P.valid = false; P.hyp_valid = false;
    hipStream_t st = c->stream;
    const int nv = map->n_views;
    if (nv <= 0) return fail(c, L3D_ERR_INVALID, "products: empty dense map");
    for (int i = 0; i < nv; ++i) {
        if (map->seg_base[i + 1] < map->seg_base[i] || (i && map->view_ids[i] <= map->view_ids[i - 1]) || map->seg_base[0] != 0)
            return fail(c, L3D_ERR_INVALID, "products: the dense map must ascend");
    }
    const int nd = map->seg_base[nv];
    if (nd <= 0) return fail(c, L3D_ERR_INVALID, "products: no segments");
    // [dv0, dv1): the dense views whose rows of the table this call builds (dv1 < 0: all) -- a rank of l3d_match_chain_blocks builds the rows of
    // its own block only, with offsets that start at 0; the pieces are all-gathered and put together by the caller
    const bool partial = dv1 >= 0;
    if (!partial) { dv0 = 0; dv1 = nv; }
    if (dv0 < 0 || dv1 > nv || dv0 > dv1) return fail(c, L3D_ERR_INVALID, "products: bad view range");
    P.seg_base.assign(map->seg_base, map->seg_base + nv + 1);
    P.view_ids.assign(map->view_ids, map->view_ids + nv);
    P.res.assign(hres, hres + n_views);
    P.n_dense = nd; P.n_views_all = nv; P.n_chain = n_views;
    P.chain_view.assign((size_t)n_views, -1);
    P.chain_verified.assign((size_t)n_views, 0);
    for (int k = 0; k < n_views; ++k) P.chain_verified[(size_t)k] = pvh[k].verified ? 1 : 0;
    P.chain_view_id.resize((size_t)n_views);
    P.early_src_index.assign((size_t)n_views, {}); P.early_src_cam.assign((size_t)n_views, {});
    auto view_of = [&](unsigned id) { auto it = std::lower_bound(P.view_ids.begin(), P.view_ids.end(), id); return it != P.view_ids.end() && *it == id ? (int)(it - P.view_ids.begin()) : -1; };

    // ---- host tables: one ProdView per chain view, one ProdSrc per (early-return view, source)
    std::vector<ProdView> pv((size_t)n_views);
    std::vector<ProdSrc> ps;
    std::vector<int> ps_alias_view;                         // dense view a ProdSrc's alias names (-1: none)
    long long total_kept = 0, early_slots = 0;
    for (int k = 0; k < n_views; ++k) if (pvh[k].verified) total_kept = std::max(total_kept, (long long)hres[k].kept_base + hres[k].n_kept);
    int max_kept = 0, maxS = 1;
    for (int k = 0; k < n_views; ++k) {
        const l3d_chain_view& v = views[k];
        const int vi = view_of(v.view_id);
        if (vi < 0 || P.seg_base[(size_t)vi + 1] - P.seg_base[(size_t)vi] != v.S_src) return fail(c, L3D_ERR_INVALID, "products: a chain view is missing from the dense map");
        P.chain_view[(size_t)k] = vi; P.chain_view_id[(size_t)k] = v.view_id;
        ProdView& o = pv[(size_t)k];
        o.kept_base = pvh[k].verified ? hres[k].kept_base : 0; o.n_kept = pvh[k].verified ? hres[k].n_kept : 0; o.R = pvh[k].verified ? hres[k].R : 0;
        o.dense_base = P.seg_base[(size_t)vi]; o.S = v.S_src; o.early = pvh[k].verified ? 0 : 1; o.pad = 0;
        o.bestpos = pvh[k].bestpos; o.best = pvh[k].best;
        max_kept = std::max(max_kept, o.n_kept); maxS = std::max(maxS, o.S);
        if (!pvh[k].verified) {
            P.early_src_index[(size_t)k].assign(v.source_index, v.source_index + v.n_sources);
            P.early_src_cam[(size_t)k].assign(v.source_cam, v.source_cam + v.n_sources);
            for (int q = 0; q < v.n_sources; ++q) {
                const int si = v.source_index[q];
                if (si < 0 || si >= k || !pvh[si].verified) continue;       // (a source is a verified earlier view)
                ProdSrc e;
                e.view = k; e.src = si; e.rank = q; e.pad = 0;     // (partitioned: an early-return view's best matches are found on EVERY rank -- the records that point at it are
                                                                    // all-gathered, in list order -- because what is filed under its local camera numbers makes it a target of far-away rows)
                const int av = view_of((unsigned)v.source_cam[q]);           // the LOCAL camera number read as a view id (line3D.cc:861-865)
                e.alias_base = av >= 0 ? P.seg_base[(size_t)av] : -1; e.alias_S = av >= 0 ? P.seg_base[(size_t)av + 1] - P.seg_base[(size_t)av] : 0;
                e.out_off = 0;
                early_slots += 2 * (long long)hres[si].n_kept;
                ps.push_back(e);
                ps_alias_view.push_back(av);
            }
        }
    }
    const long long slots_all = 2 * total_kept + early_slots;      // every key there can be: the bound of the table's entries
    const int nb = bits_for(nd);
    P.total_kept = total_kept;

    // ---- blocks of consecutive dense views.  touching[x] = the chain views / early pairs that can have a key with a source in dense
    // view x: a verified chain view touches its own view (forward keys) and its neighbours' (backward keys); an early pair its view and
    // the view its alias names.  A block is grown view by view while its key slots stay within the budget.
    std::vector<std::vector<int>> touch_view((size_t)nv), touch_src((size_t)nv);
    for (int k = 0; k < n_views; ++k) {
        if (!pvh[k].verified || hres[k].n_kept == 0) continue;
        touch_view[(size_t)P.chain_view[(size_t)k]].push_back(k);
        for (int q = 0; q < views[k].N; ++q) {
            const int t = view_of(views[k].local2global[q]);
            if (t >= 0 && t != P.chain_view[(size_t)k]) touch_view[(size_t)t].push_back(k);
        }
    }
    for (size_t q = 0; q < ps.size(); ++q) {
        if (hres[ps[q].src].n_kept == 0) continue;
        const int a = P.chain_view[(size_t)ps[q].view], b = ps_alias_view[q];
        touch_src[(size_t)a].push_back((int)q);
        if (b >= 0 && b != a) touch_src[(size_t)b].push_back((int)q);
    }
    // key slots per block: 2^28 (6.4 GB of transients at 24 B per slot) -- up to 2^30 when a quarter of the free HBM allows it: every block
    // re-reads the records of all views that touch it (its views and their neighbours), so a dense scene wants few, big blocks
    long long budget = 1ll << 28;
    if (c->opt.prod_block_keys > 0) budget = c->opt.prod_block_keys;
    else if (slots_all > budget) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) budget = std::max(budget, std::min<long long>((long long)(fr / 4 / 24), (1ll << 30) - 64));
    }
    std::vector<ProdBlock> blocks;
    std::vector<long long> off_tab;                         // per block: out_off of every chain view, then of every early pair
    {
        std::vector<int> mark_v((size_t)n_views, -1), mark_s(ps.size(), -1);
        int x = dv0;
        while (x < dv1) {
            const int bi = (int)blocks.size();
            ProdBlock B;
            B.d0 = P.seg_base[(size_t)x]; B.slots = 0;
            B.off_view = off_tab.size();
            off_tab.resize(off_tab.size() + (size_t)n_views + ps.size(), -1);
            B.off_src = B.off_view + (size_t)n_views;
            int x1 = x;
            for (; x1 < dv1; ++x1) {
                long long add = 0;
                for (int k : touch_view[(size_t)x1]) if (mark_v[(size_t)k] != bi) add += 2 * (long long)hres[k].n_kept;
                for (int q : touch_src[(size_t)x1]) if (mark_s[(size_t)q] != bi) add += 2 * (long long)hres[ps[(size_t)q].src].n_kept;
                if (x1 > x && B.slots + add > budget) break;
                for (int k : touch_view[(size_t)x1]) if (mark_v[(size_t)k] != bi) { mark_v[(size_t)k] = bi; off_tab[B.off_view + (size_t)k] = B.slots; B.slots += 2 * (long long)hres[k].n_kept; }
                for (int q : touch_src[(size_t)x1]) if (mark_s[(size_t)q] != bi) { mark_s[(size_t)q] = bi; off_tab[B.off_src + (size_t)q] = B.slots; B.slots += 2 * (long long)hres[ps[(size_t)q].src].n_kept; }
            }
            B.d1 = P.seg_base[(size_t)x1];
            if (B.slots > 0x7ffffff0ll) return fail(c, L3D_ERR_UNSUPPORTED, "products: one view and its neighbours hold more than 2^30 kept matches");
            blocks.push_back(B);
            x = x1;
        }
    }
    long long max_slots = 0;
    for (const ProdBlock& B : blocks) max_slots = std::max(max_slots, B.slots);
    const long long n_keys_max = max_slots + 1;            // (+ the sentinel)

    // ---- upload tables: [ProdView n_views][ProdSrc][ids nv][seg_base nv+1][chain view ids][out_off tables of all blocks]
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_pv = 0, o_ps = o_pv + al(pv.size() * sizeof(ProdView)), o_ids = o_ps + al(ps.size() * sizeof(ProdSrc) + 16),
                 o_sb = o_ids + al((size_t)nv * 4), o_cv = o_sb + al((size_t)(nv + 1) * 4), o_off = o_cv + al((size_t)n_views * 4), tab_total = o_off + al(off_tab.size() * 8 + 16);
    HIPCHK(c, P.tables.reserve(tab_total));
    char* tb = P.tables.as<char>();
    HIPCHK(c, hipMemcpyAsync(tb + o_pv, pv.data(), pv.size() * sizeof(ProdView), hipMemcpyHostToDevice, st));
    if (!ps.empty()) HIPCHK(c, hipMemcpyAsync(tb + o_ps, ps.data(), ps.size() * sizeof(ProdSrc), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(tb + o_ids, P.view_ids.data(), (size_t)nv * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(tb + o_sb, P.seg_base.data(), (size_t)(nv + 1) * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(tb + o_cv, P.chain_view_id.data(), (size_t)n_views * 4, hipMemcpyHostToDevice, st));
    if (!off_tab.empty()) HIPCHK(c, hipMemcpyAsync(tb + o_off, off_tab.data(), off_tab.size() * 8, hipMemcpyHostToDevice, st));
    const ProdView* dpv = reinterpret_cast<const ProdView*>(tb + o_pv);
    const ProdSrc* dps = reinterpret_cast<const ProdSrc*>(tb + o_ps);
    const unsigned* dids = reinterpret_cast<const unsigned*>(tb + o_ids);
    const int* dsb = reinterpret_cast<const int*>(tb + o_sb);
    const unsigned* dcv = reinterpret_cast<const unsigned*>(tb + o_cv);
    const long long* doff = reinterpret_cast<const long long*>(tb + o_off);

    // ---- best references, medians (once); per block: keys, sort, unique, CSR rows
    const double t_res0 = now_s();
    HIPCHK(c, P.keys.reserve((size_t)n_keys_max * 8 + 64));
    HIPCHK(c, P.keys2.reserve((size_t)n_keys_max * 8 + 64));
    HIPCHK(c, P.flag.reserve(((size_t)n_keys_max + 2) * 4));
    HIPCHK(c, P.pos.reserve(((size_t)n_keys_max + 2) * 4));
    HIPCHK(c, P.pot_start.reserve(((size_t)nd + 2) * 8));
    HIPCHK(c, P.pot_tgt.reserve(((size_t)slots_all + 2) * 4));
    HIPCHK(c, P.best_ref.reserve((size_t)nd * 8 + 64));
    HIPCHK(c, P.median.reserve((size_t)n_views * 8 + 64));          // medians | list lengths of the early-return views
    if (c->opt.timing) fprintf(stderr, "[l3d products] %zu block(s) of up to %lld keys, table of up to %lld entries: buffers reserved in %.2f ms\n", blocks.size(), n_keys_max, slots_all, (now_s() - t_res0) * 1e3);
    int* d_list_len = reinterpret_cast<int*>(P.median.as<float>() + n_views);
    HIPCHK(c, hipMemsetAsync(d_list_len, 0, (size_t)n_views * 4, st));
    unsigned long long* keys = P.keys.as<unsigned long long>();
    unsigned long long* keys2 = P.keys2.as<unsigned long long>();
    const Match* arena = c->ch_kept.as<Match>();
    HIPCHK(c, hipMemsetAsync(P.best_ref.p, 0xff, (size_t)nd * 8, st));
    const unsigned gx = (unsigned)std::max(1, std::min(512, (max_kept + 1023) / 1024));
    {
        ProfScope p(c, "prod_keys", st);
        if (!ps.empty()) hipLaunchKernelGGL(k_prod_keys_early, dim3(gx, (unsigned)ps.size()), dim3(256), 0, st, arena, dpv, dps, (const long long*)nullptr, dcv, nb, 0, 0,
                                            keys, P.best_ref.as<unsigned long long>(), d_list_len);
        hipLaunchKernelGGL(k_prod_best, dim3((maxS + 255) / 256, n_views), dim3(256), 0, st, dpv, P.best_ref.as<long long>());
        hipLaunchKernelGGL(k_prod_median, dim3(n_views), dim3(256), 0, st, dpv, P.median.as<float>());
    }
    size_t tb1 = 0, tb2 = 0;
    HIPCHK(c, sort_keys_u64(nullptr, tb1, keys, keys2, (int)n_keys_max, 0, std::min(64, 2 * nb), st));
    HIPCHK(c, exclusive_sum_int(nullptr, tb2, P.flag.as<int>(), P.pos.as<int>(), (int)n_keys_max + 1, st));
    HIPCHK(c, P.tmp.reserve(std::max(tb1, tb2) + 256));
    long long base = 0;
    for (size_t bi = 0; bi < blocks.size(); ++bi) {
        const ProdBlock& B = blocks[bi];
        const long long n_keys = B.slots + 1;
        {
            ProfScope p(c, "prod_keys", st);
            HIPCHK(c, hipMemsetAsync(keys + B.slots, 0xff, 8, st));          // the sentinel
            if (max_kept > 0) hipLaunchKernelGGL(k_prod_keys, dim3(gx, n_views), dim3(256), 0, st, arena, dpv, doff + B.off_view, dids, dsb, nv, nb, B.d0, B.d1, keys);
            if (!ps.empty()) hipLaunchKernelGGL(k_prod_keys_early, dim3(gx, (unsigned)ps.size()), dim3(256), 0, st, arena, dpv, dps, doff + B.off_src, dcv, nb, B.d0, B.d1,
                                                keys, P.best_ref.as<unsigned long long>(), d_list_len);
        }
        {
            ProfScope p(c, "prod_sort", st);
            size_t t1 = tb1, t2 = tb2;
            HIPCHK(c, sort_keys_u64(P.tmp.p, t1, keys, keys2, (int)n_keys, 0, std::min(64, 2 * nb), st));
            const unsigned nblk = (unsigned)((n_keys + 255) / 256);
            HIPCHK(c, hipMemsetAsync(P.flag.as<int>() + n_keys, 0, 4, st));
            hipLaunchKernelGGL(k_prod_flags, dim3(nblk), dim3(256), 0, st, keys2, n_keys, P.flag.as<int>());
            HIPCHK(c, exclusive_sum_int(P.tmp.p, t2, P.flag.as<int>(), P.pos.as<int>(), (int)n_keys + 1, st));
            hipLaunchKernelGGL(k_prod_csr, dim3(nblk), dim3(256), 0, st, keys2, P.flag.as<int>(), P.pos.as<int>(), n_keys, nb, B.d0, B.d1, base, P.pot_start.as<long long>(), P.pot_tgt.as<int>());
        }
        if (bi + 1 < blocks.size()) {                      // the next block's rows start behind this block's entries
            int n_unique = 0;
            HIPCHK(c, hipMemcpyAsync(&n_unique, P.pos.as<int>() + n_keys, 4, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            base += n_unique;
        }
    }
    const long long last_keys = blocks.empty() ? 0 : blocks.back().slots + 1;
    // ---- the scalars the host needs
    int n_last = 0;
    std::vector<float> med((size_t)2 * n_views, 1.0f);
    if (!blocks.empty()) HIPCHK(c, hipMemcpyAsync(&n_last, P.pos.as<int>() + last_keys, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(med.data(), P.median.p, (size_t)n_views * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(c, L3D_ERR_HIP, std::string("products: ") + hipGetErrorString(e_)); }
    const long long n_pot = base + n_last;
    P.n_pot = n_pot;
    if (held) {
        const long long r0 = blocks.empty() ? (long long)nd + 1 : (long long)P.seg_base[(size_t)dv0], r1 = blocks.empty() ? (long long)nd + 1 : (long long)P.seg_base[(size_t)dv1] + 1;
        if (r0 > 0) hipLaunchKernelGGL(k_prod_fill_rows, dim3((unsigned)((std::min<long long>(r0, nd + 1) + 255) / 256)), dim3(256), 0, st, P.pot_start.as<long long>(), 0ll, std::min<long long>(r0, nd + 1), 0ll);
        if (r1 <= nd) hipLaunchKernelGGL(k_prod_fill_rows, dim3((unsigned)(((long long)nd + 1 - r1 + 255) / 256)), dim3(256), 0, st, P.pot_start.as<long long>(), r1, (long long)nd + 1, n_pot);
        HIPCHK(c, hipStreamSynchronize(st));
        if (c->opt.part_release != 0) { P.keys.release(); P.keys2.release(); P.flag.release(); P.pos.release(); P.tmp.release(); }     // (the key blocks: transients)
    }
    for (int k = 0; k < n_views; ++k) {
        l3d_chain_summary& s = summary[k];
        s.verified = pvh[k].verified; s.n_kept = pvh[k].verified ? hres[k].n_kept : reinterpret_cast<const int*>(med.data() + n_views)[k];
        s.n_candidates = pvh[k].verified ? hres[k].R : 0;
        s.median_depth = med[(size_t)k]; s.pad = 0;
    }
    if (n_pot_out) *n_pot_out = n_pot;
    P.valid = !partial;
    return L3D_OK;
}

// =================================================================================================================================
// greedy selection on the resident products
namespace l3d {

struct ViewGeo {
    double RtKinv[9], C[3];
    float k_lower, k_upper, median_depth;
    int pad;
    const float4* segs;
};

__global__ __launch_bounds__(256) void k_hyp_flag(const long long* __restrict__ best_ref, int nd, int* __restrict__ flag)
{
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d < nd) flag[d] = best_ref[d] >= 0 ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_hyp_build(const Match* __restrict__ arena, const long long* __restrict__ best_ref, const int* __restrict__ hyp_of,
                                                   const int* __restrict__ seg_base, int nv, const ViewGeo* __restrict__ geo, int nd,
                                                   Hypothesis* __restrict__ hyp, float* __restrict__ score, int* __restrict__ hyp_dense, int* __restrict__ best_hyp,
                                                   int* __restrict__ view_hyp_begin)
{
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d <= nv) view_hyp_begin[d] = hyp_of[seg_base[d]];          // (d doubles as a view index for the first nv + 1 threads)
    if (d >= nd) return;
    const long long ref = best_ref[d];
    if (ref < 0) { best_hyp[d] = -1; return; }
    int lo = 0, hi = nv;                                          // view of the dense id: last view whose base is <= d
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_base[mid] <= d) lo = mid; else hi = mid; }
    const ViewGeo g = geo[lo];
    const Match r = arena[ref & kBestIndexMask];
    const bool rev = (ref & kBestReversed) != 0;
    const float d1 = rev ? r.depths[2] : r.depths[0], d2 = rev ? r.depths[3] : r.depths[1];
    const float conf = rev ? 0.0f : r.confidence;
    const float4 sg = g.segs[d - seg_base[lo]];
    la::M3 M;
    for (int i = 0; i < 9; ++i) M.m[i] = g.RtKinv[i];
    la::V3 P1, P2, dir;
    unproject_segment_f64(M, la::V3{ g.C[0], g.C[1], g.C[2] }, sg.x, sg.y, sg.z, sg.w, d1, d2, P1, P2, dir);
    const int h = hyp_of[d];
    Hypothesis o;
    o.P1[0] = P1.x; o.P1[1] = P1.y; o.P1[2] = P1.z; o.P2[0] = P2.x; o.P2[1] = P2.y; o.P2[2] = P2.z; o.dir[0] = dir.x; o.dir[1] = dir.y; o.dir[2] = dir.z;
    o.depth_p1 = d1; o.depth_p2 = d2; o.k_lower = g.k_lower; o.k_upper = g.k_upper; o.median_depth = g.median_depth; o.pad = 0;
    hyp[h] = o;
    score[h] = fminf(conf, 1.0f);                                 // line3D.cc:927
    hyp_dense[h] = d;
    best_hyp[d] = h;
}

// a view's best matches as records (inspection): the arena record, reversed where the reference read it reversed
__global__ __launch_bounds__(256) void k_prod_best_records(const Match* __restrict__ arena, const long long* __restrict__ best_ref, const int* __restrict__ seg_base, int nv,
                                                           const unsigned* __restrict__ early_cam, int nd, Match* __restrict__ out)
{
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= nd) return;
    const long long ref = best_ref[d];
    Match o;
    o.segID1 = 0xffffffffu; o.camID2 = 0; o.segID2 = 0; o.depths[0] = o.depths[1] = o.depths[2] = o.depths[3] = 0.0f; o.confidence = 0.0f;
    if (ref >= 0) {
        int lo = 0, hi = nv;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_base[mid] <= d) lo = mid; else hi = mid; }
        const Match r = arena[ref & kBestIndexMask];
        if (ref & kBestReversed) {
            o.segID1 = r.segID2; o.segID2 = r.segID1; o.camID2 = early_cam ? early_cam[d] : 0;
            o.depths[0] = r.depths[2]; o.depths[1] = r.depths[3]; o.depths[2] = r.depths[0]; o.depths[3] = r.depths[1];
        } else o = r;
    }
    out[d] = o;
}

}  // namespace l3d

extern "C" {

int l3d_products_hypotheses(l3d_ctx* c, const l3d_view_geometry* geometry, int n_views, int32_t* view_hyp_begin, int32_t** hyp_dense_out, int* n_hyp_out)
{
    if (!c) return L3D_ERR_INVALID;
    Products& P = c->products;
    if (!geometry || !view_hyp_begin || !hyp_dense_out || !n_hyp_out) return fail(c, L3D_ERR_INVALID, "l3d_products_hypotheses: bad argument");
    *hyp_dense_out = nullptr; *n_hyp_out = 0;
    if (!P.valid) return fail(c, L3D_ERR_INVALID, "l3d_products_hypotheses: no resident products (run l3d_match_chain_resident first)");
    if (n_views != P.n_views_all) return fail(c, L3D_ERR_INVALID, "l3d_products_hypotheses: geometry does not match the dense map");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int nd = P.n_dense, nv = n_views;
    std::vector<ViewGeo> g((size_t)nv);
    for (int i = 0; i < nv; ++i) {
        const l3d_view_geometry& s = geometry[i];
        const int S = P.seg_base[(size_t)i + 1] - P.seg_base[(size_t)i];
        if (s.n_segments != S) return fail(c, L3D_ERR_INVALID, "l3d_products_hypotheses: segment count differs from the dense map");
        ViewGeo& o = g[(size_t)i];
        memcpy(o.RtKinv, s.RtKinv, 72); memcpy(o.C, s.C, 24);
        o.k_lower = s.k_lower; o.k_upper = s.k_upper; o.median_depth = s.median_depth; o.pad = 0;
        o.segs = reinterpret_cast<const float4*>(resident_ptr(c, s.segments, (size_t)S * 16));
        if (!o.segs && S > 0) return fail(c, L3D_ERR_INVALID, "l3d_products_hypotheses: a view's segments are not registered (l3d_register_segments)");
    }
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_geo = 0, o_vhb = al((size_t)nv * sizeof(ViewGeo)), o_sb = o_vhb + al((size_t)(nv + 2) * 4);
    HIPCHK(c, P.geo.reserve(o_sb + al((size_t)(nv + 1) * 4)));
    char* gb = P.geo.as<char>();
    HIPCHK(c, hipMemcpyAsync(gb + o_geo, g.data(), (size_t)nv * sizeof(ViewGeo), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(gb + o_sb, P.seg_base.data(), (size_t)(nv + 1) * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, P.hyp_of.reserve(((size_t)nd + 2) * 8));                 // flags | exclusive sums
    int* flag = P.hyp_of.as<int>();
    int* hyp_of = flag + (nd + 2);
    HIPCHK(c, hipMemsetAsync(flag + nd, 0, 4, st));
    ProfScope p(c, "hypotheses", st);
    hipLaunchKernelGGL(k_hyp_flag, dim3((nd + 255) / 256), dim3(256), 0, st, P.best_ref.as<long long>(), nd, flag);
    size_t tb = 0;
    HIPCHK(c, exclusive_sum_int(nullptr, tb, flag, hyp_of, nd + 1, st));
    HIPCHK(c, P.tmp.reserve(tb + 256));
    HIPCHK(c, exclusive_sum_int(P.tmp.p, tb, flag, hyp_of, nd + 1, st));
    // (the number of hypotheses is at most the number of segments: the tables are sized for that, no round trip before the build)
    HIPCHK(c, c->aff_hyp.reserve((size_t)nd * sizeof(Hypothesis) + 64));
    HIPCHK(c, P.score.reserve((size_t)nd * 4 + 64));
    HIPCHK(c, P.hyp_dense.reserve((size_t)nd * 4 + 64));
    HIPCHK(c, P.best_hyp.reserve((size_t)nd * 4 + 64));
    hipLaunchKernelGGL(k_hyp_build, dim3((std::max(nd, nv + 1) + 255) / 256), dim3(256), 0, st, c->ch_kept.as<Match>(), P.best_ref.as<long long>(), hyp_of,
                       reinterpret_cast<const int*>(gb + o_sb), nv, reinterpret_cast<const ViewGeo*>(gb + o_geo), nd, c->aff_hyp.as<Hypothesis>(), P.score.as<float>(),
                       P.hyp_dense.as<int>(), P.best_hyp.as<int>(), reinterpret_cast<int*>(gb + o_vhb));
    HIPCHK(c, hipMemcpyAsync(view_hyp_begin, gb + o_vhb, (size_t)(nv + 1) * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const int nh = view_hyp_begin[nv];
    int32_t* hd = static_cast<int32_t*>(malloc((size_t)nh * 4 + 4));
    if (!hd) return fail(c, L3D_ERR_NOMEM, "malloc");
    if (nh > 0) {
        hipError_t e = hipMemcpyAsync(hd, P.hyp_dense.p, (size_t)nh * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { free(hd); return fail(c, L3D_ERR_HIP, std::string("l3d_products_hypotheses: ") + hipGetErrorString(e)); }
    }
    { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { free(hd); return fail(c, L3D_ERR_HIP, std::string("l3d_products_hypotheses: ") + hipGetErrorString(e_)); } }
    P.view_hyp_begin.assign(view_hyp_begin, view_hyp_begin + nv + 1);
    P.n_hyp = nh; P.hyp_valid = true;
    c->resident_hyp = nh;
    *hyp_dense_out = hd; *n_hyp_out = nh;
    return L3D_OK;
}

int l3d_products_hypotheses_get(l3d_ctx* c, l3d_hypothesis* hyp, float* score)
{
    if (!c) return L3D_ERR_INVALID;
    Products& P = c->products;
    if (!P.valid || !P.hyp_valid) return fail(c, L3D_ERR_INVALID, "no resident hypotheses");
    HIPCHK(c, hipSetDevice(c->device));
    if (hyp && P.n_hyp) HIPCHK(c, hipMemcpyAsync(hyp, c->aff_hyp.p, (size_t)P.n_hyp * sizeof(Hypothesis), hipMemcpyDeviceToHost, c->stream));
    if (score && P.n_hyp) HIPCHK(c, hipMemcpyAsync(score, P.score.p, (size_t)P.n_hyp * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return L3D_OK;
}

int l3d_chain_kept_list(l3d_ctx* c, int index, l3d_match** out, int* n)
{
    if (!c) return L3D_ERR_INVALID;
    Products& P = c->products;
    if (!out || !n) return fail(c, L3D_ERR_INVALID, "bad argument");
    *out = nullptr; *n = 0;
    if (!P.valid || index < 0 || index >= P.n_chain) return fail(c, L3D_ERR_INVALID, "l3d_chain_kept_list: no such view in the resident products");
    HIPCHK(c, hipSetDevice(c->device));
    auto fetch = [&](int k, std::vector<l3d_match>& v) -> int {
        const ChainResult& r = P.res[(size_t)k];
        v.resize((size_t)r.n_kept);
        if (r.n_kept) HIPCHK(c, hipMemcpy(v.data(), c->ch_kept.as<Match>() + r.kept_base, (size_t)r.n_kept * sizeof(Match), hipMemcpyDeviceToHost));
        return L3D_OK;
    };
    std::vector<l3d_match> lst;
    if (P.chain_verified[(size_t)index]) { if (int rc = fetch(index, lst)) return rc; }
    if (!P.chain_verified[(size_t)index]) {
        // cudawrapper.cu:877-878: the localized existing list -- what the earlier views pushed (line3D.cc:838-872), in push order
        lst.clear();
        std::vector<l3d_match> src;
        const unsigned vid = P.chain_view_id[(size_t)index];
        for (size_t q = 0; q < P.early_src_index[(size_t)index].size(); ++q) {
            if (int rc = fetch(P.early_src_index[(size_t)index][q], src)) return rc;
            for (const l3d_match& mp : src) {
                if (mp.camID2 != vid) continue;
                l3d_match r;
                r.segID1 = mp.segID2; r.segID2 = mp.segID1; r.confidence = 0.0f; r.camID2 = (uint32_t)P.early_src_cam[(size_t)index][q];
                r.depths[0] = mp.depths[2]; r.depths[1] = mp.depths[3]; r.depths[2] = mp.depths[0]; r.depths[3] = mp.depths[1];
                lst.push_back(r);
            }
        }
    }
    l3d_match* o = static_cast<l3d_match*>(malloc(lst.size() * sizeof(l3d_match) + 32));
    if (!o) return fail(c, L3D_ERR_NOMEM, "malloc");
    if (!lst.empty()) memcpy(o, lst.data(), lst.size() * sizeof(l3d_match));
    *out = o; *n = (int)lst.size();
    return L3D_OK;
}

int l3d_chain_products_get(l3d_ctx* c, int64_t* pot_start, int32_t* pot_tgt, l3d_match* best_match)
{
    if (!c) return L3D_ERR_INVALID;
    Products& P = c->products;
    if (!P.valid) return fail(c, L3D_ERR_INVALID, "no resident products");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    if (pot_start) HIPCHK(c, hipMemcpyAsync(pot_start, P.pot_start.p, ((size_t)P.n_dense + 1) * 8, hipMemcpyDeviceToHost, st));
    if (pot_tgt && P.n_pot) HIPCHK(c, hipMemcpyAsync(pot_tgt, P.pot_tgt.p, (size_t)P.n_pot * 4, hipMemcpyDeviceToHost, st));
    if (best_match) {
        // the LOCAL camera number of an early-return view's entry: the rank of its source, per dense id
        std::vector<unsigned> cam((size_t)P.n_dense, 0u);
        bool any_early = false;
        std::vector<long long> ref((size_t)P.n_dense);
        HIPCHK(c, hipMemcpyAsync(ref.data(), P.best_ref.p, (size_t)P.n_dense * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        for (int k = 0; k < P.n_chain; ++k) {
            if (P.chain_verified[(size_t)k]) continue;
            any_early = true;
            const int vi = P.chain_view[(size_t)k];
            // which source a reversed reference points into: the one whose slice holds the record
            for (int d = P.seg_base[(size_t)vi]; d < P.seg_base[(size_t)vi + 1]; ++d) {
                if (ref[(size_t)d] < 0) continue;
                const long long idx = ref[(size_t)d] & kBestIndexMask;
                for (size_t q = 0; q < P.early_src_index[(size_t)k].size(); ++q) {
                    const ChainResult& r = P.res[(size_t)P.early_src_index[(size_t)k][q]];
                    if (idx >= r.kept_base && idx < (long long)r.kept_base + r.n_kept) { cam[(size_t)d] = (unsigned)P.early_src_cam[(size_t)k][q]; break; }
                }
            }
        }
        HIPCHK(c, P.keys.reserve((size_t)P.n_dense * (sizeof(Match) + 4) + 256));       // (scratch: the keys are consumed)
        Match* dout = P.keys.as<Match>();
        unsigned* dcam = reinterpret_cast<unsigned*>(P.keys.as<char>() + (((size_t)P.n_dense * sizeof(Match) + 255) & ~(size_t)255));
        if (any_early) HIPCHK(c, hipMemcpyAsync(dcam, cam.data(), (size_t)P.n_dense * 4, hipMemcpyHostToDevice, st));
        HIPCHK(c, P.flag.reserve(((size_t)P.n_views_all + 2) * 4 + 256));      // (scratch: the flags are consumed)
        int* dsb = P.flag.as<int>();
        HIPCHK(c, hipMemcpyAsync(dsb, P.seg_base.data(), ((size_t)P.n_views_all + 1) * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_prod_best_records, dim3((P.n_dense + 255) / 256), dim3(256), 0, st, c->ch_kept.as<Match>(), P.best_ref.as<long long>(), dsb, P.n_views_all,
                           any_early ? dcam : nullptr, P.n_dense, dout);
        HIPCHK(c, hipMemcpyAsync(best_match, dout, (size_t)P.n_dense * sizeof(Match), hipMemcpyDeviceToHost, st));
    }
    HIPCHK(c, hipStreamSynchronize(st));
    return L3D_OK;
}

}  // extern "C"

void l3d::warm_products() { touch_kernel(reinterpret_cast<const void*>(&k_prod_flags)); }
