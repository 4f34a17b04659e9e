// l3d_products.hpp -- what Line3D::performMatching leaves behind (line3D.cc:834-884), kept on the device: shared by the resident
// chain (l3d_chain.hip), the builder (l3d_products.hip) and the affinity fill on resident tables (l3d_affinity.hip).
#pragma once

#include "l3d_ctx.hpp"

namespace l3d {

// one view of a finished chain, as the builder needs it
struct ProdChainView {
    const float2* best;     // depths of the best hypothesis per segment (k_verify_window epilogue), null for views that were not verified
    const int* bestpos;     // position (in the view's kept slice) of every segment's best kept match or -1
    int verified;
    const int* rt;          // the view's run table (l3d_runtable.hpp) when the chain's kept writer filled one, else null (rebuilt by the products)
};

// The transposes of the (view, camera) pairs done by the chain itself, view by view on a side stream (l3d_chain.hip; round 6): what the end of matchViews finds ready.
// Canonical places: pcnt / poff at [chain view * maxN + camera] (poff: the pair's first entry in E, which is aligned with the kept arena); boff_off[chain view][camera]:
// where the pair's column starts lie in boff (host copy: the builder files it in its pairs).
struct ProdEarly {
    const int* pcnt_kq = nullptr;
    const unsigned* poff_kq = nullptr;
    const int* boff = nullptr;
    const unsigned* E = nullptr;
    const int* boff_off_host = nullptr;     // n_views x maxN
    int maxN = 0;
};
// Enqueued on the context's stream behind the last view of the chain; returns after the few scalars the host needs (entries of the
// CSR, medians) have arrived.  hres: the chain's per-view result records (host copies).
// dv0, dv1 (dv1 >= 0): only the rows of the dense views [dv0, dv1), numbered from 0 -- pot_start[seg_base[dv0] .. seg_base[dv1]] and that many
// entries of pot_tgt are valid afterwards, P.valid stays false (the caller assembles the pieces: l3d_match_chain_blocks)
// held (partitioned products: l3d_match_chain_partition): per chain view, whether its records are on this rank; the rows outside [dv0, dv1) are
// then left EMPTY but well-formed (the whole pot_start array is valid)
int build_products(l3d_ctx* c, const l3d_chain_view* views, int n_views, const ProdChainView* pv, const ChainResult* hres,
                   const l3d_dense_map* map, l3d_chain_summary* summary, int64_t* n_pot, int dv0 = 0, int dv1 = -1, const char* held = nullptr,
                   const unsigned* qt_arena = nullptr, const ProdEarly* early = nullptr);
// qt_arena: the side array of the kept arena, (local camera << 16 | target) per record, when the chain's kept writer filled it together with the
// views' run tables (ProdChainView::rt); null: the products rebuild both from the records, block by block

// the chain's views as the early transposes see them (device array, one per chain view; rt == null: nothing to transpose): run table, per local camera the target
// view's segment count (0: none) and the place of its column starts in boff
struct EarlyView { const int* rt; const int* St; const int* boff_off; int k, S, N, pad; };
// the pairs of the chain views [k0, k0 + nb): counts, then transposes, on `st` (behind the kept writer of the last of them)
void launch_early_transposes(l3d_ctx* c, const EarlyView* views_dev, int k0, int nb, int maxN, int maxSt, const ChainResult* res,
                             const unsigned* qt_arena, int* pcnt_kq, unsigned* poff_kq, int* boff, unsigned* E, unsigned* T, double avg_run, hipStream_t st);

// pot_start[first_row + i] = piece[i] + base for n_rows rows (a rank's rows of the table put in place)
void launch_prod_shift_rows(const long long* piece, long long n_rows, long long base, long long* pot_start_at, hipStream_t st);

// references into the kept arena (best_ref): record index, bit 62 = "read it reversed" (the record of an earlier view that an
// early-return view got back, cudawrapper.cu:877-878: segments and depth pairs swap, confidence 0), -1 = the segment has no match
constexpr long long kBestReversed = 1ll << 62;
constexpr long long kBestIndexMask = (1ll << 40) - 1;

}  // namespace l3d
