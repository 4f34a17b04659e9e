// l3d_chain_common.hpp -- what the single-GPU resident chain (l3d_chain.hip) and the sharded one (l3d_chain_sharded.hip) share: the
// validation of the caller's view descriptions, the layout and upload of the static tables, the per-view slices of the whole-run
// arenas, the argument blocks of the kernels and the sizing rules (first capacity guesses, LDS image of the window kernel).  One copy
// of every rule; the two files differ in what happens BETWEEN the views (an arena slice vs. a slot and an exchange).
#pragma once

#include <vector>

#include "l3d_ctx.hpp"

namespace l3d {

struct ChainViewDev {               // device addresses of one view's static tables and its slices of the whole-run arenas
    const float4 *src = nullptr, *tgt = nullptr;
    size_t o_off = 0, o_F = 0, o_R = 0, o_C = 0, o_P = 0, o_Rs = 0, o_Cs = 0, o_tbm = 0, o_l2g = 0, o_sc = 0, o_si = 0, o_ss = 0;   // offsets into the table block
    unsigned long long* mask = nullptr;
    int* rowcnt = nullptr;
    int* rowA = nullptr;            // row starts of the stage-1 candidates alone (S*N + 1)
    int* rowub = nullptr;           // fused row starts: k_pair_mask's (upper-bound) counts, S*N, and their 256-row block sums (never rewritten)
    int* rowblk = nullptr;
    int* stats = nullptr;           // {raw total, raw max per segment}
    float2* best = nullptr;
    int* bestpos = nullptr;         // per segment: position (in the view's kept slice) of its best kept match or -1 (k_kept_write_chain)
    int* rt = nullptr;              // run table of the view's kept list, (N + 1) x S (l3d_runtable.hpp; single-GPU chain with run tables)
    float4* rays = nullptr;         // unit viewing rays of the target endpoints (2 per target entry), k_tgt_rays
    float4* src_rays = nullptr;     // ... of the view's own end points (2 per source segment)
    int W64 = 0, maxW = 0;
    int s0 = 0, s1 = 0;             // this rank's source-segment range ([0, S) in the single-GPU chain)
    bool verified = false;
};

struct ChainLayout {
    size_t tab_bytes = 0, mask_bytes = 0, max_mask_bytes = 0, rowcnt_ints = 0, best_elems = 0;
    size_t rowA_ints = 0, rowub_ints = 0, rowblk_ints = 0;
    int maxS = 0, maxN = 0;
    double pairs = 0, max_pairs = 0;        // stage-1 pairs of this rank's ranges: all views / the largest view
    const unsigned char* dtab = nullptr;    // the table block on the device
};

// Validates the views, makes their segment arrays resident, lays out the table block (rank, world: the source-segment ranges).
int chain_plan_views(l3d_ctx* c, const l3d_chain_view* views, int n_views, int rank, int world, std::vector<ChainViewDev>& vd, ChainLayout& L, const char* what);
// Packs the tables into the pinned block, uploads them, fills the target-ray table (one launch) -- all on `st`.
int chain_upload_tables(l3d_ctx* c, const l3d_chain_view* views, int n_views, std::vector<ChainViewDev>& vd, ChainLayout& L, bool with_rays, hipStream_t st);
// Reserves the whole-run arenas (bit rows, row counts, row starts [+ upper-bound counts and block sums], best depths [+ positions]) and
// hands every view its slices; zeroes what the kernels add into.
// mask_ring: slots of the bit-row arena (a view's bit rows live from its k_pair_mask to its k_pair_fill, both on the stage-1 stream in order:
// two slots instead of one slice per view -- 24.6 MB x 2048 views = 50 GB at 4000 segments x 24 neighbours); 0: one slice per view
// (the A/B mode that triangulates on the chain's stream, views later).
int chain_assign_arenas(l3d_ctx* c, const l3d_chain_view* views, int n_views, std::vector<ChainViewDev>& vd, ChainLayout& L, bool fused_rows, bool best_positions, int mask_ring, hipStream_t st, bool run_tables = false);
// Per-launch scratch that depends on the candidate capacity (candidate store, window scratch, stage-1 ring of `ring` slots).
int chain_reserve_candidates(l3d_ctx* c, const ChainLayout& L, size_t cand_cap, int ring);

PairArgs chain_pair_args(const l3d_ctx* c, const l3d_chain_view& v, const ChainViewDev& d, const unsigned char* dtab);
// everything of VerifyArgs that does not depend on the chain flavour (candidate arrays of the context, tables, range, parameters)
VerifyArgs chain_verify_args(l3d_ctx* c, const l3d_chain_view& v, const ChainViewDev& d, const unsigned char* dtab, size_t cand_cap);
// the window kernel's launch on those arguments (LDS image from the raw maximum per segment, or the largest the budget allows), or the
// all-pairs kernel + per-segment epilogue beyond ~50 neighbours / in all-pairs mode
void chain_launch_verify(l3d_ctx* c, VerifyArgs& va, const ChainViewDev& d, const int* exist_cams, int n_exist_cams, int raw_max_per_segment, size_t cand_cap, hipStream_t st);

// first guess of the candidate capacity from the largest view's pair count (raw density ~6 % + reverse matches; guarded on the device)
inline size_t chain_first_cand_cap(double max_pairs) { return (size_t)(max_pairs * 0.12) + 65536; }
inline size_t chain_align16(size_t x) { return (x + 15) & ~(size_t)15; }

}  // namespace l3d
