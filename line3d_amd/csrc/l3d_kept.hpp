// l3d_kept.hpp -- ordered compaction of one source segment's kept matches (conf > 1, cudawrapper.cu:1089-1110) by one
// wave: the confidences of up to 1024 candidates are fetched in ONE round of loads (16 per lane in flight) before the
// ballots, instead of one dependent load -> ballot -> store round per 64 candidates; the record loads of the few kept
// candidates (~1.6 %) follow.  Shared by the per-view, chain and sharded-chain writers.
#pragma once

#include "l3d_kernels.hpp"

namespace l3d {

__device__ __forceinline__ void write_kept_segment(const VerifyArgs& a, int y, int lane, int o, const unsigned* __restrict__ local2global,
                                                   Match* __restrict__ out)
{
    constexpr int kPre = 16;
    const int start = a.row_start[y * a.N];
    const int m = a.row_start[(y + 1) * a.N] - start;
    auto emit = [&](int i, float c, int pos) {
        const uint2 meta = a.cand_meta[start + i];
        const float4 d = a.cand_depths[start + i];
        Match r;
        r.segID1 = (unsigned)y; r.camID2 = local2global[meta.y]; r.segID2 = meta.x;
        r.depths[0] = d.x; r.depths[1] = d.y; r.depths[2] = d.z; r.depths[3] = d.w;
        r.confidence = c / 2.0f;                     // confidence_norm, cudawrapper.cu:1089,1098
        out[pos] = r;
    };
    float c[kPre];
#pragma unroll
    for (int r = 0; r < kPre; ++r) { const int i = r * 64 + lane; c[r] = i < m ? a.cand_conf[start + i] : 0.0f; }
#pragma unroll
    for (int r = 0; r < kPre; ++r) {
        if (r * 64 < m) {
            const bool k = c[r] > 1.0f;
            const unsigned long long b = __ballot(k);
            if (k) emit(r * 64 + lane, c[r], o + __popcll(b & ((1ull << lane) - 1ull)));
            o += __popcll(b);
        }
    }
    for (int i0 = kPre * 64; i0 < m; i0 += 64) {
        const int i = i0 + lane;
        const float cc = i < m ? a.cand_conf[start + i] : 0.0f;
        const bool k = cc > 1.0f;
        const unsigned long long b = __ballot(k);
        if (k) emit(i, cc, o + __popcll(b & ((1ull << lane) - 1ull)));
        o += __popcll(b);
    }
}

}  // namespace l3d
