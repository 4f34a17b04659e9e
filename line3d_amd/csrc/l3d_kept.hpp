// l3d_kept.hpp -- ordered compaction of one source segment's kept matches (conf > 1, cudawrapper.cu:1089-1110): the
// confidences of up to 2048 candidates are fetched in ONE round of loads by the 4 waves of a workgroup before the ballots,
// instead of one dependent load -> ballot -> store round per 64 candidates; the record loads of the few kept candidates
// (~1.6 %) follow.  Shared by the per-view, chain and sharded-chain writers.
#pragma once

#include "l3d_kernels.hpp"

namespace l3d {

// A whole workgroup of 256 threads (4 waves) for ONE segment: rounds of 32 chunks of 64 candidates
// (2048 candidates) whose confidences are all loaded at once; the chunks' kept counts go through LDS, every wave then knows
// the output offset of its chunks.  A segment of config 2 (about 1500 candidates) is one round of loads instead of a chain of
// dependent ones.  s_cnt: 32 ints of LDS.  All 256 threads must call.
// best_pos (optional, with s_best = 4 x 64 bits of LDS): *best_pos = position in `out` of the segment's first kept match with the
// highest confidence, or -1 -- what L3DView::addMatches(only_best) leaves of the list (view.cc:165-183: stable sort by confidence,
// front of every segment's group), found where the confidences already are in registers.
// rt (round 6, with cam_out and s_qcnt = 256 ints of LDS): the view's RUN TABLE, [camera 0 .. N][segment] with rows rt_stride apart -- rt[q * rt_stride + y] =
// position (in `out`) of the first kept match of segment y towards LOCAL camera q, row N = the end of the segment's matches: a later view finds the
// records that point at it (line3D.cc:838-872) and the products find every (view, camera) pair's records without scanning anything.  cam_out then
// holds (local camera << 16 | target segment) instead of the global camera id: all a reader of a run needs of the 32-byte record except the depths.
__device__ __forceinline__ void write_kept_segment_wg(const VerifyArgs& a, int y, int o, const unsigned* __restrict__ local2global,
                                                      Match* __restrict__ out, int* s_cnt, int* __restrict__ best_pos = nullptr,
                                                      unsigned long long* s_best = nullptr, unsigned* __restrict__ cam_out = nullptr,
                                                      int* __restrict__ rt = nullptr, int rt_stride = 0, int* s_qcnt = nullptr, bool pack_side = false)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int start = a.row_start[y * a.N];
    const int m = a.row_start[(y + 1) * a.N] - start;
    const int o_first = o;
    if (rt) s_qcnt[tid] = 0;                 // (ordered before the first atomicAdd by the first round's barrier; m == 0: not read at all)
    unsigned long long bk = 0ull;            // (confidence bits, ~position): the maximum is the first strict maximum in list order
    for (int base = 0; base < m; base += 2048) {
        float c[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { const int i = base + (wave + 4 * r) * 64 + lane; c[r] = i < m ? a.cand_conf[start + i] : 0.0f; }
        unsigned long long b[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { b[r] = __ballot(c[r] > 1.0f); if (lane == 0) s_cnt[wave + 4 * r] = __popcll(b[r]); }
        __syncthreads();
        const int v = lane < 32 ? s_cnt[lane] : 0;
        int incl = v;
        for (int d = 1; d < 32; d <<= 1) { const int u = __shfl_up(incl, d); if (lane >= d) incl += u; }
        const int excl = incl - v;
        const int total = __shfl(incl, 31);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int q = wave + 4 * r;
            const int off = __shfl(excl, q);
            if (c[r] > 1.0f) {
                const int i = base + q * 64 + lane;
                const uint2 meta = a.cand_meta[start + i];
                const float4 d = a.cand_depths[start + i];
                Match rec;
                rec.segID1 = (unsigned)y; rec.camID2 = local2global[meta.y]; rec.segID2 = meta.x;
                rec.depths[0] = d.x; rec.depths[1] = d.y; rec.depths[2] = d.z; rec.depths[3] = d.w;
                rec.confidence = c[r] / 2.0f;                    // confidence_norm, cudawrapper.cu:1089,1098
                const int pos = o + off + __popcll(b[r] & ((1ull << lane) - 1ull));
                out[pos] = rec;
                if (cam_out) cam_out[pos] = (rt || pack_side) ? ((meta.y << 16) | meta.x) : rec.camID2;   // (pack_side: the sharded chain's slots, without run tables)   // (the chain's side array: see rt above; without run tables the global camera id, scanned by later views)
                if (rt) atomicAdd(&s_qcnt[meta.y], 1);
                const unsigned long long key = ((unsigned long long)__float_as_uint(c[r]) << 32) | (0xffffffffu - (unsigned)pos);   // (c > 1: the bits order like the value)
                bk = key > bk ? key : bk;
            }
        }
        o += total;
        __syncthreads();                                       // s_cnt is rewritten by the next round
    }
    if (rt && wave == 0) {
        // kept matches per camera -> run starts: exclusive prefix over the cameras by the first wave alone (the last round's barrier put every count in
        // place; no barrier of its own -- the kernel's critical path is a launch per view)
        int run = o_first;
        for (int q0 = 0; q0 <= a.N; q0 += 64) {
            const int q = q0 + lane;
            const int v = (m > 0 && q < a.N) ? s_qcnt[q] : 0;
            int incl = v;
            for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d); if (lane >= d) incl += u; }
            if (q <= a.N) rt[(size_t)q * rt_stride + y] = run + incl - v;
            run += __shfl(incl, 63);
        }
    }
    if (best_pos) {
        for (int d = 32; d > 0; d >>= 1) {
            const unsigned lo = __shfl_down((unsigned)bk, d), hi = __shfl_down((unsigned)(bk >> 32), d);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            bk = other > bk ? other : bk;
        }
        if (lane == 0) s_best[wave] = bk;
        __syncthreads();
        if (tid == 0) {
            unsigned long long mx = s_best[0];
            for (int w = 1; w < 4; ++w) mx = s_best[w] > mx ? s_best[w] : mx;
            *best_pos = mx ? (int)(0xffffffffu - (unsigned)mx) : -1;
        }
    }
}

// One (segment, source camera) run of existing (reverse) matches, scattered in arbitrary order, into ascending target order:
// one wave, ranks by all-to-all comparison in registers (runs are short; targets inside a run are distinct).
// Runs of more than 256 entries (dense scenes: 4000 segments x 24 neighbours keeps hundreds of matches per segment and camera): with a staging
// area -- `stage`: 4 float arrays `stride` apart + `stage_key`, all indexed like the candidate arrays and unused at this point -- the wave copies the run
// out (coalesced), ranks every element against all keys (chunks of 64 keys broadcast by shuffles, four own elements at a time) and writes it to its
// place: O(n^2 / 64) compares per lane, no dependent chain.  Without one (the per-view seam path's separate launch) a single lane sorts in place.
__device__ __forceinline__ void sort_exist_run(int lane, int b, int n, int cam, uint2* meta, float4* depths, float* stage = nullptr, long long stride = 0,
                                               unsigned* stage_key = nullptr)
{
    if (n < 2) return;
    if (n > 256 && stage) {
        float *s0 = stage + b, *s1 = stage + stride + b, *s2 = stage + 2 * stride + b, *s3 = stage + 3 * stride + b;
        unsigned* sk = stage_key + b;
        for (int i = lane; i < n; i += 64) { const float4 d = depths[b + i]; s0[i] = d.x; s1[i] = d.y; s2[i] = d.z; s3[i] = d.w; sk[i] = meta[b + i].x; }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");       // (the staged copy is read back by other lanes of this wave: stores complete first)
        for (int t0 = 0; t0 < n; t0 += 256) {
            unsigned key[4];
            int rank[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int i = t0 + lane + 64 * r; key[r] = i < n ? sk[i] : 0xffffffffu; }
            for (int c0 = 0; c0 < n; c0 += 64) {
                const unsigned mine = c0 + lane < n ? sk[c0 + lane] : 0xffffffffu;
                const int cnt = min(64, n - c0);
                for (int l = 0; l < cnt; ++l) {
                    const unsigned other = __shfl(mine, l);
#pragma unroll
                    for (int q = 0; q < 4; ++q) rank[q] += other < key[q];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = t0 + lane + 64 * r;
                if (i < n) { meta[b + rank[r]] = make_uint2(key[r], (unsigned)cam); depths[b + rank[r]] = make_float4(s0[i], s1[i], s2[i], s3[i]); }
            }
        }
        return;
    }
    if (n > 256) {                                  // pathological run without a staging area: one lane, in place
        if (lane == 0)
            for (int i = b + 1; i < b + n; ++i) {
                const uint2 m = meta[i];
                const float4 d = depths[i];
                int j = i;
                for (; j > b && meta[j - 1].x > m.x; --j) { meta[j] = meta[j - 1]; depths[j] = depths[j - 1]; }
                meta[j] = m; depths[j] = d;
            }
        return;
    }
    unsigned key[4];
    float4 d[4];
    int rank[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = lane + 64 * r;
        key[r] = i < n ? meta[b + i].x : 0xffffffffu;
        d[r] = i < n ? depths[b + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                   // (all indices static: the arrays stay in registers)
        if (r * 64 < n) {
            const int cnt = min(64, n - r * 64);
            for (int l = 0; l < cnt; ++l) {
                const unsigned other = __shfl(key[r], l);
#pragma unroll
                for (int q = 0; q < 4; ++q) rank[q] += other < key[q];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = lane + 64 * r;
        if (i < n) { meta[b + rank[r]] = make_uint2(key[r], (unsigned)cam); depths[b + rank[r]] = d[r]; }
    }
}

}  // namespace l3d
