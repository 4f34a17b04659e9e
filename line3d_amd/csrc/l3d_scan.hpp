// l3d_scan.hpp -- exclusive prefix sums of small int arrays: wg_scan_excl (one workgroup, for kernels that continue with the
// total in the same launch) and wg_scan_excl_tile (independent workgroups, one per tile: the row scans of the chains).
//
// The arrays scanned on the matching path are small (rows of one view: S*N ints, a few ten thousand), so one
// workgroup and no second pass beat a multi-block scan whose launches would sit on the per-view critical path.
// Tiles of 4096 ints are read coalesced (4 consecutive ints per thread, 16-byte loads when the pointers allow),
// scanned with wave shuffles + 16 wave totals in LDS, and written back the same way; a running carry links tiles.
// Kernels that need the total (allocation of the kept slice, slot headers) call this and continue in the same launch.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace l3d {

constexpr int kScanThreads = 1024;

// out[i] = sum(in[0..i-1]) for i in [0,n], i.e. n+1 entries; `zero` (optional) gets n zeros.  Returns the total in
// every thread.  s_w: NT/64 ints of LDS.  All NT threads of the workgroup must call.
template <int NT = kScanThreads>
__device__ __forceinline__ int wg_scan_excl(const int* __restrict__ in, int* __restrict__ out, int n, int* __restrict__ zero, int* s_w)
{
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool vec_in = (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    const bool vec_out = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    const bool vec_zero = zero && (reinterpret_cast<uintptr_t>(zero) & 15) == 0;
    int carry = 0;
    for (int base = 0; base < n; base += 4 * NT) {
        const int i = base + tid * 4;
        int4 v = make_int4(0, 0, 0, 0);
        const bool full = i + 3 < n;
        if (full && vec_in) v = *reinterpret_cast<const int4*>(in + i);
        else {
            if (i < n) v.x = in[i];
            if (i + 1 < n) v.y = in[i + 1];
            if (i + 2 < n) v.z = in[i + 2];
            if (i + 3 < n) v.w = in[i + 3];
        }
        const int t = v.x + v.y + v.z + v.w;
        int incl = t;
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        int woff = 0, tile = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const int x = s_w[w]; tile += x; if (w < wave) woff += x; }
        const int e0 = carry + woff + incl - t;
        const int4 e = make_int4(e0, e0 + v.x, e0 + v.x + v.y, e0 + v.x + v.y + v.z);
        if (full && vec_out) *reinterpret_cast<int4*>(out + i) = e;
        else {
            if (i < n) out[i] = e.x;
            if (i + 1 < n) out[i + 1] = e.y;
            if (i + 2 < n) out[i + 2] = e.z;
            if (i + 3 < n) out[i + 3] = e.w;
        }
        if (zero) {
            if (full && vec_zero) *reinterpret_cast<int4*>(zero + i) = make_int4(0, 0, 0, 0);
            else for (int k = 0; k < 4; ++k) if (i + k < n) zero[i + k] = 0;
        }
        carry += tile;
        __syncthreads();                       // s_w is rewritten by the next tile
    }
    if (tid == 0) out[n] = carry;
    return carry;
}

// The same prefix sum by several INDEPENDENT workgroups of 256 threads, one per tile of 4096 ints: workgroup t scans its own
// tile and, instead of waiting for its predecessors, sums everything in front of its tile itself (all loads issued up front;
// t is at most a few dozen, the array is L2 resident).  No inter-workgroup synchronisation.  Small workgroups on purpose:
// this launch sits on the per-view critical path while another stream keeps the CUs full, and a 1024-thread workgroup has
// to wait until one CU has 16 free wave slots (measured 40 us for a 9 us kernel).  s_w: 5 ints of LDS.
constexpr int kTileThreads = 256;
constexpr int kTileInts = 4096;
__device__ __forceinline__ void wg_scan_excl_tile(const int* __restrict__ in, int* __restrict__ out, int n, int* __restrict__ zero, int tile, int* s_w,
                                                  int* __restrict__ total_out = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;       // 4 waves
    const bool vec_in = (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    const bool vec_out = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    const bool vec_zero = zero && (reinterpret_cast<uintptr_t>(zero) & 15) == 0;
    const int base = tile * kTileInts;
    // own tile: 4 sub-tiles of 1024 ints, one int4 per thread each (issued before the front sum is needed)
    int4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = base + (k * kTileThreads + tid) * 4;
        v[k] = make_int4(0, 0, 0, 0);
        if (i + 3 < n && vec_in) v[k] = *reinterpret_cast<const int4*>(in + i);
        else {
            if (i < n) v[k].x = in[i];
            if (i + 1 < n) v[k].y = in[i + 1];
            if (i + 2 < n) v[k].z = in[i + 2];
            if (i + 3 < n) v[k].w = in[i + 3];
        }
    }
    // everything in front of the tile (full tiles: always in range)
    int front = 0;
    for (int j = tid * 4; j < base; j += kTileThreads * 4) {
        if (vec_in) { const int4 u = *reinterpret_cast<const int4*>(in + j); front += u.x + u.y + u.z + u.w; }
        else front += in[j] + in[j + 1] + in[j + 2] + in[j + 3];
    }
    for (int o = 32; o > 0; o >>= 1) front += __shfl_down(front, o);
    if (tid == 0) s_w[4] = 0;
    __syncthreads();
    if (lane == 0) atomicAdd(&s_w[4], front);
    __syncthreads();
    int carry = s_w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = base + (k * kTileThreads + tid) * 4;
        const int t4 = v[k].x + v[k].y + v[k].z + v[k].w;
        int incl = t4;
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
        __syncthreads();                                   // s_w[0..3] of the previous sub-tile have been read
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        int woff = 0, sub = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int x = s_w[w]; sub += x; if (w < wave) woff += x; }
        const int e0 = carry + woff + incl - t4;
        const int4 e = make_int4(e0, e0 + v[k].x, e0 + v[k].x + v[k].y, e0 + v[k].x + v[k].y + v[k].z);
        if (i + 3 < n && vec_out) *reinterpret_cast<int4*>(out + i) = e;
        else {
            if (i < n) out[i] = e.x;
            if (i + 1 < n) out[i + 1] = e.y;
            if (i + 2 < n) out[i + 2] = e.z;
            if (i + 3 < n) out[i + 3] = e.w;
        }
        if (zero) {
            if (i + 3 < n && vec_zero) *reinterpret_cast<int4*>(zero + i) = make_int4(0, 0, 0, 0);
            else for (int c = 0; c < 4; ++c) if (i + c < n) zero[i + c] = 0;
        }
        carry += sub;
    }
    if (base + kTileInts >= n && tid == 0) { out[n] = carry; if (total_out) *total_out = carry; }     // the last tile also writes the total
}

// Segments [seg_begin, seg_end) ordered by descending candidate count (64 bins of 32): the verification kernel takes its
// workgroups in this order, longest first, so that the last round of workgroups is made of short ones (the order changes
// nothing in the results).  One extra workgroup of the row scan's launch.  s_hist: 130 ints of LDS.
__device__ __forceinline__ void wg_segment_order(const int* __restrict__ rowcnt, int N, int seg_begin, int seg_end,
                                                 int* __restrict__ order, int* s_hist)
{
    const int tid = threadIdx.x;
    if (tid < 130) s_hist[tid] = 0;
    __syncthreads();
    // (a segment's candidate count straight from its N row counts: independent of the prefix sums)
    for (int y = seg_begin + tid; y < seg_end; y += (int)blockDim.x) {
        int m = 0;
        for (int r = 0; r < N; ++r) m += rowcnt[y * N + r];
        atomicAdd(&s_hist[63 - min(63, m >> 5)], 1);
    }
    __syncthreads();
    if (tid == 0) { int run = 0; for (int b = 0; b < 64; ++b) { s_hist[65 + b] = run; run += s_hist[b]; } }
    __syncthreads();
    for (int y = seg_begin + tid; y < seg_end; y += (int)blockDim.x) {
        int m = 0;
        for (int r = 0; r < N; ++r) m += rowcnt[y * N + r];
        order[atomicAdd(&s_hist[65 + 63 - min(63, m >> 5)], 1)] = y;
    }
}

// Sum and per-segment maximum of the candidate counts of segments [seg_begin, seg_end) (sizes the verification launch of
// the view; written straight into host-mapped memory).  One more workgroup of the row scan's launch.  s_red: 8 ints of LDS.
__device__ __forceinline__ void wg_raw_stats(const int* __restrict__ rowcnt, int N, int seg_begin, int seg_end, int* __restrict__ out2, int* s_red)
{
    int tot = 0, mx = 0;
    for (int s = seg_begin + (int)threadIdx.x; s < seg_end; s += (int)blockDim.x) {
        int c = 0;
        for (int k = 0; k < N; ++k) c += rowcnt[s * N + k];
        tot += c; mx = max(mx, c);
    }
    for (int o = 32; o > 0; o >>= 1) { tot += __shfl_down(tot, o); mx = max(mx, __shfl_down(mx, o)); }
    const int wave = threadIdx.x >> 6, nw = ((int)blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0 && wave < 4) { s_red[wave] = tot; s_red[4 + wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < min(nw, 4); ++w) { tot += s_red[w]; mx = max(mx, s_red[4 + w]); }
        out2[0] = tot; out2[1] = mx;
    }
}

}  // namespace l3d
