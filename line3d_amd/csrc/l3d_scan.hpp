// l3d_scan.hpp -- exclusive prefix sum of n ints by ONE workgroup of 1024 threads (16 waves).
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
// every thread.  s_w: 16 ints of LDS.  All 1024 threads must call.
__device__ __forceinline__ int wg_scan_excl(const int* __restrict__ in, int* __restrict__ out, int n, int* __restrict__ zero, int* s_w)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool vec_in = (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    const bool vec_out = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    const bool vec_zero = zero && (reinterpret_cast<uintptr_t>(zero) & 15) == 0;
    int carry = 0;
    for (int base = 0; base < n; base += 4 * kScanThreads) {
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
        for (int w = 0; w < 16; ++w) { const int x = s_w[w]; tile += x; if (w < wave) woff += x; }
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

// Segments [seg_begin, seg_end) ordered by descending candidate count (64 bins of 32): the verification kernel takes its
// workgroups in this order, longest first, so that the last round of workgroups is made of short ones (the order changes
// nothing in the results).  Runs in the tail of the single-workgroup scan that produced row_start.  s_hist: 130 ints of LDS.
__device__ __forceinline__ void wg_segment_order(const int* __restrict__ row_start, int N, int seg_begin, int seg_end,
                                                 int* __restrict__ order, int* s_hist)
{
    const int tid = threadIdx.x;
    if (tid < 130) s_hist[tid] = 0;
    __syncthreads();
    for (int y = seg_begin + tid; y < seg_end; y += kScanThreads) {
        const int m = row_start[(y + 1) * N] - row_start[y * N];
        atomicAdd(&s_hist[63 - min(63, m >> 5)], 1);
    }
    __syncthreads();
    if (tid == 0) { int run = 0; for (int b = 0; b < 64; ++b) { s_hist[65 + b] = run; run += s_hist[b]; } }
    __syncthreads();
    for (int y = seg_begin + tid; y < seg_end; y += kScanThreads) {
        const int m = row_start[(y + 1) * N] - row_start[y * N];
        order[atomicAdd(&s_hist[65 + 63 - min(63, m >> 5)], 1)] = y;
    }
}

}  // namespace l3d
