// l3d_chain_sharded.hip -- the resident matchViews chain (l3d_chain.hip) with every view's SOURCE segments sharded over
// the GPUs of one node.
//
// matchViews is a dependency chain over views, so views are not distributed.  Each rank (one process per GPU) runs the
// chain on its source-segment range [S*r/W, S*(r+1)/W) of every view: stage 1 and the verification of a source segment
// only touch that segment's rows, so the ranges are independent and need no exchange.  The single exchange per view
// is its kept list: a later view pulls its reverse matches (line3D.cc:838-872) from ALL ranks' kept records, so after a
// view's kept records are written into this rank's fixed-size SLOT the caller all-gathers the slots of that view
// (RCCL over xGMI, `torch.distributed.all_gather_into_tensor` on this context's stream, see
// line3d_amd/distributed.py).  Nothing waits for the host: the library only enqueues (l3d_shard_chain_enqueue), the
// framework enqueues the collective on the same stream, the host trails behind on events (l3d_shard_chain_fetch).
// The concatenation of the ranks' kept lists in rank order is the sorted list of the unsharded run, bit for bit.
//
// slot layout: [SlotHeader 32 B][best depth pairs float2 x seg_cap][position of every segment's best kept match int x seg_cap][kept records
// l3d_match x slot_records]
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "l3d_ctx.hpp"
#include "l3d_scan.hpp"
#include "l3d_kept.hpp"
#include "l3d_chain_common.hpp"
#include "l3d_products.hpp"

using namespace l3d;

// Host-side waits of the enqueue threads: yield for a while, then sleep -- a thread that is far ahead (the stage-1 thread
// almost always is) must not burn a core of a CPU-quota'd container shared by eight ranks.
template <class Pred>
static inline void wait_until(Pred pred, int spins_before_sleep)
{
    int spins = 0;
    while (!pred()) {
        if (++spins < spins_before_sleep) std::this_thread::yield();
        else { const int us = tunables().wait_sleep_us.load(std::memory_order_relaxed); if (us > 0) std::this_thread::sleep_for(std::chrono::microseconds(us)); else std::this_thread::yield(); }
    }
}

namespace l3d {

struct SlotHeader { int n_kept, R, overflow, s0, s1, pad[3]; };
static_assert(sizeof(SlotHeader) == 32, "slot header");

// ring: view blocks the gathered buffer holds (view k lives in block k % ring); n_views when every block is kept
struct SlotGeom { size_t slot_bytes, best_off, bpos_off, rec_off, cam_off, rt_off; int seg_cap, slot_records, world, ring, max_n; };   // rt_off: 0 = no run table in the slot   // cam_off: 0 = the slot carries no side array of target cameras

// reverse matches for view `view_id`, source-segment range [s0,s1), out of the gathered slots of earlier views
// (blockIdx.y = source * world + rank)
__global__ void k_exist_count_slots(const unsigned char* __restrict__ G, SlotGeom g, const int* __restrict__ src_index,
                                    const int* __restrict__ src_cam, const int* __restrict__ src_slot, unsigned view_id, int N, int s0, int s1, int* __restrict__ rowcnt)
{
    const int src = blockIdx.y / g.world, r = blockIdx.y % g.world;
    const unsigned char* slot = G + ((size_t)(src_index[src] % g.ring) * g.world + r) * g.slot_bytes;
    const SlotHeader* hd = reinterpret_cast<const SlotHeader*>(slot);
    const int n = hd->overflow ? 0 : hd->n_kept;
    const Match* kept = reinterpret_cast<const Match*>(slot + g.rec_off);
    // side array (dense scenes): (the record's LOCAL camera << 16 | its target segment) -- camera and segment range are decided on 4 bytes, the count reads no
    // record (round 5: the global camera id; the 1 record in N that matched was read, 7 in 8 of those for another rank's range)
    const unsigned* side = g.cam_off ? reinterpret_cast<const unsigned*>(slot + g.cam_off) : nullptr;
    const int cam = src_cam[src];
    if (src_slot[src] < 0) return;                              // (this view is not among the source's neighbours: no record of it points here)
    const unsigned want = (unsigned)src_slot[src];
    const int stride = gridDim.x * blockDim.x;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (side && g.rt_off && (int)want <= g.max_n) {
        if (n <= 0) return;                                     // (an overflowed slot holds no records and no run table)
        // run tables: the runs of this view's camera in the rank's slot, sixteen lanes per run
        const int* rt = reinterpret_cast<const int*>(slot + g.rt_off);
        const int* r0 = rt + (size_t)want * g.seg_cap;
        const int* r1 = r0 + g.seg_cap;
        const int nsl = hd->s1 - hd->s0;
        const int grp = i >> 4, gl = i & 15, ngrp = stride >> 4;
        for (int sl = grp; sl < nsl; sl += ngrp) {
            const int a = r0[sl], b = r1[sl];
            for (int k = a + gl; k < b; k += 16) { const unsigned w = side[k]; const int u = (int)(w & 0xffffu); if (u >= s0 && u < s1) atomicAdd(&rowcnt[u * N + cam], 1); }
        }
        return;
    }
    if (side) {
        auto count = [&](unsigned w) { const int u = (int)(w & 0xffffu); if ((w >> 16) == want && u >= s0 && u < s1) atomicAdd(&rowcnt[u * N + cam], 1); };
        for (; i + 3 * stride < n; i += 4 * stride) {      // four words in flight per thread
            const unsigned c0 = side[i], c1 = side[i + stride], c2 = side[i + 2 * stride], c3 = side[i + 3 * stride];
            count(c0); count(c1); count(c2); count(c3);
        }
        for (; i < n; i += stride) count(side[i]);
        return;
    }
    for (; i < n; i += stride) {
        const Match m = kept[i];
        if (m.camID2 == view_id && (int)m.segID2 >= s0 && (int)m.segID2 < s1) atomicAdd(&rowcnt[m.segID2 * N + cam], 1);
    }
}

// Both writers of the combined candidate arrays in one launch (they are independent: stage-1 candidates go to the rows of the
// cameras to be matched, reverse matches to the rows of the source cameras): the first `blocks_move` workgroups copy the
// stage-1 candidates of the rank's rows (a wave per row), the others scatter the reverse matches
// (k_exist_scatter_slots, wps workgroups per (source view, rank) list).
__global__ __launch_bounds__(256) void k_place_slots(int blocks_move, int wps, const int* __restrict__ tbm, int n_tbm, const int* __restrict__ rowA,
                                                     const uint2* __restrict__ metaA, const float4* __restrict__ depthsA,
                                                     const unsigned char* __restrict__ G, SlotGeom g, const int* __restrict__ src_index,
                                                     const int* __restrict__ src_cam, const int* __restrict__ src_slot, unsigned view_id, int N, int S, int s0, int s1,
                                                     const int* __restrict__ row_start, int* __restrict__ cursor,
                                                     uint2* __restrict__ meta, float4* __restrict__ depths, int cap)
{
    if (row_start[(size_t)S * N] > cap) return;                  // overflow: the chain is re-run with more room
    if ((int)blockIdx.x < blocks_move) {
        const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (row >= (s1 - s0) * n_tbm) return;
        const int y = s0 + row / n_tbm, cam = tbm[row % n_tbm];
        const int a = rowA[y * N + cam], b = row_start[y * N + cam], n = row_start[y * N + cam + 1] - b;   // (rowA may have been laid out for an upper bound of the row)
        for (int j = lane; j < n; j += 64) { meta[b + j] = metaA[a + j]; depths[b + j] = depthsA[a + j]; }
        return;
    }
    const int e = (int)blockIdx.x - blocks_move, list = e / wps, bx = e % wps;
    const int src = list / g.world, r = list % g.world;
    const unsigned char* slot = G + ((size_t)(src_index[src] % g.ring) * g.world + r) * g.slot_bytes;
    const SlotHeader* hd = reinterpret_cast<const SlotHeader*>(slot);
    const int n = hd->overflow ? 0 : hd->n_kept;
    const Match* kept = reinterpret_cast<const Match*>(slot + g.rec_off);
    const unsigned* side = g.cam_off ? reinterpret_cast<const unsigned*>(slot + g.cam_off) : nullptr;     // (local camera << 16 | target segment): k_exist_count_slots
    const int cam = src_cam[src];
    if (src_slot[src] < 0) return;                              // (this view is not among the source's neighbours: no record of it points here)
    const unsigned want = (unsigned)src_slot[src];
    auto place = [&](int i) {
        const Match m = kept[i];
        if (m.camID2 == view_id && (int)m.segID2 >= s0 && (int)m.segID2 < s1) {
            const int row = m.segID2 * N + cam;
            const int slotpos = row_start[row] + atomicAdd(&cursor[row], 1);
            meta[slotpos] = make_uint2(m.segID1, (unsigned)cam);
            depths[slotpos] = make_float4(m.depths[2], m.depths[3], m.depths[0], m.depths[1]);
        }
    };
    auto mine = [&](unsigned w) { const int u = (int)(w & 0xffffu); return (w >> 16) == want && u >= s0 && u < s1; };
    const int stride = wps * 256;
    int i = bx * 256 + (int)threadIdx.x;
    if (side && g.rt_off && (int)want <= g.max_n) {     // run tables: k_exist_count_slots
        if (n <= 0) return;
        const int* rt = reinterpret_cast<const int*>(slot + g.rt_off);
        const int* r0 = rt + (size_t)want * g.seg_cap;
        const int* r1 = r0 + g.seg_cap;
        const int nsl = hd->s1 - hd->s0;
        const int grp = i >> 4, gl = i & 15, ngrp = stride >> 4;
        for (int sl = grp; sl < nsl; sl += ngrp) {
            const int a = r0[sl], b = r1[sl];
            for (int k = a + gl; k < b; k += 16) { const unsigned w = side[k]; const int u = (int)(w & 0xffffu); if (u >= s0 && u < s1) place(k); }
        }
        return;
    }
    if (side) {     // (four words in flight per thread; a record is read only where the word says it is this rank's)
        for (; i + 3 * stride < n; i += 4 * stride) {
            const unsigned c0 = side[i], c1 = side[i + stride], c2 = side[i + 2 * stride], c3 = side[i + 3 * stride];
            if (mine(c0)) place(i);
            if (mine(c1)) place(i + stride);
            if (mine(c2)) place(i + 2 * stride);
            if (mine(c3)) place(i + 3 * stride);
        }
        for (; i < n; i += stride) if (mine(side[i])) place(i);
        return;
    }
    for (; i < n; i += stride) place(i);
}

// Kept records of this rank's source-segment range [s0,s1) into its slot, in ONE launch behind the verification (every
// launch of the sharded chain sits on the per-view critical path): each workgroup (one segment) sums the kept counts in
// front of its segment itself -- a few hundred to 2000 ints out of L2 -- instead of waiting for a scan launch;
// workgroup 0 also writes the slot header (count, #candidates, overflow).  kept_cnt points at the range's first segment.
__global__ __launch_bounds__(256) void k_slot_write(VerifyArgs a, const int* __restrict__ kept_cnt, int nrow, int slot_records,
                                                    const unsigned* __restrict__ local2global, const float2* __restrict__ best, SlotGeom g,
                                                    unsigned char* __restrict__ slot)
{
    __shared__ int s_red[8];
    __shared__ int s_cnt[32];
    __shared__ int s_qcnt[256];
    __shared__ unsigned long long s_best[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nseg = a.seg_end - a.seg_begin;
    const int yl = blockIdx.x;
    int before = 0, total = 0;
    for (int i = tid; i < nseg; i += 256) { const int v = kept_cnt[i]; total += v; if (i < yl) before += v; }
    for (int o = 32; o > 0; o >>= 1) { before += __shfl_down(before, o); total += __shfl_down(total, o); }
    if (lane == 0) { s_red[wave] = before; s_red[4 + wave] = total; }
    __syncthreads();
    before = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    total = s_red[4] + s_red[5] + s_red[6] + s_red[7];
    SlotHeader h;
    h.R = a.row_start[nrow];
    h.overflow = h.R > a.cand_cap ? 1 : 0;
    h.n_kept = h.overflow ? 0 : total;
    if (h.n_kept > slot_records) { h.overflow |= 2; h.n_kept = 0; }
    h.s0 = a.seg_begin; h.s1 = a.seg_end; h.pad[0] = total; h.pad[1] = h.pad[2] = 0;      // pad[0]: the kept count even when it did not fit
    if (blockIdx.x == 0 && tid == 0) *reinterpret_cast<SlotHeader*>(slot) = h;
    if (yl >= nseg) return;                                  // (an empty range still launches workgroup 0 for the header)
    const int y = a.seg_begin + yl;
    if (tid == 0) reinterpret_cast<float2*>(slot + g.best_off)[yl] = best[y];
    int* bpos = reinterpret_cast<int*>(slot + g.bpos_off) + yl;              // (position in this slot's records; -1: the segment kept nothing)
    if (h.overflow) { if (tid == 0) *bpos = -1; return; }
    // (round 6) the side array holds (local camera << 16 | target segment): a reader decides camera AND segment range on 4 bytes -- the count reads no record at all
    // the run table's rows are seg_cap apart and indexed by the segment's position in the rank's range (the writer indexes by the segment: the base is shifted)
    int* rt = g.rt_off ? reinterpret_cast<int*>(slot + g.rt_off) - a.seg_begin : nullptr;
    write_kept_segment_wg(a, y, before, local2global, reinterpret_cast<Match*>(slot + g.rec_off), s_cnt, bpos, s_best, g.cam_off ? reinterpret_cast<unsigned*>(slot + g.cam_off) : nullptr,
                          rt, g.seg_cap, s_qcnt, true);
}

// Hand-over of one finished view on a committing rank: the ranks' kept records, concatenated in rank (= segment) order
// behind the raw depth-pair arrays, are packed into a contiguous device staging buffer (one SDMA copy of the exact size
// brings them to the host); the summary goes straight into host-mapped pinned memory.
struct PackHeader { long long R; int n_kept, overflow, pad[4]; };
static_assert(sizeof(PackHeader) == 32, "pack header");
__global__ __launch_bounds__(256) void k_pack_view(const unsigned char* __restrict__ block, SlotGeom g, unsigned char* __restrict__ out,
                                                   PackHeader* __restrict__ hdr_host)
{
    const int r = blockIdx.y;
    int base = 0, bad = 0;
    long long R = 0;
    for (int q = 0; q < g.world; ++q) {
        const SlotHeader* hq = reinterpret_cast<const SlotHeader*>(block + (size_t)q * g.slot_bytes);
        const bool ok = !hq->overflow && hq->n_kept >= 0 && hq->n_kept <= g.slot_records;
        bad |= !ok;
        if (q < r && ok) base += hq->n_kept;
        R += hq->R;
    }
    const unsigned char* slot = block + (size_t)r * g.slot_bytes;
    const SlotHeader* hd = reinterpret_cast<const SlotHeader*>(slot);
    const size_t best_bytes = (size_t)g.seg_cap * 8;
    unsigned char* o_best = out + (size_t)r * best_bytes;
    unsigned char* o_rec = out + (size_t)g.world * best_bytes;
    const int n = bad ? 0 : hd->n_kept;
    if (r == g.world - 1 && blockIdx.x == 0 && threadIdx.x == 0) {
        PackHeader ph;
        ph.R = R; ph.n_kept = bad ? 0 : base + n; ph.overflow = bad; ph.pad[0] = ph.pad[1] = ph.pad[2] = ph.pad[3] = 0;
        *hdr_host = ph;
    }
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    // depth pairs of this rank's segment range (fixed stride seg_cap; the host skips the -1 markers)
    const int nb = (hd->s1 - hd->s0);
    for (int i = tid; i < nb; i += nt)
        reinterpret_cast<float2*>(o_best)[i] = reinterpret_cast<const float2*>(slot + g.best_off)[i];
    const float4* src = reinterpret_cast<const float4*>(slot + g.rec_off);
    float4* dst = reinterpret_cast<float4*>(o_rec + (size_t)base * sizeof(Match));
    for (int i = tid; i < 2 * n; i += nt) dst[i] = src[i];
}

// ---- matchViews' products from the gathered slots (l3d_shard_chain_products): every rank holds every view's kept records, so every
// rank can build what the single-GPU chain builds from its kept arena -- no rank hands lists to the host.
// per view: kept matches and candidates summed over the ranks' slots
// (hdr, hdr_stride: the slot headers where they are -- in front of every slot of the gathered blocks, or the compact header table the ring
// mode keeps)
__global__ void k_shard_totals(const unsigned char* __restrict__ hdr, size_t hdr_stride, SlotGeom g, const unsigned char* __restrict__ verified, int n_views, int2* __restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_views) return;
    int n = 0; long long R = 0;
    if (verified[k])
        for (int r = 0; r < g.world; ++r) {
            const SlotHeader* hd = reinterpret_cast<const SlotHeader*>(hdr + ((size_t)k * g.world + r) * hdr_stride);
            n += hd->overflow ? 0 : hd->n_kept; R += hd->R;
        }
    out[k] = make_int2(n, (int)min(R, 0x7fffffffll));
}
// all views' slots -> ONE kept arena (views back to back, ranks in segment order: the sorted list of the unsharded run) + the whole
// view's best depth pairs and best positions (relative to the view's slice).  grid (x, rank, view).
__global__ __launch_bounds__(256) void k_shard_pack_all(const unsigned char* __restrict__ G, SlotGeom g, const unsigned char* __restrict__ verified,
                                                        const unsigned* __restrict__ kept_base, const long long* __restrict__ best_off, Match* __restrict__ arena,
                                                        float2* __restrict__ best_all, int* __restrict__ bestpos_all)
{
    const int k = blockIdx.z, r = blockIdx.y;
    if (!verified[k]) return;
    const unsigned char* block = G + (size_t)k * g.world * g.slot_bytes;
    int base = 0;
    for (int q = 0; q < r; ++q) { const SlotHeader* hq = reinterpret_cast<const SlotHeader*>(block + (size_t)q * g.slot_bytes); base += hq->overflow ? 0 : hq->n_kept; }
    const unsigned char* slot = block + (size_t)r * g.slot_bytes;
    const SlotHeader* hd = reinterpret_cast<const SlotHeader*>(slot);
    const int n = hd->overflow ? 0 : hd->n_kept;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    const float4* src = reinterpret_cast<const float4*>(slot + g.rec_off);
    float4* dst = reinterpret_cast<float4*>(arena + (size_t)kept_base[k] + base);
    for (int i = tid; i < 2 * n; i += nt) dst[i] = src[i];
    const float2* sb = reinterpret_cast<const float2*>(slot + g.best_off);
    const int* sp = reinterpret_cast<const int*>(slot + g.bpos_off);
    for (int i = tid; i < hd->s1 - hd->s0; i += nt) {
        const long long o = best_off[k] + hd->s0 + i;
        best_all[o] = sb[i];
        const int p = sp[i];
        bestpos_all[o] = p < 0 || n == 0 ? -1 : base + p;
    }
}

// Ring mode (l3d_shard_chain_run when the gathered blocks of all views would not fit a budget: 2048 views x 8 ranks x 3.8 MB slots = 63 GB at
// 4000 segments x 24 neighbours): the gathered buffer holds only the views later views still read (the neighbour window of the schedule);
// older blocks are RETIRED a batch at a time -- their used records into the compact kept arena in the order of the unsharded run (what
// k_shard_pack_all does for all views at the end), their headers into a small table for the shared verdict -- before their ring block is
// overwritten.  grid (x, rank, view of the batch).  base_dev[k]: the arena offset of view k (running sum kept on the device: the host never
// learns a count on the way); the block of the batch's last view publishes the offset of the next batch.
// keep (partitioned job, l3d_shard_chain_partition; null: every view): a view that is not flagged leaves its headers in the table (the shared
// verdict reads them) and nothing in this rank's arena.
__global__ __launch_bounds__(256) void k_shard_retire(const unsigned char* __restrict__ G, SlotGeom g, const unsigned char* __restrict__ verified, const unsigned char* __restrict__ keep, int k0, int n_batch,
                                                      long long* __restrict__ base_dev, long long arena_cap, const long long* __restrict__ best_off, Match* __restrict__ arena,
                                                      float2* __restrict__ best_all, int* __restrict__ bestpos_all, SlotHeader* __restrict__ hdr_all, int* __restrict__ overflow,
                                                      unsigned* __restrict__ qt_arena, int* __restrict__ rt_all, const long long* __restrict__ rt_off, const int2* __restrict__ view_sn)
{
    const int k = k0 + blockIdx.z, r = blockIdx.y;
    long long base = base_dev[k0];
    for (int q = k0; q < k; ++q) {
        if (!verified[q] || (keep && !keep[q])) continue;
        const unsigned char* bq = G + (size_t)(q % g.ring) * g.world * g.slot_bytes;
        for (int w = 0; w < g.world; ++w) { const SlotHeader* hq = reinterpret_cast<const SlotHeader*>(bq + (size_t)w * g.slot_bytes); base += hq->overflow ? 0 : hq->n_kept; }
    }
    const unsigned char* block = G + (size_t)(k % g.ring) * g.world * g.slot_bytes;
    const bool kept_here = !keep || keep[k];
    int in_front = 0, all = 0;
    if (verified[k] && kept_here)
        for (int w = 0; w < g.world; ++w) {
            const SlotHeader* hq = reinterpret_cast<const SlotHeader*>(block + (size_t)w * g.slot_bytes);
            const int n = hq->overflow ? 0 : hq->n_kept;
            if (w < r) in_front += n;
            all += n;
        }
    const bool first = blockIdx.x == 0 && threadIdx.x == 0;
    if (first && r == 0) base_dev[k] = base;                                          // (k0 itself: rewritten with the same value)
    if (first && r == 0 && (int)blockIdx.z == n_batch - 1) base_dev[k + 1] = base + all;
    SlotHeader* ho = hdr_all + (size_t)k * g.world + r;
    if (!verified[k]) { if (first) { SlotHeader z; z.n_kept = z.R = z.overflow = z.s0 = z.s1 = 0; z.pad[0] = z.pad[1] = z.pad[2] = 0; *ho = z; } return; }
    const unsigned char* slot = block + (size_t)r * g.slot_bytes;
    const SlotHeader* hd = reinterpret_cast<const SlotHeader*>(slot);
    if (first) *ho = *hd;
    if (!kept_here) return;
    const int n = hd->overflow ? 0 : hd->n_kept;
    if (base + all > arena_cap) { if (first) atomicMax(overflow, 1); return; }      // the arena's first guess was too small: the run ends with a capacity verdict
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    const float4* src = reinterpret_cast<const float4*>(slot + g.rec_off);
    float4* dst = reinterpret_cast<float4*>(arena + base + in_front);
    for (int i = tid; i < 2 * n; i += nt) dst[i] = src[i];
    const float2* sb = reinterpret_cast<const float2*>(slot + g.best_off);
    const int* sp = reinterpret_cast<const int*>(slot + g.bpos_off);
    for (int i = tid; i < hd->s1 - hd->s0; i += nt) {
        const long long o = best_off[k] + hd->s0 + i;
        best_all[o] = sb[i];
        const int p = sp[i];
        bestpos_all[o] = p < 0 || n == 0 ? -1 : in_front + p;
    }
    // (round 6) the slot's side words and run table travel with its records: the view's (local camera << 16 | target) array and its run table, (N + 1) x S with
    // positions counted from the view's first record -- what the single-GPU chain's kept writer leaves, so the products transpose without rebuilding either
    if (qt_arena) {
        const unsigned* sq = reinterpret_cast<const unsigned*>(slot + g.cam_off);
        unsigned* dq = qt_arena + base + in_front;
        for (int i = tid; i < n; i += nt) dq[i] = sq[i];
        const int S = view_sn[k].x, N = view_sn[k].y, ns = hd->s1 - hd->s0;
        int* rt = rt_all + rt_off[k];
        const int* srt = reinterpret_cast<const int*>(slot + g.rt_off);
        for (int i = tid; i < (N + 1) * ns; i += nt) {
            const int q = i / ns, j = i - q * ns;
            rt[(size_t)q * S + hd->s0 + j] = in_front + (n > 0 ? srt[(size_t)q * g.seg_cap + j] : 0);
        }
    }
}

// After the last exchange: the OR of the overflow bits of every gathered slot header (1 candidate capacity, 2 slot records,
// 4 a rank gave up) and, for sizing a retry, the largest candidate / kept count any rank reported.  Every rank holds the same
// gathered blocks, so every rank computes the same answer: the ranks agree on the outcome without another collective.
__global__ void k_check_slots(const unsigned char* __restrict__ hdr, size_t hdr_stride, SlotGeom g, const unsigned char* __restrict__ verified, int n_views, int* __restrict__ out3)
{
    int bits = 0, maxR = 0, maxK = 0;
    const int n = n_views * g.world;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!verified[i / g.world]) continue;
        const SlotHeader* hd = reinterpret_cast<const SlotHeader*>(hdr + (size_t)i * hdr_stride);
        bits |= hd->overflow & 7;
        maxR = max(maxR, hd->R);
        maxK = max(maxK, hd->pad[0]);
    }
    for (int o = 32; o > 0; o >>= 1) { bits |= __shfl_down(bits, o); maxR = max(maxR, __shfl_down(maxR, o)); maxK = max(maxK, __shfl_down(maxK, o)); }
    if ((threadIdx.x & 63) == 0) { if (bits) atomicOr(&out3[0], bits); atomicMax(&out3[1], maxR); atomicMax(&out3[2], maxK); }
}

}  // namespace l3d

namespace {

inline unsigned __float_as_uint_host(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
typedef l3d::ChainViewDev SViewDev;           // (s0, s1: this rank's source-segment range)

size_t salign(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

struct l3d_shard_chain {
    l3d_ctx* c = nullptr;
    const l3d_chain_view* views = nullptr;
    int n_views = 0, rank = 0, world = 1;
    SlotGeom geom;
    std::vector<SViewDev> vd;
    const unsigned char* dtab = nullptr;
    int* hstats = nullptr;
    int* hstats_dev = nullptr;
    std::vector<hipEvent_t> ev1, ev2;
    int k_p1 = 0;
    size_t cand_cap = 0;
    static constexpr int kStage1Ahead = 8, kRingA = kStage1Ahead + 2;   // ring of stage-1 candidate buffers
    int maxS = 0, maxN = 0;
    const unsigned char* gathered = nullptr;
    double pairs = 0, raw_sum = 0, kept_total = 0;
    std::vector<float> best_scratch;
    // hand-over (k_pack_view): a ring of device staging buffers, per-view summaries in host-mapped pinned memory
    static constexpr int kRing = 16;
    size_t stage_bytes = 0;
    PackHeader* hdr_host = nullptr;
    PackHeader* hdr_dev = nullptr;
    bool eager_pack = false;                 // l3d_shard_chain_run on a committing rank: pack at mark time, on the chain's stream
    std::vector<char> packed;                // per view: its pack kernel is enqueued (and covered by ev2)
    hipEvent_t ev3 = nullptr;                // lazy pack (step-wise protocol): enqueued by fetch
    std::atomic<int> marked{0};              // views [0, marked) carry their completion event
    std::atomic<int> s1_done{0};             // l3d_shard_chain_run with a stage-1 thread: views [0, s1_done) have their stage 1 enqueued
    bool s1_thread = false;
    std::atomic<int> fetched{0};             // views [0, fetched) have left their staging buffer
    int copy_issued = -1;                    // view whose D2H copy the previous fetch has already issued (fetch thread only)
    l3d_match* copy_dst = nullptr;           // ... and where its records go (pinned arena)
    double t_wait = 0, t_copy = 0, t_cb = 0, t_enq = 0, t_ex = 0;   // host-side phase timers (L3D_TIMING=1)
    int outcome[3] = { 0, 0, 0 };            // after l3d_shard_chain_run: OR of the ranks' overflow bits (8: the compact arena of the ring mode), largest candidate / kept count of a slot
    // ring mode of l3d_shard_chain_run (commit on the device only): the gathered buffer holds `ring` view blocks, older ones are retired
    std::mutex cap_mu;                       // held while the chain's stream is capturing: HIP refuses another thread's wait on an event of a capturing stream
    bool defer_stats = false, use_graphs = false;   // l3d_shard_chain_run: no host wait for a view's stage-1 statistics; graph replay of repeated passes
    bool ring_mode = false;
    int window = 0;                          // max over views of (index - smallest source index): how far back a view reads
    long long arena_cap = 0, arena_needed = 0;
    long long* base_dev = nullptr;           // arena offset of every view (+ the total behind the last), kept on the device
    SlotHeader* hdr_all = nullptr;           // the headers of all slots of all views (32 B each)
    const long long* best_off_dev = nullptr;
    const unsigned char* ver_dev = nullptr;
    bool retire_tables = false;              // the retire kernel also files the slots' side words and run tables (ch_keptcam, ch_rt): the products need not rebuild them
    std::vector<long long> rt_off;           // where view k's run table starts in ch_rt (ints)
    // partitioned retirement (l3d_shard_chain_partition): only the views flagged in `keep` go to this rank's compact arena
    bool partition = false;
    int part_own0 = 0, part_own1 = 0, part_reach = 0;
    std::vector<unsigned char> keep;
    const unsigned char* keep_dev = nullptr;
    l3d_exchange_fn part_exchange = nullptr;   // the run's exchange: l3d_shard_chain_products' one status exchange
    void* part_user = nullptr;
    unsigned char* part_status = nullptr;
};

extern "C" {

void* l3d_ctx_stream(l3d_ctx* c) { return c ? (void*)c->stream : nullptr; }

int l3d_shard_chain_open(l3d_ctx* c, const l3d_chain_view* views, int n_views, int rank, int world, int slot_records,
                         l3d_shard_chain** out, size_t* slot_bytes)
{
    if (!c) return L3D_ERR_INVALID;
    if (!out || !slot_bytes || n_views < 0 || world < 1 || rank < 0 || rank >= world || (n_views > 0 && !views) || slot_records < 1)
        return fail(c, L3D_ERR_INVALID, "l3d_shard_chain_open: bad argument");
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    (void)hipGetLastError();
    l3d_shard_chain* h = new l3d_shard_chain();
    h->c = c; h->views = views; h->n_views = n_views; h->rank = rank; h->world = world;
    ChainLayout L;
    { int rc = chain_plan_views(c, views, n_views, rank, world, h->vd, L, "l3d_shard_chain_open"); if (rc) { delete h; return rc; } }
    h->maxS = L.maxS; h->maxN = L.maxN; h->pairs = L.pairs;
    const double max_pairs = L.max_pairs;
    h->geom.world = world;
    h->geom.ring = std::max(1, n_views);
    for (int k = 0; k < n_views; ++k)
        for (int q = 0; q < views[k].n_sources; ++q) h->window = std::max(h->window, k - views[k].source_index[q]);
    h->geom.seg_cap = (h->maxS + world - 1) / world + 1;
    h->geom.slot_records = slot_records;
    h->geom.best_off = sizeof(SlotHeader);
    h->geom.bpos_off = h->geom.best_off + (size_t)h->geom.seg_cap * 8;
    h->geom.rec_off = salign(h->geom.bpos_off + (size_t)h->geom.seg_cap * 4, 32);
    // dense scenes (slots of 65 536 records and more): a side array of the records' target cameras travels with the slot -- the two scans every later
    // neighbour makes of it (k_exist_count_slots, k_place_slots) read 4 bytes per record instead of 32; + 12.5 % on the all-gather.  Measured on one
    // emulated rank of eight at 64 x 4000 x 24 (profiles/r5_emulated_rank_64x4000x24_w8_partition.txt)
    h->geom.cam_off = slot_records >= c->opt.slot_cams_min ? salign(h->geom.rec_off + (size_t)slot_records * sizeof(Match), 32) : 0;
    // ... and (round 6) the slot's RUN TABLE, (N + 1) x seg_cap ints (l3d_runtable.hpp; positions in the slot's records): a later neighbour reads the runs of its
    // camera -- 1/N of the slot -- instead of scanning the side array (188 MB per view and rank at 64 x 4000 x 24 on eight ranks)
    h->geom.max_n = h->maxN;
    h->geom.rt_off = h->geom.cam_off && c->opt.run_tables != 0 ? salign(h->geom.cam_off + (size_t)slot_records * 4, 32) : 0;
    h->geom.slot_bytes = salign(h->geom.rt_off ? h->geom.rt_off + ((size_t)h->maxN + 1) * h->geom.seg_cap * 4
                                : (h->geom.cam_off ? h->geom.cam_off + (size_t)slot_records * 4 : h->geom.rec_off + (size_t)slot_records * sizeof(Match)), 256);
    *slot_bytes = h->geom.slot_bytes;

    auto bail = [&](int rc) { delete h; return rc; };
#define OCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fail(c, L3D_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); return bail(L3D_ERR_HIP); } } while (0)
    // tables, target rays, per-view slices of the whole-run arenas (l3d_chain_common.hip: shared with the single-GPU chain)
    { int rc = chain_upload_tables(c, views, n_views, h->vd, L, true, st); if (rc) return bail(rc); }
    { int rc = chain_assign_arenas(c, views, n_views, h->vd, L, false, true, l3d_shard_chain::kRingA, st); if (rc) return bail(rc); }   // (+ best positions: l3d_shard_chain_products)
    h->dtab = L.dtab;
    OCHK(c->ch_pin_res.reserve((size_t)n_views * 8 + 64));
    h->hstats = c->ch_pin_res.as<int>();
    OCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->hstats_dev), h->hstats, 0));
    {   // stage 1 (own stream) starts after the tables and the zeroed row counts are in place
        hipEvent_t ready = get_event(c);
        OCHK(hipEventRecord(ready, st));
        OCHK(hipStreamWaitEvent(c->stage1_stream, ready, 0));
        put_event(c, ready);
    }
    h->cand_cap = c->test_cand_cap ? c->test_cand_cap : chain_first_cand_cap(max_pairs);     // (l3d_set_chain_capacities: tests, retries with more room)
    { int rc = chain_reserve_candidates(c, L, h->cand_cap, l3d_shard_chain::kRingA); if (rc) return bail(rc); }
    h->stage_bytes = salign((size_t)world * ((size_t)h->geom.seg_cap * 8 + (size_t)slot_records * sizeof(Match)), 256);
    OCHK(c->ch_pin_kept.reserve(2 * ((size_t)world * (size_t)h->geom.seg_cap * 8 + 64) + 64));
    c->pin_arena.reset();
    OCHK(c->ch_pin_best.reserve((size_t)n_views * sizeof(PackHeader) + 64));
    h->hdr_host = c->ch_pin_best.as<PackHeader>();
    OCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->hdr_dev), h->hdr_host, 0));
    h->packed.assign((size_t)n_views, 0);
#undef OCHK
    h->ev1.assign((size_t)n_views, nullptr);
    h->ev2.assign((size_t)n_views, nullptr);
    for (int k = 0; k < n_views; ++k) { h->ev2[(size_t)k] = get_event(c); h->ev1[(size_t)k] = get_event(c); }
    h->ev3 = get_event(c);
    c->stats[0] = h->pairs;
    *out = h;
    return L3D_OK;
}

static PairArgs shard_pair_args(l3d_shard_chain* h, int k) { return chain_pair_args(h->c, h->views[k], h->vd[(size_t)k], h->dtab); }

static int shard_stage1(l3d_shard_chain* h, int k)
{
    l3d_ctx* c = h->c;
    hipStream_t s1 = c->stage1_stream;
    const SViewDev& d = h->vd[(size_t)k];
    if (!d.verified) return L3D_OK;
    h->hstats[2 * k] = h->hstats[2 * k + 1] = 0;
    if (d.s1 > d.s0) {
        const PairArgs pa = shard_pair_args(h, k);
        {   // bit rows + row counts (added into the rows zeroed when the chain was opened) in one launch
            PairArgs pm = pa;
            pm.rowcnt = d.rowcnt;
            ProfScope p(c, "pair_mask", s1);
            launch_pair_mask(pm, d.maxW, s1, c->opt.pair_spb);
        }
        // row starts of the stage-1 candidates of the rank's rows + their statistics straight into host-mapped memory (one launch)
        { ProfScope p(c, "scan", s1); launch_scan_range(d.rowcnt, d.rowA, h->views[k].N, d.s0, d.s1, h->views[k].S_src * h->views[k].N, nullptr, nullptr, s1, h->hstats_dev + 2 * k); }
        // depth records of the stage-1 candidates, in their own row order, into the ring slot last used by view k - kRingA
        // (its completion event is recorded by l3d_shard_chain_mark before this view's stage 1 is enqueued)
        if (k - l3d_shard_chain::kRingA >= 0) {
            std::lock_guard<std::mutex> lk(h->cap_mu);          // (the event belongs to the chain's stream: not while that stream captures a view's launches)
            HIPCHK(c, hipStreamWaitEvent(s1, h->ev2[(size_t)(k - l3d_shard_chain::kRingA)], 0));
        }
        PairArgs pf = pa;
        pf.cand_cap = (int)h->cand_cap;
        pf.rowcnt = d.rowcnt;           // (the row's true count replaces k_pair_mask's upper bound)
        { ProfScope p(c, "pair_fill", s1); launch_pair_fill(pf, d.rowA, c->ch_ringA_meta.as<uint2>() + (size_t)(k % l3d_shard_chain::kRingA) * h->cand_cap,
                                                             c->ch_ringA_depths.as<float4>() + (size_t)(k % l3d_shard_chain::kRingA) * h->cand_cap, s1); }
    }
    HIPCHK(c, hipEventRecord(h->ev1[(size_t)k], s1));
    return L3D_OK;
}

// Enqueue view k up to (and including) the write of this rank's slot.  send_slot: device buffer of slot_bytes that the
// caller all-gathers into gathered_base + (k*world + r)*slot_bytes for r = 0..world-1 (stream ordered, same stream).
int l3d_shard_chain_enqueue(l3d_shard_chain* h, int k, void* send_slot, const void* gathered_base)
{
    if (!h || k < 0 || k >= h->n_views || !gathered_base) return L3D_ERR_INVALID;
    l3d_ctx* c = h->c;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    (void)hipGetLastError();            // the framework shares this thread: its (benign) sticky errors are not ours
    if (!h->s1_thread)
        while (h->k_p1 < h->n_views && h->k_p1 <= k + l3d_shard_chain::kStage1Ahead) { int rc = shard_stage1(h, h->k_p1); if (rc) return rc; ++h->k_p1; }
    else
        wait_until([&]() { return h->s1_done.load(std::memory_order_acquire) > k; }, 4000);     // the stage-1 thread enqueues it (normally long done)
    const l3d_chain_view& v = h->views[k];
    const SViewDev& d = h->vd[(size_t)k];
    h->gathered = reinterpret_cast<const unsigned char*>(gathered_base);
    if (!d.verified) return L3D_OK;
    if (!send_slot) return fail(c, L3D_ERR_INVALID, "l3d_shard_chain_enqueue: null slot");
    // The native run (l3d_shard_chain_run) never waits for a view's stage-1 statistics on the host: they only size the LDS image of the
    // verification, which the budget caps anyway (-1: the largest image the budget allows), and the candidate total is summed when the run
    // is over.  The step-wise protocol keeps the wait (its callers read the statistics between the steps).
    const bool deferred = h->defer_stats;
    if (!deferred) {
        HIPCHK(c, hipEventSynchronize(h->ev1[(size_t)k]));
        h->raw_sum += h->hstats[2 * k];
    }
    HIPCHK(c, hipStreamWaitEvent(st, h->ev1[(size_t)k], 0));
    PairArgs pa = shard_pair_args(h, k);
    pa.cand_cap = (int)h->cand_cap;
    const int S = v.S_src, N = v.N;
    const size_t nrow = (size_t)S * N;
    const unsigned char* dtab = h->dtab;
    const int* d_sc = reinterpret_cast<const int*>(dtab + d.o_sc);
    const int* d_si = reinterpret_cast<const int*>(dtab + d.o_si);
    unsigned char* slot = reinterpret_cast<unsigned char*>(send_slot);
    int mmax = 0;
    // this rank's kernels of view k: the same sequence whether it is launched call by call or replayed as a graph
    // workgroups per (source view, rank) list of the two scans of the sources' slots: 16 for the lists of a sparse scene (a few thousand records: 5 steps
    // per thread), one per 4096 records of the slot's capacity on a dense one -- 650 k records per slot at 4000 segments x 24 neighbours were 160
    // dependent loads per thread (measured on one emulated rank of eight, 64 x 4000 x 24: exist + cand_move 47 -> see profiles/r5_emulated_rank_*)
    // workgroups per (source, rank) list: a scan wants them by the slot's records; with run tables a list is seg_cap runs of sixteen lanes
    const int wps = h->geom.rt_off ? std::max(1, std::min(256, (h->geom.seg_cap * 16 + 255) / 256))
                                   : std::max(16, std::min(256, h->geom.slot_records / (h->c->opt.slot_scan_grain > 0 ? h->c->opt.slot_scan_grain : 4096)));
    auto issue = [&]() {
        if (v.n_sources) {
            ProfScope p(c, "exist");
            hipLaunchKernelGGL(k_exist_count_slots, dim3(wps, v.n_sources * h->world), dim3(256), 0, st, h->gathered, h->geom, d_si, d_sc, reinterpret_cast<const int*>(h->dtab + d.o_ss), v.view_id, N, d.s0, d.s1, d.rowcnt);
        }
        // row starts of this rank's rows only (+ zeroed scatter cursors, segment order); row_start[nrow] = their total
        { ProfScope p(c, "scan"); launch_scan_range(d.rowcnt, c->row_start.as<int>(), N, d.s0, d.s1, (int)nrow, c->ch_cursor.as<int>(), c->ch_segorder.as<int>(), st); }
        {
            ProfScope p(c, "cand_move");
            const int blocks_move = ((d.s1 - d.s0) * v.n_tbm + 3) / 4;
            const int blocks = blocks_move + wps * v.n_sources * h->world;
            if (blocks > 0)
                hipLaunchKernelGGL(k_place_slots, dim3(blocks), dim3(256), 0, st, blocks_move, wps, pa.tbm, v.n_tbm, d.rowA,
                                   c->ch_ringA_meta.as<uint2>() + (size_t)(k % l3d_shard_chain::kRingA) * h->cand_cap,
                                   c->ch_ringA_depths.as<float4>() + (size_t)(k % l3d_shard_chain::kRingA) * h->cand_cap,
                                   h->gathered, h->geom, d_si, d_sc, reinterpret_cast<const int*>(h->dtab + d.o_ss), v.view_id, N, S, d.s0, d.s1, c->row_start.as<int>(), c->ch_cursor.as<int>(),
                                   c->cand_meta.as<uint2>(), c->cand_depths.as<float4>(), (int)h->cand_cap);
        }
        // (the window kernel orders the runs itself -- except on a rank's small launch of a dense scene: a segment's workgroup ranks its 12 runs three per
        // wave, one after the other, and the launch waits for the heaviest segment; a launch of one wave per run is balanced.  Same rule as the split verification)
        const bool sort_apart = v.n_sources && c->opt.exist_sort_apart != 0 && (c->opt.exist_sort_apart > 0 ||
                                ((d.s1 - d.s0) <= c->opt.vw_wide_max && h->cand_cap / (size_t)std::max(1, d.s1 - d.s0) >= (size_t)c->opt.vw_split_avg));
        if (v.n_sources && (!(c->verify_mode == 0 && verify_window_supported(N)) || sort_apart)) {
            ProfScope p(c, "exist");
            launch_exist_sort_runs(d_sc, v.n_sources, N, S, c->row_start.as<int>(), c->cand_meta.as<uint2>(), c->cand_depths.as<float4>(), (int)h->cand_cap, st, d.s0, d.s1,
                                   c->vw_scratch.as<float>(), (long long)h->cand_cap + kVWSlack, c->cand_conf.as<unsigned>());
        }
        VerifyArgs va = chain_verify_args(c, v, d, dtab, h->cand_cap);
        chain_launch_verify(c, va, d, d_sc, sort_apart ? 0 : v.n_sources, deferred ? -1 : h->hstats[2 * k + 1], h->cand_cap, st);
        mmax = va.mmax;
        {
            ProfScope p(c, "kept_write");
            hipLaunchKernelGGL(k_slot_write, dim3(std::max(1, d.s1 - d.s0)), dim3(256), 0, st, va, c->kept_cnt.as<int>() + d.s0, (int)nrow,
                               h->geom.slot_records, reinterpret_cast<const unsigned*>(dtab + d.o_l2g), d.best, h->geom, slot);
        }
    };
    // Passes over the same scene (bench.py's timed steps, a caller that re-runs matchViews) repeat the very same launches: the third pass on
    // replays the five launches of a view as ONE graph launch (captured during the second pass, when the arguments proved stable; keyed by a
    // checksum of everything the launches depend on).  A single compute3Dmodel never captures anything.  L3D_GRAPH=0: A/B.
    ShardGraph* G = nullptr;
    bool capture = false;
    if (h->use_graphs && !c->prof_on && c->opt.graph != 0) {
        if ((int)c->shard_graphs.size() < h->n_views) c->shard_graphs.resize((size_t)h->n_views);
        G = &c->shard_graphs[(size_t)k];
        unsigned long long sig = 1469598103934665603ull;
        auto mix = [&](unsigned long long x) { sig = (sig ^ x) * 1099511628211ull; };
        auto mixp = [&](const void* p) { mix((unsigned long long)(uintptr_t)p); };
        mix((unsigned long long)k); mixp(send_slot); mixp(gathered_base); mix(h->geom.slot_bytes); mix((unsigned long long)h->geom.ring); mix((unsigned long long)h->geom.slot_records);
        mix((unsigned long long)h->world); mix((unsigned long long)h->rank); mix(h->cand_cap); mix((unsigned long long)h->n_views);
        mixp(c->row_start.p); mixp(c->ch_cursor.p); mixp(c->ch_segorder.p); mixp(c->cand_meta.p); mixp(c->cand_depths.p); mixp(c->cand_conf.p); mixp(c->vw_scratch.p);
        mixp(c->kept_cnt.p); mixp(c->ch_ringA_meta.p); mixp(c->ch_ringA_depths.p); mixp(d.rowcnt); mixp(d.rowA); mixp(d.best); mixp(d.bestpos); mixp(dtab); mixp(d.src); mixp(d.tgt);
        mix((unsigned long long)S); mix((unsigned long long)N); mix((unsigned long long)v.n_tbm); mix((unsigned long long)v.n_sources); mix((unsigned long long)v.view_id);
        mix((unsigned long long)d.s0); mix((unsigned long long)d.s1); mix((unsigned long long)d.o_sc); mix((unsigned long long)d.o_si); mix((unsigned long long)d.o_l2g);
        mix((unsigned long long)__float_as_uint_host(v.sigma_p)); mix((unsigned long long)__float_as_uint_host(v.sigma_a)); mix((unsigned long long)__float_as_uint_host(v.spatial_k));
        mix((unsigned long long)c->verify_mode); mix((unsigned long long)verify_window_max_lds(c->opt.vw_lds)); mix((unsigned long long)c->opt.vw_wide_max); mix((unsigned long long)c->opt.vw_debug);
        mix((unsigned long long)c->opt.vw_split); mix((unsigned long long)c->opt.vw_unit); mix((unsigned long long)c->opt.vw_split_avg); mix((unsigned long long)c->opt.exist_sort_apart);
        if (G->exec && G->sig == sig) {
            if (hipGraphLaunch(G->exec, st) == hipSuccess) { ++c->shard_graph_launches; return L3D_OK; }
            (void)hipGetLastError();
            (void)hipGraphExecDestroy(G->exec); G->exec = nullptr; G->seen = -1000000;          // (never again on this context: launch call by call)
        }
        if (G->sig != sig) { if (G->exec) { (void)hipGraphExecDestroy(G->exec); G->exec = nullptr; } G->sig = sig; G->seen = 1; }
        else if (++G->seen == 2 && !G->exec) capture = true;
    }
    std::unique_lock<std::mutex> cap_lk(h->cap_mu, std::defer_lock);
    if (capture) cap_lk.lock();
    if (capture && hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        issue();
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipError_t ce = hipStreamEndCapture(st, &graph);
        cap_lk.unlock();
        if (ce == hipSuccess && graph) ce = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (ce == hipSuccess && exec && hipGraphLaunch(exec, st) == hipSuccess) { G->exec = exec; ++c->shard_graph_launches; return L3D_OK; }
        (void)hipGetLastError();
        if (exec) (void)hipGraphExecDestroy(exec);
        G->seen = -1000000;                                                                     // capture is not available here: call by call from now on
    } else if (capture) {
        (void)hipGetLastError();
        G->seen = -1000000;
    }
    if (cap_lk.owns_lock()) cap_lk.unlock();
    issue();
    { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail(c, L3D_ERR_HIP, std::string("shard enqueue view ") + std::to_string(k) + " (mmax " + std::to_string(mmax) + ", lds " + std::to_string(verify_window_lds_bytes(mmax, N)) + ", range " + std::to_string(d.s0) + "-" + std::to_string(d.s1) + "): " + hipGetErrorString(e_)); }
    return L3D_OK;
}

// Enqueue the hand-over of view k (k_pack_view) into staging buffer k % kRing.
static int shard_pack(l3d_shard_chain* h, int k, hipStream_t st)
{
    l3d_ctx* c = h->c;
    const size_t block = (size_t)h->world * h->geom.slot_bytes;
    // the staging ring of the host hand-over: 16 views x the ranks' slots -- only a rank that commits on the host ever packs (13 GB at 4000 segments x
    // 24 neighbours: a rank that builds the products on its device never pays it)
    HIPCHK(c, c->ch_stage.reserve((size_t)l3d_shard_chain::kRing * h->stage_bytes + 64));
    hipLaunchKernelGGL(k_pack_view, dim3(8, h->world), dim3(256), 0, st, h->gathered + (size_t)k * block, h->geom,
                       c->ch_stage.as<unsigned char>() + (size_t)(k % l3d_shard_chain::kRing) * h->stage_bytes, h->hdr_dev + k);
    h->packed[(size_t)k] = 1;
    return L3D_OK;
}

// Record "view k is complete" on the stream -- call after the all-gather of view k has been enqueued.
int l3d_shard_chain_mark(l3d_shard_chain* h, int k)
{
    if (!h || k < 0 || k >= h->n_views) return L3D_ERR_INVALID;
    l3d_ctx* c = h->c;
    if (h->eager_pack && h->vd[(size_t)k].verified) { int rc = shard_pack(h, k, c->stream); if (rc) return rc; }
    HIPCHK(c, hipEventRecord(h->ev2[(size_t)k], c->stream));
    h->marked.store(k + 1, std::memory_order_release);
    return L3D_OK;
}

// Wait for view k (host side only) and hand the ranks' kept lists, concatenated in rank (= segment) order, and the depth
// pairs to the callback.  Views must be fetched in ascending order.
int l3d_shard_chain_fetch(l3d_shard_chain* h, int k, l3d_chain_callback cb, void* user)
{
    if (!h || k < 0 || k >= h->n_views || !cb) return L3D_ERR_INVALID;
    l3d_ctx* c = h->c;
    if (!h->vd[(size_t)k].verified) {
        h->fetched.store(k + 1, std::memory_order_release);
        return cb(user, k, 0, nullptr, 0, nullptr, 0, 0) ? fail(c, L3D_ERR_INVALID, "callback failed") : L3D_OK;
    }
    if (h->marked.load(std::memory_order_acquire) <= k) return fail(c, L3D_ERR_INVALID, "l3d_shard_chain_fetch: view not marked");
    HIPCHK(c, hipSetDevice(c->device));
    const double tf0 = now_s();
    if (h->packed[(size_t)k]) {
        HIPCHK(c, hipEventSynchronize(h->ev2[(size_t)k]));
    } else {                                        // step-wise protocol: pack now, behind whatever the stream has queued
        int rc = shard_pack(h, k, c->stream); if (rc) return rc;
        HIPCHK(c, hipEventRecord(h->ev3, c->stream));
        HIPCHK(c, hipEventSynchronize(h->ev3));
    }
    const double tf1 = now_s();
    const PackHeader ph = h->hdr_host[k];
    if (ph.overflow) return fail(c, L3D_ERR_NOMEM, "l3d_shard_chain: view " + std::to_string(k) + ": a rank ran out of candidate or slot capacity, or gave up (l3d_shard_chain_info tells which)");
    const size_t best_bytes = (size_t)h->geom.seg_cap * 8;
    const size_t pin_bytes = (size_t)h->world * best_bytes + 64;
    unsigned char* host = c->ch_pin_kept.as<unsigned char>() + (size_t)(k & 1) * pin_bytes;      // depth pairs: two small pinned buffers
    // the records go straight into the pinned arena (valid for the caller until the next chain), the depth pairs to staging
    auto issue_copy = [&](int v, const PackHeader& hd, l3d_match** dst) -> int {
        const unsigned char* stage = c->ch_stage.as<unsigned char>() + (size_t)(v % l3d_shard_chain::kRing) * h->stage_bytes;
        hipError_t ae = hipSuccess;
        *dst = static_cast<l3d_match*>(c->pin_arena.alloc((size_t)hd.n_kept * sizeof(Match) + 16, &ae));
        HIPCHK(c, ae);
        HIPCHK(c, hipMemcpyAsync(c->ch_pin_kept.as<unsigned char>() + (size_t)(v & 1) * pin_bytes, stage, (size_t)h->world * best_bytes,
                                 hipMemcpyDeviceToHost, c->copy_stream));
        if (hd.n_kept) HIPCHK(c, hipMemcpyAsync(*dst, stage + (size_t)h->world * best_bytes, (size_t)hd.n_kept * sizeof(Match), hipMemcpyDeviceToHost, c->copy_stream));
        return L3D_OK;
    };
    l3d_match* kept_dst = h->copy_dst;
    if (h->copy_issued != k) { int rc = issue_copy(k, ph, &kept_dst); if (rc) return rc; }       // (otherwise the previous fetch has issued it already)
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    h->fetched.store(k + 1, std::memory_order_release);          // the staging buffer of view k may be reused
    // the next view's copy travels while this view's bookkeeping runs, if the GPU is already done with it
    if (k + 1 < h->n_views && h->vd[(size_t)(k + 1)].verified && h->marked.load(std::memory_order_acquire) > k + 1 && h->packed[(size_t)(k + 1)] &&
        hipEventQuery(h->ev2[(size_t)(k + 1)]) == hipSuccess && !h->hdr_host[k + 1].overflow) {
        int rc = issue_copy(k + 1, h->hdr_host[k + 1], &h->copy_dst); if (rc) return rc;
        h->copy_issued = k + 1;
    } else {
        (void)hipGetLastError();                                 // hipEventQuery reports "not ready" as an error
    }
    const l3d_match* kept = kept_dst;
    std::vector<float>& best = h->best_scratch;
    best.clear();
    if (ph.R > 0)
        for (int r = 0; r < h->world; ++r) {
            const int s0 = (int)(((long long)h->views[k].S_src * r) / h->world), s1 = (int)(((long long)h->views[k].S_src * (r + 1)) / h->world);
            const float* b = reinterpret_cast<const float*>(host + (size_t)r * best_bytes);
            for (int s = 0; s < s1 - s0; ++s)
                if (b[2 * s] != -1.0f) { best.push_back(b[2 * s]); best.push_back(b[2 * s + 1]); }
        }
    const double tf2 = now_s();
    h->kept_total += (double)ph.n_kept;
    if (cb(user, k, 1, kept, ph.n_kept, best.data(), (int)(best.size() / 2), (int)std::min<long long>(ph.R, 0x7fffffff)))
        return fail(c, L3D_ERR_INVALID, "callback failed");
    h->t_wait += tf1 - tf0; h->t_copy += tf2 - tf1; h->t_cb += now_s() - tf2;
    return L3D_OK;
}

// ---- the whole sharded chain as ONE native call -------------------------------------------------------------------
// The calling thread only enqueues: this rank's kernels of view k, then the exchange of the ranks' slots on the same
// stream (RCCL all-gather over xGMI through l3d_exchange_rccl), then the "view complete" event.  A second host thread
// trails behind on those events and does the host bookkeeping (fetch -> callback) of the ranks that commit.  No
// interpreter, no host synchronisation on the enqueue path: per view the stream sees ~9 kernels + one collective.
int l3d_shard_chain_run(l3d_shard_chain* h, l3d_exchange_fn exchange, void* exchange_user, l3d_chain_callback cb, void* cb_user)
{
    if (!h || !exchange) return L3D_ERR_INVALID;
    l3d_ctx* c = h->c;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t slot = h->geom.slot_bytes, block = slot * (size_t)h->world;
    // Ring mode: when the gathered blocks of all views exceed the budget (or L3D_SLOT_RING=1) and nobody commits on the host, the buffer
    // holds window + batch + 2 view blocks; a batch of older views is retired into the compact kept arena (k_shard_retire) before its
    // blocks are reused.  L3D_SLOT_RING=0 keeps every block (what the recording tests read back through l3d_shard_chain_gathered).
    constexpr int kRetireBatch = 16;
    const int ring_views = std::min(h->window + kRetireBatch + 2, std::max(1, h->n_views));
    const size_t gathered_budget = (size_t)8 << 30;
    h->ring_mode = !cb && c->opt.slot_ring != 0 && ring_views < h->n_views && (c->opt.slot_ring > 0 || (size_t)h->n_views * block > gathered_budget);
    if (h->partition) {      // a partitioned job retires by definition: what a rank does not keep is gone when its ring block is reused
        if (cb) return fail(c, L3D_ERR_UNSUPPORTED, "l3d_shard_chain_run: a partitioned run hands nothing to the host (cb must be NULL)");
        h->ring_mode = true;  // (a schedule that reads further back than it has views -- scattered neighbourhoods -- gets a ring that never wraps: all retired at the end)
    }
    h->geom.ring = h->ring_mode ? ring_views : std::max(1, h->n_views);
    const int send_ring = h->ring_mode ? ring_views : h->n_views;            // (a slot is read by its exchange only)
    if (h->ring_mode) {      // (rings of whole slots: GBs whose size is exact)
        HIPCHK(c, c->ch_send.reserve_exact((size_t)send_ring * slot + 256));
        HIPCHK(c, c->ch_gathered.reserve_exact((size_t)h->geom.ring * block + 256));
    } else {
        HIPCHK(c, c->ch_send.reserve((size_t)send_ring * slot + 256));
        HIPCHK(c, c->ch_gathered.reserve((size_t)h->geom.ring * block + 256));
    }
    h->eager_pack = cb != nullptr;
    h->defer_stats = c->opt.defer_stats != 0;
    h->use_graphs = cb == nullptr && h->defer_stats;                  // (a committing rank's pack kernel belongs to the mark, not to the view's sequence)
    unsigned char* send = c->ch_send.as<unsigned char>();
    unsigned char* gathered = c->ch_gathered.as<unsigned char>();
    // every verified view's block is fully written by its exchange before anything reads it; only the blocks of views
    // that are never verified (nothing to match) must read as "no kept records"
    if (!h->ring_mode)
        for (int k = 0; k < h->n_views; ++k)
            if (!h->vd[(size_t)k].verified) HIPCHK(c, hipMemsetAsync(gathered + (size_t)k * block, 0, block, c->stream));
    std::vector<unsigned char> ver_h((size_t)h->n_views);
    std::vector<long long> best_off_h((size_t)h->n_views, 0);
    int* ring_overflow = nullptr;
    unsigned char* part_status = nullptr;      // partitioned: [own word | the ranks' words], 256 B each
    size_t o_rto_ = 0, o_sn_ = 0;              // (ring mode: the run tables' offsets / the views' (S, N) in the header block)
    h->retire_tables = false;
    if (h->ring_mode) {
        // compact arena (first guess like the single-GPU chain's, or what an earlier pass / a capacity verdict taught), the per-view offsets,
        // the header table, the flags the retire kernel reads
        h->arena_cap = c->test_arena_cap ? (long long)c->test_arena_cap : std::max((long long)(h->pairs * h->world * 0.004) + 1048576, (long long)c->chain_seen_arena_cap);
        if (!h->partition && !c->test_arena_cap && c->opt.arena_guess > 4) {
            // (as the single-GPU chain: a generous first guess within 35 % of the free HBM -- here a guess that is too small costs a capacity verdict and the whole chain again)
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
                const long long want = (long long)(h->pairs * h->world * 0.001 * c->opt.arena_guess) + 1048576, fits = (long long)((double)fr * 0.35 / 40.0);
                h->arena_cap = std::min<long long>(0xfffffff0ll, std::max(h->arena_cap, std::min(want, fits)));
            }
            (void)hipGetLastError();
        }
        if (!h->partition) HIPCHK(c, c->ch_kept.reserve(((size_t)h->arena_cap + 64) * sizeof(Match)));
        auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t nvs = (size_t)h->n_views;
        const size_t o_base = 0, o_hdr = al((nvs + 2) * 8), o_bo = o_hdr + al(nvs * h->world * sizeof(SlotHeader)), o_ver = o_bo + al(nvs * 8), o_ovf = o_ver + al(nvs), o_keep = o_ovf + 256, o_stat = o_keep + al(nvs),
                     o_rto = o_stat + 256 * ((size_t)h->world + 1), o_sn = o_rto + al(nvs * 8), tot = o_sn + al(nvs * 8);
        HIPCHK(c, c->ch_hdr.reserve(tot));
        unsigned char* hb = c->ch_hdr.as<unsigned char>();
        h->base_dev = reinterpret_cast<long long*>(hb + o_base);
        h->hdr_all = reinterpret_cast<SlotHeader*>(hb + o_hdr);
        for (int k = 0; k < h->n_views; ++k) {
            ver_h[(size_t)k] = h->vd[(size_t)k].verified ? 1 : 0;
            best_off_h[(size_t)k] = h->vd[(size_t)k].verified ? (long long)(h->vd[(size_t)k].best - c->ch_best.as<float2>()) : 0;
        }
        HIPCHK(c, hipMemsetAsync(hb, 0, o_bo, c->stream));
        HIPCHK(c, hipMemsetAsync(hb + o_ovf, 0, 256, c->stream));
        HIPCHK(c, hipMemcpyAsync(hb + o_bo, best_off_h.data(), nvs * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(hb + o_ver, ver_h.data(), nvs, hipMemcpyHostToDevice, c->stream));
        h->best_off_dev = reinterpret_cast<const long long*>(hb + o_bo);
        h->ver_dev = hb + o_ver;
        h->keep_dev = nullptr;
        if (h->partition) {
            HIPCHK(c, hipMemcpyAsync(hb + o_keep, h->keep.data(), nvs, hipMemcpyHostToDevice, c->stream));
            h->keep_dev = hb + o_keep;
            // (the first guess of the arena: this rank's share of the scene's)
            if (!c->test_arena_cap) { long long kv = 0; for (unsigned char x : h->keep) kv += x; h->arena_cap = std::max<long long>((long long)c->part_arena_seen, h->arena_cap * kv / std::max(1, h->n_views) + 1048576); }
            HIPCHK(c, c->ch_kept.reserve_exact(((size_t)h->arena_cap + 64) * sizeof(Match)));
        }
        // the retired views' side words and run tables (slots that carry both: L3D_RUN_TABLES, the default)
        h->retire_tables = h->geom.rt_off != 0 && h->geom.cam_off != 0 && c->opt.prod_transpose != 0 && c->opt.retire_tables != 0;
        if (h->retire_tables) {
            h->rt_off.assign(nvs, 0);
            std::vector<int2> sn(nvs);
            long long rt_ints = 0;
            for (int k = 0; k < h->n_views; ++k) {
                sn[(size_t)k] = make_int2(h->views[k].S_src, h->views[k].N);
                h->rt_off[(size_t)k] = rt_ints;
                if (h->vd[(size_t)k].verified && (!h->partition || h->keep[(size_t)k])) rt_ints += ((long long)h->views[k].N + 1) * h->views[k].S_src;
            }
            HIPCHK(c, c->ch_rt.reserve((size_t)rt_ints * 4 + 256));
            HIPCHK(c, c->ch_keptcam.reserve(((size_t)h->arena_cap + 64) * 4));
            HIPCHK(c, hipMemcpyAsync(hb + o_rto, h->rt_off.data(), nvs * 8, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(hb + o_sn, sn.data(), nvs * 8, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));         // (`sn` leaves scope)
        }
        o_rto_ = o_rto; o_sn_ = o_sn;
        ring_overflow = reinterpret_cast<int*>(hb + o_ovf);
        part_status = hb + o_stat;
        h->part_status = part_status; h->part_exchange = exchange; h->part_user = exchange_user;
    }
    int retired = 0;                        // views [0, retired) are in the compact arena (ring mode)
    // The retirement copies a batch's records out of the ring (9.6 GB of traffic per pass of an emulated rank of eight at 64 x 4000 x 24: 4.7 of its 63 ms) and
    // nothing of the chain reads what it writes: it runs on a side stream behind the batch's exchanges (round 6; L3D_RETIRE_APART=0: on the chain's stream).
    // The chain waits for it only before the first exchange that overwrites a block of the batch (view a + ring for a batch that starts at view a)
    hipStream_t rs = c->stream;
    if (h->ring_mode && c->opt.retire_apart != 0) {
        if (!c->prod_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->prod_stream, hipStreamNonBlocking));
        rs = c->prod_stream;
    }
    hipEvent_t retire_done = nullptr;       // the last batch's retirement on the side stream, not yet waited for
    int retire_need = 0;                    // ... the first view whose exchange needs it done
    auto retire_wait = [&]() {
        if (!retire_done) return;
        (void)hipStreamWaitEvent(c->stream, retire_done, 0);
        put_local_event(c, retire_done);
        retire_done = nullptr;
    };
    auto retire_to = [&](int upto) {        // enqueue the retirement of views [retired, upto) behind their exchanges
        while (retired < upto) {
            const int nb = std::min(kRetireBatch, upto - retired);
            if (rs != c->stream) {
                retire_wait();
                hipEvent_t e = get_local_event(c);
                (void)hipEventRecord(e, c->stream);
                (void)hipStreamWaitEvent(rs, e, 0);
                put_local_event(c, e);
            }
            ProfScope p(c, "retire", rs);
            hipLaunchKernelGGL(k_shard_retire, dim3(8, h->world, nb), dim3(256), 0, rs, gathered, h->geom, h->ver_dev, h->keep_dev, retired, nb, h->base_dev, h->arena_cap,
                               h->best_off_dev, c->ch_kept.as<Match>(), c->ch_best.as<float2>(), c->ch_bestpos.as<int>(), h->hdr_all, ring_overflow,
                               h->retire_tables ? c->ch_keptcam.as<unsigned>() : nullptr, h->retire_tables ? c->ch_rt.as<int>() : nullptr,
                               reinterpret_cast<const long long*>(c->ch_hdr.as<unsigned char>() + o_rto_), reinterpret_cast<const int2*>(c->ch_hdr.as<unsigned char>() + o_sn_));
            if (rs != c->stream) {
                retire_done = get_local_event(c);
                (void)hipEventRecord(retire_done, rs);
                retire_need = retired + h->geom.ring;
            }
            retired += nb;
        }
    };

    // the "gave up" header of the failure protocol below -- allocated BEFORE any helper thread exists: an early return with
    // joinable threads would terminate the process and leave the peers waiting in their collectives
    SlotHeader* abort_hdr = nullptr;
    {
        hipError_t ae = hipSuccess;
        abort_hdr = static_cast<SlotHeader*>(c->pin_arena.alloc(sizeof(SlotHeader), &ae));
        HIPCHK(c, ae);
        memset(abort_hdr, 0, sizeof(SlotHeader));
        abort_hdr->overflow = 4;
    }
    std::mutex mu;
    std::condition_variable cv;
    int marked = 0;                         // views [0, marked) carry their "complete" event
    bool stop = false;
    int fetch_rc = L3D_OK;
    std::string fetch_err;
    std::thread fetcher;
    if (cb) {
        fetcher = std::thread([&]() {
            (void)hipSetDevice(c->device);
            for (int k = 0; k < h->n_views; ++k) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&]() { return marked > k || stop; });
                    if (marked <= k) return;
                }
                const int rc = l3d_shard_chain_fetch(h, k, cb, cb_user);
                if (rc) { std::lock_guard<std::mutex> lk(mu); fetch_rc = rc; fetch_err = c->err; return; }
            }
        });
    }
    // a second enqueue thread feeds the stage-1 stream (7 of the ~17 calls per view): with small per-rank slices the host
    // launch rate, not the GPU, limits the chain otherwise.  It stays kStage1Ahead views ahead of the chain thread; the ring
    // slot it writes was released by view k - kRingA, whose completion event the chain thread has recorded by then.
    std::atomic<int> s1_rc{L3D_OK};
    std::atomic<bool> s1_stop{false};
    std::thread stage1;
    h->s1_thread = true;
    h->s1_done.store(0);
    stage1 = std::thread([&]() {
        (void)hipSetDevice(c->device);
        for (int k = 0; k < h->n_views && !s1_stop.load(); ++k) {
            wait_until([&]() { return s1_stop.load() || h->marked.load(std::memory_order_acquire) >= k - l3d_shard_chain::kStage1Ahead; }, 64);
            if (s1_stop.load()) break;
            const int r = shard_stage1(h, k);
            if (r) { s1_rc.store(r); break; }
            h->s1_done.store(k + 1, std::memory_order_release);
        }
        h->s1_done.store(h->n_views + 1, std::memory_order_release);      // (never leave the chain thread waiting)
    });
    int rc = L3D_OK;
    std::string rc_msg;
    const double t_run0 = now_s();
    // A rank that fails -- its own enqueue, its stage-1 thread, or (committing rank) the bookkeeping thread reading a slot that
    // overflowed -- must NOT stop calling the exchange: the other ranks keep enqueueing one collective per verified view and would
    // wait for it forever.  From the failure on it sends a header that says "gave up" for every remaining view (no kernels) and
    // keeps exchanging; k_check_slots then gives every rank the same verdict.  Only a failing exchange call itself ends the loop.
    bool draining = false, exchange_broken = false;
    for (int k = 0; k < h->n_views; ++k) {
        if (!draining) {
            // a committing rank stays less than a staging ring ahead of its bookkeeping thread (which trails the GPU closely)
            if (s1_rc.load()) { rc = s1_rc.load(); rc_msg = "l3d_shard_chain_run: stage 1 failed: " + c->err; draining = true; }
            while (!draining && cb && h->fetched.load(std::memory_order_acquire) < k - (l3d_shard_chain::kRing - 2)) {
                { std::lock_guard<std::mutex> lk(mu); if (fetch_rc) { draining = true; break; } }
                std::this_thread::yield();
            }
            { std::lock_guard<std::mutex> lk(mu); if (fetch_rc) draining = true; }
        }
        const bool verified = h->vd[(size_t)k].verified;
        if (!draining) {
            const double te0 = now_s();
            // ring mode: block k % ring is about to be overwritten by this view's exchange -- whatever lived there (view k - ring) and every
            // older view must be in the arena first (a batch at a time; no later view reads them: they are further back than the window)
            // (views in front of k - window are read by nobody from view k on; with ring = window + batch + 2 the block this view's exchange
            // overwrites, k - ring, is always among the retired)
            if (h->ring_mode && (k - h->window) - retired >= kRetireBatch) retire_to(k - h->window);
            if (retire_done && k >= retire_need) retire_wait();
            const int r2 = l3d_shard_chain_enqueue(h, k, send + (size_t)(k % send_ring) * slot, gathered);
            h->t_enq += now_s() - te0;
            if (r2) { rc = r2; rc_msg = c->err; draining = true; }
        }
        if (h->ring_mode && !verified && !draining)          // (a view that is never verified must read as "no kept records", whatever lived in its ring block)
            if (hipMemsetAsync(gathered + (size_t)(k % h->geom.ring) * block, 0, block, c->stream) != hipSuccess) { (void)hipGetLastError(); }
        if (draining && verified) {
            if (hipMemcpyAsync(send + (size_t)(k % send_ring) * slot, abort_hdr, sizeof(SlotHeader), hipMemcpyHostToDevice, c->stream) != hipSuccess) { (void)hipGetLastError(); }
        }
        if (verified) {
            const double te1 = now_s();
            if (const int exr = exchange(exchange_user, k, send + (size_t)(k % send_ring) * slot, gathered + (size_t)(k % h->geom.ring) * block, slot, h->world, (void*)c->stream)) {
                if (rc == L3D_OK) { rc = L3D_ERR_HIP; rc_msg = "l3d_shard_chain_run: the exchange of view " + std::to_string(k) + " failed (code " + std::to_string(exr) + ", last HIP error: " + hipGetErrorString(hipGetLastError()) + ")"; }
                exchange_broken = true;
                break;
            }
            h->t_ex += now_s() - te1;
        }
        if (!draining) {
            const int r3 = l3d_shard_chain_mark(h, k);
            if (r3) { rc = r3; rc_msg = c->err; draining = true; }
            else { { std::lock_guard<std::mutex> lk(mu); marked = k + 1; } cv.notify_one(); }
        }
    }
    s1_stop.store(true);
    stage1.join();
    h->s1_thread = false;
    if (rc == L3D_OK && s1_rc.load()) rc = fail(c, s1_rc.load(), "l3d_shard_chain_run: stage 1 failed");
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_one();
    const double t_run1 = now_s();
    (void)hipStreamSynchronize(c->stream);
    const double t_run2 = now_s();
    for (int k = 0; k < h->n_views; ++k) if (h->vd[(size_t)k].verified) h->raw_sum += h->hstats[2 * k];     // (deferred: the statistics are final now)
    h->defer_stats = false; h->use_graphs = false;
    if (fetcher.joinable()) fetcher.join();
    if (c->opt.timing)
        fprintf(stderr, "[l3d shard chain run] enqueue loop %.2f ms, stream drained after %.2f ms, bookkeeping thread done after %.2f ms\n",
                (t_run1 - t_run0) * 1e3, (t_run2 - t_run0) * 1e3, (now_s() - t_run0) * 1e3);
    if (rc == L3D_OK && fetch_rc) { rc = fetch_rc; rc_msg = fetch_err; }
    // the verdict every rank shares: the overflow bits of all gathered headers
    if (!exchange_broken) {
        std::vector<unsigned char> ver((size_t)h->n_views);
        for (int k = 0; k < h->n_views; ++k) ver[(size_t)k] = h->vd[(size_t)k].verified ? 1 : 0;
        hipError_t e = c->ch_flags.reserve((size_t)h->n_views + 64);
        int host3[3] = { 0, 0, 0 };
        int ring_ovf_h = 0;
        long long arena_total_h = 0;
        if (e == hipSuccess) e = hipMemsetAsync(c->ch_flags.p, 0, 16, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(c->ch_flags.as<unsigned char>() + 16, ver.data(), (size_t)h->n_views, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            if (h->ring_mode) {
                if (!draining) retire_to(h->n_views);                  // the views still in the ring
                retire_wait();
                if (retired == h->n_views) {
                    hipLaunchKernelGGL(k_check_slots, dim3(8), dim3(256), 0, c->stream, reinterpret_cast<const unsigned char*>(h->hdr_all), sizeof(SlotHeader), h->geom,
                                       c->ch_flags.as<unsigned char>() + 16, h->n_views, c->ch_flags.as<int>());
                    e = hipMemcpyAsync(&ring_ovf_h, ring_overflow, 4, hipMemcpyDeviceToHost, c->stream);
                    if (e == hipSuccess) e = hipMemcpyAsync(&arena_total_h, h->base_dev + h->n_views, 8, hipMemcpyDeviceToHost, c->stream);
                } else {
                    host3[0] = 4;                                      // this rank gave up before every view was exchanged: so did the verdict
                }
            } else {
                hipLaunchKernelGGL(k_check_slots, dim3(8), dim3(256), 0, c->stream, gathered, slot, h->geom, c->ch_flags.as<unsigned char>() + 16, h->n_views, c->ch_flags.as<int>());
            }
            if (e == hipSuccess && !(h->ring_mode && retired != h->n_views)) e = hipMemcpyAsync(host3, c->ch_flags.p, 12, hipMemcpyDeviceToHost, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (h->partition && part_status) {
            // every rank's arena is its own: "too small" is no shared verdict by itself -- one more exchange makes it one (entered by every rank,
            // whatever happened to it: tag -3, the status words of the block modes)
            long long w[2] = { e == hipSuccess ? (long long)ring_ovf_h : 1, 0 };
            hipError_t e2 = hipMemcpyAsync(part_status, w, 16, hipMemcpyHostToDevice, c->stream);
            if (e2 == hipSuccess) e2 = hipStreamSynchronize(c->stream);
            if (exchange(exchange_user, -3, part_status, part_status + 256, 256, h->world, (void*)c->stream)) { if (rc == L3D_OK) { rc = L3D_ERR_HIP; rc_msg = "l3d_shard_chain_run: the exchange of the arena verdicts failed"; } }
            else {
                for (int r = 0; r < h->world && e2 == hipSuccess; ++r) {
                    long long x = 0;
                    e2 = hipMemcpyAsync(&x, part_status + 256 * ((size_t)r + 1), 8, hipMemcpyDeviceToHost, c->stream);
                    if (e2 == hipSuccess) e2 = hipStreamSynchronize(c->stream);
                    if (e2 == hipSuccess && x) ring_ovf_h = 1;
                }
            }
            if (e == hipSuccess) e = e2;
        }
        if (e != hipSuccess) { if (rc == L3D_OK) { rc = L3D_ERR_HIP; rc_msg = std::string("l3d_shard_chain_run: reading the slot headers: ") + hipGetErrorString(e); } }
        else {
            if (ring_ovf_h) host3[0] |= 8;                          // the compact arena's first guess was too small (same data, same verdict on every rank)
            h->arena_needed = arena_total_h;
            h->outcome[0] = host3[0]; h->outcome[1] = host3[1]; h->outcome[2] = host3[2];
            if (host3[0] && (rc == L3D_OK || rc == L3D_ERR_NOMEM)) {
                rc = L3D_ERR_NOMEM;
                rc_msg = std::string("l3d_shard_chain_run:") + ((host3[0] & 1) ? " candidate capacity exceeded on a rank (" + std::to_string(host3[1]) + " candidates in one view's range; l3d_set_chain_capacities);" : "") +
                         ((host3[0] & 2) ? " slot_records too small (" + std::to_string(host3[2]) + " kept matches in one view's range);" : "") +
                         ((host3[0] & 8) ? " the compact kept arena of the slot ring is too small (" + std::to_string(arena_total_h) + " kept matches);" : "") +
                         ((host3[0] & 4) ? " a rank gave up;" : "");
            }
        }
    }
    if (retire_done) { (void)hipStreamSynchronize(rs); put_local_event(c, retire_done); retire_done = nullptr; }      // (a run that ended early: nothing of it stays in flight)
    return rc ? fail(c, rc, rc_msg) : L3D_OK;
}

// exchange adapters.  RCCL: user = l3d_rccl_link {communicator, address of ncclAllGather}; the library does not link
// against RCCL, the caller hands over what its process already has loaded (line3d_amd/distributed.py).
int l3d_exchange_rccl(void* user, int, const void* send_slot, void* recv_block, size_t slot_bytes, int, void* stream)
{
    const l3d_rccl_link* L = static_cast<const l3d_rccl_link*>(user);
    if (!L || !L->comm || !L->all_gather) return 1;
    typedef int (*all_gather_t)(const void*, void*, size_t, int, void*, void*);       // ncclAllGather(send, recv, count, dtype, comm, stream)
    return reinterpret_cast<all_gather_t>(L->all_gather)(send_slot, recv_block, slot_bytes, 1 /* ncclUint8 */, L->comm, stream);
}
// a single rank: the gathered block is the slot itself
int l3d_exchange_local(void*, int, const void* send_slot, void* recv_block, size_t slot_bytes, int world, void* stream)
{
    if (world != 1) return 1;
    return (int)hipMemcpyAsync(recv_block, send_slot, slot_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
}
// replay of a recorded run (user = device address of its gathered blocks): one rank of a world-W job measured on one GPU
int l3d_exchange_replay(void* user, int view, const void* send_slot, void* recv_block, size_t slot_bytes, int world, void* stream)
{
    const size_t block = slot_bytes * (size_t)world;
    //  the only negative tag a replay can answer is -3, the status words of a partitioned run, whose readers ask "did anybody report something"
    // and "how much in all": the replayed rank's words land in slot 0, its peers' stay zero.  Any other collective of the block modes indexes the block by
    // rank and would read the wrong slot: refused
    if (view < 0 && view != -3) { fprintf(stderr, "[l3d exchange_replay] tag %d: a replay holds gathered slots of views and answers status exchanges (-3) only\n", view); return 1; }
    if (view < 0) {      // a status-word exchange of a partitioned run (tag -3): the replayed rank alone, its peers report "nothing to report"
        hipError_t e0 = hipMemsetAsync(recv_block, 0, block, (hipStream_t)stream);
        if (e0 == hipSuccess) e0 = hipMemcpyAsync(recv_block, send_slot, slot_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        return (int)e0;
    }
    const hipError_t e = hipMemcpyAsync(recv_block, static_cast<const unsigned char*>(user) + (size_t)view * block, block, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) fprintf(stderr, "[l3d exchange_replay] view %d: hipMemcpyAsync(dst %p, src %p + %zu, %zu bytes): %s\n", view, recv_block, user, (size_t)view * block, block, hipGetErrorString(e));
    return (int)e;
}
const void* l3d_shard_chain_gathered(l3d_shard_chain* h) { return h && !h->ring_mode ? h->c->ch_gathered.p : nullptr; }     // (ring mode: most blocks are gone)
long long l3d_shard_chain_arena_needed(l3d_shard_chain* h) { return h ? h->arena_needed : 0; }
int l3d_shard_chain_info(l3d_shard_chain* h, size_t* cand_cap, int* slot_records, int* overflow_bits, int* max_candidates, int* max_kept)
{
    if (!h) return L3D_ERR_INVALID;
    if (cand_cap) *cand_cap = h->cand_cap;
    if (slot_records) *slot_records = h->geom.slot_records;
    if (overflow_bits) *overflow_bits = h->outcome[0];
    if (max_candidates) *max_candidates = h->outcome[1];
    if (max_kept) *max_kept = h->outcome[2];
    return L3D_OK;
}

// What matchViews leaves behind (l3d_match_chain_resident's products: potential correspondences, best matches, medians), built on THIS
// rank's device from the gathered slots of a finished, successful l3d_shard_chain_run: all views' kept records go into one arena in the
// order of the unsharded run, then the very builder of the single-GPU chain runs on it.  Every rank may call it (each then holds the
// full products and can run greedy selection / affinity fill / clustering); none hands a kept list to the host.
static int shard_products_local(l3d_shard_chain* h, const l3d_dense_map* map, l3d_chain_summary* summary, int64_t* n_pot, ProductsPart* part_out);

int l3d_shard_chain_products(l3d_shard_chain* h, const l3d_dense_map* map, l3d_chain_summary* summary, int64_t* n_pot)
{
    if (!h) return L3D_ERR_INVALID;
    l3d_ctx* c = h->c;
    if (!map || !summary) return fail(c, L3D_ERR_INVALID, "l3d_shard_chain_products: bad argument");
    if (!h->gathered || h->outcome[0]) return fail(c, L3D_ERR_INVALID, "l3d_shard_chain_products: no finished run without overflow on this chain");     // (a verdict every rank shares)
    if (!h->partition) return shard_products_local(h, map, summary, n_pot, nullptr);
    // partitioned: whatever this rank fails in on its own travels with ONE status exchange (tag -3) that every rank enters -- the ranks go on to the
    // collective finish together or not at all
    ProductsPart part;
    int64_t n_local = 0;
    const int rc = shard_products_local(h, map, summary, &n_local, &part);
    std::string err_local;
    if (rc) { std::lock_guard<std::mutex> lk(c->err_mu); err_local = c->err; }
    hipStream_t st = c->stream;
    long long w[2] = { rc ? -(long long)std::abs(rc) : (long long)n_local, 0 };
    hipError_t e = hipMemcpyAsync(h->part_status, w, 16, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) (void)hipMemsetAsync(h->part_status, 0xff, 16, st);
    if (!h->part_exchange || h->part_exchange(h->part_user, -3, h->part_status, h->part_status + 256, 256, h->world, (void*)st)) return fail(c, L3D_ERR_HIP, "l3d_shard_chain_products: the exchange of the status words failed");
    if (rc) return fail(c, rc, err_local);
    part.n_pot_all = 0;
    for (int r = 0; r < h->world; ++r) {
        long long x = 0;
        HIPCHK(c, hipMemcpyAsync(&x, h->part_status + 256 * ((size_t)r + 1), 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (x < 0) return fail(c, L3D_ERR_HIP, "l3d_shard_chain_products: rank " + std::to_string(r) + " failed (code " + std::to_string(-x) + ") while building its rows of the products");
        part.n_pot_all += x;
    }
    c->products.part = part;
    c->products.n_pot = n_local;
    c->products.valid = true;
    if (n_pot) *n_pot = n_local;
    return L3D_OK;
}

// the rank-local part: totals, (partitioned: release of the chain's scratch,) the builder of the single-GPU chain on this rank's arena
static int shard_products_local(l3d_shard_chain* h, const l3d_dense_map* map, l3d_chain_summary* summary, int64_t* n_pot, ProductsPart* part_out)
{
    l3d_ctx* c = h->c;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int nv = h->n_views;
    const size_t nvs = (size_t)nv;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    // scratch: verified flags | totals int2 | kept_base int | best_off i64
    const size_t o_ver = 0, o_tot = al(nvs), o_kb = o_tot + al(nvs * 8), o_bo = o_kb + al(nvs * 4), bytes = o_bo + al(nvs * 8);
    HIPCHK(c, c->g7.reserve(bytes + 64));
    unsigned char* sc = c->g7.as<unsigned char>();
    std::vector<unsigned char> ver(nvs);
    for (int k = 0; k < nv; ++k) ver[(size_t)k] = h->vd[(size_t)k].verified ? 1 : 0;
    HIPCHK(c, hipMemcpyAsync(sc + o_ver, ver.data(), nvs, hipMemcpyHostToDevice, st));
    if (h->ring_mode) hipLaunchKernelGGL(k_shard_totals, dim3((nv + 255) / 256), dim3(256), 0, st, reinterpret_cast<const unsigned char*>(h->hdr_all), sizeof(SlotHeader), h->geom, sc + o_ver, nv, reinterpret_cast<int2*>(sc + o_tot));
    else hipLaunchKernelGGL(k_shard_totals, dim3((nv + 255) / 256), dim3(256), 0, st, h->gathered, h->geom.slot_bytes, h->geom, sc + o_ver, nv, reinterpret_cast<int2*>(sc + o_tot));
    std::vector<int2> tot(nvs);
    HIPCHK(c, hipMemcpyAsync(tot.data(), sc + o_tot, nvs * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    std::vector<ChainResult> hres(nvs);
    std::vector<ProdChainView> pvh(nvs);
    std::vector<unsigned> kept_base(nvs, 0);
    std::vector<long long> best_off(nvs, 0);
    long long total = 0;
    for (int k = 0; k < nv; ++k) {
        const SViewDev& d = h->vd[(size_t)k];
        ChainResult& r = hres[(size_t)k];
        const bool here = !h->partition || h->keep[(size_t)k];            // (partitioned: the views this rank retired)
        r.kept_base = (unsigned)total; r.n_kept = d.verified && here ? tot[(size_t)k].x : 0; r.R = d.verified && here ? tot[(size_t)k].y : 0; r.overflow = 0;
        kept_base[(size_t)k] = (unsigned)total;
        total += r.n_kept;
        if (total > 0xfffffff0ll) return fail(c, L3D_ERR_UNSUPPORTED, "l3d_shard_chain_products: more than 2^32 kept matches on this rank (l3d_shard_chain_partition keeps a rank's share only)");
        best_off[(size_t)k] = d.verified ? (long long)(d.best - c->ch_best.as<float2>()) : 0;
        pvh[(size_t)k].verified = d.verified ? 1 : 0;
        pvh[(size_t)k].best = d.verified && here ? d.best : nullptr;
        pvh[(size_t)k].bestpos = d.verified && here ? d.bestpos : nullptr;
        pvh[(size_t)k].rt = h->ring_mode && h->retire_tables && d.verified && here ? c->ch_rt.as<int>() + h->rt_off[(size_t)k] : nullptr;
    }
    const unsigned* qt_arena = h->ring_mode && h->retire_tables ? c->ch_keptcam.as<unsigned>() : nullptr;
    if (h->ring_mode) {
        // the records are in the arena already (k_shard_retire, view by view in this very order)
        if (total != h->arena_needed) return fail(c, L3D_ERR_INVALID, "l3d_shard_chain_products: the retired records do not add up to the slot headers");
        h->kept_total = (double)total;
        if (h->partition) {
            // this rank's share: rows of the views within reach of its block, best matches / medians of every view it kept
            auto dense_of = [&](int k) { if (k >= nv) return map->n_views; const uint32_t* it = std::lower_bound(map->view_ids, map->view_ids + map->n_views, h->views[k].view_id); return (int)(it - map->view_ids); };
            ProductsPart part;
            part.active = true; part.rank = h->rank; part.world = h->world;
            part.own_dv0 = dense_of(h->part_own0); part.own_dv1 = dense_of(h->part_own1);
            part.row_dv0 = dense_of(std::max(0, h->part_own0 - h->part_reach)); part.row_dv1 = dense_of(std::min(nv, h->part_own1 + h->part_reach));
            part.held_dv0 = dense_of(std::max(0, h->part_own0 - 2 * h->part_reach)); part.held_dv1 = dense_of(std::min(nv, h->part_own1 + 2 * h->part_reach));
            int64_t n_local = 0;
            if (c->opt.part_release != 0) {
                // the job is sized by memory: what only the running chain needed -- the rings of slots, the candidate store and its ring, window scratch,
                // bit rows, viewing rays, row counters -- is given back before the products are built (as l3d_match_chain_partition does)
                (void)hipStreamSynchronize(st); (void)hipStreamSynchronize(c->stage1_stream);
                DevBuf* b[] = { &c->ch_gathered, &c->ch_send, &c->ch_ringA_meta, &c->ch_ringA_depths, &c->cand_meta, &c->cand_depths, &c->cand_conf, &c->vw_scratch, &c->ch_mask, &c->ch_rays, &c->ch_rowcnt, &c->ch_rowA };
                for (DevBuf* x : b) x->release();
                h->gathered = nullptr;
            }
            const int rc = build_products(c, h->views, nv, pvh.data(), hres.data(), map, summary, &n_local, part.row_dv0, part.row_dv1, reinterpret_cast<const char*>(h->keep.data()), qt_arena);
            if (rc) return rc;
            c->part_arena_seen = std::max(c->part_arena_seen, (size_t)total + (size_t)total / 8 + 65536);
            if (n_pot) *n_pot = n_local;
            if (part_out) *part_out = part;
            memcpy(c->ch_pin_res.as<ChainResult>(), hres.data(), (size_t)nv * sizeof(ChainResult));      // (what l3d_chain_kept_list reads)
            return L3D_OK;
        }
        c->chain_seen_arena_cap = std::max(c->chain_seen_arena_cap, (size_t)total + (size_t)total / 8 + 65536);
        return build_products(c, h->views, nv, pvh.data(), hres.data(), map, summary, n_pot, 0, -1, nullptr, qt_arena);
    }
    HIPCHK(c, c->ch_kept.reserve(((size_t)total + 64) * sizeof(Match)));
    HIPCHK(c, hipMemcpyAsync(sc + o_kb, kept_base.data(), nvs * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(sc + o_bo, best_off.data(), nvs * 8, hipMemcpyHostToDevice, st));
    if (nv > 0)
        hipLaunchKernelGGL(k_shard_pack_all, dim3(8, h->world, nv), dim3(256), 0, st, h->gathered, h->geom, sc + o_ver, reinterpret_cast<const unsigned*>(sc + o_kb),
                           reinterpret_cast<const long long*>(sc + o_bo), c->ch_kept.as<Match>(), c->ch_best.as<float2>(), c->ch_bestpos.as<int>());
    HIPCHK(c, hipStreamSynchronize(st));                                 // (the upload sources above are locals)
    HIPCHK(c, hipGetLastError());
    h->kept_total = (double)total;
    return build_products(c, h->views, nv, pvh.data(), hres.data(), map, summary, n_pot);
}

// A partitioned job on top of the segment-sharded run (round 5): every rank sees every view's gathered slots go by, so WHAT IT KEEPS is its own
// choice -- with this call: the views its block [own_begin, own_end) of the chain needs (2 x reach either side: DESIGN.md section 6 iii), the sources
// of the early-return views and the views their local camera numbers name (cudawrapper.cu:877-878, line3D.cc:861-865: whole lists, a few views).
// The run then retires only those into this rank's arena; l3d_shard_chain_products builds the rows of the block and its neighbours, the best
// matches and medians of the kept views -- the state l3d_match_chain_partition leaves, computed WITHOUT speculation (a scene whose chain never
// forgets a cold start -- the box scene at 4000 segments x 24 neighbours keeps half of its candidates and does not within 120 views -- gets the
// segment-sharded run's speed and the partition's memory).  l3d_affinity_fill_sharded follows as there.
int l3d_partition_keep_views(const l3d_chain_view* views, int nv, int own_begin, int own_end, unsigned char* keep, int* reach_out)
{
    // (host logic only: no context, no device -- tests/test_partition_keep_cpu.py runs it without a GPU)
    if (!views || !keep || nv < 0 || own_begin < 0 || own_end > nv || own_begin > own_end) return L3D_ERR_INVALID;
    std::vector<std::pair<unsigned, int>> idx((size_t)nv);
    for (int k = 0; k < nv; ++k) idx[(size_t)k] = { views[k].view_id, k };
    std::sort(idx.begin(), idx.end());
    auto chain_of = [&](unsigned id) { auto it = std::lower_bound(idx.begin(), idx.end(), std::make_pair(id, -1)); return it != idx.end() && it->first == id ? it->second : -1; };
    int reach = 1;
    for (int k = 0; k < nv; ++k)
        for (int q = 0; q < views[k].N; ++q) { const int j = views[k].local2global ? chain_of(views[k].local2global[q]) : -1; if (j >= 0) reach = std::max(reach, std::abs(j - k)); }
    std::fill(keep, keep + nv, (unsigned char)0);
    for (int k = std::max(0, own_begin - 2 * reach); k < std::min(nv, own_end + 2 * reach); ++k) keep[k] = 1;
    for (int k = 0; k < nv; ++k) {
        if (views[k].n_tbm != 0 || views[k].n_sources == 0) continue;          // an early-return view (cudawrapper.cu:877-878) ...
        keep[k] = 1;
        for (int q = 0; q < views[k].n_sources; ++q) {
            const int si = views[k].source_index[q];                            // ... its sources, whose lists hold the records that point at it ...
            if (si >= 0 && si < nv) keep[si] = 1;
            const int av = chain_of((unsigned)views[k].source_cam[q]);          // ... and the view its LOCAL camera number names (line3D.cc:861-865)
            if (av >= 0) keep[av] = 1;
        }
    }
    if (reach_out) *reach_out = reach;
    return L3D_OK;
}

int l3d_shard_chain_partition(l3d_shard_chain* h, int own_begin, int own_end)
{
    if (!h) return L3D_ERR_INVALID;
    l3d_ctx* c = h->c;
    const int nv = h->n_views;
    if (own_begin < 0 || own_end > nv || own_begin > own_end) return fail(c, L3D_ERR_INVALID, "l3d_shard_chain_partition: bad view range");
    h->keep.assign((size_t)nv, 0);
    int reach = 1;
    if (int rc = l3d_partition_keep_views(h->views, nv, own_begin, own_end, h->keep.data(), &reach)) return fail(c, rc, "l3d_shard_chain_partition: bad schedule");
    h->partition = true; h->part_own0 = own_begin; h->part_own1 = own_end; h->part_reach = reach;
    return L3D_OK;
}

int l3d_shard_chain_close(l3d_shard_chain* h)
{
    if (!h) return L3D_ERR_INVALID;
    l3d_ctx* c = h->c;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stage1_stream);
    (void)hipStreamSynchronize(c->stream);
    for (hipEvent_t e : h->ev1) put_event(c, e);
    for (hipEvent_t e : h->ev2) put_event(c, e);
    put_event(c, h->ev3);
    c->stats[1] = h->raw_sum; c->stats[3] = h->kept_total;
    if (c->opt.timing)
        fprintf(stderr, "[l3d shard chain rank %d/%d] enqueue %.2f  exchange-call %.2f | fetch: wait %.2f  d2h %.2f  callback %.2f ms\n",
                h->rank, h->world, h->t_enq * 1e3, h->t_ex * 1e3, h->t_wait * 1e3, h->t_copy * 1e3, h->t_cb * 1e3);
    delete h;
    return L3D_OK;
}

}  // extern "C"

void l3d::warm_chain_sharded() { touch_kernel(reinterpret_cast<const void*>(&l3d::k_exist_count_slots)); }
