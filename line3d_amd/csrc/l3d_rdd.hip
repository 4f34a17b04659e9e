// l3d_rdd.hip -- replicator_dynamics_diffusion (cudawrapper.h:73-74, cudawrapper.cu:1131-1191) with the SparseMatrix
// construction of performDiffusion (line3D.cc:1258, sparsematrix.cc:63-191) on the device.
//
// The reference builds W (entries sorted by column, a stable list sort by (j,i), sparsematrix.cc:81-86) and P (the column-sorted
// entries re-sorted by row, sparsematrix.cc:157-167) on the host and uploads three copies.  Here the edge list is uploaded once; the
// two orders are stable LSD radix sorts on the device (hipCUB; a stable sort has exactly one result, the one the reference's stable
// list sorts produce), the float4 entries and the first-entry tables (sparsematrix.cc:99-131) are built by kernels, and the result
// comes back as (i, j, w) records.  Round 1 sorted on 16 host threads: 13.6 ms of the 35 ms a diffusion of the config-2 affinity
// list (978 k entries) took; the device sorts take well under a millisecond.
#include "l3d_sort.hpp"

#include "l3d_ctx.hpp"
#include "l3d_geometry.hpp"

using namespace l3d;

namespace l3d {

// keys of the column order: (j, i); values: the entry's position in the input list
__global__ void k_rdd_keys_w(const l3d_edge* __restrict__ A, int nnz, int shift, unsigned long long* __restrict__ key, unsigned* __restrict__ val)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) { key[k] = ((unsigned long long)(unsigned)A[k].j << shift) | (unsigned)A[k].i; val[k] = (unsigned)k; }
}
// keys of the row order over the column-sorted list: (i, j); values: the position in the column order
__global__ void k_rdd_keys_p(const l3d_edge* __restrict__ A, const unsigned* __restrict__ ordW, int nnz, int shift, unsigned long long* __restrict__ key,
                             unsigned* __restrict__ val)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) { const l3d_edge e = A[ordW[k]]; key[k] = ((unsigned long long)(unsigned)e.i << shift) | (unsigned)e.j; val[k] = (unsigned)k; }
}
// The reference keeps three copies of the entries as float4 (row, col, val, 0) -- W by column, P and P' by row (cudawrapper.cu:1148)
// -- and every thread of every iteration walks them entry by entry: index tests to find the end of a row / column run, a
// search for the slot of the result.  The pattern never changes during the iteration, so here it is resolved ONCE: first entry and
// LENGTH of every row of P and column of W, the slot each entry's result goes to (tpos), and the values alone (4 bytes per entry
// instead of 16) in arrays of their own.  The arithmetic -- which products, in which order, into which slot -- is the reference's.
__global__ void k_rdd_build(const l3d_edge* __restrict__ A, const unsigned* __restrict__ ordW, const unsigned* __restrict__ ordP, int nnz,
                            float* __restrict__ Wval, int* __restrict__ Wcol, int2* __restrict__ Pij, float* __restrict__ Pval, float* __restrict__ Pval2,
                            int* __restrict__ startW, int* __restrict__ startP)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const l3d_edge w = A[ordW[k]];
    Wval[k] = w.w; Wcol[k] = w.j;
    if (k == 0 || A[ordW[k - 1]].j != w.j) startW[w.j] = k;
    const l3d_edge q = A[ordW[ordP[k]]];
    Pij[k] = make_int2(q.i, q.j); Pval[k] = q.w; Pval2[k] = q.w;
    if (k == 0 || A[ordW[ordP[k - 1]]].i != q.i) startP[q.i] = k;
}
// run lengths: the entry that ends a run knows where it started
__global__ void k_rdd_lens(const int* __restrict__ Wcol, const int2* __restrict__ Pij, int nnz, const int* __restrict__ startW, const int* __restrict__ startP,
                           int* __restrict__ lenW, int* __restrict__ lenP)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const int c = Wcol[k];
    if (k == nnz - 1 || Wcol[k + 1] != c) lenW[c] = k + 1 - startW[c];
    const int r = Pij[k].x;
    if (k == nnz - 1 || Pij[k + 1].x != r) lenP[r] = k + 1 - startP[r];
}
// K_sparseMat_diffusion_step stores the product of entry (x, y) in the FIRST entry of row y with column x (cudawrapper.cu:809-826): its slot
__global__ void k_rdd_tpos(const int2* __restrict__ Pij, int nnz, const int* __restrict__ startP, const int* __restrict__ lenP, int* __restrict__ tpos)
{
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= nnz) return;
    const int r = Pij[y].y, c = Pij[y].x;
    int t = -1;
    const int s0 = startP[r];
    if (s0 >= 0) {
        int lo = s0, hi = s0 + lenP[r];
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (Pij[mid].y < c) lo = mid + 1; else hi = mid; }
        if (lo < s0 + lenP[r] && Pij[lo].y == c) t = lo;
    }
    tpos[y] = t;
}
// K_sparseMat_row_normalization (cudawrapper.cu:717-762): one wave per row; the first 64 values are loaded by the lanes at once, the
// sum runs over them in entry order, one addition after the other (the order is part of the result)
__global__ __launch_bounds__(256) void k_rdd_rownorm(float* __restrict__ val, const int* __restrict__ start, const int* __restrict__ len, int num_rows)
{
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (y >= num_rows) return;
    const int s = start[y];
    if (s < 0) return;
    const int n = len[y];
    const float v = lane < n ? val[s + lane] : 0.0f;
    float sum = 0.0f;
    const int head = n < 64 ? n : 64;
    for (int k = 0; k < head; ++k) sum += __shfl(v, k);
    if (n > 64) {
        if (lane == 0) for (int i = s + 64; i < s + n; ++i) sum += val[i];
        sum = __shfl(sum, 0);
    }
    if (sum < kEpsG) sum = kEpsG;
    if (lane < n) val[s + lane] = v / sum;
    for (int i = s + 64 + lane; i < s + n; i += 64) val[i] = val[i] / sum;
}
// K_sparseMat_diffusion_step (cudawrapper.cu:765-829): one thread per entry (x, y): positional lock-step product of row y of P with
// column x of W -- as many terms as the shorter of the two runs --, times the entry's own value, stored in slot (y, x) of P'
__global__ void k_rdd_step(const int2* __restrict__ Pij, const float* __restrict__ Pval, const float* __restrict__ Wval, const int* __restrict__ startP,
                           const int* __restrict__ lenP, const int* __restrict__ startW, const int* __restrict__ lenW, const int* __restrict__ tpos,
                           float* __restrict__ Pout, int nnz)
{
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= nnz) return;
    const int2 ij = Pij[y];
    const int r = ij.y, c = ij.x;
    float mul = 0.0f;
    const int sp = startP[r], sw = startW[c];
    if (sp >= 0 && sw >= 0) {
        const int n = min(lenP[r], lenW[c]);
        for (int k = 0; k < n; ++k) mul += (Pval[sp + k] * Wval[sw + k]);
    }
    mul *= Pval[y];
    if (mul < kEpsG) mul = kEpsG;
    const int t = tpos[y];
    if (t >= 0) Pout[t] = mul;
}
__global__ void k_rdd_result(const int2* __restrict__ Pij, const float* __restrict__ Pval, int nnz, l3d_edge* __restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) { l3d_edge r; r.i = Pij[k].x; r.j = Pij[k].y; r.w = Pval[k]; out[k] = r; }
}

// after the diffusion (line3D.cc:1275-1301): A(i,j) = A(j,i) = min(W(i,j), W(j,i)); W sorted by (row, column), entries unique.
// The transposed entry is looked up by binary search; `bad` is raised when the list is not strictly ascending or an entry has no
// transposed partner (the caller then takes the reference's literal map path on the host).
__global__ void k_rdd_symmetrise(const l3d_edge* __restrict__ W, int nnz, l3d_edge* __restrict__ out, int* __restrict__ bad)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const l3d_edge e = W[k];
    if (k > 0) { const l3d_edge p = W[k - 1]; if (!(p.i < e.i || (p.i == e.i && p.j < e.j))) *bad = 1; }
    int lo = 0, hi = nnz;
    while (lo < hi) {                                                   // first entry >= (e.j, e.i)
        const int mid = (lo + hi) >> 1;
        const l3d_edge m = W[mid];
        if (m.i < e.j || (m.i == e.j && m.j < e.i)) lo = mid + 1; else hi = mid;
    }
    if (lo >= nnz || W[lo].i != e.j || W[lo].j != e.i) { *bad = 1; return; }
    const float t = W[lo].w;
    l3d_edge r = e;
    r.w = e.i <= e.j ? __builtin_fminf(t, e.w) : __builtin_fminf(e.w, t);   // the visit of the later entry decides
    out[k] = r;
}
// stable ascending weight order of performClustering (clustering.cc:14, CLEdge::operator<): monotone key, -0 == +0
__global__ void k_edge_weight_keys(const l3d_edge* __restrict__ E, int nnz, unsigned* __restrict__ key, unsigned* __restrict__ val)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    float w = E[k].w;
    if (w == 0.0f) w = 0.0f;
    const unsigned u = __float_as_uint(w);
    key[k] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    val[k] = (unsigned)k;
}
// ---- connected components of the edge list (labels = smallest node id of the component): Felzenszwalb-Huttenlocher never looks
// across components, so each one can be segmented on its own (l3d_clustering_edges_grouped).  Hooking by atomicMin on the roots +
// full path compression, repeated until a hooking round changes nothing.
__global__ void k_cc_init(int* __restrict__ comp, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) comp[i] = i;
}
__global__ void k_cc_hook(const l3d_edge* __restrict__ E, int nnz, int* __restrict__ comp, int* __restrict__ changed)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    int a = E[k].i, b = E[k].j;
    // roots (the labels only ever decrease: a stale read costs another round, nothing else)
    int ra = __hip_atomic_load(comp + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (ra != a) { a = ra; ra = __hip_atomic_load(comp + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    int rb = __hip_atomic_load(comp + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (rb != b) { b = rb; rb = __hip_atomic_load(comp + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if (a != b) { atomicMin(&comp[max(a, b)], min(a, b)); *changed = 1; }
}
__global__ void k_cc_compress(int* __restrict__ comp, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int r = comp[i];
    while (true) { const int q = comp[r]; if (q == r) break; r = q; }
    comp[i] = r;
}
// keys of the grouped order: (component, weight key); values: the entry's position
__global__ void k_edge_group_keys(const l3d_edge* __restrict__ E, const int* __restrict__ comp, int nnz, unsigned long long* __restrict__ key, unsigned* __restrict__ val)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    float w = E[k].w;
    if (w == 0.0f) w = 0.0f;
    const unsigned u = __float_as_uint(w);
    key[k] = ((unsigned long long)(unsigned)comp[E[k].i] << 32) | ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
    val[k] = (unsigned)k;
}
// first entry of every group in the sorted key array
__global__ void k_group_flags(const unsigned long long* __restrict__ key, int nnz, int* __restrict__ flag)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) flag[k] = (k == 0 || (key[k] >> 32) != (key[k - 1] >> 32)) ? 1 : 0;
}
__global__ void k_group_starts(const int* __restrict__ flag, const int* __restrict__ rank, int nnz, int* __restrict__ start)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz && flag[k]) start[rank[k]] = k;
}
__global__ void k_edge_gather(const l3d_edge* __restrict__ E, const unsigned* __restrict__ order, int nnz, l3d_edge* __restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) out[k] = E[order[k]];
}

// ---- the merge loop of performClustering (clustering.cc:21-40, universe.h:59-115) on the device, one WAVE per connected component.
// Felzenszwalb-Huttenlocher is sequential by definition -- every merge decision depends on the thresholds the earlier merges left --
// but only INSIDE a component: each one is walked by lane 0 of its wave with the unions, ranks and roots of the one sequential walk,
// its state (parent, rank, size, threshold per node) in LDS under LOCAL node numbers (the component's nodes sorted by id), the edges
// prefetched 64 at a time by the whole wave.  All components run at once: the launch lasts as long as its largest component
// (config 2: 1326 edges, 127 nodes).  Components beyond kUfLds nodes keep their state in global arrays (same code, slower).
constexpr int kUfLds = 2048;
__global__ void k_uf_node_keys(const int* __restrict__ comp, int n, unsigned* __restrict__ key, unsigned* __restrict__ val, int* __restrict__ labels)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) { key[v] = (unsigned)comp[v]; val[v] = (unsigned)v; labels[v] = v; }   // (a node without edges is its own cluster)
}
// position of every node in the (component, id) order, and the node count of every component under its label
__global__ void k_uf_pos(const unsigned* __restrict__ key_sorted, const unsigned* __restrict__ node_sorted, int n, int* __restrict__ pos, int* __restrict__ ncnt)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) { pos[node_sorted[p]] = p; atomicAdd(&ncnt[key_sorted[p]], 1); }
}
// one component's walk; called once with the LDS arrays and once with global ones, so that either gets its own address space
__device__ __forceinline__ void uf_component_walk(const l3d_edge* __restrict__ E, int e0, int e1, const unsigned* __restrict__ node_sorted,
                                                  const int* __restrict__ pos, int n0, int k, float c, int* cid, int* rnk, int* size, float* thr,
                                                  int* s_ei, int* s_ej, float* s_ew, int* __restrict__ labels)
{
    const int lane = threadIdx.x;
    for (int t = lane; t < k; t += 64) { cid[t] = t; rnk[t] = 0; size[t] = 1; thr[t] = c; }
    __threadfence_block();
    for (int base = e0; base < e1; base += 64) {
        const int q = base + lane;
        if (q < e1) { const l3d_edge e = E[q]; s_ei[lane] = pos[e.i] - n0; s_ej[lane] = pos[e.j] - n0; s_ew[lane] = e.w; }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            // clustering.cc:21-40 with universe.h's find (path halving finds the same root: unions go by rank, which no compression touches)
            const int cnt = min(64, e1 - base);
            for (int q2 = 0; q2 < cnt; ++q2) {
                const float w = s_ew[q2];
                int a = s_ei[q2], b = s_ej[q2];
                while (a != cid[a]) { cid[a] = cid[cid[a]]; a = cid[a]; }
                while (b != cid[b]) { cid[b] = cid[cid[b]]; b = cid[b]; }
                if (a != b && w <= thr[a] && w <= thr[b]) {
                    if (rnk[a] > rnk[b]) { cid[b] = a; size[a] += size[b]; }
                    else { cid[a] = b; size[b] += size[a]; if (rnk[a] == rnk[b]) rnk[b]++; a = b; }
                    thr[a] = w + c / (float)size[a];
                }
                // the list holds every edge in both directions and the stable order keeps the two together: whatever the first one did
                // (merged, found merged, failed a threshold), the reversed twin right behind it meets the same state and changes nothing
                if (q2 + 1 < cnt && s_ei[q2 + 1] == s_ej[q2] && s_ej[q2 + 1] == s_ei[q2] && s_ew[q2 + 1] == w) ++q2;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __threadfence_block();
    // labels: CLUniverse::find of every node, as global node ids (clustering.h:125 hands back find(k))
    for (int t = lane; t < k; t += 64) {
        int y = t;
        while (y != cid[y]) y = cid[y];
        labels[node_sorted[n0 + t]] = (int)node_sorted[n0 + y];
    }
}
__global__ __launch_bounds__(64) void k_uf_component(const l3d_edge* __restrict__ E, const int* __restrict__ gstart, int n_groups, int nnz,
                                                      const int* __restrict__ comp, const unsigned* __restrict__ node_sorted, const int* __restrict__ pos,
                                                      const int* __restrict__ ncnt, int n, float c, int* __restrict__ g_state, int* __restrict__ labels)
{
    __shared__ int s_cid[kUfLds], s_rank[kUfLds], s_size[kUfLds];
    __shared__ float s_thr[kUfLds];
    __shared__ int s_ei[64], s_ej[64];
    __shared__ float s_ew[64];
    const int g = blockIdx.x;
    const int e0 = gstart[g], e1 = g + 1 < n_groups ? gstart[g + 1] : nnz;
    // the component's label is its smallest node id, and the node order inside a component is ascending: its first node is the label
    const int label = comp[E[e0].i], n0 = pos[label], k = ncnt[label];
    if (k <= kUfLds) uf_component_walk(E, e0, e1, node_sorted, pos, n0, k, c, s_cid, s_rank, s_size, s_thr, s_ei, s_ej, s_ew, labels);
    // (a component too big for LDS: its slices of four global arrays, indexed by the same local numbers)
    else uf_component_walk(E, e0, e1, node_sorted, pos, n0, k, c, g_state + n0, g_state + (size_t)n + n0, g_state + 2 * (size_t)n + n0,
                           reinterpret_cast<float*>(g_state + 3 * (size_t)n) + n0, s_ei, s_ej, s_ew, labels);
}

}  // namespace l3d

namespace {

// replicator dynamics on the list at dA (device, c->g6): the diffused entries, sorted by (row, column), replace it
int rdd_resident(l3d_ctx* c, int nnz, int n, int iters, hipStream_t st, bool timing);

}  // namespace

extern "C" int l3d_replicator_dynamics_diffusion(l3d_ctx* c, const l3d_edge* A, int nnz, int n, int iters, l3d_edge* out)
{
    if (!c) return L3D_ERR_INVALID;
    if (nnz < 0 || n < 0 || iters < 0 || (nnz > 0 && (!A || !out))) return fail(c, L3D_ERR_INVALID, "bad argument");
    if (nnz == 0 || n == 0) return L3D_OK;      // sparsematrix.cc:77-78: empty matrix, nothing to do
    for (int k = 0; k < nnz; ++k)
        if (A[k].i < 0 || A[k].i >= n || A[k].j < 0 || A[k].j >= n) return fail(c, L3D_ERR_INVALID, "edge index out of range");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const bool timing = c->opt.timing != 0;
    const size_t ab = (size_t)nnz * sizeof(l3d_edge);
    HIPCHK(c, c->g6.reserve(ab + 64));
    c->resident_edges = 0; c->resident_nodes = 0; c->resident_labels = 0;   // (whatever list was resident there is gone; g4 is scratch)
    HIPCHK(c, hipMemcpyAsync(c->g6.p, A, ab, hipMemcpyHostToDevice, st));
    if (int rc = rdd_resident(c, nnz, n, iters, st, timing)) return rc;
    HIPCHK(c, hipMemcpyAsync(out, c->g6.p, ab, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    return L3D_OK;
}

namespace {

int rdd_resident(l3d_ctx* c, int nnz, int n, int iters, hipStream_t st, bool timing)
{
    double t_last = now_s();
    auto lap = [&](const char* what) { if (timing) { (void)hipStreamSynchronize(st); const double t = now_s(); fprintf(stderr, "[l3d rdd] %-24s %8.2f ms\n", what, (t - t_last) * 1e3); t_last = t; } };

    int shift = 1;
    while ((1ll << shift) < (long long)n) ++shift;                       // index bits: keys are (major << shift) | minor
    const size_t vb4 = ((size_t)nnz * 4 + 255) & ~(size_t)255, sb = ((size_t)n * 4 + 255) & ~(size_t)255;
    HIPCHK(c, c->g0.reserve(3 * vb4)); HIPCHK(c, c->g1.reserve(2 * vb4)); HIPCHK(c, c->g2.reserve(2 * vb4));
    HIPCHK(c, c->g3.reserve(2 * sb)); HIPCHK(c, c->g4.reserve(2 * sb));
    // sort scratch: two key arrays, four index arrays, hipCUB's temporary storage
    size_t temp_bytes = 0;
    HIPCHK(c, sort_pairs_u64_u32(nullptr, temp_bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                                 (const unsigned*)nullptr, (unsigned*)nullptr, nnz, 0, 2 * shift, st));
    const size_t kb = ((size_t)nnz * 8 + 255) & ~(size_t)255, vb = ((size_t)nnz * 4 + 255) & ~(size_t)255;
    HIPCHK(c, c->g7.reserve(2 * kb + 4 * vb + temp_bytes + 256));
    unsigned char* sc = c->g7.as<unsigned char>();
    unsigned long long* key_in = reinterpret_cast<unsigned long long*>(sc);
    unsigned long long* key_out = reinterpret_cast<unsigned long long*>(sc + kb);
    unsigned* val_in = reinterpret_cast<unsigned*>(sc + 2 * kb);
    unsigned* ordW = reinterpret_cast<unsigned*>(sc + 2 * kb + vb);
    unsigned* val_in2 = reinterpret_cast<unsigned*>(sc + 2 * kb + 2 * vb);
    unsigned* ordP = reinterpret_cast<unsigned*>(sc + 2 * kb + 3 * vb);
    void* temp = sc + 2 * kb + 4 * vb;

    float* Wval = c->g0.as<float>();
    int* Wcol = reinterpret_cast<int*>(c->g0.as<unsigned char>() + vb4);
    int* tpos = reinterpret_cast<int*>(c->g0.as<unsigned char>() + 2 * vb4);
    int2* Pij = c->g1.as<int2>();
    float* Pval = c->g2.as<float>();
    float* Pval2 = reinterpret_cast<float*>(c->g2.as<unsigned char>() + vb4);
    int *startW = c->g3.as<int>(), *lenW = reinterpret_cast<int*>(c->g3.as<unsigned char>() + sb);
    int *startP = c->g4.as<int>(), *lenP = reinterpret_cast<int*>(c->g4.as<unsigned char>() + sb);
    l3d_edge* dA = c->g6.as<l3d_edge>();
    const dim3 grid((nnz + 255) / 256), block(256);
    // W: column-sorted (line3D.cc:1258 -> sparsematrix.cc:81-86, stable list sort by (j,i)); P: the column-sorted entries
    // re-sorted by row (cudawrapper.cu:1145 -> sparsematrix.cc:157-167)
    hipLaunchKernelGGL(k_rdd_keys_w, grid, block, 0, st, dA, nnz, shift, key_in, val_in);
    HIPCHK(c, sort_pairs_u64_u32(temp, temp_bytes, key_in, key_out, val_in, ordW, nnz, 0, 2 * shift, st));
    hipLaunchKernelGGL(k_rdd_keys_p, grid, block, 0, st, dA, ordW, nnz, shift, key_in, val_in2);
    HIPCHK(c, sort_pairs_u64_u32(temp, temp_bytes, key_in, key_out, val_in2, ordP, nnz, 0, 2 * shift, st));
    HIPCHK(c, hipMemsetAsync(startW, 0xff, (size_t)n * 4, st));           // -1: no entry in that column / row
    HIPCHK(c, hipMemsetAsync(startP, 0xff, (size_t)n * 4, st));
    HIPCHK(c, hipMemsetAsync(lenW, 0, (size_t)n * 4, st));
    HIPCHK(c, hipMemsetAsync(lenP, 0, (size_t)n * 4, st));
    hipLaunchKernelGGL(k_rdd_build, grid, block, 0, st, dA, ordW, ordP, nnz, Wval, Wcol, Pij, Pval, Pval2, startW, startP);
    hipLaunchKernelGGL(k_rdd_lens, grid, block, 0, st, Wcol, Pij, nnz, startW, startP, lenW, lenP);
    hipLaunchKernelGGL(k_rdd_tpos, grid, block, 0, st, Pij, nnz, startP, lenP, tpos);
    lap("sort + sparse build (device)");

    float *cur = Pval, *nxt = Pval2;                                     // P and P' (cudawrapper.cu:1148: P' starts as a copy of P)
    const dim3 rgrid((n + 3) / 4);
    { ProfScope p(c, "rownorm"); hipLaunchKernelGGL(k_rdd_rownorm, rgrid, block, 0, st, cur, startP, lenP, n); }
    for (int it = 0; it < iters; ++it) {
        { ProfScope p(c, "diffusion_step"); hipLaunchKernelGGL(k_rdd_step, grid, block, 0, st, Pij, cur, Wval, startP, lenP, startW, lenW, tpos, nxt, nnz); }
        std::swap(cur, nxt);
        if (it < iters - 1) { ProfScope p(c, "rownorm"); hipLaunchKernelGGL(k_rdd_rownorm, rgrid, block, 0, st, cur, startP, lenP, n); }
    }
    lap("kernels");
    hipLaunchKernelGGL(k_rdd_result, grid, block, 0, st, Pij, cur, nnz, dA);   // (the input copy is no longer needed)
    return L3D_OK;
}

}  // namespace

// The edge list performClustering walks (clustering.cc:14-40), prepared on the device: optional performDiffusion
// (line3D.cc:1255-1303: replicator dynamics, symmetrise by the minimum, (i,j) order) and the stable ascending weight order.
// merge_loop: the merge loop runs on the device as well (k_uf_component); the n labels stay there and come back when labels_out != nullptr
static int clustering_edges_impl(l3d_ctx* c, const l3d_edge* A, int nnz, int n, int perform_diffusion, int iters, l3d_edge* sorted_out,
                                 int32_t** group_start_out, int* n_groups_out, bool merge_loop = false, int32_t* labels_out = nullptr, float cl_c = 0.0f)
{
    if (!c) return L3D_ERR_INVALID;
    if (nnz < 0 || n < 0 || iters < 0 || (nnz > 0 && !sorted_out && !merge_loop)) return fail(c, L3D_ERR_INVALID, "bad argument");
    if (merge_loop && nnz == 0) {
        if (!labels_out && n > 0) return fail(c, L3D_ERR_INVALID, "clustering: no edges and no place for the labels");
        for (int v = 0; v < n; ++v) labels_out[v] = v;
        return L3D_OK;
    }
    if (group_start_out) { *group_start_out = nullptr; *n_groups_out = 0; }
    if (nnz == 0) return L3D_OK;
    if (!A && c->resident_edges != nnz) return fail(c, L3D_ERR_INVALID, "no resident edge list of that size (l3d_affinity_fill)");
    if (A) for (int k = 0; k < nnz; ++k)
        if (A[k].i < 0 || A[k].i >= n || A[k].j < 0 || A[k].j >= n) return fail(c, L3D_ERR_INVALID, "edge index out of range");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const bool timing = c->opt.timing != 0;
    double t_last = now_s();
    auto lap = [&](const char* what) { if (timing) { (void)hipStreamSynchronize(st); const double t = now_s(); fprintf(stderr, "[l3d edges] %-24s %8.2f ms\n", what, (t - t_last) * 1e3); t_last = t; } };
    const size_t ab = (size_t)nnz * sizeof(l3d_edge);
    c->resident_labels = 0;                                              // (g4 is scratch of the diffusion and of the merge loop)
    if (A) {
        c->resident_nodes = 0;                                           // (g6 is overwritten)
        HIPCHK(c, c->g6.reserve(ab + 64));
        HIPCHK(c, hipMemcpyAsync(c->g6.p, A, ab, hipMemcpyHostToDevice, st));
    }
    if (!A) {                                                            // (a device copy stays for l3d_resident_edges_get: 12 B per edge, microseconds)
        HIPCHK(c, c->edges_keep.reserve(ab + 64));
        HIPCHK(c, hipMemcpyAsync(c->edges_keep.p, c->g6.p, ab, hipMemcpyDeviceToDevice, st));
        c->kept_edges = nnz;
    }
    c->resident_edges = 0;                                               // the list is consumed (diffusion overwrites it)
    const dim3 grid((nnz + 255) / 256), block(256);
    l3d_edge* E = c->g6.as<l3d_edge>();
    if (perform_diffusion && n > 0) {
        if (int rc = rdd_resident(c, nnz, n, iters, st, timing)) return rc;
        // symmetrised list (g0 is free again after the iteration)
        HIPCHK(c, c->g0.reserve(ab + 64));
        HIPCHK(c, c->g3.reserve(64));
        int* bad = c->g3.as<int>();
        HIPCHK(c, hipMemsetAsync(bad, 0, 4, st));
        hipLaunchKernelGGL(k_rdd_symmetrise, grid, block, 0, st, E, nnz, c->g0.as<l3d_edge>(), bad);
        int h_bad = 0;
        HIPCHK(c, hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (h_bad) return fail(c, L3D_ERR_UNSUPPORTED, "diffused list is not a symmetric pattern of unique entries");
        E = c->g0.as<l3d_edge>();
        lap("diffusion + symmetrise");
    }
    bool grouped = group_start_out != nullptr || merge_loop;
    if (grouped) {
        // ---- grouped by connected component: labels, then ONE stable sort by (component, weight key)
        HIPCHK(c, c->g3.reserve((size_t)n * 4 + 256));
        int* comp = c->g3.as<int>();
        int* changed = comp + n;                                           // (n * 4 + 256 reserved)
        const dim3 ngrid((n + 255) / 256);
        hipLaunchKernelGGL(k_cc_init, ngrid, block, 0, st, comp, n);
        const int max_rounds = c->opt.cc_max_rounds > 0 ? c->opt.cc_max_rounds : 64;      // (test hook: 1 forces the not-converged path)
        for (int round = 0; round < max_rounds; ++round) {
            HIPCHK(c, hipMemsetAsync(changed, 0, 4, st));
            for (int r = 0; r < 3; ++r) {                                  // a few hooking rounds per look at the flag
                hipLaunchKernelGGL(k_cc_hook, grid, block, 0, st, E, nnz, comp, changed + (r == 2 ? 0 : 1));
                hipLaunchKernelGGL(k_cc_compress, ngrid, block, 0, st, comp, n);
            }
            int h_changed = 0;
            HIPCHK(c, hipMemcpyAsync(&h_changed, changed, 4, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            if (!h_changed) break;
            if (round == max_rounds - 1) {                                 // not converged: one group, ...
                if (merge_loop) HIPCHK(c, hipMemsetAsync(comp, 0, (size_t)n * 4, st));   // ... here as the single component 0
                else grouped = false;                                      // ... the plain order below
            }
        }
        lap("connected components");
    }
    if (grouped) {
        const int* comp = c->g3.as<int>();
        int shift = 1;
        while ((1ll << shift) < (long long)n) ++shift;
        size_t tb = 0;
        HIPCHK(c, sort_pairs_u64_u32(nullptr, tb, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (const unsigned*)nullptr, (unsigned*)nullptr, nnz, 0, 32 + shift, st));
        size_t tb2 = 0;
        HIPCHK(c, exclusive_sum_int(nullptr, tb2, (const int*)nullptr, (int*)nullptr, nnz + 1, st));
        tb = std::max(tb, tb2);
        const size_t kb = ((size_t)nnz * 8 + 255) & ~(size_t)255, vb4 = ((size_t)nnz * 4 + 255 + 4) & ~(size_t)255;
        HIPCHK(c, c->g7.reserve(2 * kb + 5 * vb4 + tb + 256));
        unsigned char* sc = c->g7.as<unsigned char>();
        unsigned long long* key_in = reinterpret_cast<unsigned long long*>(sc);
        unsigned long long* key_out = reinterpret_cast<unsigned long long*>(sc + kb);
        unsigned* val_in = reinterpret_cast<unsigned*>(sc + 2 * kb);
        unsigned* order = reinterpret_cast<unsigned*>(sc + 2 * kb + vb4);
        int* flag = reinterpret_cast<int*>(sc + 2 * kb + 2 * vb4);
        int* frank = reinterpret_cast<int*>(sc + 2 * kb + 3 * vb4);
        int* gstart = reinterpret_cast<int*>(sc + 2 * kb + 4 * vb4);
        void* temp = sc + 2 * kb + 5 * vb4;
        hipLaunchKernelGGL(k_edge_group_keys, grid, block, 0, st, E, comp, nnz, key_in, val_in);
        HIPCHK(c, sort_pairs_u64_u32(temp, tb, key_in, key_out, val_in, order, nnz, 0, 32 + shift, st));
        HIPCHK(c, c->g1.reserve(ab + 64));
        hipLaunchKernelGGL(k_edge_gather, grid, block, 0, st, E, order, nnz, c->g1.as<l3d_edge>());
        if (sorted_out) HIPCHK(c, hipMemcpyAsync(sorted_out, c->g1.p, ab, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemsetAsync(flag + nnz, 0, 4, st));
        hipLaunchKernelGGL(k_group_flags, grid, block, 0, st, key_out, nnz, flag);
        HIPCHK(c, exclusive_sum_int(temp, tb, flag, frank, nnz + 1, st));
        hipLaunchKernelGGL(k_group_starts, grid, block, 0, st, flag, frank, nnz, gstart);
        int n_groups = 0;
        HIPCHK(c, hipMemcpyAsync(&n_groups, frank + nnz, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (merge_loop) {
            // nodes in (component, id) order -> local numbers; then one wave per component
            size_t tbn = 0;
            HIPCHK(c, sort_pairs_u32_u32(nullptr, tbn, (const unsigned*)nullptr, (unsigned*)nullptr, (const unsigned*)nullptr, (unsigned*)nullptr, n, 0, shift, st));
            const size_t sb = ((size_t)n * 4 + 255) & ~(size_t)255;
            HIPCHK(c, c->g4.reserve(11 * sb + tbn + 256));
            unsigned char* ns = c->g4.as<unsigned char>();
            unsigned *nkey_in = reinterpret_cast<unsigned*>(ns), *nkey = reinterpret_cast<unsigned*>(ns + sb);
            unsigned *nval_in = reinterpret_cast<unsigned*>(ns + 2 * sb), *node_sorted = reinterpret_cast<unsigned*>(ns + 3 * sb);
            int *pos = reinterpret_cast<int*>(ns + 4 * sb), *ncnt = reinterpret_cast<int*>(ns + 5 * sb), *labels = reinterpret_cast<int*>(ns + 6 * sb);
            int* g_state = reinterpret_cast<int*>(ns + 7 * sb);
            const dim3 ngrid((n + 255) / 256);
            hipLaunchKernelGGL(k_uf_node_keys, ngrid, block, 0, st, comp, n, nkey_in, nval_in, labels);
            HIPCHK(c, sort_pairs_u32_u32(ns + 11 * sb, tbn, nkey_in, nkey, nval_in, node_sorted, n, 0, shift, st));
            HIPCHK(c, hipMemsetAsync(ncnt, 0, (size_t)n * 4, st));
            hipLaunchKernelGGL(k_uf_pos, ngrid, block, 0, st, nkey, node_sorted, n, pos, ncnt);
            { ProfScope p(c, "uf_components");
              hipLaunchKernelGGL(k_uf_component, dim3(n_groups), dim3(64), 0, st, c->g1.as<l3d_edge>(), gstart, n_groups, nnz, comp, node_sorted, pos, ncnt, n, cl_c, g_state, labels); }
            if (labels_out) HIPCHK(c, hipMemcpyAsync(labels_out, labels, (size_t)n * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            HIPCHK(c, hipGetLastError());
            c->resident_labels = n; c->resident_labels_p = labels;        // (for l3d_fit_labelled_clusters)
            lap("grouped order + merge loop (device)");
            if (n_groups_out) *n_groups_out = n_groups;
            return L3D_OK;
        }
        int32_t* gs = static_cast<int32_t*>(malloc(((size_t)n_groups + 1) * 4));
        if (!gs) return fail(c, L3D_ERR_NOMEM, "malloc");
        hipError_t e1 = hipMemcpyAsync(gs, gstart, (size_t)n_groups * 4, hipMemcpyDeviceToHost, st);
        hipError_t e2 = hipStreamSynchronize(st);
        hipError_t e3 = hipGetLastError();
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { free(gs); return fail(c, L3D_ERR_HIP, "clustering edges: read-back failed"); }
        gs[n_groups] = nnz;
        *group_start_out = gs; *n_groups_out = n_groups;
        lap("grouped weight order + download");
        return L3D_OK;
    }
    // stable radix sort by the weight key, then gather
    size_t temp_bytes = 0;
    HIPCHK(c, sort_pairs_u32_u32(nullptr, temp_bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const unsigned*)nullptr, (unsigned*)nullptr, nnz, 0, 32, st));
    const size_t vb = ((size_t)nnz * 4 + 255) & ~(size_t)255;
    HIPCHK(c, c->g7.reserve(4 * vb + temp_bytes + 256));
    unsigned char* sc = c->g7.as<unsigned char>();
    unsigned* key_in = reinterpret_cast<unsigned*>(sc);
    unsigned* key_out = reinterpret_cast<unsigned*>(sc + vb);
    unsigned* val_in = reinterpret_cast<unsigned*>(sc + 2 * vb);
    unsigned* order = reinterpret_cast<unsigned*>(sc + 3 * vb);
    hipLaunchKernelGGL(k_edge_weight_keys, grid, block, 0, st, E, nnz, key_in, val_in);
    HIPCHK(c, sort_pairs_u32_u32(sc + 4 * vb, temp_bytes, key_in, key_out, val_in, order, nnz, 0, 32, st));
    HIPCHK(c, c->g1.reserve(ab + 64));
    hipLaunchKernelGGL(k_edge_gather, grid, block, 0, st, E, order, nnz, c->g1.as<l3d_edge>());
    HIPCHK(c, hipMemcpyAsync(sorted_out, c->g1.p, ab, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    lap("weight order + download");
    if (group_start_out) {                                                 // (components not available: the whole list is one group)
        int32_t* gs = static_cast<int32_t*>(malloc(2 * sizeof(int32_t)));
        if (!gs) return fail(c, L3D_ERR_NOMEM, "malloc");
        gs[0] = 0; gs[1] = nnz;
        *group_start_out = gs; *n_groups_out = 1;
    }
    return L3D_OK;
}

extern "C" int l3d_clustering_edges(l3d_ctx* c, const l3d_edge* A, int nnz, int n, int perform_diffusion, int iters, l3d_edge* sorted_out)
{
    return clustering_edges_impl(c, A, nnz, n, perform_diffusion, iters, sorted_out, nullptr, nullptr);
}

// The same list GROUPED BY CONNECTED COMPONENT of the (diffused) graph, ascending weight (stable) inside every group: the merge loop of
// performClustering never relates nodes of different components, so the groups can be walked independently -- and in parallel.
extern "C" int l3d_clustering_edges_grouped(l3d_ctx* c, const l3d_edge* A, int nnz, int n, int perform_diffusion, int iters, l3d_edge* sorted_out,
                                            int32_t** group_start, int* n_groups)
{
    if (!group_start || !n_groups) return c ? fail(c, L3D_ERR_INVALID, "bad argument") : L3D_ERR_INVALID;
    return clustering_edges_impl(c, A, nnz, n, perform_diffusion, iters, sorted_out, group_start, n_groups);
}

// clusterSegments2D's tail in one call: [performDiffusion] + performClustering(A, n, c) with only the labels coming back
// (line3D.cc:1239-1246 -> clustering.cc:6-47).  labels[k] = CLUniverse::find(k), the same roots as the host walk.
extern "C" int l3d_perform_clustering_device(l3d_ctx* c, const l3d_edge* A, int nnz, int n, int perform_diffusion, int iters, float cl_c, int32_t* labels,
                                             int* n_components)
{
    if (n_components) *n_components = 0;
    return clustering_edges_impl(c, A, nnz, n, perform_diffusion, iters, nullptr, nullptr, n_components, true, labels, cl_c);
}

// the edge list of the last l3d_affinity_fill / l3d_affinity_fill_resident, copied to the host: from g6 while it is resident, from the
// copy the clustering took when it consumed it afterwards
extern "C" int l3d_resident_edges_get(l3d_ctx* c, l3d_edge* out, int nnz)
{
    if (!c) return L3D_ERR_INVALID;
    if (nnz < 0 || (nnz > 0 && !out)) return fail(c, L3D_ERR_INVALID, "bad argument");
    if (nnz == 0) return L3D_OK;
    const void* src = c->resident_edges == nnz ? c->g6.p : c->kept_edges == nnz ? c->edges_keep.p : nullptr;
    if (!src) return fail(c, L3D_ERR_INVALID, "no resident edge list of that size (l3d_affinity_fill)");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(out, src, (size_t)nnz * sizeof(l3d_edge), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return L3D_OK;
}

void l3d::warm_rdd() { touch_kernel(reinterpret_cast<const void*>(&k_cc_init)); }
