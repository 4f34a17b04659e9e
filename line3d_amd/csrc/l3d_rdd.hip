// l3d_rdd.hip -- replicator_dynamics_diffusion (cudawrapper.h:73-74, cudawrapper.cu:1131-1191) with the SparseMatrix
// construction of performDiffusion (line3D.cc:1258, sparsematrix.cc:63-191) on the device.
//
// The reference builds W (entries sorted by column, a stable list sort by (j,i), sparsematrix.cc:81-86) and P (the column-sorted
// entries re-sorted by row, sparsematrix.cc:157-167) on the host and uploads three copies.  Here the edge list is uploaded once; the
// two orders are stable LSD radix sorts on the device (hipCUB; a stable sort has exactly one result, the one the reference's stable
// list sorts produce), the float4 entries and the first-entry tables (sparsematrix.cc:99-131) are built by kernels, and the result
// comes back as (i, j, w) records.  Round 1 sorted on 16 host threads: 13.6 ms of the 35 ms a diffusion of the config-2 affinity
// list (978 k entries) took; the device sorts take well under a millisecond.
#include <hipcub/hipcub.hpp>

#include "l3d_ctx.hpp"

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
// entries (row, col, val, 0) as floats like the reference's SparseMatrix, in both orders (+ P' = copy of P, cudawrapper.cu:1148), and
// the first entry of every column of W / row of P (tables preset to -1)
__global__ void k_rdd_build(const l3d_edge* __restrict__ A, const unsigned* __restrict__ ordW, const unsigned* __restrict__ ordP, int nnz,
                            float4* __restrict__ W, float4* __restrict__ P, float4* __restrict__ Pp, int* __restrict__ startW, int* __restrict__ startP)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const l3d_edge w = A[ordW[k]];
    W[k] = make_float4((float)w.i, (float)w.j, w.w, 0.0f);
    if (k == 0 || A[ordW[k - 1]].j != w.j) startW[w.j] = k;
    const l3d_edge q = A[ordW[ordP[k]]];
    const float4 e = make_float4((float)q.i, (float)q.j, q.w, 0.0f);
    P[k] = e; Pp[k] = e;
    if (k == 0 || A[ordW[ordP[k - 1]]].i != q.i) startP[q.i] = k;
}
__global__ void k_rdd_result(const float4* __restrict__ P, int nnz, l3d_edge* __restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) { const float4 e = P[k]; l3d_edge r; r.i = (int)e.x; r.j = (int)e.y; r.w = e.z; out[k] = r; }
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
__global__ void k_edge_gather(const l3d_edge* __restrict__ E, const unsigned* __restrict__ order, int nnz, l3d_edge* __restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) out[k] = E[order[k]];
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
    const bool timing = getenv("L3D_TIMING") != nullptr;
    const size_t ab = (size_t)nnz * sizeof(l3d_edge);
    HIPCHK(c, c->g6.reserve(ab + 64));
    c->resident_edges = 0;                                               // (whatever list was resident there is gone)
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
    const size_t eb = (size_t)nnz * 16, sb = (size_t)n * 4;
    HIPCHK(c, c->g0.reserve(eb)); HIPCHK(c, c->g1.reserve(eb)); HIPCHK(c, c->g2.reserve(eb));
    HIPCHK(c, c->g3.reserve(sb)); HIPCHK(c, c->g4.reserve(sb)); HIPCHK(c, c->g5.reserve(sb));
    // sort scratch: two key arrays, four index arrays, hipCUB's temporary storage
    size_t temp_bytes = 0;
    HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
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

    float4 *dW = c->g0.as<float4>(), *dP = c->g1.as<float4>(), *dPp = c->g2.as<float4>();
    int *dWc = c->g3.as<int>(), *dPr = c->g4.as<int>(), *dPpr = c->g5.as<int>();
    l3d_edge* dA = c->g6.as<l3d_edge>();
    const dim3 grid((nnz + 255) / 256), block(256);
    // W: column-sorted (line3D.cc:1258 -> sparsematrix.cc:81-86, stable list sort by (j,i)); P: the column-sorted entries
    // re-sorted by row (cudawrapper.cu:1145 -> sparsematrix.cc:157-167)
    hipLaunchKernelGGL(k_rdd_keys_w, grid, block, 0, st, dA, nnz, shift, key_in, val_in);
    HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, key_in, key_out, val_in, ordW, nnz, 0, 2 * shift, st));
    hipLaunchKernelGGL(k_rdd_keys_p, grid, block, 0, st, dA, ordW, nnz, shift, key_in, val_in2);
    HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, key_in, key_out, val_in2, ordP, nnz, 0, 2 * shift, st));
    HIPCHK(c, hipMemsetAsync(dWc, 0xff, sb, st));                       // -1: no entry in that column / row
    HIPCHK(c, hipMemsetAsync(dPr, 0xff, sb, st));
    hipLaunchKernelGGL(k_rdd_build, grid, block, 0, st, dA, ordW, ordP, nnz, dW, dP, dPp, dWc, dPr);
    HIPCHK(c, hipMemcpyAsync(dPpr, dPr, sb, hipMemcpyDeviceToDevice, st));
    lap("sort + sparse build (device)");

    { ProfScope p(c, "rownorm"); launch_rownorm(dP, dPr, n, nnz, st); }
    for (int it = 0; it < iters; ++it) {
        { ProfScope p(c, "diffusion_step"); launch_diffusion_step(dP, dW, dPr, dWc, dPp, dPpr, nnz, st); }
        std::swap(dP, dPp);
        std::swap(dPr, dPpr);
        if (it < iters - 1) { ProfScope p(c, "rownorm"); launch_rownorm(dP, dPr, n, nnz, st); }
    }
    lap("kernels");
    hipLaunchKernelGGL(k_rdd_result, grid, block, 0, st, dP, nnz, dA);   // (the input copy is no longer needed)
    return L3D_OK;
}

}  // namespace

// The edge list performClustering walks (clustering.cc:14-40), prepared on the device: optional performDiffusion
// (line3D.cc:1255-1303: replicator dynamics, symmetrise by the minimum, (i,j) order) and the stable ascending weight order.
extern "C" int l3d_clustering_edges(l3d_ctx* c, const l3d_edge* A, int nnz, int n, int perform_diffusion, int iters, l3d_edge* sorted_out)
{
    if (!c) return L3D_ERR_INVALID;
    if (nnz < 0 || n < 0 || iters < 0 || (nnz > 0 && !sorted_out)) return fail(c, L3D_ERR_INVALID, "bad argument");
    if (nnz == 0) return L3D_OK;
    if (!A && c->resident_edges != nnz) return fail(c, L3D_ERR_INVALID, "no resident edge list of that size (l3d_affinity_fill)");
    if (A) for (int k = 0; k < nnz; ++k)
        if (A[k].i < 0 || A[k].i >= n || A[k].j < 0 || A[k].j >= n) return fail(c, L3D_ERR_INVALID, "edge index out of range");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const bool timing = getenv("L3D_TIMING") != nullptr;
    double t_last = now_s();
    auto lap = [&](const char* what) { if (timing) { (void)hipStreamSynchronize(st); const double t = now_s(); fprintf(stderr, "[l3d edges] %-24s %8.2f ms\n", what, (t - t_last) * 1e3); t_last = t; } };
    const size_t ab = (size_t)nnz * sizeof(l3d_edge);
    if (A) {
        HIPCHK(c, c->g6.reserve(ab + 64));
        HIPCHK(c, hipMemcpyAsync(c->g6.p, A, ab, hipMemcpyHostToDevice, st));
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
    // stable radix sort by the weight key, then gather
    size_t temp_bytes = 0;
    HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const unsigned*)nullptr, (unsigned*)nullptr, nnz, 0, 32, st));
    const size_t vb = ((size_t)nnz * 4 + 255) & ~(size_t)255;
    HIPCHK(c, c->g7.reserve(4 * vb + temp_bytes + 256));
    unsigned char* sc = c->g7.as<unsigned char>();
    unsigned* key_in = reinterpret_cast<unsigned*>(sc);
    unsigned* key_out = reinterpret_cast<unsigned*>(sc + vb);
    unsigned* val_in = reinterpret_cast<unsigned*>(sc + 2 * vb);
    unsigned* order = reinterpret_cast<unsigned*>(sc + 3 * vb);
    hipLaunchKernelGGL(k_edge_weight_keys, grid, block, 0, st, E, nnz, key_in, val_in);
    HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(sc + 4 * vb, temp_bytes, key_in, key_out, val_in, order, nnz, 0, 32, st));
    HIPCHK(c, c->g1.reserve(ab + 64));
    hipLaunchKernelGGL(k_edge_gather, grid, block, 0, st, E, order, nnz, c->g1.as<l3d_edge>());
    HIPCHK(c, hipMemcpyAsync(sorted_out, c->g1.p, ab, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    lap("weight order + download");
    return L3D_OK;
}
