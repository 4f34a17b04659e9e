// l3d_linefit.hip -- the line fit of processClusteredSegments (line3D.cc:1306-1597: getLineEquation3D, projectToLine) on the device
// (SURVEY.md 8f4): one wave per cluster, the arithmetic of l3d_linefit.hpp -- the very functions the host fit calls.
//
// A cluster is a list of hypothesis indices in key order (= ascending index: hypotheses are numbered by (camera, segment)).  The wave
// takes the members' 3-D end points back to the caller's coordinates, fits the line (every lane runs the same short sequential
// sums: the order of the additions is part of the result), gives every point its float distance along the line, ranks the points
// (all-to-all comparison: the rank IS the stable order), and one lane sweeps them, emitting the stretches seen by >= 3 cameras.
#include "l3d_sort.hpp"

#include "l3d_ctx.hpp"
#include "l3d_linefit.hpp"

using namespace l3d;

namespace l3d {

struct FitArgs {
    int n_groups;
    const int* group_start;             // n_groups + 1 (members)
    const int* member_hyp;              // hypothesis indices, ascending per group
    const Hypothesis* hyp;
    const unsigned* hyp_cam;            // camera id per hypothesis
    la::M3 Rinv; double scale_inv; la::V3 tneg;
    la::V3* pts;                        // 2 per member
    float* dist;                        // 2 per member
    int* order;                         // 2 per member
    unsigned char* line_open;           // 1 per member
    unsigned* cam_ids; unsigned* cam_cnt;   // 1 per member
    double* out;                        // 6 doubles per emitted segment, at most one per member
    int* out_cnt;                       // per group
};

constexpr int kFitLds = 128;            // members of a cluster whose sweep state fits the per-wave LDS arrays

__global__ __launch_bounds__(256) void k_fit_clusters(FitArgs a)
{
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= a.n_groups) return;
    const int m0 = a.group_start[g], members = a.group_start[g + 1] - m0, n2 = 2 * members;
    // points, order and sweep state in LDS for clusters of up to kFitLds members (nearly all): with everything in global memory the
    // sequential sums and the sweep are chains of dependent L2 round trips (0.86 ms per launch at config 2); bigger clusters keep
    // the global scratch
    __shared__ double s_pts[4][2 * kFitLds][3];
    const int wv = threadIdx.x >> 6;
    const bool small = members <= kFitLds;
    la::V3* gpts = a.pts + 2 * (size_t)m0;
    for (int i = lane; i < n2; i += 64) {
        const Hypothesis& h = a.hyp[a.member_hyp[m0 + (i >> 1)]];
        const la::V3 P = (i & 1) ? la::V3{ h.P2[0], h.P2[1], h.P2[2] } : la::V3{ h.P1[0], h.P1[1], h.P1[2] };
        const la::V3 Q = fit::inverse_transform(a.Rinv, a.scale_inv, a.tneg, P);
        if (small) { s_pts[wv][i][0] = Q.x; s_pts[wv][i][1] = Q.y; s_pts[wv][i][2] = Q.z; }
        else gpts[i] = Q;
    }
    __threadfence_block();
    auto get = [&](int i) { return small ? la::V3{ s_pts[wv][i][0], s_pts[wv][i][1], s_pts[wv][i][2] } : gpts[i]; };
    la::V3 Pc, dir, min_point;
    fit::line_of_points(get, n2, Pc, dir, min_point);                     // (every lane: the same sums in the same order)
    __shared__ float s_dist[4][2 * kFitLds];
    __shared__ int s_order[4][2 * kFitLds];
    float* dist = small ? s_dist[wv] : a.dist + 2 * (size_t)m0;
    int* order = small ? s_order[wv] : a.order + 2 * (size_t)m0;
    for (int i = lane; i < n2; i += 64) { dist[i] = fit::point_dist(get(i), min_point); order[i] = i; }   // (order: every slot valid even when NaN distances make ranks collide)
    __threadfence_block();
    for (int i = lane; i < n2; i += 64) {                                 // stable order = rank by (distance, point index)
        const float d = dist[i];
        int r = 0;
        for (int j = 0; j < n2; ++j) { const float e = dist[j]; r += (e < d) || (e == d && j < i); }
        order[r] = i;
    }
    __threadfence_block();
    __shared__ unsigned s_cam[4][kFitLds], s_cid[4][kFitLds], s_ccnt[4][kFitLds];
    __shared__ unsigned char s_open[4][kFitLds];
    const int* mh = a.member_hyp + m0;
    if (small) {
        for (int i = lane; i < members; i += 64) s_cam[wv][i] = a.hyp_cam[mh[i]];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    double* out = a.out + 6 * (size_t)m0;
    int n_out = 0;
    auto emit = [&](la::V3 s0, la::V3 e0) { double* o = out + 6 * (size_t)n_out++; o[0] = s0.x; o[1] = s0.y; o[2] = s0.z; o[3] = e0.x; o[4] = e0.y; o[5] = e0.z; };
    bool swept = false;
    if (small) {
        // The sweep of fit::sweep_line with its state in registers: lane c owns camera slot c (id, open segments), the members' open
        // flags are two wave-uniform 64-bit masks, the walk is wave-uniform.  (One lane walking LDS arrays spent 0.55 ms of the launch
        // on dependent LDS round trips.)  Pure bookkeeping -- the same decisions as the sequential version; a cluster with more
        // than 64 cameras takes that one.
        unsigned my_cam = 0xffffffffu, my_cnt = 0;
        int n_cams = 0;
        unsigned long long open_lo = 0ull, open_hi = 0ull;
        bool opened = false, overflow = false;
        la::V3 start;
        for (int k = 0; k < n2 && !overflow; ++k) {
            const int p = s_order[wv][k], member = p >> 1;
            const unsigned cam = s_cam[wv][member];
            const unsigned long long hit = __ballot(lane < n_cams && my_cam == cam);
            int ci = hit ? __ffsll((long long)hit) - 1 : n_cams;
            if (!hit) { if (n_cams == 64) { overflow = true; break; } if (lane == n_cams) { my_cam = cam; my_cnt = 0; } ++n_cams; }
            unsigned long long& om = member < 64 ? open_lo : open_hi;
            const unsigned long long bit = 1ull << (member & 63);
            if (!(om & bit)) { om |= bit; if (lane == ci) ++my_cnt; }
            else { om &= ~bit; if (lane == ci) --my_cnt; }
            const int n_open_cams = __popcll(__ballot(lane < n_cams && my_cnt > 0));
            if (opened && n_open_cams < 3) { if (lane == 0) emit(start, get(p)); else ++n_out; opened = false; }
            else if (!opened && n_open_cams >= 3) { start = get(p); opened = true; }
        }
        swept = !overflow;
        if (overflow) n_out = 0;
    }
    if (!swept && lane == 0) {
        n_out = 0;
        if (small) fit::sweep_line(s_order[wv], n2, get, [&](int member) { return s_cam[wv][member]; }, s_open[wv], s_cid[wv], s_ccnt[wv], emit);
        else fit::sweep_line(order, n2, get, [&](int member) { return a.hyp_cam[mh[member]]; }, a.line_open + m0, a.cam_ids + m0, a.cam_cnt + m0, emit);
    }
    if (lane == 0) a.out_cnt[g] = n_out;
}

// the emitted segments of all groups, back to back
__global__ void k_fit_gather(const int* __restrict__ group_start, const int* __restrict__ out_off, const double* __restrict__ out, int n_groups,
                             double* __restrict__ packed)
{
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= n_groups) return;
    const int n = (out_off[g + 1] - out_off[g]) * 6;
    const double* src = out + 6 * (size_t)group_start[g];
    double* dst = packed + 6 * (size_t)out_off[g];
    for (int i = lane; i < n; i += 64) dst[i] = src[i];
}

}  // namespace l3d

namespace l3d {

// ---- processClusteredSegments' grouping (line3D.cc:1306-1368) on the device: clusters in ascending label order, their members in key
// order (= ascending hypothesis index), only those with >= 4 members seen from >= 4 cameras.
__global__ void k_lab_keys(const int* __restrict__ labels, const int* __restrict__ node_hyp, int n, unsigned long long* __restrict__ key)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) key[v] = ((unsigned long long)(unsigned)labels[v] << 32) | (unsigned)node_hyp[v];
}
__global__ void k_lab_flags(const unsigned long long* __restrict__ key, int n, int* __restrict__ flag)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) flag[p] = (p == 0 || (key[p] >> 32) != (key[p - 1] >> 32)) ? 1 : 0;
    if (p == n) flag[p] = 0;
}
__global__ void k_lab_starts(const int* __restrict__ flag, const int* __restrict__ rank, int n, int* __restrict__ start)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n && flag[p]) start[rank[p]] = p;
    if (p == n) start[rank[n]] = n;                                        // (rank[n] = number of clusters)
}
// a thread per cluster: members (0 unless the cluster qualifies: >= 4 members, >= 4 cameras -- line3D.cc:1324-1340)
__global__ void k_lab_valid(const unsigned long long* __restrict__ key, const int* __restrict__ start, const int* __restrict__ n_all, const unsigned* __restrict__ hyp_cam,
                            int* __restrict__ vflag, int* __restrict__ vsize)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int ng = *n_all;
    if (g > ng) return;
    if (g == ng) { vflag[g] = 0; vsize[g] = 0; return; }
    const int p0 = start[g], p1 = start[g + 1];
    int ncam = 1;
    for (int p = p0 + 1; p < p1; ++p) ncam += hyp_cam[(unsigned)key[p]] != hyp_cam[(unsigned)key[p - 1]];
    const bool ok = p1 - p0 >= 4 && ncam >= 4;
    vflag[g] = ok ? 1 : 0; vsize[g] = ok ? p1 - p0 : 0;
}
__global__ void k_lab_compact(const unsigned long long* __restrict__ key, const int* __restrict__ start, const int* __restrict__ n_all, const int* __restrict__ vflag,
                              const int* __restrict__ vrank, const int* __restrict__ voff, int* __restrict__ group_start, int* __restrict__ member_hyp)
{
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ng = *n_all;
    if (g > ng) return;
    if (g == ng) { if (lane == 0) group_start[vrank[ng]] = voff[ng]; return; }   // (the closing entry)
    if (!vflag[g]) return;
    const int p0 = start[g], n = start[g + 1] - p0, o = voff[g];
    if (lane == 0) group_start[vrank[g]] = o;
    for (int i = lane; i < n; i += 64) member_hyp[o + i] = (int)(unsigned)key[p0 + i];
}

}  // namespace l3d

namespace {

struct FitStage { const int* gs; const int* mh; const Hypothesis* hyp; const unsigned* cam; };

// the fits of clusters whose tables are on the device already
int fit_core(l3d_ctx* c, FitStage in, int n_groups, int n_members, const double* Rinv, double scale_inv, const double* tneg,
             int32_t** seg_count, double** segs, int* n_segs)
{
    hipStream_t st = c->stream;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t nm = (size_t)n_members, ng = (size_t)n_groups;
    const size_t o_pts = 0, o_dist = al(2 * nm * sizeof(la::V3) + 64), o_ord = o_dist + al(2 * nm * 4 + 4), o_open = o_ord + al(2 * nm * 4 + 4), o_ci = o_open + al(nm + 4),
                 o_cc = o_ci + al(nm * 4 + 4), o_out = o_cc + al(nm * 4 + 4), o_cnt = o_out + al(6 * nm * 8 + 8), o_off = o_cnt + al((ng + 1) * 4), sc_bytes = o_off + al((ng + 2) * 4);
    HIPCHK(c, c->g1.reserve(sc_bytes));
    char* sb = c->g1.as<char>();
    FitArgs a;
    a.n_groups = n_groups;
    a.group_start = in.gs; a.member_hyp = in.mh; a.hyp = in.hyp; a.hyp_cam = in.cam;
    for (int i = 0; i < 9; ++i) a.Rinv.m[i] = Rinv[i];
    a.scale_inv = scale_inv; a.tneg = la::V3{ tneg[0], tneg[1], tneg[2] };
    a.pts = reinterpret_cast<la::V3*>(sb + o_pts); a.dist = reinterpret_cast<float*>(sb + o_dist); a.order = reinterpret_cast<int*>(sb + o_ord);
    a.line_open = reinterpret_cast<unsigned char*>(sb + o_open); a.cam_ids = reinterpret_cast<unsigned*>(sb + o_ci); a.cam_cnt = reinterpret_cast<unsigned*>(sb + o_cc);
    a.out = reinterpret_cast<double*>(sb + o_out); a.out_cnt = reinterpret_cast<int*>(sb + o_cnt);
    int* out_off = reinterpret_cast<int*>(sb + o_off);
    HIPCHK(c, hipMemsetAsync(a.out_cnt, 0, (ng + 1) * 4, st));
    { ProfScope p(c, "fit_clusters", st); hipLaunchKernelGGL(k_fit_clusters, dim3((n_groups + 3) / 4), dim3(256), 0, st, a); }
    size_t tb = 0;
    HIPCHK(c, exclusive_sum_int(nullptr, tb, a.out_cnt, out_off, n_groups + 1, st));
    HIPCHK(c, c->g7.reserve(tb + 256));
    HIPCHK(c, exclusive_sum_int(c->g7.p, tb, a.out_cnt, out_off, n_groups + 1, st));
    int32_t* cnt = static_cast<int32_t*>(malloc((ng + 1) * 4));
    if (!cnt) return fail(c, L3D_ERR_NOMEM, "malloc");
    int total = 0;
    hipError_t e1 = hipMemcpyAsync(cnt, a.out_cnt, ng * 4, hipMemcpyDeviceToHost, st);
    hipError_t e2 = hipMemcpyAsync(&total, out_off + n_groups, 4, hipMemcpyDeviceToHost, st);
    hipError_t e3 = hipStreamSynchronize(st);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { free(cnt); return fail(c, L3D_ERR_HIP, "line fit: read-back failed"); }
    double* packed_host = static_cast<double*>(malloc(((size_t)total * 6 + 1) * 8));
    if (!packed_host) { free(cnt); return fail(c, L3D_ERR_NOMEM, "malloc"); }
    if (total > 0) {
        if (c->g2.reserve((size_t)total * 48 + 64) != hipSuccess) { free(cnt); free(packed_host); return fail(c, L3D_ERR_NOMEM, "line fit: device buffer for the packed segments"); }
        hipLaunchKernelGGL(k_fit_gather, dim3((n_groups + 3) / 4), dim3(256), 0, st, a.group_start, out_off, a.out, n_groups, c->g2.as<double>());
        e1 = hipMemcpyAsync(packed_host, c->g2.p, (size_t)total * 48, hipMemcpyDeviceToHost, st);
        e2 = hipStreamSynchronize(st);
        e3 = hipGetLastError();
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { free(cnt); free(packed_host); return fail(c, L3D_ERR_HIP, "line fit: read-back failed"); }
    }
    *seg_count = cnt; *segs = packed_host; *n_segs = total;
    return L3D_OK;
}

}  // namespace

// clusters: group_start (n_groups + 1) into member_hyp (hypothesis indices in key order); hyp / hyp_cam: all hypotheses (3-D end
// points in the normalised scene) and their camera ids; Rinv (3x3 row-major), scale_inv, tneg: Line3D::inverseTransform.
// Out (callee-allocated, l3d_free): seg_count[g] = 3-D segments of cluster g, segs = 6 doubles per segment, groups back to back.
extern "C" int l3d_fit_clusters(l3d_ctx* c, const int32_t* group_start, int n_groups, const int32_t* member_hyp, const l3d_hypothesis* hyp, const uint32_t* hyp_cam,
                                int n_hyp, const double* Rinv, double scale_inv, const double* tneg, int32_t** seg_count, double** segs, int* n_segs)
{
    if (!c) return L3D_ERR_INVALID;
    if (!seg_count || !segs || !n_segs || n_groups < 0 || n_hyp < 0 || (n_groups > 0 && (!group_start || !member_hyp || !hyp_cam || !Rinv || !tneg)))
        return fail(c, L3D_ERR_INVALID, "bad argument");
    if (n_groups > 0 && !hyp && c->resident_hyp != n_hyp) return fail(c, L3D_ERR_INVALID, "line fit: no resident hypothesis table of that size (l3d_affinity_fill)");
    *seg_count = nullptr; *segs = nullptr; *n_segs = 0;
    if (n_groups == 0) return L3D_OK;
    const int n_members = group_start[n_groups];
    if (group_start[0] != 0 || n_members < 0) return fail(c, L3D_ERR_INVALID, "line fit: group table must start at 0");
    for (int g = 0; g < n_groups; ++g) if (group_start[g + 1] < group_start[g]) return fail(c, L3D_ERR_INVALID, "line fit: group table must ascend");
    for (int i = 0; i < n_members; ++i) if (member_hyp[i] < 0 || member_hyp[i] >= n_hyp) return fail(c, L3D_ERR_INVALID, "line fit: member out of range");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t nm = (size_t)n_members, ng = (size_t)n_groups;
    const size_t o_gs = 0, o_mh = al((ng + 1) * 4), o_hyp = o_mh + al(nm * 4 + 4), o_cam = o_hyp + (hyp ? al((size_t)n_hyp * sizeof(Hypothesis)) : 0),
                 in_bytes = o_cam + al((size_t)n_hyp * 4 + 4);
    HIPCHK(c, c->g0.reserve(in_bytes));
    char* ib = c->g0.as<char>();
    HIPCHK(c, hipMemcpyAsync(ib + o_gs, group_start, (ng + 1) * 4, hipMemcpyHostToDevice, st));
    if (nm) HIPCHK(c, hipMemcpyAsync(ib + o_mh, member_hyp, nm * 4, hipMemcpyHostToDevice, st));
    if (n_hyp && hyp) HIPCHK(c, hipMemcpyAsync(ib + o_hyp, hyp, (size_t)n_hyp * sizeof(Hypothesis), hipMemcpyHostToDevice, st));
    if (n_hyp) HIPCHK(c, hipMemcpyAsync(ib + o_cam, hyp_cam, (size_t)n_hyp * 4, hipMemcpyHostToDevice, st));
    FitStage in{ reinterpret_cast<const int*>(ib + o_gs), reinterpret_cast<const int*>(ib + o_mh),
                 hyp ? reinterpret_cast<const Hypothesis*>(ib + o_hyp) : c->aff_hyp.as<Hypothesis>(), reinterpret_cast<const unsigned*>(ib + o_cam) };
    return fit_core(c, in, n_groups, n_members, Rinv, scale_inv, tneg, seg_count, segs, n_segs);
}

// processClusteredSegments from the LABELS (line3D.cc:1306-1368): the grouping on the device as well.  labels / node_hyp (n_nodes each: the
// cluster label and the hypothesis index of every node) -- NULL: the arrays l3d_perform_clustering_device / l3d_affinity_fill* left on
// the device.  Out: the clusters that were fitted, in ascending label order -- group_start (n_groups + 1), member_hyp -- and their segments.
extern "C" int l3d_fit_labelled_clusters(l3d_ctx* c, const int32_t* labels, const int32_t* node_hyp, int n_nodes, const l3d_hypothesis* hyp, const uint32_t* hyp_cam, int n_hyp,
                                         const double* Rinv, double scale_inv, const double* tneg, int32_t** group_start, int32_t** member_hyp, int* n_groups,
                                         int32_t** seg_count, double** segs, int* n_segs)
{
    if (!c) return L3D_ERR_INVALID;
    if (!group_start || !member_hyp || !n_groups || !seg_count || !segs || !n_segs || n_nodes < 0 || n_hyp < 0 || (n_nodes > 0 && (!hyp_cam || !Rinv || !tneg)))
        return fail(c, L3D_ERR_INVALID, "bad argument");
    *group_start = nullptr; *member_hyp = nullptr; *n_groups = 0; *seg_count = nullptr; *segs = nullptr; *n_segs = 0;
    if (n_nodes == 0) return L3D_OK;
    if (!hyp && c->resident_hyp != n_hyp) return fail(c, L3D_ERR_INVALID, "line fit: no resident hypothesis table of that size (l3d_affinity_fill)");
    if (!labels && c->resident_labels != n_nodes) return fail(c, L3D_ERR_INVALID, "line fit: no resident labels of that size (l3d_perform_clustering_device)");
    if (!node_hyp && c->resident_nodes != n_nodes) return fail(c, L3D_ERR_INVALID, "line fit: no resident node table of that size (l3d_affinity_fill)");
    if (labels) for (int v = 0; v < n_nodes; ++v) if (labels[v] < 0 || labels[v] >= n_nodes) return fail(c, L3D_ERR_INVALID, "line fit: label out of range");
    if (node_hyp) for (int v = 0; v < n_nodes; ++v) if (node_hyp[v] < 0 || node_hyp[v] >= n_hyp) return fail(c, L3D_ERR_INVALID, "line fit: node hypothesis out of range");
    // k_lab_valid counts a cluster's cameras as camera CHANGES between neighbouring members in hypothesis order (line3D.cc:1320,1334:
    // cluster2cameras.size() >= 4): that is the number of distinct cameras only if the hypotheses are numbered camera by camera
    for (int k = 1; k < n_hyp; ++k)
        if (hyp_cam[k] < hyp_cam[k - 1]) return fail(c, L3D_ERR_INVALID, "line fit: hyp_cam must be non-decreasing in hypothesis index (hypotheses numbered view by view)");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t nn = (size_t)n_nodes;
    // staging (g0): group table and members of the fitted clusters (upper bounds: n/4 clusters, n members), hypotheses, cameras, uploads
    const size_t o_gs = 0, o_mh = al((nn / 4 + 2) * 4), o_hyp = o_mh + al(nn * 4 + 4), o_cam = o_hyp + (hyp ? al((size_t)n_hyp * sizeof(Hypothesis)) : 0),
                 o_lab = o_cam + al((size_t)n_hyp * 4 + 4), o_nh = o_lab + (labels ? al(nn * 4) : 0), in_bytes = o_nh + (node_hyp ? al(nn * 4) : 0);
    HIPCHK(c, c->g0.reserve(in_bytes));
    char* ib = c->g0.as<char>();
    if (hyp) HIPCHK(c, hipMemcpyAsync(ib + o_hyp, hyp, (size_t)n_hyp * sizeof(Hypothesis), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(ib + o_cam, hyp_cam, (size_t)n_hyp * 4, hipMemcpyHostToDevice, st));
    if (labels) HIPCHK(c, hipMemcpyAsync(ib + o_lab, labels, nn * 4, hipMemcpyHostToDevice, st));
    if (node_hyp) HIPCHK(c, hipMemcpyAsync(ib + o_nh, node_hyp, nn * 4, hipMemcpyHostToDevice, st));
    const int* d_lab = labels ? reinterpret_cast<const int*>(ib + o_lab) : c->resident_labels_p;
    const int* d_nh = node_hyp ? reinterpret_cast<const int*>(ib + o_nh) : c->resident_nodes_p;
    const unsigned* d_cam = reinterpret_cast<const unsigned*>(ib + o_cam);
    int* d_gs = reinterpret_cast<int*>(ib + o_gs);
    int* d_mh = reinterpret_cast<int*>(ib + o_mh);
    // scratch (g5): two key arrays, flags / ranks / starts / valid / sizes and their scans, hipCUB's temporary storage
    int bits = 1;
    while ((1ll << bits) < (long long)n_nodes) ++bits;
    size_t tb = 0, tb2 = 0;
    HIPCHK(c, sort_keys_u64(nullptr, tb, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, n_nodes, 0, 32 + bits, st));
    HIPCHK(c, exclusive_sum_int(nullptr, tb2, (const int*)nullptr, (int*)nullptr, n_nodes + 1, st));
    tb = std::max(tb, tb2);
    const size_t kb = al(nn * 8), ib4 = al((nn + 2) * 4);
    HIPCHK(c, c->g5.reserve(2 * kb + 8 * ib4 + tb + 256));
    char* sc = c->g5.as<char>();
    unsigned long long *key_in = reinterpret_cast<unsigned long long*>(sc), *key = reinterpret_cast<unsigned long long*>(sc + kb);
    int* flag = reinterpret_cast<int*>(sc + 2 * kb);
    int* rank = reinterpret_cast<int*>(sc + 2 * kb + ib4);
    int* start = reinterpret_cast<int*>(sc + 2 * kb + 2 * ib4);
    int* vflag = reinterpret_cast<int*>(sc + 2 * kb + 3 * ib4);
    int* vsize = reinterpret_cast<int*>(sc + 2 * kb + 4 * ib4);
    int* vrank = reinterpret_cast<int*>(sc + 2 * kb + 5 * ib4);
    int* voff = reinterpret_cast<int*>(sc + 2 * kb + 6 * ib4);
    void* temp = sc + 2 * kb + 8 * ib4;
    const dim3 block(256), grid((n_nodes + 1 + 255) / 256);
    hipLaunchKernelGGL(k_lab_keys, grid, block, 0, st, d_lab, d_nh, n_nodes, key_in);
    HIPCHK(c, sort_keys_u64(temp, tb, key_in, key, n_nodes, 0, 32 + bits, st));
    hipLaunchKernelGGL(k_lab_flags, grid, block, 0, st, key, n_nodes, flag);
    HIPCHK(c, exclusive_sum_int(temp, tb, flag, rank, n_nodes + 1, st));
    hipLaunchKernelGGL(k_lab_starts, grid, block, 0, st, flag, rank, n_nodes, start);
    const int* n_all = rank + n_nodes;                                       // number of clusters (device)
    hipLaunchKernelGGL(k_lab_valid, grid, block, 0, st, key, start, n_all, d_cam, vflag, vsize);
    // (the scans run over n_nodes + 1 entries; beyond the clusters the flags are whatever k_lab_valid left -- it writes 0 at index n_all,
    // and nothing behind it is read)
    int h_all = 0;
    HIPCHK(c, hipMemcpyAsync(&h_all, n_all, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, exclusive_sum_int(temp, tb, vflag, vrank, h_all + 1, st));
    HIPCHK(c, exclusive_sum_int(temp, tb, vsize, voff, h_all + 1, st));
    hipLaunchKernelGGL(k_lab_compact, dim3((h_all + 1 + 3) / 4), block, 0, st, key, start, n_all, vflag, vrank, voff, d_gs, d_mh);
    int tot[2] = { 0, 0 };
    HIPCHK(c, hipMemcpyAsync(&tot[0], vrank + h_all, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(&tot[1], voff + h_all, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    const int ng = tot[0], nm = tot[1];
    if (ng == 0) return L3D_OK;
    int32_t* gs_h = static_cast<int32_t*>(malloc(((size_t)ng + 1) * 4));
    int32_t* mh_h = static_cast<int32_t*>(malloc(((size_t)nm + 1) * 4));
    if (!gs_h || !mh_h) { free(gs_h); free(mh_h); return fail(c, L3D_ERR_NOMEM, "malloc"); }
    hipError_t e1 = hipMemcpyAsync(gs_h, d_gs, ((size_t)ng + 1) * 4, hipMemcpyDeviceToHost, st);
    hipError_t e2 = hipMemcpyAsync(mh_h, d_mh, (size_t)nm * 4, hipMemcpyDeviceToHost, st);
    if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipStreamSynchronize(st); free(gs_h); free(mh_h); return fail(c, L3D_ERR_HIP, "line fit: read-back failed"); }
    FitStage in{ d_gs, d_mh, hyp ? reinterpret_cast<const Hypothesis*>(ib + o_hyp) : c->aff_hyp.as<Hypothesis>(), d_cam };
    const int rc = fit_core(c, in, ng, nm, Rinv, scale_inv, tneg, seg_count, segs, n_segs);   // (synchronises: the two copies above are done)
    if (rc) { (void)hipStreamSynchronize(st); free(gs_h); free(mh_h); return rc; }
    *group_start = gs_h; *member_hyp = mh_h; *n_groups = ng;
    return L3D_OK;
}

void l3d::warm_linefit() { touch_kernel(reinterpret_cast<const void*>(&l3d::k_fit_gather)); }
