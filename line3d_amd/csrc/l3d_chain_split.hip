// l3d_chain_split.hip -- the per-view step of Line3D::matchViews' dependency chain, reduced to what really depends on it.
//
// K_verify_matches (cudawrapper.cu:614-714) scores a hypothesis by summing, over the OTHER cameras in ascending order, the
// best confidence any witness of that camera gives it.  The candidates of a view are its stage-1 candidates (cameras still to
// be matched, known without any earlier result) and the reverse matches handed over by earlier views (line3D.cc:838-872: the
// chain).  Per-camera maxima are order independent and a camera is either one or the other, so the evaluation splits exactly:
//
//   part A  (no dependency on earlier views; k_verify_window with `max_out`, run ahead on its own stream)
//           stage-1 hypotheses x stage-1 witnesses: per hypothesis the maxima of the cameras to be matched, and their sum in
//           ascending camera order (the final confidence when no reverse match supports the hypothesis -- "+0.0f" terms are
//           exact no-ops).  ~97 % of the pair evaluations of a view.
//   chain   (this file; ONE launch per view on the chain stream, plus the kept-list writer off the critical path)
//           k_chain_verify: stage-1 hypotheses x reverse witnesses, reverse hypotheses x all witnesses, the sums over ALL
//           cameras in ascending order (maxima of part A inserted at their place), per-segment best hypothesis / kept count,
//           and the hand-over: every kept stage-1 hypothesis is appended to a bin (this view, target camera, target segment)
//           from which the later view's workgroup of that segment reads its reverse matches directly -- the reference's
//           round trip through the host and a file (line3D.cc:868-872, view.cc:185-224) and the count / scan / scatter /
//           run-ordering launches of the first resident chain are gone.
//           k_chain_kept: ordered compaction of the kept matches (conf > 1, cudawrapper.cu:1089-1110) of both candidate
//           kinds into the view's slice of the kept arena, (segment, camera, target) order, and the view's result record.
//
// Results are bit-identical to the per-view entry point and to the first resident chain (tests).
#include <algorithm>
#include <mutex>

#include "l3d_geometry.hpp"
#include "l3d_kernels.hpp"
#include "l3d_verify_eval.hpp"

namespace l3d {

constexpr int kRevLds = 128;                 // reverse matches of one segment held in LDS (more: read from the L2-resident store)
constexpr int kCQ = 256;                     // per-wave ring of (hypothesis, witness) pairs that passed the 1-D depth tests (16-byte entries)

// One workgroup per source segment y of the view (longest segments first).
//   1. gather: the reverse matches of y out of the sources' bins -- two rounds of loads (a thread per source reads its bin
//      counter, a thread per record reads the record) --, ordered by their first depth (rank by counting; in LDS when there are
//      at most kRevLds, else in the ring slot's store in L2).  The order only serves the window walk below; k_chain_kept
//      restores the reference's (camera, target) order for the few kept ones.
//   2. rounds of 256 hypotheses, one per thread: first the stage-1 candidates, then the reverse matches.  A hypothesis walks the
//      reverse matches r whose first depth lies in a conservative window around its own (binary search + a few steps); pairs that
//      pass the 1-D pre-tests go into a per-wave ring -- (hypothesis, witness r) and, for a stage-1 candidate i, also (hypothesis
//      r, witness i) -- and are evaluated 64 at a time with the reference's float sequence (ONE evaluation site in the code).
//      After its round a stage-1 candidate's confidence is the sum over all cameras in ascending order, part A's maxima at the
//      cameras to be matched, this launch's at the source cameras (nothing to add for most: part A's sum stands); a kept
//      hypothesis is appended to its bin towards the later view.
//   3. the sums of the reverse hypotheses, the per-segment epilogue.
__global__ __launch_bounds__(256) void k_chain_verify(ChainSplitArgs a)
{
    extern __shared__ __align__(16) unsigned char s_raw[];
    __shared__ unsigned u_key[kRevLds];                                      // unsorted staging of the reverse matches
    __shared__ float u_d[4][kRevLds];
    __shared__ unsigned c_key[kRevLds];                                      // sorted by first depth: (camera << 16) | id
    __shared__ float c_d1[kRevLds], c_d2[kRevLds];
    __shared__ int s_g[kSplitMaxSrc], s_c[kSplitMaxSrc], s_pre[kSplitMaxSrc + 1], s_cam[kSplitMaxSrc], s_view[kSplitMaxSrc];
    __shared__ int s_n, s_base, s_ovf;
    __shared__ int s_rk[4];
    __shared__ float s_rb[4], s_rd1[4], s_rd2[4];
    __shared__ unsigned s_rkey[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int y = a.seg_order ? a.seg_order[blockIdx.x] : (int)blockIdx.x;
    const int N = a.N, n_src = a.n_src, B = a.B;
    float* smaxB = reinterpret_cast<float*>(s_raw);                        // [n_src][256]
    float* smaxR_lds = smaxB + max(n_src, 1) * 256;                         // [kRevLds][N]
    uint4* q = reinterpret_cast<uint4*>(smaxR_lds + kRevLds * N) + wave * kCQ;   // this wave's ring (the layout keeps it 16-byte aligned)
    float* sP = reinterpret_cast<float*>(reinterpret_cast<uint4*>(smaxR_lds + kRevLds * N) + 4 * kCQ);   // [N][12]
    int* sOff = reinterpret_cast<int*>(sP + N * 12);                        // [N]
    int* slotmap = sOff + N;                                                // [N]: tbm slot of a camera, or -1 (a source camera)
    int* srcslot = slotmap + N;                                             // [N]: source slot of a camera (0 for the others: never used)
    int* sBso = srcslot + N;                                                // [N]: first bin (relative) of a camera to be matched

    // everything the segment needs from global memory that does not depend on anything else is requested first
    const int rowA_total = a.rowA[(size_t)a.S * N];
    const int startA = a.rowA[y * N];
    const int endA = a.rowA[(y + 1) * N];
    const float4 sseg = a.src_segs[y];
    SplitSource mysrc;
    mysrc.bin_first = -1; mysrc.n_bins = 0; mysrc.view = 0; mysrc.cam = 0;
    if (tid < n_src) mysrc = tid < kSplitInlineSrc ? a.src_inl[tid] : a.sources[tid];
    const bool overflow = rowA_total > a.cand_cap;                          // part A had no room: the chain is re-run from this view
    if (overflow || a.debug == 1) {
        if (tid == 0) { a.kept_cnt[y] = 0; a.best_depths[y] = make_float2(-1.0f, -1.0f); a.rev_seg[y] = make_int2(0, 0); if (blockIdx.x == 0 && overflow) atomicOr(a.flags, 1); }
        return;
    }
    // ---- 1. gather.  The bins' counters by a thread per source (the first sources travel in the kernel arguments: no table
    // load in front of the counter load) ...
    if (tid < n_src) {
        const int g = (mysrc.bin_first >= 0 && y < mysrc.n_bins) ? mysrc.bin_first + y : -1;
        s_g[tid] = g; s_c[tid] = g >= 0 ? a.bin_cnt[g] : 0; s_cam[tid] = mysrc.cam; s_view[tid] = mysrc.view;
    }
    for (int i = tid; i < N * 12; i += 256) sP[i] = a.P[i];
    for (int i = tid; i < N; i += 256) { sOff[i] = a.offsets[i].x; slotmap[i] = -1; srcslot[i] = 0; sBso[i] = 0; }
    if (tid == 0) s_ovf = 0;
    __syncthreads();
    for (int j = tid; j < a.n_tbm; j += 256) { const int cam = a.tbm[j]; slotmap[cam] = j; sBso[cam] = a.bin_slot_off[j]; }
    if (tid < n_src) srcslot[s_cam[tid]] = tid;
    int cnt_bins = 0, anyovf = 0;                                           // (every thread sums the few counters itself: no second barrier)
    for (int si = 0; si < n_src; ++si) { const int c = s_c[si]; if (tid == 0) s_pre[si] = cnt_bins; cnt_bins += min(c, B); anyovf |= c > B; }
    if (tid == 0) s_pre[n_src] = cnt_bins;
    if (anyovf) {                                                           // (rare) records beyond a bin's capacity sit in the source view's overflow list
        for (int si = 0; si < n_src; ++si) {
            if (s_c[si] <= B) continue;
            const int g = s_g[si];
            const int no = min(a.ovf_cnt_all[s_view[si]], a.ovf_cap);
            const int2* ok = a.ovf_key_all + (size_t)s_view[si] * a.ovf_cap;
            int mine = 0;
            for (int e = tid; e < no; e += 256) mine += ok[e].x == g;
            if (mine) atomicAdd(&s_ovf, mine);
        }
        __syncthreads();
    }
    const int n_rev_all = cnt_bins + (anyovf ? s_ovf : 0);
    // the segment's slice of the reverse store: its own fixed slot of rev_stride entries, or (more than that) a piece of the shared tail
    int base = y * a.rev_stride;
    if (n_rev_all > a.rev_stride) {
        if (tid == 0) {
            int b2 = a.S * a.rev_stride + atomicAdd(a.rev_total, n_rev_all);
            if (b2 + n_rev_all > a.rev_cap) { atomicOr(a.flags, 8); b2 = -1; }
            s_base = b2;
        }
        __syncthreads();
        base = s_base;
    }
    if (tid == 0) s_n = cnt_bins;                                           // overflow records are appended behind the bins' records
    const int n_rev = base < 0 ? 0 : n_rev_all;                             // (no room: the view is re-run; keep going without them)
    if (tid == 0) a.rev_seg[y] = make_int2(max(base, 0), n_rev);
    const bool in_lds = n_rev <= kRevLds;
    float* smaxR = in_lds ? smaxR_lds : a.rev_max + (size_t)max(base, 0) * N;   // [n_rev][N] per-(reverse hypothesis, camera) maxima
    if (n_rev > 0) {
        __syncthreads();                                                    // s_pre, s_n
        // ... then the records by a thread per record (position = prefix of the bins' counts: no atomics, no order dependence)
        for (int idx = tid; idx < cnt_bins; idx += 256) {
            int si = 0;
            while (s_pre[si + 1] <= idx) ++si;
            const int k = idx - s_pre[si];
            const size_t o = (size_t)s_g[si] * B + k;
            const unsigned key = ((unsigned)s_cam[si] << 16) | a.bin_id[o];
            const float4 d = a.bin_depth[o];
            if (in_lds) { u_key[idx] = key; u_d[0][idx] = d.x; u_d[1][idx] = d.y; u_d[2][idx] = d.z; u_d[3][idx] = d.w; }
            else { a.rev_tmp_meta[base + idx] = make_uint2(key, 0u); a.rev_tmp_depth[base + idx] = d; }
        }
        if (anyovf) {
            for (int si = 0; si < n_src; ++si) {
                if (s_c[si] <= B) continue;
                const int g = s_g[si];
                const int no = min(a.ovf_cnt_all[s_view[si]], a.ovf_cap);
                const int2* ok = a.ovf_key_all + (size_t)s_view[si] * a.ovf_cap;
                const float4* od = a.ovf_depth_all + (size_t)s_view[si] * a.ovf_cap;
                for (int e = tid; e < no; e += 256) {
                    const int2 k2 = ok[e];
                    if (k2.x != g) continue;
                    const float4 d = od[e];
                    const int pos = atomicAdd(&s_n, 1);
                    const unsigned key = ((unsigned)s_cam[si] << 16) | (unsigned)k2.y;
                    if (in_lds) { u_key[pos] = key; u_d[0][pos] = d.x; u_d[1][pos] = d.y; u_d[2][pos] = d.z; u_d[3][pos] = d.w; }
                    else { a.rev_tmp_meta[base + pos] = make_uint2(key, 0u); a.rev_tmp_depth[base + pos] = d; }
                }
            }
        }
        for (int i = tid; i < n_rev * N; i += 256) smaxR[i] = 0.0f;
        if (!in_lds) __threadfence_block();
        __syncthreads();
        // order by (first depth, key): rank by counting (keys are distinct)
        if (in_lds) {
            for (int r = tid; r < n_rev; r += 256) {
                const float d1 = u_d[0][r];
                const unsigned key = u_key[r];
                int rk = 0;
                for (int i = 0; i < n_rev; ++i) { const float o1 = u_d[0][i]; rk += o1 < d1 || (o1 == d1 && u_key[i] < key); }
                a.rev_meta[base + rk] = make_uint2(key & 0xffffu, key >> 16);
                a.rev_depth[base + rk] = make_float4(d1, u_d[1][r], u_d[2][r], u_d[3][r]);
                c_key[rk] = key; c_d1[rk] = d1; c_d2[rk] = u_d[1][r];
            }
        } else {
            for (int r = tid; r < n_rev; r += 256) {
                const float4 d = a.rev_tmp_depth[base + r];
                const unsigned key = a.rev_tmp_meta[base + r].x;
                int rk = 0;
                for (int i = 0; i < n_rev; ++i) { const float o1 = a.rev_tmp_depth[base + i].x; rk += o1 < d.x || (o1 == d.x && a.rev_tmp_meta[base + i].x < key); }
                a.rev_meta[base + rk] = make_uint2(key & 0xffffu, key >> 16);
                a.rev_depth[base + rk] = d;
            }
            __threadfence_block();
        }
    }
    __syncthreads();

    const f3 C = mk3(a.C_src[0], a.C_src[1], a.C_src[2]);
    const f3 ray1 = normalize(mat3_apply(a.RtKinv_src, mk3(sseg.x, sseg.y, 1.0f)));
    const f3 ray2 = normalize(mat3_apply(a.RtKinv_src, mk3(sseg.z, sseg.w, 1.0f)));
    const float c_inf = __builtin_fmaxf(__builtin_fabsf(C.x), __builtin_fmaxf(__builtin_fabsf(C.y), __builtin_fabsf(C.z)));
    const float two_sig_d = 2.0f * (a.sigma_p * a.sigma_p);
    const float two_sig_a = 2.0f * (a.sigma_a * a.sigma_a);
    const bool gate = a.spatial_k > 0.0f;
    // 1-D pre-tests (l3d_verify_eval.hpp, window_margin): a witness that passes the reference's gate for a hypothesis at depth d_y
    // has |d_y - d_i| <= unc*1.00001 + 2e-6 (|d_y| + |d_i| + |C|inf) with unc = spatial_k |C - X| <= spatial_k |d_y| (1 + 1e-6)
    const float kw = a.spatial_k * 1.00002f + 2.0e-6f;
    const float cw = 2.0e-6f * c_inf;
    const bool windowed = gate && kw < 0.25f;                               // else every reverse match is walked
    const float kwalk = 1.001f / (1.0f - kw);                               // the walk's range covers the windows of BOTH roles

    // reverse match number p (first-depth order): key, depths
    auto rev_get = [&](int p, unsigned& key, float& d1, float& d2) {
        if (in_lds) { key = c_key[p]; d1 = c_d1[p]; d2 = c_d2[p]; }
        else { const uint2 m = a.rev_meta[base + p]; const float4 dd = a.rev_depth[base + p]; key = (m.y << 16) | m.x; d1 = dd.x; d2 = dd.y; }
    };
    auto rev_d1 = [&](int p) { return in_lds ? c_d1[p] : a.rev_depth[base + p].x; };

    const int mA = endA - startA;
    const bool walk = n_rev > 0 && a.debug != 3;
    const int roundsA = (mA + 255) >> 8, roundsR = walk ? (n_rev + 255) >> 8 : 0;

    int kept_l = 0;
    float best_l = 0.0f, bestd1_l = 0.0f, bestd2_l = 0.0f;
    unsigned bestk_l = 0xffffffffu;
    auto consider = [&](float conf, unsigned key, float d1, float d2) {    // kept count + first strict maximum in candidate order
        kept_l += conf > 1.0f;
        if (conf > 0.5f && (conf > best_l || (conf == best_l && key < bestk_l))) { best_l = conf; bestk_l = key; bestd1_l = d1; bestd2_l = d2; }
    };

    // the next round's records are requested while the current round is walked
    float cf_n = 0.0f;
    float4 dd_n = make_float4(0.f, 0.f, 0.f, 0.f);
    uint2 mm_n = make_uint2(0u, 0xffu);
    auto request = [&](int round) {
        const int i = round * 256 + tid;
        cf_n = 0.0f; dd_n = make_float4(0.f, 0.f, 0.f, 0.f); mm_n = make_uint2(0u, 0xffu);
        if (round < roundsA && i < mA) { cf_n = a.confA[startA + i]; if (walk) { dd_n = a.depthsA[startA + i]; mm_n = a.metaA[startA + i]; } }
    };
    request(0);
    for (int round = 0; round < roundsA + roundsR; ++round) {
        const bool modeA = round < roundsA;
        // this round's hypothesis of the thread: a stage-1 candidate (mode A) or a reverse match (mode R)
        const int i = round * 256 + tid;                                     // mode A: index in the segment's stage-1 candidates
        const int rown = (round - roundsA) * 256 + tid;                      // mode R: index in the reverse matches
        float conf = cf_n;
        float4 d = dd_n;
        uint2 meta = mm_n;
        bool hv = modeA ? i < mA : rown < n_rev;
        if (!modeA) { unsigned kh = 0; d = make_float4(0.f, 0.f, 0.f, 0.f); if (hv) rev_get(rown, kh, d.x, d.y); meta = make_uint2(kh & 0xffffu, kh >> 16); }
        request(round + 1);
        if (modeA && !walk && conf > 0.5f) { d = a.depthsA[startA + i]; meta = a.metaA[startA + i]; }
        if (walk) {
            if (modeA) for (int si = 0; si < n_src; ++si) smaxB[si * 256 + tid] = 0.0f;
            const unsigned cam_h = meta.y;
            const float ad1 = __builtin_fabsf(d.x), ad2 = __builtin_fabsf(d.y);
            const float wi1 = kw * ad1 + cw, wi2 = kw * ad2 + cw;               // own windows (as hypothesis), without the witness term
            const unsigned wct_h = (cam_h << 16) | meta.x;
            const unsigned dest_h = modeA ? (unsigned)tid : (0x80000000u | (unsigned)rown);
            int p = 0, pend = 0;
            float hi = 0.0f;
            if (hv) {
                pend = n_rev; hi = __builtin_inff();
                if (windowed) {
                    const float W = (kw * ad1 + cw) * kwalk + 4.0e-6f * ad1;
                    const float x = d.x - W;
                    int lo = 0, up = n_rev;                                      // first p with d1[p] >= x
                    while (lo < up) { const int mid = (lo + up) >> 1; if (rev_d1(mid) < x) lo = mid + 1; else up = mid; }
                    p = lo; hi = d.x + W;
                }
            }
            int head = 0, count = 0;                                            // wave-uniform ring state
            for (;;) {
                const bool act = p < pend;
                unsigned key = 0; float rd1 = 0.0f, rd2 = 0.0f;
                if (act) rev_get(p, key, rd1, rd2);
                const bool in = act && rd1 <= hi;
                const bool more = __any(in);
                if (more) {
                    const bool pair = in && (key >> 16) != cam_h;              // :674 (other cameras only)
                    const float df1 = __builtin_fabsf(d.x - rd1), df2 = __builtin_fabsf(d.y - rd2);
                    const float ar1 = __builtin_fabsf(rd1), ar2 = __builtin_fabsf(rd2);
                    // own hypothesis, r witness
                    const bool pb = pair && (!gate || (df1 <= wi1 + 2.0e-6f * ar1 && df2 <= wi2 + 2.0e-6f * ar2));
                    const unsigned long long mb = __ballot(pb);
                    if (pb) q[(head + count + __popcll(mb & ((1ull << lane) - 1ull))) & (kCQ - 1)] = make_uint4(dest_h, key, __float_as_uint(rd1), __float_as_uint(rd2));
                    count += __popcll(mb);
                    // r hypothesis, the stage-1 candidate witness
                    const bool pc = modeA && pair && (!gate || (df1 <= kw * ar1 + cw + 2.0e-6f * ad1 && df2 <= kw * ar2 + cw + 2.0e-6f * ad2));
                    const unsigned long long mc = __ballot(pc);
                    if (pc) q[(head + count + __popcll(mc & ((1ull << lane) - 1ull))) & (kCQ - 1)] = make_uint4(0x80000000u | (unsigned)p, wct_h, __float_as_uint(d.x), __float_as_uint(d.y));
                    count += __popcll(mc);
                    if (in) ++p; else pend = p;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                }
                if (count >= 64 || (!more && count > 0)) {
                    // ---- the ONE evaluation site: up to 64 queued pairs, one per lane.  An entry carries the destination of the
                    // per-camera maximum (bit 31 clear: this round's hypothesis of thread `dest` -- its depths come from that lane's
                    // registers --, smaxB[source slot][thread]; set: reverse hypothesis number dest & 0x7fffffff, smaxR[number][camera])
                    // and the witness (camera, target id, two depths); everything else about the hypothesis is recomputed from its
                    // two depths with the reference's operations (cudawrapper.cu:644-645,390-394)
                    const int n = min(count, 64);
                    uint4 e = make_uint4(0u, 0u, 0u, 0u);
                    if (lane < n) e = q[(head + lane) & (kCQ - 1)];
                    const bool forR = (e.x & 0x80000000u) != 0;
                    const int rh = (int)(e.x & 0x7fffffffu);
                    float hd1 = __shfl(d.x, (int)(e.x & 63u)), hd2 = __shfl(d.y, (int)(e.x & 63u));
                    if (lane < n && forR) {
                        if (in_lds) { hd1 = c_d1[rh]; hd2 = c_d2[rh]; }
                        else { const float4 hdd = a.rev_depth[base + rh]; hd1 = hdd.x; hd2 = hdd.y; }
                    }
                    if (lane < n && a.debug != 4) {
                        const int cam = (int)(e.y >> 16), tgt = (int)(e.y & 0xffffu);
                        const float4 tq = a.tgt_segs[sOff[cam] + tgt];
                        const f3 X1 = C + hd1 * ray1;                          // D_unproject_point_src, cudawrapper.cu:644-645
                        const f3 X2 = C + hd2 * ray2;
                        const f3 v1 = normalize(X1 - X2);
                        float T1 = 0.0f, T2 = 0.0f;
                        if (gate) {
                            T1 = sq_threshold(a.spatial_k * length(C - X1));   // cudawrapper.cu:390-394
                            T2 = sq_threshold(a.spatial_k * length(C - X2));
                        }
                        const float cfw = witness_conf(C, ray1, ray2, X1, X2, v1, T1, T2, gate, __uint_as_float(e.z), __uint_as_float(e.w), sP + cam * 12, tq, two_sig_d, two_sig_a);
                        if (cfw > 0.5f) {                                        // :699-704 (max over the camera's witnesses)
                            float* slot = forR ? &smaxR[(size_t)rh * N + cam] : &smaxB[srcslot[cam] * 256 + (int)e.x];
                            atomicMax(reinterpret_cast<int*>(slot), __float_as_int(cfw));
                        }
                    }
                    head = (head + n) & (kCQ - 1);
                    count -= n;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                }
                if (!more && count == 0) break;
            }
            if (modeA && hv) {
                // the sum over ALL cameras in ascending order (cudawrapper.cu:677-687,709): part A's maxima at the cameras to be
                // matched (all zero when its sum is zero: the row is only stored otherwise), this launch's at the source cameras;
                // without any reverse support part A's sum stands, bit for bit
                bool any = false;
                for (int si = 0; si < n_src; ++si) any = any || smaxB[si * 256 + tid] != 0.0f;
                if (any) {
                    const float* mo = a.maxA + (size_t)(startA + i) * a.max_stride;
                    const bool rowA_nonzero = conf != 0.0f;
                    conf = 0.0f;
                    for (int c = 0; c < N; ++c) { const int j = slotmap[c]; conf += j >= 0 ? (rowA_nonzero ? mo[j] : 0.0f) : smaxB[srcslot[c] * 256 + tid]; }
                    a.confA[startA + i] = conf;
                }
            }
        }
        if (modeA && conf > 0.5f) {
            consider(conf, (meta.y << 16) | meta.x, d.x, d.y);
            if (conf > 1.0f) {
                // hand-over to the later view `meta.y` (line3D.cc:838-858): (seg, tgt) swap roles, the depth pairs swap
                const int g = a.bin_first + sBso[meta.y] + (int)meta.x;
                const int pos = atomicAdd(&a.bin_cnt[g], 1);
                const float4 rd = make_float4(d.z, d.w, d.x, d.y);
                if (pos < B) { a.bin_id[(size_t)g * B + pos] = (unsigned)y; a.bin_depth[(size_t)g * B + pos] = rd; }
                else {
                    const int e = atomicAdd(&a.ovf_cnt_all[a.view_index], 1);
                    if (e < a.ovf_cap) { a.ovf_key_all[(size_t)a.view_index * a.ovf_cap + e] = make_int2(g, y); a.ovf_depth_all[(size_t)a.view_index * a.ovf_cap + e] = rd; }
                    else atomicOr(a.flags, 4);
                }
            }
        }
    }
    if (!in_lds) __threadfence_block();
    __syncthreads();
    // ---- 3. confidences of the reverse hypotheses
    for (int r = tid; r < n_rev; r += 256) {
        float conf = 0.0f;
        for (int c = 0; c < N; ++c) conf += smaxR[(size_t)r * N + c];
        a.rev_conf[base + r] = conf;
        unsigned key; float d1, d2;
        rev_get(r, key, d1, d2);
        consider(conf, key, d1, d2);
    }
    // ---- per-segment epilogue (cudawrapper.cu:1037-1062, :1096): kept count, depths of the first best hypothesis
    for (int o = 32; o > 0; o >>= 1) {
        kept_l += __shfl_down(kept_l, o);
        const float ob = __shfl_down(best_l, o), od1 = __shfl_down(bestd1_l, o), od2 = __shfl_down(bestd2_l, o);
        const unsigned ok = __shfl_down(bestk_l, o);
        if (ob > best_l || (ob == best_l && ok < bestk_l)) { best_l = ob; bestk_l = ok; bestd1_l = od1; bestd2_l = od2; }
    }
    if (lane == 0) { s_rk[wave] = kept_l; s_rb[wave] = best_l; s_rkey[wave] = bestk_l; s_rd1[wave] = bestd1_l; s_rd2[wave] = bestd2_l; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) {
            kept_l += s_rk[w];
            if (s_rb[w] > best_l || (s_rb[w] == best_l && s_rkey[w] < bestk_l)) { best_l = s_rb[w]; bestk_l = s_rkey[w]; bestd1_l = s_rd1[w]; bestd2_l = s_rd2[w]; }
        }
        a.kept_cnt[y] = kept_l;
        a.best_depths[y] = best_l > 0.5f ? make_float2(bestd1_l, bestd2_l) : make_float2(-1.0f, -1.0f);   // (marker: not in the median list)
    }
}

// Kept matches of the view (conf > 1, cudawrapper.cu:1089-1110) into its slice of the kept arena in (segment, camera, target)
// order: one workgroup per segment sums the kept counts in front of its segment itself; inside the segment the stage-1
// candidates (cameras to be matched) are already in order, so a record's place is its rank among the kept stage-1 candidates
// plus the number of kept reverse matches at lower cameras; a kept reverse match (first-depth order in the store, few per
// segment) is ranked by (camera, target) among its kind by counting, plus the kept stage-1 candidates at lower cameras.
// Workgroup 0 writes the view's result record (device copy for the arena chain, host-mapped copy for the host).
__global__ __launch_bounds__(256) void k_chain_kept(ChainSplitArgs a, const ChainResult* __restrict__ prev, int arena_cap, ChainResult* __restrict__ res,
                                                    ChainResult* __restrict__ res_host, Match* __restrict__ arena)
{
    __shared__ int s_red[12];
    __shared__ int s_keptA[256], s_keptR[256], s_baseA[256], s_baseR[256];   // per local camera (N <= 255)
    __shared__ int s_cnt[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int y = blockIdx.x, N = a.N, S = a.S;
    int before = 0, total = 0, nrev = 0;
    for (int i = tid; i < S; i += 256) { const int v = a.kept_cnt[i]; total += v; if (i < y) before += v; nrev += a.rev_seg[i].y; }
    for (int o = 32; o > 0; o >>= 1) { before += __shfl_down(before, o); total += __shfl_down(total, o); nrev += __shfl_down(nrev, o); }
    if (lane == 0) { s_red[wave] = before; s_red[4 + wave] = total; s_red[8 + wave] = nrev; }
    for (int c = tid; c < N; c += 256) { s_keptA[c] = 0; s_keptR[c] = 0; }
    __syncthreads();
    before = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    total = s_red[4] + s_red[5] + s_red[6] + s_red[7];
    nrev = s_red[8] + s_red[9] + s_red[10] + s_red[11];
    const int flags = *a.flags;
    ChainResult r;
    r.R = a.rowA[(size_t)S * N] + nrev;
    r.overflow = flags;
    r.n_kept = flags ? 0 : total;
    r.kept_base = prev ? prev->kept_base + prev->n_kept : 0;
    if (r.kept_base + r.n_kept > arena_cap) { r.overflow |= 2; r.n_kept = 0; }
    if (y == 0 && tid == 0) { *res = r; *res_host = r; }
    if (r.overflow || y >= S) return;
    if (a.kept_cnt[y] == 0) return;
    Match* out = arena + r.kept_base + before;

    const int startA = a.rowA[y * N];
    const int mA = a.rowA[(y + 1) * N] - startA;
    const int2 rs = a.rev_seg[y];
    // kept reverse matches per camera
    for (int r2 = tid; r2 < rs.y; r2 += 256) if (a.rev_conf[rs.x + r2] > 1.0f) atomicAdd(&s_keptR[a.rev_meta[rs.x + r2].y], 1);
    if (mA <= 2048) {
        // one round of loads for the confidences of the whole segment, one for the records of the kept ones
        float c[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int i = (wave + 4 * q) * 64 + lane; c[q] = i < mA ? a.confA[startA + i] : 0.0f; }
        unsigned long long b[8];
        uint2 meta[8];
        float4 d[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            b[q] = __ballot(c[q] > 1.0f);
            if (lane == 0) s_cnt[wave + 4 * q] = __popcll(b[q]);
            if (c[q] > 1.0f) { const int i = (wave + 4 * q) * 64 + lane; meta[q] = a.metaA[startA + i]; d[q] = a.depthsA[startA + i]; }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) if (c[q] > 1.0f) atomicAdd(&s_keptA[meta[q].y], 1);
        __syncthreads();
        if (tid == 0) {
            int ra = 0, rr = 0;
            for (int cc = 0; cc < N; ++cc) { s_baseA[cc] = rr; s_baseR[cc] = ra; ra += s_keptA[cc]; rr += s_keptR[cc]; }
        }
        const int v = lane < 32 ? s_cnt[lane] : 0;
        int incl = v;
        for (int o = 1; o < 32; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
        const int excl = incl - v;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int off = __shfl(excl, wave + 4 * q);
            if (c[q] > 1.0f) {
                Match rec;
                rec.segID1 = (unsigned)y; rec.camID2 = a.local2global[meta[q].y]; rec.segID2 = meta[q].x;
                rec.depths[0] = d[q].x; rec.depths[1] = d[q].y; rec.depths[2] = d[q].z; rec.depths[3] = d[q].w;
                rec.confidence = c[q] / 2.0f;                                // confidence_norm, cudawrapper.cu:1089,1098
                out[off + __popcll(b[q] & ((1ull << lane) - 1ull)) + s_baseA[meta[q].y]] = rec;
            }
        }
    } else {
        for (int i = tid; i < mA; i += 256) if (a.confA[startA + i] > 1.0f) atomicAdd(&s_keptA[a.metaA[startA + i].y], 1);
        __syncthreads();
        if (tid == 0) {
            int ra = 0, rr = 0;
            for (int cc = 0; cc < N; ++cc) { s_baseA[cc] = rr; s_baseR[cc] = ra; ra += s_keptA[cc]; rr += s_keptR[cc]; }
        }
        __syncthreads();
        int run = 0;
        for (int b0 = 0; b0 < mA; b0 += 256) {                              // ordered compaction, 256 candidates per round
            const int i = b0 + tid;
            const float c = i < mA ? a.confA[startA + i] : 0.0f;
            const unsigned long long bm = __ballot(c > 1.0f);
            if (lane == 0) s_cnt[wave] = __popcll(bm);
            __syncthreads();
            int off = run;
            for (int w = 0; w < wave; ++w) off += s_cnt[w];
            const int tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            if (c > 1.0f) {
                const uint2 meta = a.metaA[startA + i];
                const float4 d = a.depthsA[startA + i];
                Match rec;
                rec.segID1 = (unsigned)y; rec.camID2 = a.local2global[meta.y]; rec.segID2 = meta.x;
                rec.depths[0] = d.x; rec.depths[1] = d.y; rec.depths[2] = d.z; rec.depths[3] = d.w;
                rec.confidence = c / 2.0f;
                out[off + __popcll(bm & ((1ull << lane) - 1ull)) + s_baseA[meta.y]] = rec;
            }
            run += tot;
            __syncthreads();
        }
    }
    for (int i = tid; i < rs.y; i += 256) {
        const float c = a.rev_conf[rs.x + i];
        if (!(c > 1.0f)) continue;
        const uint2 meta = a.rev_meta[rs.x + i];
        const unsigned key = (meta.y << 16) | meta.x;
        int rk = 0;
        for (int j = 0; j < rs.y; ++j) {
            if (!(a.rev_conf[rs.x + j] > 1.0f)) continue;
            const uint2 mj = a.rev_meta[rs.x + j];
            rk += ((mj.y << 16) | mj.x) < key;
        }
        const float4 d = a.rev_depth[rs.x + i];
        Match rec;
        rec.segID1 = (unsigned)y; rec.camID2 = a.local2global[meta.y]; rec.segID2 = meta.x;
        rec.depths[0] = d.x; rec.depths[1] = d.y; rec.depths[2] = d.z; rec.depths[3] = d.w;
        rec.confidence = c / 2.0f;
        out[s_baseR[meta.y] + rk] = rec;                       // kept stage-1 candidates of lower cameras + kept reverse matches with a smaller key come first
    }
}

size_t chain_split_lds_bytes(int N, int n_src) { return ((size_t)std::max(n_src, 1) * 256 + (size_t)kRevLds * N + 4 * 4 * kCQ + (size_t)N * 16) * 4 + 64; }
bool chain_split_supported(int N) { return N <= kSplitMaxSrc && chain_split_lds_bytes(N, N) <= 60 * 1024; }

static bool split_lds_opt_in()
{
    static std::mutex mu;
    static unsigned char done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lk(mu);
    if (done[dev]) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_verify), hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024) != hipSuccess) return false;
    done[dev] = 1;
    return true;
}

void launch_chain_verify(const ChainSplitArgs& a, hipStream_t st)
{
    if (a.S <= 0 || !split_lds_opt_in()) return;
    hipLaunchKernelGGL(k_chain_verify, dim3(a.S), dim3(256), chain_split_lds_bytes(a.N, a.n_src), st, a);
}
void launch_chain_kept(const ChainSplitArgs& a, const ChainResult* prev, int arena_cap, ChainResult* res, ChainResult* res_host, Match* arena, hipStream_t st)
{
    hipLaunchKernelGGL(k_chain_kept, dim3(std::max(1, a.S)), dim3(256), 0, st, a, prev, arena_cap, res, res_host, arena);
}

}  // namespace l3d
