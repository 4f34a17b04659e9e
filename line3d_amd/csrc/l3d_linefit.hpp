// l3d_linefit.hpp -- getLineEquation3D + projectToLine (line3D.cc:1392-1597) as functions the host and the device share: the
// arithmetic (which sums in which order, which comparisons) is written once, so the two give the same bits.
//
// A cluster's members arrive in key order (camera, segment) with their 3-D end points already taken back to the caller's
// coordinates (inverseTransform, line3D.cc:1782-1786).  Points are numbered 2*member + {0: P1, 1: P2}.
#pragma once

#include "l3d_linalg.hpp"

namespace l3d {
namespace fit {

using la::M3;
using la::V3;

// Line3D::inverseTransform
L3D_LA_HD inline V3 inverse_transform(const M3& Rinv, double scale_inv, V3 tneg, V3 P) { return la::mul(Rinv, P * scale_inv + tneg); }

// Centre, principal direction (largest eigenvalue of the scatter matrix; sign: largest |component| positive -- Eigen's sign is an
// implementation detail) and the projected end point farthest against the direction (:1479-1540).  get(i) = point i, n2 = 2 x members.
template <class Get>
L3D_LA_HD inline void line_of_points(Get get, int n2, V3& Pc, V3& dir, V3& min_point)
{
    Pc = V3();
    for (int i = 0; i < n2; ++i) Pc = Pc + get(i);
    Pc = Pc / (double)n2;
    M3 Sc;
    for (int i = 0; i < n2; ++i) {
        const V3 d = get(i) - Pc;
        const double dv[3] = { d.x, d.y, d.z };
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) Sc(r, c) += dv[r] * dv[c];
        }
    }
    double w[3]; M3 V;
    la::eig_sym3(Sc, w, V);
    int mx = 0;
    if (w[1] > w[mx]) mx = 1;
    if (w[2] > (mx == 1 ? w[1] : w[0])) mx = 2;
    dir = mx == 0 ? V3{ V(0, 0), V(1, 0), V(2, 0) } : (mx == 1 ? V3{ V(0, 1), V(1, 1), V(2, 1) } : V3{ V(0, 2), V(1, 2), V(2, 2) });
    dir = dir / la::norm(dir);
    {
        const double a[3] = { fabs(dir.x), fabs(dir.y), fabs(dir.z) };
        int k = 0; if (a[1] > a[k]) k = 1; if (a[2] > a[k]) k = 2;
        const double c = k == 0 ? dir.x : (k == 1 ? dir.y : dir.z);
        if (c < 0) dir = dir * -1.0;
    }
    min_point = V3();
    double min_length = 0.0;
    const double dn2 = la::norm(dir) * la::norm(dir);
    for (int i = 0; i < n2; ++i) {
        const V3 proj = Pc + (la::dot(dir, get(i) - Pc) / dn2) * dir;
        const double loc = la::dot(dir, Pc - proj);
        if (loc <= min_length) { min_length = loc; min_point = proj; }
    }
}

// the sort key of a point (:1527-1541: float distance from the first end of the line)
L3D_LA_HD inline float point_dist(V3 P, V3 min_point) { return (float)la::norm(P - min_point); }

// The sweep of projectToLine (:1543-1594) over the points in STABLE ascending order of their distance (order[k] = point): a 3-D
// segment (member) opens at its first end point and closes at its second; while at least three cameras have an open segment the
// line exists.  line_open: one byte per member, cam_ids / cam_cnt: scratch for up to `members` cameras.  emit(start, end).
template <class Get, class Cam, class Emit>
L3D_LA_HD inline int sweep_line(const int* order, int n2, Get get, Cam cam_of_member, unsigned char* line_open, unsigned* cam_ids, unsigned* cam_cnt, Emit emit)
{
    const int members = n2 / 2;
    for (int i = 0; i < members; ++i) line_open[i] = 0;
    int n_cams = 0, n_open_cams = 0, emitted = 0;
    bool opened = false;
    V3 start;
    for (int k = 0; k < n2; ++k) {
        const int p = order[k], member = p >> 1;
        const unsigned cam = cam_of_member(member);
        int ci = 0;
        while (ci < n_cams && cam_ids[ci] != cam) ++ci;
        if (ci == n_cams) { cam_ids[n_cams] = cam; cam_cnt[n_cams] = 0; ++n_cams; }
        if (!line_open[member]) { line_open[member] = 1; if (cam_cnt[ci]++ == 0) ++n_open_cams; }
        else { line_open[member] = 0; if (--cam_cnt[ci] == 0) --n_open_cams; }
        if (opened && n_open_cams < 3) { emit(start, get(p)); ++emitted; opened = false; }
        else if (!opened && n_open_cams >= 3) { start = get(p); opened = true; }
    }
    return emitted;
}

}  // namespace fit
}  // namespace l3d
