// l3d_linalg.hpp -- dependency-free double-precision 3x3 / 4x4 algebra for the host pipeline
// (the reference uses Eigen 3, which is not a dependency of this build).
#pragma once

#include <cmath>
#include <cstring>

// (the line fit runs these on the device too: l3d_linefit.hip)
#if defined(__HIPCC__)
#define L3D_LA_HD __host__ __device__
#else
#define L3D_LA_HD
#endif

namespace l3d {
namespace la {

struct V3 { double x = 0, y = 0, z = 0; };
L3D_LA_HD inline V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
L3D_LA_HD inline V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
L3D_LA_HD inline V3 operator*(V3 a, double s) { return { a.x * s, a.y * s, a.z * s }; }
L3D_LA_HD inline V3 operator*(double s, V3 a) { return { a.x * s, a.y * s, a.z * s }; }
L3D_LA_HD inline V3 operator/(V3 a, double s) { return { a.x / s, a.y / s, a.z / s }; }
L3D_LA_HD inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
L3D_LA_HD inline double norm(V3 a) { return std::sqrt(dot(a, a)); }
L3D_LA_HD inline V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }

struct M3 {
    double m[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    L3D_LA_HD double& operator()(int r, int c) { return m[r * 3 + c]; }
    L3D_LA_HD double operator()(int r, int c) const { return m[r * 3 + c]; }
};
L3D_LA_HD inline M3 identity3() { M3 r; r(0, 0) = r(1, 1) = r(2, 2) = 1.0; return r; }
L3D_LA_HD inline M3 transpose(const M3& a) { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = a(j, i); return r; }
L3D_LA_HD inline M3 mul(const M3& a, const M3& b)
{
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j);
    return r;
}
L3D_LA_HD inline V3 mul(const M3& a, V3 v)
{
    return { a(0, 0) * v.x + a(0, 1) * v.y + a(0, 2) * v.z, a(1, 0) * v.x + a(1, 1) * v.y + a(1, 2) * v.z,
             a(2, 0) * v.x + a(2, 1) * v.y + a(2, 2) * v.z };
}
L3D_LA_HD inline double det(const M3& a)
{
    return a(0, 0) * (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) + a(0, 1) * (a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2)) +
           a(0, 2) * (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0));
}
// cofactor inverse (what a fixed-size 3x3 inverse does)
L3D_LA_HD inline M3 inverse(const M3& a)
{
    const double c00 = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1);
    const double c01 = a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2);
    const double c02 = a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0);
    const double id = 1.0 / (a(0, 0) * c00 + a(0, 1) * c01 + a(0, 2) * c02);
    M3 r;
    r(0, 0) = c00 * id;
    r(0, 1) = (a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2)) * id;
    r(0, 2) = (a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1)) * id;
    r(1, 0) = c01 * id;
    r(1, 1) = (a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0)) * id;
    r(1, 2) = (a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2)) * id;
    r(2, 0) = c02 * id;
    r(2, 1) = (a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1)) * id;
    r(2, 2) = (a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0)) * id;
    return r;
}

// Cyclic Jacobi eigen-decomposition of a symmetric 3x3: A = V diag(w) V^T (columns of V).
L3D_LA_HD inline void eig_sym3(const M3& A, double w[3], M3& V)
{
    M3 a = A;
    V = identity3();
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = a(0, 1) * a(0, 1) + a(0, 2) * a(0, 2) + a(1, 2) * a(1, 2);
        const double diag = a(0, 0) * a(0, 0) + a(1, 1) * a(1, 1) + a(2, 2) * a(2, 2);
        if (off <= 1e-300 || off <= 1e-32 * diag) break;
        // (fixed trip counts, unrolled: on the device the matrices then live in registers instead of indexed scratch memory)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                if (a(p, q) == 0.0) continue;
                const double theta = (a(q, q) - a(p, p)) / (2.0 * a(p, q));
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                for (int k = 0; k < 3; ++k) {   // A <- A J
                    const double akp = a(k, p), akq = a(k, q);
                    a(k, p) = c * akp - s * akq;
                    a(k, q) = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {   // A <- J^T A
                    const double apk = a(p, k), aqk = a(q, k);
                    a(p, k) = c * apk - s * aqk;
                    a(q, k) = s * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V(k, p), vkq = V(k, q);
                    V(k, p) = c * vkp - s * vkq;
                    V(k, q) = s * vkp + c * vkq;
                }
            }
    }
    w[0] = a(0, 0); w[1] = a(1, 1); w[2] = a(2, 2);
}

// One-sided Jacobi SVD of a general 3x3: A = U diag(s) V^T, s sorted descending, U and V orthogonal
// (columns belonging to vanishing singular values are completed to an orthonormal basis).
inline void svd3(const M3& A, M3& U, double s[3], M3& V)
{
    M3 B = A;           // columns get orthogonalised: B = A V
    V = identity3();
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int k = 0; k < 3; ++k) { al += B(k, p) * B(k, p); be += B(k, q) * B(k, q); ga += B(k, p) * B(k, q); }
                if (ga == 0.0 || std::fabs(ga) <= 1e-17 * std::sqrt(al * be)) continue;
                rotated = true;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < 3; ++k) {
                    const double bp = B(k, p), bq = B(k, q);
                    B(k, p) = c * bp - sn * bq;
                    B(k, q) = sn * bp + c * bq;
                    const double vp = V(k, p), vq = V(k, q);
                    V(k, p) = c * vp - sn * vq;
                    V(k, q) = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double n[3];
    for (int j = 0; j < 3; ++j) n[j] = std::sqrt(B(0, j) * B(0, j) + B(1, j) * B(1, j) + B(2, j) * B(2, j));
    int ord[3] = { 0, 1, 2 };
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (n[ord[j]] < n[ord[j + 1]]) { int t = ord[j]; ord[j] = ord[j + 1]; ord[j + 1] = t; }
    M3 Vs;
    V3 u[3];
    const double tol = 1e-14 * (n[ord[0]] > 0 ? n[ord[0]] : 1.0);
    int rank = 0;
    for (int j = 0; j < 3; ++j) {
        const int c = ord[j];
        s[j] = n[c];
        for (int k = 0; k < 3; ++k) Vs(k, j) = V(k, c);
        if (n[c] > tol) { u[j] = V3{ B(0, c), B(1, c), B(2, c) } / n[c]; rank = j + 1; }
    }
    if (rank == 0) { u[0] = { 1, 0, 0 }; u[1] = { 0, 1, 0 }; u[2] = { 0, 0, 1 }; }
    else if (rank == 1) {
        V3 a = std::fabs(u[0].x) < 0.9 ? V3{ 1, 0, 0 } : V3{ 0, 1, 0 };
        u[1] = cross(u[0], a); u[1] = u[1] / norm(u[1]);
        u[2] = cross(u[0], u[1]);
    } else if (rank == 2) {
        u[2] = cross(u[0], u[1]); u[2] = u[2] / norm(u[2]);
    }
    for (int j = 0; j < 3; ++j) { U(0, j) = u[j].x; U(1, j) = u[j].y; U(2, j) = u[j].z; }
    V = Vs;
}

// general 4x4 inverse, Gauss-Jordan with partial pivoting; returns false if singular
inline bool inverse4(const double* a, double* inv)
{
    double m[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { m[i][j] = a[i * 4 + j]; m[i][4 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(m[r][c]) > std::fabs(m[piv][c])) piv = r;
        if (m[piv][c] == 0.0) return false;
        if (piv != c) for (int j = 0; j < 8; ++j) { double t = m[c][j]; m[c][j] = m[piv][j]; m[piv][j] = t; }
        const double d = m[c][c];
        for (int j = 0; j < 8; ++j) m[c][j] /= d;
        for (int r = 0; r < 4; ++r) if (r != c) {
            const double f = m[r][c];
            if (f != 0.0) for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j];
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) inv[i * 4 + j] = m[i][4 + j];
    return true;
}

}  // namespace la
}  // namespace l3d
