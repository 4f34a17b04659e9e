// l3d_contract.hpp -- the numeric contract of the HIP path (product side).
//
// The reference's device functions (cudawrapper.cu:44-427) are float expressions whose
// results feed hard thresholds (overlap > 0.1/0.3, conf > 0.5, conf > 1.0, depth > 0).
// To make those decisions reproducible between this GPU path and the CPU oracle, every
// expression is evaluated exactly as written in IEEE binary32: no fused multiply-add
// (the translation unit is built with -ffp-contract=off), correctly rounded division and
// square root (-fhip-fp32-correctly-rounded-divide-sqrt), normalize() = v * (1.0f/sqrtf(.))
// as in the host definition of rsqrtf (helper_math.h:61-64), and the two transcendentals
// (expf, acosf; double acos for similarity_coll3D) are the fixed operation sequences below
// instead of a vendor libm.  DESIGN.md section "Numeric contract" states the sequences;
// the oracle keeps its own independent copy (oracle/l3d_oracle_math.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define L3D_HD __host__ __device__ __forceinline__

namespace l3d {

L3D_HD float pow2i(int k)  // 2^k for -126 <= k <= 127
{
    union { uint32_t u; float f; } v;
    v.u = (uint32_t)(k + 127) << 23;
    return v.f;
}

// exp(x): x = k*ln2 + r, |r| <= ln2/2; exp(r) = 1 + r + r^2 * poly5(r); result * 2^k.
L3D_HD float c_expf(float x)
{
    if (!(x > -87.0f)) return 0.0f;
    if (x > 88.0f) return __builtin_inff();
    const float kf = __builtin_rintf(x * 1.44269504088896341f);
    float r = x - kf * 0.693359375f;
    r = r - kf * -2.12194440e-4f;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    const float r2 = r * r;
    float y = p * r2 + r;
    y = y + 1.0f;
    return y * pow2i((int)kf);
}

L3D_HD float c_asin_small(float x)  // |x| <= 0.5
{
    const float z = x * x;
    float p = 4.2163199048e-2f;
    p = p * z + 2.4181311049e-2f;
    p = p * z + 4.5470025998e-2f;
    p = p * z + 7.4953002686e-2f;
    p = p * z + 1.6666752422e-1f;
    p = p * z;
    p = p * x;
    return p + x;
}

// acos(x) on [-1,1]
L3D_HD float c_acosf(float x)
{
    if (x < -0.5f) {
        const float s = __builtin_sqrtf(0.5f * (1.0f + x));
        return 3.14159265358979323846f - 2.0f * c_asin_small(s);
    }
    if (x > 0.5f) {
        const float s = __builtin_sqrtf(0.5f * (1.0f - x));
        return 2.0f * c_asin_small(s);
    }
    return 1.5707963267948966f - c_asin_small(x);
}

L3D_HD double c_acos_ratio(double z)
{
    double p = 3.47933107596021167570e-05;
    p = p * z + 7.91534994289814532176e-04;
    p = p * z + -4.00555345006794114027e-02;
    p = p * z + 2.01212532134862925881e-01;
    p = p * z + -3.25565818622400915405e-01;
    p = p * z + 1.66666666666666657415e-01;
    p = p * z;
    double q = 7.70381505559019352791e-02;
    q = q * z + -6.88283971605453293030e-01;
    q = q * z + 2.02094576023350569471e+00;
    q = q * z + -2.40339491173441421878e+00;
    q = q * z + 1.0;
    return p / q;
}

// double acos on [-1,1] (similarity_coll3D, line3D.cc:1668)
L3D_HD double c_acos(double x)
{
    const double pi = 3.14159265358979311600e+00;
    const double pio2 = 1.57079632679489655800e+00;
    if (x >= 1.0) return 0.0;
    if (x <= -1.0) return pi;
    if (x < -0.5) {
        const double z = (1.0 + x) * 0.5;
        const double s = __builtin_sqrt(z);
        const double w = c_acos_ratio(z) * s;
        return pi - 2.0 * (s + w);
    }
    if (x > 0.5) {
        const double z = (1.0 - x) * 0.5;
        const double s = __builtin_sqrt(z);
        const double w = c_acos_ratio(z) * s;
        return 2.0 * (s + w);
    }
    const double z = x * x;
    return pio2 - (x + x * c_acos_ratio(z));
}

// ---- float3 algebra with the operation order of helper_math.h -------------------------
struct f3 { float x, y, z; };

L3D_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
L3D_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
L3D_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
L3D_HD f3 operator*(float b, f3 a) { return mk3(b * a.x, b * a.y, b * a.z); }
L3D_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
L3D_HD float length(f3 v) { return __builtin_sqrtf(dot(v, v)); }
L3D_HD f3 normalize(f3 v)
{
    const float inv = 1.0f / __builtin_sqrtf(dot(v, v));
    return mk3(v.x * inv, v.y * inv, v.z * inv);
}
L3D_HD f3 cross(f3 a, f3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// M (3x3 row-major) * p with the accumulation order of D_get_ray_src (cudawrapper.cu:270-285)
L3D_HD f3 mat3_apply(const float* M, f3 p)
{
    f3 r;
    r.x = ((0.0f + M[0] * p.x) + M[1] * p.y) + M[2] * p.z;
    r.y = ((0.0f + M[3] * p.x) + M[4] * p.y) + M[5] * p.z;
    r.z = ((0.0f + M[6] * p.x) + M[7] * p.y) + M[8] * p.z;
    return r;
}
L3D_HD f3 mat3T_apply(const float* M, f3 p)  // transpose (D_epipolar_line with transpose=true)
{
    f3 r;
    r.x = ((0.0f + M[0] * p.x) + M[3] * p.y) + M[6] * p.z;
    r.y = ((0.0f + M[1] * p.x) + M[4] * p.y) + M[7] * p.z;
    r.z = ((0.0f + M[2] * p.x) + M[5] * p.y) + M[8] * p.z;
    return r;
}

}  // namespace l3d
