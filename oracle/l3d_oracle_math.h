/*
 * oracle/l3d_oracle_math.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * The "numeric contract" restated for the CPU oracle: every float expression of
 * the reference's device code (cudawrapper.cu:44-829) is evaluated in IEEE
 * binary32 with NO fused multiply-add, correctly rounded + - * / sqrt, and the
 * two transcendentals the path uses (expf, acosf) replaced by the fixed
 * operation sequences below.  The reference's CUDA build uses nvcc's expf/acosf
 * (2 ulp / 2 ulp, contraction compiler-chosen), so it has no canonical bit
 * pattern of its own (SURVEY.md section 7 hard part 2); this contract is the one
 * both the oracle and the HIP kernels follow so that threshold decisions
 * (conf > 0.5, conf > 1.0, overlap > 0.1 ...) are bit-reproducible.
 *
 * This header is written independently of line3d_amd/csrc/l3d_contract.hpp
 * (the product-side copy); tests/test_contract_math.py checks (a) both against
 * libm within 2 ulp and (b) GPU-vs-oracle bit equality.
 *
 * Build with -DL3DO_LIBM to use glibc expf/acosf/acos instead (what a plain
 * host compile of the reference text would call); tests compare both modes.
 */
#ifndef L3D_ORACLE_MATH_H
#define L3D_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float l3do_bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* exp(x), x <= 0 is the only range the path produces (-d*d/(2 s^2)). Cephes-style
 * range reduction x = k ln2 + r, |r| <= ln2/2, degree-5 polynomial in r for
 * (exp(r)-1-r)/r^2, separate mul/add (no fma). */
static inline float l3do_expf(float x)
{
#ifdef L3DO_LIBM
    return expf(x);
#else
    if (!(x > -87.0f)) return 0.0f;      /* also catches NaN -> 0 (never produced) */
    if (x > 88.0f) return INFINITY;
    float kf = rintf(x * 1.44269504088896341f);
    float r = x - kf * 0.693359375f;     /* ln2 hi: 0x3f318000, 9 significant bits */
    r = r - kf * -2.12194440e-4f;        /* ln2 lo */
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    float r2 = r * r;
    float y = p * r2 + r;
    y = y + 1.0f;
    int k = (int)kf;                     /* -126 <= k <= 127 here */
    return y * l3do_bits2f((uint32_t)(k + 127) << 23);
#endif
}

/* asin on |x| <= 0.5 (Cephes asinf kernel) */
static inline float l3do_asinf_kernel(float x)
{
    float z = x * x;
    float p = 4.2163199048e-2f;
    p = p * z + 2.4181311049e-2f;
    p = p * z + 4.5470025998e-2f;
    p = p * z + 7.4953002686e-2f;
    p = p * z + 1.6666752422e-1f;
    p = p * z;
    p = p * x;
    return p + x;
}

/* acos(x), x in [-1,1] (callers clamp first) */
static inline float l3do_acosf(float x)
{
#ifdef L3DO_LIBM
    return acosf(x);
#else
    if (x < -0.5f) {
        float s = sqrtf(0.5f * (1.0f + x));
        return 3.14159265358979323846f - 2.0f * l3do_asinf_kernel(s);
    }
    if (x > 0.5f) {
        float s = sqrtf(0.5f * (1.0f - x));
        return 2.0f * l3do_asinf_kernel(s);
    }
    return 1.5707963267948966f - l3do_asinf_kernel(x);
#endif
}

/* double acos for similarity_coll3D (line3D.cc:1668). fdlibm-structured rational
 * approximation without the hi/lo split (result is rounded to float by the caller,
 * ~1e-15 relative is enough). */
static inline double l3do_acos_R(double z)
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

static inline double l3do_acos(double x)
{
#ifdef L3DO_LIBM
    return acos(x);
#else
    const double pi = 3.14159265358979311600e+00;
    const double pio2 = 1.57079632679489655800e+00;
    if (x >= 1.0) return 0.0;
    if (x <= -1.0) return pi;
    if (x < -0.5) {
        double z = (1.0 + x) * 0.5;
        double s = sqrt(z);
        double w = l3do_acos_R(z) * s;
        return pi - 2.0 * (s + w);
    }
    if (x > 0.5) {
        double z = (1.0 - x) * 0.5;
        double s = sqrt(z);
        double w = l3do_acos_R(z) * s;
        return 2.0 * (s + w);
    }
    double z = x * x;
    return pio2 - (x + x * l3do_acos_R(z));
#endif
}

#endif
