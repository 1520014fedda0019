// scvod_math.h -- arithmetic spec of the SCV-OD hot path, shared by the HIP kernels and
// the host side of libscvod (compiled by hipcc for gfx950 and by g++ for the CPU unit
// tests of the spec itself).
//
// Everything that decides an INTEGER (bin index, patch id, ground/non-ground class) is
// written here with explicit IEEE-754 operations only: no libm / ocml transcendental is
// called, FMA contraction is disabled for the whole library (-ffp-contract=off), and
// fp32/fp64 division and sqrt are the correctly rounded HIP defaults.  The reference
// calls glibc here:
//   * atan2f  (include/utility.h:382,385,391 -- float overload)
//   * atan2   (include/patchwork.h:419,421 -- double)
// glibc 2.27-2.40 implement atan2f/atanf with the Sun fdlibm algorithm in pure fp32
// arithmetic, which is restated below operation by operation, so it is bit-identical to
// glibc on the CPU (tests/test_math_spec.py checks that against the libm of this image)
// and, because it only uses IEEE add/mul/div, bit-identical on the GPU.
// For the double atan2 the fdlibm double algorithm is used (error < 1 ulp); it feeds
// only `int(theta / sector_size)` in pc2czm, where a last-bit difference to glibc can
// matter only within 1 ulp of a sector boundary (counted by the same test: 0 flips).
#ifndef SCVOD_MATH_H_
#define SCVOD_MATH_H_

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SCVOD_HD __host__ __device__ __forceinline__
#else
#define SCVOD_HD inline
#endif

namespace scvod {

SCVOD_HD uint32_t f2u(float f) {
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(f);
#else
    memcpy(&u, &f, 4);
#endif
    return u;
}
SCVOD_HD float u2f(uint32_t u) {
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(u);
#else
    memcpy(&f, &u, 4);
#endif
    return f;
}
SCVOD_HD uint64_t d2u(double d) {
    uint64_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = (uint64_t)__double_as_longlong(d);
#else
    memcpy(&u, &d, 8);
#endif
    return u;
}
SCVOD_HD double u2d(uint64_t u) {
    double d;
#if defined(__HIP_DEVICE_COMPILE__)
    d = __longlong_as_double((long long)u);
#else
    memcpy(&d, &u, 8);
#endif
    return d;
}

SCVOD_HD float fabs_f(float x) { return u2f(f2u(x) & 0x7fffffffu); }
SCVOD_HD double fabs_d(double x) { return u2d(d2u(x) & 0x7fffffffffffffffull); }

// correctly rounded sqrt (IEEE): sqrtf / sqrt map to v_sqrt + fixup on gfx950 under the
// default -fhip-fp32-correctly-rounded-divide-sqrt; on the host they are SSE sqrtss/sd.
SCVOD_HD float sqrt_f(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_sqrtf(x);  // llvm.sqrt.f32, correctly rounded (HIP's __fsqrt_rn is the 1-ulp native sqrt)
#else
    return __builtin_sqrtf(x);
#endif
}
SCVOD_HD double sqrt_d(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_sqrt(x);
#else
    return __builtin_sqrt(x);
#endif
}
SCVOD_HD float floor_f(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return ::floorf(x);
#else
    return __builtin_floorf(x);
#endif
}
SCVOD_HD float ceil_f(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return ::ceilf(x);
#else
    return __builtin_ceilf(x);
#endif
}

// ---- fdlibm atanf / atan2f in fp32 (glibc sysdeps/ieee754/flt-32/{s_atanf,e_atan2f}.c
// algorithm: argument reduction to [0, 7/16] around 0.5, 1, 1.5, inf and an 11-term odd
// polynomial split into even/odd halves) ------------------------------------------------
SCVOD_HD float atan_f32(float x) {
    const float atanhi[4] = {u2f(0x3eed6338u), u2f(0x3f490fdau), u2f(0x3f7b985eu), u2f(0x3fc90fdau)};
    const float atanlo[4] = {u2f(0x31ac3769u), u2f(0x33222168u), u2f(0x33140fb4u), u2f(0x33a22168u)};
    const float aT0 = u2f(0x3eaaaaabu), aT1 = u2f(0xbe4ccccdu), aT2 = u2f(0x3e124925u),
                aT3 = u2f(0xbde38e38u), aT4 = u2f(0x3dba2e6eu), aT5 = u2f(0xbd9d8795u),
                aT6 = u2f(0x3d886b35u), aT7 = u2f(0xbd6ef16bu), aT8 = u2f(0x3d4bda59u),
                aT9 = u2f(0xbd15a221u), aT10 = u2f(0x3c8569d7u);
    const float one = 1.0f;
    uint32_t hx = f2u(x);
    uint32_t ix = hx & 0x7fffffffu;
    int id;
    if (ix >= 0x4c000000u) { /* |x| >= 2^25 */
        if (ix > 0x7f800000u) return x + x; /* NaN */
        if ((int32_t)hx > 0) return atanhi[3] + atanlo[3];
        return -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000u) { /* |x| < 0.4375 */
        if (ix < 0x31000000u) return x; /* |x| < 2^-29 */
        id = -1;
    } else {
        x = fabs_f(x);
        if (ix < 0x3f980000u) {     /* |x| < 1.1875 */
            if (ix < 0x3f300000u) { /* 7/16 <= |x| < 11/16 */
                id = 0;
                x = (2.0f * x - one) / (2.0f + x);
            } else { /* 11/16 <= |x| < 19/16 */
                id = 1;
                x = (x - one) / (x + one);
            }
        } else {
            if (ix < 0x401c0000u) { /* |x| < 2.4375 */
                id = 2;
                x = (x - 1.5f) / (one + 1.5f * x);
            } else { /* 2.4375 <= |x| < 2^25 */
                id = 3;
                x = -1.0f / x;
            }
        }
    }
    float z = x * x;
    float w = z * z;
    float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return ((int32_t)hx < 0) ? -z : z;
}

SCVOD_HD float atan2_f32(float y, float x) {
    const float tiny = 1.0e-30f;
    const float pi_o_4 = u2f(0x3f490fdbu), pi_o_2 = u2f(0x3fc90fdbu), pi = u2f(0x40490fdbu),
                pi_lo = u2f(0xb3bbbd2eu);
    uint32_t hx = f2u(x), hy = f2u(y);
    uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    if (ix > 0x7f800000u || iy > 0x7f800000u) return x + y; /* NaN */
    if (hx == 0x3f800000u) return atan_f32(y);              /* x = 1.0 */
    int m = (int)((hy >> 31) & 1u) | (int)((hx >> 30) & 2u); /* 2*sign(x)+sign(y) */
    if (iy == 0) {
        switch (m) {
            case 0:
            case 1: return y;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (ix == 0) return ((int32_t)hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000u) {
        if (iy == 0x7f800000u) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
                case 0: return 0.0f;
                case 1: return -0.0f;
                case 2: return pi + tiny;
                default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000u) return ((int32_t)hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    int32_t k = ((int32_t)iy - (int32_t)ix) >> 23;
    float z;
    if (k > 60)
        z = pi_o_2 + 0.5f * pi_lo;
    else if ((int32_t)hx < 0 && k < -60)
        z = 0.0f;
    else
        z = atan_f32(fabs_f(y / x));
    switch (m) {
        case 0: return z;
        case 1: return u2f(f2u(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// ---- fdlibm atan / atan2 in fp64 (Sun s_atan.c / e_atan2.c algorithm) ------------------
SCVOD_HD double atan_f64(double x) {
    const double atanhi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01,
                              9.82793723247329054082e-01, 1.57079632679489655800e+00};
    const double atanlo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17,
                              1.39033110312309984516e-17, 6.12323399573676603587e-17};
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
                 aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
                 aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
                 aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
                 aT10 = 1.62858201153657823623e-02;
    uint64_t bx = d2u(x);
    int32_t hx = (int32_t)(bx >> 32);
    int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x44100000) { /* |x| >= 2^66 */
        uint32_t lx = (uint32_t)bx;
        if (ix > 0x7ff00000 || (ix == 0x7ff00000 && lx != 0)) return x + x;
        if (hx > 0) return atanhi[3] + atanlo[3];
        return -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3fdc0000) {               /* |x| < 0.4375 */
        if (ix < 0x3e200000) return x;   /* |x| < 2^-29 */
        id = -1;
    } else {
        x = fabs_d(x);
        if (ix < 0x3ff30000) {     /* |x| < 1.1875 */
            if (ix < 0x3fe60000) { /* 7/16 <= |x| < 11/16 */
                id = 0;
                x = (2.0 * x - 1.0) / (2.0 + x);
            } else {
                id = 1;
                x = (x - 1.0) / (x + 1.0);
            }
        } else {
            if (ix < 0x40038000) { /* |x| < 2.4375 */
                id = 2;
                x = (x - 1.5) / (1.0 + 1.5 * x);
            } else {
                id = 3;
                x = -1.0 / x;
            }
        }
    }
    double z = x * x;
    double w = z * z;
    double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return (hx < 0) ? -z : z;
}

SCVOD_HD double atan2_f64(double y, double x) {
    const double tiny = 1.0e-300;
    const double pi_o_4 = 7.8539816339744827900E-01, pi_o_2 = 1.5707963267948965580E+00,
                 pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
    uint64_t bx = d2u(x), by = d2u(y);
    int32_t hx = (int32_t)(bx >> 32), hy = (int32_t)(by >> 32);
    uint32_t lx = (uint32_t)bx, ly = (uint32_t)by;
    int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (((uint32_t)ix | ((lx | (0u - lx)) >> 31)) > 0x7ff00000u ||
        ((uint32_t)iy | ((ly | (0u - ly)) >> 31)) > 0x7ff00000u)
        return x + y; /* NaN */
    if (((uint32_t)(hx - 0x3ff00000) | lx) == 0) return atan_f64(y); /* x = 1.0 */
    int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (((uint32_t)iy | ly) == 0) {
        switch (m) {
            case 0:
            case 1: return y;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (((uint32_t)ix | lx) == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7ff00000) {
        if (iy == 0x7ff00000) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0 * pi_o_4 + tiny;
                default: return -3.0 * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
                case 0: return 0.0;
                case 1: return -0.0;
                case 2: return pi + tiny;
                default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7ff00000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    int32_t k = (iy - ix) >> 20;
    double z;
    if (k > 60)
        z = pi_o_2 + 0.5 * pi_lo;
    else if (hx < 0 && k < -60)
        z = 0.0;
    else
        z = atan_f64(fabs_d(y / x));
    switch (m) {
        case 0: return z;
        case 1: return -z;
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// ---- curved-voxel binning: SSC::makeApriVec (src/ssc.cpp:155-195) with
// Utility::pointDistance2d / getPolarAngle / getAzimuth / rad2deg
// (include/utility.h:346-349, 371-392) ------------------------------------------------------
struct BinParams {
    float min_dis, max_dis, min_angle, max_angle, min_azimuth, max_azimuth;
    float range_res, sector_res, azimuth_res;
    int32_t range_num, sector_num, azimuth_num, bin_num;
};

#define SCVOD_M_PI 3.14159265358979323846

// rad2deg: `(float)radians * 180.0 / M_PI` evaluated in double, returned as float
// (utility.h:346-349)
SCVOD_HD float rad2deg_f(float radians) { return (float)((double)radians * 180.0 / SCVOD_M_PI); }

SCVOD_HD float point_distance2d(float x, float y) {
    // (float)sqrt(x*x + y*y): fp32 products and sum (no FMA), correctly rounded sqrt
    float s = x * x + y * y;
    return sqrt_f(s);
}

SCVOD_HD float polar_angle_deg(float x, float y) {
    if (x == 0.0f && y == 0.0f) return 0.0f;
    if (y >= 0.0f) return rad2deg_f(atan2_f32(y, x));
    // (float)rad2deg((float)atan2(y,x) + 2*M_PI): the sum is a double, rad2deg casts it to
    // float first (utility.h:348,385)
    double s = (double)atan2_f32(y, x) + 2.0 * SCVOD_M_PI;
    return rad2deg_f((float)s);
}

SCVOD_HD float azimuth_deg(float z, float dis) { return rad2deg_f(atan2_f32(z, dis)); }

struct Apri {
    float x, y, z, range, angle, azimuth, intensity;
    int32_t range_idx, sector_idx, azimuth_idx, voxel_idx;
};

// returns 1 when the point passes the range / FOV test of ssc.cpp:161-172
SCVOD_HD int apri_of_point(const BinParams& g, float x, float y, float z, float intensity,
                           Apri& a) {
    float dis = point_distance2d(x, y);
    float angle = polar_angle_deg(x, y);
    float azimuth = azimuth_deg(z, dis);
    int keep = 1;
    if (dis < g.min_dis || dis > g.max_dis) keep = 0;
    if (angle < g.min_angle || angle > g.max_angle) keep = 0;
    if (azimuth < g.min_azimuth || azimuth > g.max_azimuth) keep = 0;
    a.x = x;
    a.y = y;
    a.z = z;
    a.range = dis;
    a.angle = angle;
    a.azimuth = azimuth;
    a.intensity = intensity;
    // std::ceil(float) - 1 -> float -> int (ssc.cpp:185-187)
    a.range_idx = (int32_t)(ceil_f((dis - g.min_dis) / g.range_res) - 1.0f);
    a.sector_idx = (int32_t)(ceil_f((angle - g.min_angle) / g.sector_res) - 1.0f);
    a.azimuth_idx = (int32_t)(ceil_f((azimuth - g.min_azimuth) / g.azimuth_res) - 1.0f);
    a.voxel_idx = a.azimuth_idx * g.range_num * g.sector_num + a.range_idx * g.sector_num + a.sector_idx;
    return keep;
}

// ---- voxel index of a point WITHOUT the reference's arithmetic where that provably cannot matter (SSC::tracking re-bins
// every transformed point, ssc.cpp:1280-1286; only the index is used).  The two angles are estimated with a polynomial
// arctangent (Abramowitz & Stegun 4.4.49, |error| <= 2e-8 rad) in fp32: the estimate and the reference's
// float(double(atan2f(..)) * 180 / pi) both lie within 2.5e-4 degrees of each other (measured over 10^9 points,
// tests/test_math_spec.py: < 6e-5), so when the estimate is farther than 1.5e-3 degrees (plus the rounding of the scaled
// value) from every bin edge the index is the reference's; otherwise -- a few points per thousand -- the caller evaluates
// apri_of_point.  y == +-0 (angle exactly 0 / 180 / -180: sector -1 and the signed-zero cases of atan2f) and the origin
// always take the reference arithmetic.  The range index is the reference's own computation (one sqrt, one division).
struct BinFast {
    float inv_sector_res, inv_azimuth_res;
    float m_sector, m_azimuth;  // distance to a bin edge, in bins, below which the estimate decides nothing
};
inline BinFast bin_fast_of(const BinParams& g) {
    BinFast f;
    f.inv_sector_res = 1.0f / g.sector_res;
    f.inv_azimuth_res = 1.0f / g.azimuth_res;
    f.m_sector = 1.5e-3f * f.inv_sector_res + 2.0e-4f;
    f.m_azimuth = 1.5e-3f * f.inv_azimuth_res + 2.0e-4f;
    return f;
}
SCVOD_HD float atan01_poly(float t) {  // t in [0, 1]
    const float s = t * t;
    float q = 0.0028662257f;
    q = q * s + -0.0161657367f;
    q = q * s + 0.0429096138f;
    q = q * s + -0.0752896400f;
    q = q * s + 0.1065626393f;
    q = q * s + -0.1420889944f;
    q = q * s + 0.1999355085f;
    q = q * s + -0.3333314528f;
    q = q * s + 1.0f;
    return t * q;
}
SCVOD_HD float rcp_fast(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
// atan2(|y|, x) in [0, pi] for |y| > 0, within 6.1e-7 rad of the exact value: atan01_poly is within 1.05e-7 rad of atan over ALL
// 2^30 floats of [0, 1] (checked exhaustively on the CPU, the same unfused arithmetic), the quotient within 1.8e-7 of mn / mx
// (v_rcp_f32: 1 ulp, + the product's rounding), the two folds add 1.1e-7 and 2.1e-7 (constants + roundings).  NaN in, NaN out.
SCVOD_HD float atan2_abs_rad_fast(float ay, float x) {
    const float ax = fabs_f(x);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    float a = atan01_poly(mn * rcp_fast(mx));
    if (ay > ax) a = 1.57079632679f - a;
    if (x < 0.0f) a = 3.14159265359f - a;
    return a;
}
// degrees in [0, 180] of atan2(|y|, x) for |y| > 0
SCVOD_HD float atan2_abs_deg_fast(float ay, float x) { return atan2_abs_rad_fast(ay, x) * 57.2957795131f; }
// true: (*ri, *si, *ai) are the reference's range / sector / azimuth indices of the point; false: undecided (use apri_of_point)
SCVOD_HD bool idx3_fast(const BinParams& g, const BinFast& f, float x, float y, float z, int32_t* ri_out, int32_t* si_out, int32_t* ai_out) {
    const float ay = fabs_f(y);
    if (!(ay > 0.0f)) return false;  // y == +-0 (and NaN)
    const float dis = point_distance2d(x, y);
    float ang = atan2_abs_deg_fast(ay, x);
    if (y < 0.0f) ang = 360.0f - ang;
    const float us = (ang - g.min_angle) * f.inv_sector_res;
    const float fs = floor_f(us);
    const float ds = us - fs;
    if (!(ds > f.m_sector && ds < 1.0f - f.m_sector)) return false;
    const float az = (z == 0.0f) ? 0.0f : atan2_abs_deg_fast(fabs_f(z), dis);
    const float azs = z < 0.0f ? -az : az;
    const float ua = (azs - g.min_azimuth) * f.inv_azimuth_res;
    const float fa = floor_f(ua);
    const float da = ua - fa;
    if (!(da > f.m_azimuth && da < 1.0f - f.m_azimuth)) return false;
    if (!(fabs_f(us) < 1.0e6f && fabs_f(ua) < 1.0e6f)) return false;
    *ri_out = (int32_t)(ceil_f((dis - g.min_dis) / g.range_res) - 1.0f);
    *si_out = (int32_t)fs;
    *ai_out = (int32_t)fa;
    return true;
}
// true: *voxel_idx is SSC::tracking's index of the point; false: undecided (use apri_of_point)
SCVOD_HD bool voxel_idx_fast(const BinParams& g, const BinFast& f, float x, float y, float z, int32_t* voxel_idx) {
    int32_t ri, si, ai;
    if (!idx3_fast(g, f, x, y, z, &ri, &si, &ai)) return false;
    *voxel_idx = ai * g.range_num * g.sector_num + ri * g.sector_num + si;
    return true;
}

// ---- range / FOV verdict of makeApriVec WITHOUT the two atan2f of apri_of_point, for callers that only need the
// verdict (k_pw_arrange).  Same result by construction:
//   * range: the same fp32 sqrt and comparisons.
//   * angle: with min_angle <= 0 and max_angle >= 360 the test cannot fail -- polar_angle_deg returns at most
//     fl32(fl32(2 pi) * 180 / pi) = 360.0f (360.00001 rounds down, the float spacing there is 3e-5) and never a
//     negative value; NaN compares false in both forms.
//   * azimuth: atan2f(z, dis) and the conversion to degrees are within 1.2e-5 deg of atan(z / dis) * 180 / pi, so when
//     t = z / dis is farther than 2e-6 * (1 + t^2) (six times that error, in units of t) from tan(min_azimuth) and
//     tan(max_azimuth) the decision is known; only points inside that sliver evaluate apri_of_point.
struct KeepFast {
    int32_t ok;      // the shortcut is valid for this parameter set
    float tan_lo, tan_hi;
};
inline KeepFast keep_fast_of(const BinParams& g) {
    KeepFast k;
    k.ok = (g.min_angle <= 0.0f && g.max_angle >= 360.0f && g.min_azimuth > -89.0f && g.max_azimuth < 89.0f &&
            g.min_azimuth < g.max_azimuth)
               ? 1
               : 0;
    k.tan_lo = (float)__builtin_tan((double)g.min_azimuth * SCVOD_M_PI / 180.0);
    k.tan_hi = (float)__builtin_tan((double)g.max_azimuth * SCVOD_M_PI / 180.0);
    return k;
}
SCVOD_HD int keep_of_point(const BinParams& g, const KeepFast& k, float x, float y, float z) {
    if (k.ok) {
        const float dis = point_distance2d(x, y);
        if (dis < g.min_dis || dis > g.max_dis) return 0;
        const float t = z / dis;  // dis == 0: inf / NaN, which fails every comparison below -> reference arithmetic
        const float m = 2.0e-6f * (1.0f + t * t);
        if (t > k.tan_lo + m && t < k.tan_hi - m) return 1;
        if (t < k.tan_lo - m || t > k.tan_hi + m) return 0;
    }
    Apri a;
    return apri_of_point(g, x, y, z, 0.f, a);
}

// ---- Patchwork concentric zone model: pc2czm / xy2theta / xy2radius
// (include/patchwork.h:416-459) ------------------------------------------------------------
struct CzmParams {
    double min_range, max_range;
    double zone_min[4];     // min_ranges (patchwork.h:87)
    double ring_size[4];    // patchwork.h:88-91
    double sector_size[4];  // patchwork.h:92-94
    int32_t num_rings[4], num_sectors[4];
    int32_t patch_base[4];  // first patch id of each zone in (zone, ring, sector) order
    int32_t num_patches;
    double z_cut;           // -1.8 * sensor_height (patchwork.h:304)
    double seed_margin_z;   // adaptive_seed_selection_margin * sensor_height (patchwork.h:247)
    double th_seeds, th_dist, uprightness_thr;
    double elevation_thr[4], flatness_thr[4];
    int32_t num_iter, num_lpr, num_min_pts, num_rings_of_interest;
    // derived constants of the cheap (fp32 / reciprocal) evaluation, czm_finalize(); they only decide WHEN the cheap
    // value is trusted, never what the result is
    int32_t fast_ok;           // ranges small enough for the fp32 error bound below
    float r_margin;            // |r_fp32 - r_fp64| stays far below this (1.2e-7 * r for r <= 1000 m)
    float min_range_f, max_range_f, zone_min_f[4], inv_ring_f[4], ring_margin_f[4];
    double inv_sector[4], sector_margin[4];
    double sector_margin_est[4];  // the same for the branch-free estimate of the angle (atan2_abs_rad_fast)
};

inline void czm_finalize(CzmParams& c) {
    c.fast_ok = (c.max_range <= 1000.0 && c.min_range >= 0.0) ? 1 : 0;
    c.r_margin = 1.0e-3f;
    c.min_range_f = (float)c.min_range;
    c.max_range_f = (float)c.max_range;
    for (int k = 0; k < 4; ++k) {
        c.zone_min_f[k] = (float)c.zone_min[k];
        c.inv_ring_f[k] = (float)(1.0 / c.ring_size[k]);
        c.ring_margin_f[k] = (float)(2.0e-3 / c.ring_size[k] + 1.0e-4);
        if (!(c.ring_size[k] > 1.0e-2) || c.num_rings[k] > 256) c.fast_ok = 0;  // keeps (r - zone_min) / ring_size <= 256
        c.inv_sector[k] = 1.0 / c.sector_size[k];
        c.sector_margin[k] = 1.0e-6 / c.sector_size[k] + 1.0e-9;
        c.sector_margin_est[k] = 4.0e-6 / c.sector_size[k] + 1.0e-9;  // 6.5 x the estimate's error bound (6.1e-7 rad)
    }
}

// patch id in (zone, ring, sector) emission order, or -1 when the point is not binned
SCVOD_HD int32_t czm_patch_of(const CzmParams& c, float xf, float yf, float zf) {
    if ((double)zf < c.z_cut) return -1;  // erased prefix of the z-sorted cloud (patchwork.h:302-310)
    double x = (double)xf, y = (double)yf;
    // Zone and ring from the fp32 radius when it is farther than r_margin from every boundary that matters (its error
    // against the reference's double sqrt is below 1.3e-7 * r): same decisions, no fp64 sqrt / division.  Points next
    // to a boundary, NaNs and exotic parameter sets take the reference arithmetic below.
    int k = -1;
    int32_t ring = 0;
    if (c.fast_ok) {
        const float rf = sqrt_f(xf * xf + yf * yf);
        const float m = c.r_margin;
        if (fabs_f(rf - c.max_range_f) > m && fabs_f(rf - c.min_range_f) > m && fabs_f(rf - c.zone_min_f[1]) > m &&
            fabs_f(rf - c.zone_min_f[2]) > m && fabs_f(rf - c.zone_min_f[3]) > m) {
            if (!((rf <= c.max_range_f) && (rf > c.min_range_f))) return -1;
            const int kf = (rf < c.zone_min_f[1]) ? 0 : (rf < c.zone_min_f[2]) ? 1 : (rf < c.zone_min_f[3]) ? 2 : 3;
            const float v = (rf - c.zone_min_f[kf]) * c.inv_ring_f[kf];
            const int32_t rv = (int32_t)v;
            const float fr = v - (float)rv;
            if (fr > c.ring_margin_f[kf] && fr < 1.0f - c.ring_margin_f[kf]) {
                k = kf;
                ring = rv;
            }
        }
    }
    if (k < 0) {
        double r = sqrt_d(x * x + y * y);  // pow(x,2) == x*x exactly for float-valued doubles
        if (!((r <= c.max_range) && (r > c.min_range))) return -1;
        if (r < c.zone_min[1])
            k = 0;
        else if (r < c.zone_min[2])
            k = 1;
        else if (r < c.zone_min[3])
            k = 2;
        else
            k = 3;
        ring = (int32_t)((r - c.zone_min[k]) / c.ring_size[k]);
    }
    if (ring > c.num_rings[k] - 1) ring = c.num_rings[k] - 1;
    // sector = int(theta / sector_size) with theta = atan2(y, x) in double (+2 pi for y < 0).  The fp32 fdlibm
    // atan2 of the same (float-valued) arguments is within 3e-7 rad of it, so whenever theta_f / sector_size is
    // farther than that from an integer the cheap value decides the SAME index; only the rare points next to a
    // sector boundary (and the exact axis directions) take the fp64 evaluation.
    int32_t sector = 0;
    bool have_sector = false;
    // first a branch-free estimate (one reciprocal, a degree-17 polynomial: k_pw_classify is bound by VALU issue, and the fdlibm
    // evaluation below is five argument ranges with a division each, executed one after the other by a wave whose lanes differ):
    // it decides the sector whenever theta / sector_size is farther than 6.5 x its error bound from an integer
    if (c.fast_ok && fabs_f(yf) > 0.0f) {  // (y == +-0: the signed-zero cases of atan2 take the reference arithmetic)
        const float a = atan2_abs_rad_fast(fabs_f(yf), xf);
        const double te = (yf < 0.0f) ? 2.0 * SCVOD_M_PI - (double)a : (double)a;
        const double u = te * c.inv_sector[k];
        const int32_t su = (int32_t)u;
        const double fr = u - (double)su;
        const double margin = c.sector_margin_est[k];
        if (fr > margin && fr < 1.0 - margin) {  // (false for NaN)
            sector = su;
            have_sector = true;
        }
    }
    if (!have_sector) {
        double tf = (double)atan2_f32(yf, xf);
        if (y < 0.0) tf += 2.0 * SCVOD_M_PI;
        const double u = tf * c.inv_sector[k];  // within 1e-15 of tf / sector_size: far inside the margin
        const int32_t su = (int32_t)u;
        const double fr = u - (double)su;
        const double margin = c.sector_margin[k];
        if (fr > margin && fr < 1.0 - margin) {
            sector = su;
        } else {
            const double theta = (y >= 0.0) ? atan2_f64(y, x) : 2.0 * SCVOD_M_PI + atan2_f64(y, x);
            sector = (int32_t)(theta / c.sector_size[k]);
        }
    }
    if (sector > c.num_sectors[k] - 1) sector = c.num_sectors[k] - 1;
    if (sector < 0) sector = 0;  // y == -0.0f with x < 0 gives theta = -pi: out-of-bounds indexing in the reference
    return c.patch_base[k] + ring * c.num_sectors[k] + sector;
}

// monotone map float -> uint32 (ascending float order == ascending unsigned order;
// -0.0 sorts before +0.0, which std::sort's `a.z < b.z` treats as equal: ties)
SCVOD_HD uint32_t float_sort_key(float f) {
    uint32_t u = f2u(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---- 3x3 SVD: Eigen 3.3 JacobiSVD<MatrixXf>(cov, ComputeFullU) restated (two-sided Jacobi,
// sweep order (p,q) = (1,0),(2,0),(2,1); Eigen/src/SVD/JacobiSVD.h, Jacobi/Jacobi.h).
// Outputs: singular values sorted descending, U columns. -----------------------------------
struct Svd3 {
    float sv[3];
    float U[9];  // row-major U(r,c) = U[3*r+c]
};

SCVOD_HD void jacobi_make(float x, float y, float z, float& c, float& s) {
    // JacobiRotation::makeJacobi(x, y, z)
    float deno = 2.0f * fabs_f(y);
    if (deno < 1.17549435e-38f) {
        c = 1.0f;
        s = 0.0f;
    } else {
        float tau = (x - z) / deno;
        float w = sqrt_f(tau * tau + 1.0f);
        float t;
        if (tau > 0.0f)
            t = 1.0f / (tau + w);
        else
            t = 1.0f / (tau - w);
        float sign_t = t > 0.0f ? 1.0f : -1.0f;
        float n = 1.0f / sqrt_f(t * t + 1.0f);
        s = -sign_t * (y / fabs_f(y)) * fabs_f(t) * n;
        c = n;
    }
}

SCVOD_HD void svd3_jacobi(const float cov[9], Svd3& out) {
    const float precision = 2.0f * 1.1920929e-07f;  // 2 * epsilon
    const float considerAsZero = 1.17549435e-38f;    // numeric_limits<float>::min()
    float W[9], U[9];
    float scale = 0.0f;
    for (int i = 0; i < 9; ++i) {
        float a = fabs_f(cov[i]);
        if (a > scale) scale = a;
    }
    if (scale == 0.0f) scale = 1.0f;
    for (int i = 0; i < 9; ++i) W[i] = cov[i] / scale;
    for (int i = 0; i < 9; ++i) U[i] = 0.0f;
    U[0] = U[4] = U[8] = 1.0f;
    float maxDiag = fabs_f(W[0]);
    if (fabs_f(W[4]) > maxDiag) maxDiag = fabs_f(W[4]);
    if (fabs_f(W[8]) > maxDiag) maxDiag = fabs_f(W[8]);
    bool finished = false;
    int guard = 0;
    while (!finished && guard < 64) {
        ++guard;
        finished = true;
        for (int p = 1; p < 3; ++p) {
            for (int q = 0; q < p; ++q) {
                float thr = precision * maxDiag;
                if (considerAsZero > thr) thr = considerAsZero;
                float wpq = W[3 * p + q], wqp = W[3 * q + p];
                if (fabs_f(wpq) > thr || fabs_f(wqp) > thr) {
                    finished = false;
                    // real_2x2_jacobi_svd
                    float m00 = W[3 * p + p], m01 = wpq, m10 = wqp, m11 = W[3 * q + q];
                    float t = m00 + m11;
                    float d = m10 - m01;
                    float r1c, r1s;
                    if (fabs_f(d) < considerAsZero) {
                        r1s = 0.0f;
                        r1c = 1.0f;
                    } else {
                        float u = t / d;
                        float tmp = sqrt_f(1.0f + u * u);
                        r1s = 1.0f / tmp;
                        r1c = u / tmp;
                    }
                    // m.applyOnTheLeft(0,1,rot1): row0' = c*row0 + s*row1 ; row1' = -s*row0 + c*row1
                    if (!(r1c == 1.0f && r1s == 0.0f)) {
                        float a0 = m00, a1 = m01, b0 = m10, b1 = m11;
                        m00 = r1c * a0 + r1s * b0;
                        m01 = r1c * a1 + r1s * b1;
                        m10 = -r1s * a0 + r1c * b0;
                        m11 = -r1s * a1 + r1c * b1;
                    }
                    float jrc, jrs;
                    jacobi_make(m00, m01, m11, jrc, jrs);
                    // j_left = rot1 * j_right.transpose();  transpose = (c, -s)
                    float tc = jrc, ts = -jrs;
                    float jlc = r1c * tc - r1s * ts;
                    float jls = r1c * ts + r1s * tc;
                    // m_workMatrix.applyOnTheLeft(p,q,j_left): rows p,q
                    if (!(jlc == 1.0f && jls == 0.0f)) {
                        for (int i = 0; i < 3; ++i) {
                            float xi = W[3 * p + i], yi = W[3 * q + i];
                            W[3 * p + i] = jlc * xi + jls * yi;
                            W[3 * q + i] = -jls * xi + jlc * yi;
                        }
                    }
                    // m_matrixU.applyOnTheRight(p,q,j_left.transpose()): columns p,q rotated
                    // by (j_left.transpose()).transpose() == j_left
                    if (!(jlc == 1.0f && jls == 0.0f)) {
                        for (int i = 0; i < 3; ++i) {
                            float xi = U[3 * i + p], yi = U[3 * i + q];
                            U[3 * i + p] = jlc * xi + jls * yi;
                            U[3 * i + q] = -jls * xi + jlc * yi;
                        }
                    }
                    // m_workMatrix.applyOnTheRight(p,q,j_right): columns p,q by j_right.transpose()
                    if (!(tc == 1.0f && ts == 0.0f)) {
                        for (int i = 0; i < 3; ++i) {
                            float xi = W[3 * i + p], yi = W[3 * i + q];
                            W[3 * i + p] = tc * xi + ts * yi;
                            W[3 * i + q] = -ts * xi + tc * yi;
                        }
                    }
                    float ap = fabs_f(W[3 * p + p]), aq = fabs_f(W[3 * q + q]);
                    float mx = ap > aq ? ap : aq;
                    if (mx > maxDiag) maxDiag = mx;
                }
            }
        }
    }
    float sv[3];
    for (int i = 0; i < 3; ++i) {
        float a = W[4 * i];
        sv[i] = fabs_f(a);
        if (a < 0.0f) {
            U[i] = -U[i];
            U[3 + i] = -U[3 + i];
            U[6 + i] = -U[6 + i];
        }
    }
    for (int i = 0; i < 3; ++i) sv[i] *= scale;
    // sort descending (first maximum wins), stop at a zero remainder
    for (int i = 0; i < 3; ++i) {
        int pos = i;
        float mx = sv[i];
        for (int j = i + 1; j < 3; ++j)
            if (sv[j] > mx) {
                mx = sv[j];
                pos = j;
            }
        if (mx == 0.0f) break;
        if (pos != i) {
            float t = sv[i];
            sv[i] = sv[pos];
            sv[pos] = t;
            for (int r = 0; r < 3; ++r) {
                float tu = U[3 * r + i];
                U[3 * r + i] = U[3 * r + pos];
                U[3 * r + pos] = tu;
            }
        }
    }
    for (int i = 0; i < 3; ++i) out.sv[i] = sv[i];
    for (int i = 0; i < 9; ++i) out.U[i] = U[i];
}

}  // namespace scvod
#endif  // SCVOD_MATH_H_
