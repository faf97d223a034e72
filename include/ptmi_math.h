/*
 * ptmi_math.h — the canonical fp32 arithmetic of the path-tracing hot path.
 *
 * Why this exists (SURVEY.md §7 "Hard parts", Appendix C): the reference shader
 * (Assets/Resources/util/ *.hlsl) uses intrinsics whose rounding is backend-defined
 * (rcp, rsqrt, normalize, sin, cos, log, acos, pow, mad contraction).  A path tracer
 * takes a discrete branch on almost every float it computes (hit/miss, lobe pick,
 * Russian roulette), so an accumulation-buffer L-inf bound of 1e-4 between two
 * implementations is only attainable when both evaluate the SAME fp32 expression
 * tree.  This header fixes that tree: IEEE-754 binary32, round-to-nearest-even,
 * correctly-rounded + - * / sqrt, no FMA contraction except where written as
 * pt_fma, and elementary functions defined by the explicit polynomial kernels below
 * (classic Cephes single-precision forms, restated; libm / ocml are never called).
 *
 * Both the MI355X kernels (hipcc, device side) and the CPU oracle (g++) include this
 * file and are compiled with -ffp-contract=off, so a path computes bit-identical
 * values on both.  tests/test_math.py pins every function against libm (<= 4 ulp).
 *
 * Nothing in here calls into oracle/; the oracle includes this header, not vice versa.
 */
#ifndef PTMI_MATH_H
#define PTMI_MATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define PT_HD __host__ __device__ __forceinline__
#else
#define PT_HD static inline
#endif

#define PT_PI          3.14159265358979323f   /* util/common.hlsl:5-9, rounded to fp32 */
#define PT_INV_PI      0.31830988618379067f
#define PT_TWO_PI      6.28318530717958648f
#define PT_INV_TWO_PI  0.15915494309189533f
#define PT_INV_4_PI    0.07957747154594766f

/* ---- bit casts ----------------------------------------------------------------- */
PT_HD uint32_t pt_asuint(float f) { return __builtin_bit_cast(uint32_t, f); }
PT_HD float    pt_asfloat(uint32_t u) { return __builtin_bit_cast(float, u); }

/* ---- correctly rounded primitives ------------------------------------------------ */
PT_HD float pt_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PT_HD float pt_sqrt(float x) { return __builtin_sqrtf(x); }
PT_HD float pt_abs(float x) { return __builtin_fabsf(x); }
PT_HD float pt_floor(float x) { return __builtin_floorf(x); }
PT_HD float pt_ceil(float x) { return __builtin_ceilf(x); }
PT_HD float pt_rint(float x) { return __builtin_rintf(x); }          /* ties-to-even */
/* HLSL min/max: if one operand is NaN the other is returned (IEEE minNum/maxNum). */
PT_HD float pt_min(float a, float b) { return __builtin_fminf(a, b); }
PT_HD float pt_max(float a, float b) { return __builtin_fmaxf(a, b); }
PT_HD float pt_clamp(float x, float lo, float hi) { return pt_min(pt_max(x, lo), hi); }
PT_HD float pt_saturate(float x) { return pt_min(pt_max(x, 0.0f), 1.0f); }
PT_HD float pt_rcp(float x) { return 1.0f / x; }                      /* HLSL rcp   */
PT_HD float pt_rsqrt(float x) { return 1.0f / pt_sqrt(x); }           /* HLSL rsqrt */
PT_HD float pt_lerp(float a, float b, float t) { return a + t * (b - a); }
PT_HD int   pt_isnan(float x) { return x != x; }

/* util/texture.hlsl:41-48 wraps a texture coordinate with `while (u > 1) u -= 1; while (u < 0) u += 1;`.
 * Below 2^24 every intermediate u -+ 1 of those loops is exact (the result is a multiple of ulp(u) of smaller
 * magnitude) except possibly the last step of the upward loop, so ONE subtraction / addition of the step count gives
 * the loops' result bit for bit: k = ceil(u) - 1 steps down for u > 1, k = ceil(-u) steps up for u < 0.
 * From 2^24 on (and for +-inf) u -+ 1 == u and the reference's loops never terminate (a GPU hang / TDR): those
 * inputs are DEFINED here as 0.  NaN fails both loop conditions and passes through.
 * tests/test_math.py checks this against the literal loops. */
PT_HD float pt_wrap01(float u)
{
    if (pt_abs(u) >= 16777216.0f) return 0.0f;
    if (u > 1.0f) return u - (pt_ceil(u) - 1.0f);
    if (u < 0.0f) return u + pt_ceil(-u);
    return u;
}

/* An 8-bit texel channel as a float: util/texture.hlsl divides by 255.  (float)b / 255.0f for the 256 possible inputs equals
 * one multiplication by fl(1/255) followed by one fma residual and one fma correction -- checked exhaustively by
 * tests/test_math.py against the division -- at 3 instructions instead of the 11 of a correctly rounded fp32 division
 * (16 of them per bilinear texture sample). */
PT_HD float pt_unorm8(uint32_t b)
{
    const float x = (float)(b & 0xFFu);
    const float r = 1.0f / 255.0f;               /* a constant: folded at compile time */
    const float q = x * r;
    return pt_fma(pt_fma(-255.0f, q, x), r, q);
}

/* float -> integer conversions with D3D/WebGPU semantics (truncate, saturate, NaN -> 0);
 * C leaves the out-of-range cases undefined, gfx950 v_cvt saturates, x86 does not. */
PT_HD uint32_t pt_f2u(float x)
{
    if (!(x >= 0.0f)) return 0u;                 /* negative and NaN */
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
PT_HD int32_t pt_f2i(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (int32_t)0x80000000u;
    return (int32_t)x;
}

/* ---- sin / cos ---------------------------------------------------------------------
 * Cody-Waite reduction by pi/2 in three exactly-representable pieces (each product
 * k*piece is exact for |k| < 2^13, i.e. |x| < ~1.2e4; the shader only passes angles in
 * [0, 2pi]), then the degree-7 / degree-8 minimax kernels on [-pi/4, pi/4].          */
PT_HD float pt_sincos_reduce(float x, int32_t* quadrant)
{
    float k = pt_rint(x * 0.636619772367581343f);          /* x * 2/pi */
    float y = ((x - k * 1.5703125f) - k * 4.837512969970703125e-4f) - k * 7.54978995489188216e-8f;
    *quadrant = pt_f2i(k) & 3;
    return y;
}
PT_HD float pt_sin_kernel(float y)
{
    float z = y * y;
    float p = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * y;
    return p + y;
}
PT_HD float pt_cos_kernel(float y)
{
    float z = y * y;
    float p = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    return (p - 0.5f * z) + 1.0f;
}
PT_HD float pt_sin(float x)
{
    int32_t q;
    float y = pt_sincos_reduce(x, &q);
    float r = (q & 1) ? pt_cos_kernel(y) : pt_sin_kernel(y);
    return (q & 2) ? -r : r;
}
PT_HD float pt_cos(float x)
{
    int32_t q;
    float y = pt_sincos_reduce(x, &q);
    float r = (q & 1) ? pt_sin_kernel(y) : pt_cos_kernel(y);
    return ((q + 1) & 2) ? -r : r;
}

/* ---- log / log2 --------------------------------------------------------------------
 * x = m * 2^e with m in [sqrt(1/2), sqrt(2)); degree-9 kernel in (m - 1).            */
PT_HD float pt_log_split(float x, float* e_out)
{
    /* returns f = m - 1, m in [sqrt(.5), sqrt(2)); *e_out = exponent as float */
    uint32_t u = pt_asuint(x);
    int32_t e = 0;
    if (u < 0x00800000u) {                 /* subnormal: renormalise */
        x = x * 8388608.0f;                /* 2^23 */
        u = pt_asuint(x);
        e = -23;
    }
    e += (int32_t)(u >> 23) - 126;         /* x = m * 2^e, m in [0.5, 1) */
    float m = pt_asfloat((u & 0x007FFFFFu) | 0x3F000000u);
    float f;
    if (m < 0.707106781186547524f) { e -= 1; f = (m + m) - 1.0f; }
    else                            { f = m - 1.0f; }
    *e_out = (float)e;
    return f;
}
PT_HD float pt_log_poly(float f)
{
    /* returns y such that log(1+f) = f - 0.5 f^2 + y */
    float z = f * f;
    float y = ((((((((7.0376836292e-2f * f - 1.1514610310e-1f) * f + 1.1676998740e-1f) * f
               - 1.2420140846e-1f) * f + 1.4249322787e-1f) * f - 1.6668057665e-1f) * f
               + 2.0000714765e-1f) * f - 2.4999993993e-1f) * f + 3.3333331174e-1f) * f * z;
    return y;
}
PT_HD float pt_log(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return pt_asfloat(0x7FC00000u);
    if (x == 0.0f) return pt_asfloat(0xFF800000u);
    if (x == pt_asfloat(0x7F800000u)) return x;
    float fe;
    float f = pt_log_split(x, &fe);
    float z = f * f;
    float y = pt_log_poly(f);
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    float r = f + y;
    r = r + 0.693359375f * fe;
    return r;
}
PT_HD float pt_log2(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return pt_asfloat(0x7FC00000u);
    if (x == 0.0f) return pt_asfloat(0xFF800000u);
    if (x == pt_asfloat(0x7F800000u)) return x;
    float fe;
    float f = pt_log_split(x, &fe);
    float z = f * f;
    float y = pt_log_poly(f);
    y = y - 0.5f * z;
    /* log2(1+f) = (f + y) * log2(e), log2(e) = 1 + 0.44269504088896340736 */
    float r = y * 0.44269504088896340736f;
    r = r + f * 0.44269504088896340736f;
    r = r + y;
    r = r + f;
    r = r + fe;
    return r;
}

/* ---- exp2 ---------------------------------------------------------------------------- */
PT_HD float pt_exp2(float x)
{
    if (x != x) return x;
    if (x >= 128.0f) return pt_asfloat(0x7F800000u);
    if (x < -150.0f) return 0.0f;
    float fi = pt_floor(x);
    float f = x - fi;
    int32_t i = pt_f2i(fi);
    if (f > 0.5f) { i += 1; f = f - 1.0f; }
    float p = (((((1.535336188319500e-4f * f + 1.339887440266574e-3f) * f + 9.618437357674640e-3f) * f
              + 5.550332471162809e-2f) * f + 2.402264791363012e-1f) * f + 6.931472028550421e-1f) * f + 1.0f;
    /* scale by 2^i in two exact steps so that results below the normal range round once */
    int32_t i1 = i < -126 ? -126 : i;
    int32_t i2 = i - i1;                                   /* 0 or negative, >= -25 */
    float s1 = pt_asfloat((uint32_t)(i1 + 127) << 23);
    float r = p * s1;
    if (i2 != 0) r = r * pt_asfloat((uint32_t)(i2 + 127) << 23);
    return r;
}

/* HLSL pow(x, y) = exp2(y * log2(x)); x < 0 -> NaN, pow(0, y>0) = 0. */
PT_HD float pt_pow(float x, float y) { return pt_exp2(y * pt_log2(x)); }

/* ---- asin / acos --------------------------------------------------------------------- */
PT_HD float pt_asin_kernel(float x, float z)
{
    /* asin(x) for |x| <= 0.5, z = x*x */
    return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z
             + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * x + x;
}
PT_HD float pt_asin(float x)
{
    float a = pt_abs(x);
    if (!(a <= 1.0f)) return pt_asfloat(0x7FC00000u);
    float r;
    if (a > 0.5f) {
        float z = 0.5f * (1.0f - a);
        float s = pt_sqrt(z);
        r = 1.5707963267948966192f - 2.0f * pt_asin_kernel(s, z);
    } else {
        r = pt_asin_kernel(a, a * a);
    }
    return x < 0.0f ? -r : r;
}
PT_HD float pt_acos(float x)
{
    if (!(pt_abs(x) <= 1.0f)) return pt_asfloat(0x7FC00000u);
    if (x > 0.5f) {
        float z = 0.5f * (1.0f - x);
        float s = pt_sqrt(z);
        return 2.0f * pt_asin_kernel(s, z);
    }
    if (x < -0.5f) {
        float z = 0.5f * (1.0f + x);
        float s = pt_sqrt(z);
        return 3.14159265358979323846f - 2.0f * pt_asin_kernel(s, z);
    }
    return 1.5707963267948966192f - pt_asin_kernel(x, x * x);
}

/* ---- atan / atan2 / fmod  (environment map, util/sky.hlsl:45-53) ----------------------- */
PT_HD float pt_atan(float x)
{
    /* range reduction at tan(pi/8) and tan(3pi/8), degree-4 odd polynomial on the reduced argument (|error| < 2 ulp) */
    float a = pt_abs(x), y = 0.0f;
    if (a > 2.414213562373095f) { y = 1.5707963267948966192f; a = -(1.0f / a); }
    else if (a > 0.4142135623730950f) { y = 0.7853981633974483096f; a = (a - 1.0f) / (a + 1.0f); }
    float z = a * a;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * a + a);
    return x < 0.0f ? -y : y;
}
PT_HD float pt_atan2(float y, float x)
{
    /* HLSL atan2(y, x): quadrant-correct angle in [-pi, pi]; (0, 0) -> 0 */
    if (pt_isnan(x) || pt_isnan(y)) return pt_asfloat(0x7FC00000u);
    if (x == 0.0f) {
        if (y == 0.0f) return 0.0f;
        return y > 0.0f ? 1.5707963267948966192f : -1.5707963267948966192f;
    }
    float r = pt_atan(y / x);
    if (x < 0.0f) r = (y < 0.0f) ? r - 3.14159265358979323846f : r + 3.14159265358979323846f;
    return r;
}
PT_HD float pt_trunc(float x) { return __builtin_truncf(x); }
PT_HD float pt_fmod(float x, float y) { return x - y * pt_trunc(x / y); }     /* HLSL fmod: sign of x */

/* ---- the reference's hash RNG.  util/random.hlsl:5-16 (uint32 exact) ------------------ */
PT_HD void pt_rng_next(uint32_t* state)
{
    uint32_t old = *state + 747796405u + 2891336453u;          /* add, not multiply: as the shader */
    uint32_t word = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
    *state = (word >> 22u) ^ word;
}
PT_HD float pt_random_float(uint32_t* state)
{
    pt_rng_next(state);
    return (float)(*state) / 4294967296.0f;     /* (float)0xffffffffu == 2^32; result in [0,1] inclusive */
}

#endif /* PTMI_MATH_H */
