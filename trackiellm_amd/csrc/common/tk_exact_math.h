/*
 * tk_exact_math.h — the numerics contract shared by the HIP kernels and the CPU oracle.
 *
 * Every function here is built only from IEEE-754 binary32 operations that are
 * correctly rounded on both x86-64 (g++ -ffp-contract=off -mfma) and gfx950
 * (hipcc -ffp-contract=off): add, sub, mul, fma, div, sqrt, int<->float
 * conversion and integer bit manipulation.  No libm / no ocml transcendental is
 * called, so a value computed on the host is BIT-IDENTICAL to the value computed
 * in a kernel.  That is what lets the parity tests demand 0-ulp agreement for
 * the LLM / detector / ASR streams instead of a tolerance.
 *
 * The reference delegates all of this arithmetic to llama.cpp / ONNX Runtime /
 * whisper.cpp, so there is no reference formula to follow; the
 * accuracy of each routine versus libm is pinned in tests/test_exact_math.py.
 */
#ifndef TK_EXACT_MATH_H
#define TK_EXACT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TK_HD __host__ __device__ __forceinline__
#else
#define TK_HD static inline
#endif

TK_HD float tk_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

TK_HD float tk_divf(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fdiv_rn(a, b);
#else
    return a / b;
#endif
}

TK_HD uint32_t tk_f32_bits(float f);
TK_HD float tk_bits_f32(uint32_t u);

/*
 * sqrt(a), a >= 0 (normal range): NOT the hardware instruction (gfx950's v_sqrt_f32 is a 1-ulp
 * approximation and __fsqrt_rn lowers to it), but a fixed Newton sequence of IEEE mul/fma from an
 * integer seed, so host and device agree bit for bit.  <= 1 ulp of the true root (tests pin it).
 */
TK_HD float tk_sqrtf(float a) {
    if (!(a > 0.0f)) return 0.0f;
    float y = tk_bits_f32(0x5f3759dfu - (tk_f32_bits(a) >> 1)); /* ~1/sqrt(a), 3.4 % */
    const float h = 0.5f * a;
    y = y * tk_fmaf(-h * y, y, 1.5f);
    y = y * tk_fmaf(-h * y, y, 1.5f);
    y = y * tk_fmaf(-h * y, y, 1.5f);
    y = y * tk_fmaf(-h * y, y, 1.5f);
    float s = a * y;
    float r = tk_fmaf(-s, s, a);
    return tk_fmaf(r, 0.5f * y, s);
}

TK_HD uint32_t tk_f32_bits(float f) {
    union { float f; uint32_t u; } v; v.f = f; return v.u;
}
TK_HD float tk_bits_f32(uint32_t u) {
    union { float f; uint32_t u; } v; v.u = u; return v.f;
}

/* round-to-nearest-even to an integer value, |x| < 2^22 */
TK_HD float tk_rintf(float x) {
    const float magic = 12582912.0f; /* 1.5 * 2^23 */
    float t = x + magic;
    return t - magic;
}

TK_HD float tk_fabsf(float x) { return tk_bits_f32(tk_f32_bits(x) & 0x7fffffffu); }
TK_HD float tk_fmaxf(float a, float b) { return a > b ? a : b; }
TK_HD float tk_fminf(float a, float b) { return a < b ? a : b; }

/* exp(x): Cody-Waite reduction + degree-6 polynomial, <= 1 ulp on [-87, 88]. */
TK_HD float tk_expf(float x) {
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) return 0.0f;
    const float log2e = 1.44269504088896341f;
    const float ln2_hi = 0.693145751953125f;       /* 12 significant bits */
    const float ln2_lo = 1.42860682030941723e-6f;
    float n = tk_rintf(x * log2e);
    float r = tk_fmaf(-n, ln2_hi, x);
    r = tk_fmaf(-n, ln2_lo, r);
    float p = 1.0f / 720.0f;
    p = tk_fmaf(p, r, 1.0f / 120.0f);
    p = tk_fmaf(p, r, 1.0f / 24.0f);
    p = tk_fmaf(p, r, 1.0f / 6.0f);
    p = tk_fmaf(p, r, 0.5f);
    p = tk_fmaf(p, r, 1.0f);
    p = tk_fmaf(p, r, 1.0f);
    int32_t ni = (int32_t)n;
    /* scale by 2^n in two steps so n in [-126, 127] never over/underflows the bias */
    int32_t n1 = ni / 2, n2 = ni - n1;
    float s1 = tk_bits_f32((uint32_t)(n1 + 127) << 23);
    float s2 = tk_bits_f32((uint32_t)(n2 + 127) << 23);
    return (p * s1) * s2;
}

/* natural log for x > 0 (normal range): x = m * 2^e, m in [sqrt(.5), sqrt(2)) */
TK_HD float tk_logf(float x) {
    uint32_t u = tk_f32_bits(x);
    int32_t e = (int32_t)(u >> 23) - 127;
    uint32_t mb = (u & 0x007fffffu) | 0x3f800000u;
    float m = tk_bits_f32(mb);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float s = tk_divf(f, 2.0f + f);
    float z = s * s;
    /* atanh series: log(m) = 2s(1 + z/3 + z^2/5 + z^3/7 + z^4/9) */
    float p = 1.0f / 9.0f;
    p = tk_fmaf(p, z, 1.0f / 7.0f);
    p = tk_fmaf(p, z, 1.0f / 5.0f);
    p = tk_fmaf(p, z, 1.0f / 3.0f);
    p = tk_fmaf(p, z, 1.0f);
    float lm = (2.0f * s) * p;
    const float ln2_hi = 0.693145751953125f;
    const float ln2_lo = 1.42860682030941723e-6f;
    float fe = (float)e;
    return tk_fmaf(fe, ln2_hi, tk_fmaf(fe, ln2_lo, lm));
}

TK_HD float tk_log10f(float x) { return tk_logf(x) * 0.434294481903251828f; }

TK_HD float tk_sigmoidf(float x) { return tk_divf(1.0f, 1.0f + tk_expf(-x)); }
TK_HD float tk_siluf(float x) { return x * tk_sigmoidf(x); }

TK_HD float tk_tanhf(float x) {
    /* tanh(x) = 1 - 2/(exp(2x)+1); saturates cleanly through tk_expf's clamps */
    float e = tk_expf(2.0f * x);
    return 1.0f - tk_divf(2.0f, e + 1.0f);
}

/* GELU, tanh form (the form ggml/whisper.cpp evaluates) */
TK_HD float tk_geluf(float x) {
    const float k0 = 0.797884560802865356f; /* sqrt(2/pi) */
    const float k1 = 0.044715f;
    float x3 = (x * x) * x;
    float inner = k0 * tk_fmaf(k1, x3, x);
    return (0.5f * x) * (1.0f + tk_tanhf(inner));
}

/* IEEE binary16 <-> binary32, round-to-nearest-even, subnormals handled. */
TK_HD float tk_f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    if (exp == 0) {
        if (man == 0) return tk_bits_f32(sign);
        /* subnormal: value = man * 2^-24 */
        float f = (float)man * 5.9604644775390625e-8f;
        return tk_bits_f32(tk_f32_bits(f) | sign);
    }
    if (exp == 31) return tk_bits_f32(sign | 0x7f800000u | (man << 13));
    return tk_bits_f32(sign | ((exp + 112u) << 23) | (man << 13));
}

TK_HD uint16_t tk_f32_to_f16(float f) {
    uint32_t u = tk_f32_bits(f);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((a > 0x7f800000u) ? 0x200u : 0u));
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); /* rounds to inf */
    if (a < 0x33000001u) return (uint16_t)sign;              /* rounds to zero */
    int32_t e = (int32_t)(a >> 23) - 127;
    uint32_t m = (a & 0x007fffffu) | 0x00800000u;
    uint32_t shift, hexp;
    if (e < -14) { shift = (uint32_t)(13 + (-14 - e)); hexp = 0; }
    else { shift = 13; hexp = (uint32_t)(e + 15); }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    /* q carries the implicit bit when normal; adding it to the exponent field handles carry-out */
    uint32_t out = (hexp == 0) ? q : (((hexp - 1u) << 10) + q);
    return (uint16_t)(sign | out);
}

#endif /* TK_EXACT_MATH_H */
