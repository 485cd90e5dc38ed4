// s2r_math.h — exact-arithmetic building blocks shared by the gfx950 kernels and the
// host side of libs2r (PRODUCT code; the CPU oracle under oracle/ does NOT use this file,
// it calls the host libm / its own SLEEF restatement, so the two are independent).
//
// Every routine here is written only in IEEE-754 +,-,*,/,fma and integer ops so that the
// same source yields the same bits under gcc (host) and hipcc (gfx950).  Translation
// units including it MUST be compiled with -ffp-contract=off and without fast-math.
//
// What each routine reproduces (reference call sites, all under
// /root/reference/components/s2_lib/src/try3/):
//   s2r_expf        Rust `f32::exp` = glibc `expf`           filters.rs:21
//   s2r_pow2_sleef  `sleef::Sleef::pow(2.0, y)` on f32x16    process.rs:244
//   s2r_pow2_libm   Rust `2_f32.powf(y)` = glibc `powf`      process.rs:227
//   s2r_fmod1 / s2r_fmod_period  Rust `%` on f32 = fmodf     oscillators.rs:379,66,105,154
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define S2R_HD __host__ __device__ __forceinline__
#else
#define S2R_HD static inline
#endif

#define S2R_EXP2F_N 32

// 2^(i/32) as IEEE binary64 bits minus (i << 47): the table of glibc's expf/exp2f/powf
// (Szabolcs Nagy's "optimized-routines" single-precision exp family, glibc >= 2.27).
// Values are recomputed from scratch (correctly rounded 2^(i/32)) by tools/gen_exp2f_table.py.
#define S2R_EXP2F_TABLE_INIT { \
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, \
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, \
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull, \
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, \
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull, \
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, \
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, \
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull }

S2R_HD uint32_t s2r_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
S2R_HD float s2r_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
S2R_HD uint64_t s2r_d2u(double d) { uint64_t u; __builtin_memcpy(&u, &d, 8); return u; }
S2R_HD double s2r_u2d(uint64_t u) { double d; __builtin_memcpy(&d, &u, 8); return d; }

// ---------------------------------------------------------------------------------------
// glibc expf (sysdeps/ieee754/flt-32/e_expf.c, FMA ifunc variant): double-precision
// evaluation of 2^(k/32) * p(r), rounded once to float.  T = S2R_EXP2F_TABLE (host array
// or LDS copy).
// ---------------------------------------------------------------------------------------
typedef struct s2r_expf_partial { double r; uint64_t ki; uint64_t t; } s2r_expf_partial;

// first half: argument reduction and the table read (on the GPU an LDS read whose latency the
// caller hides behind independent work before calling s2r_expf_end)
S2R_HD s2r_expf_partial s2r_expf_begin(float x, const uint64_t* T) {
    const double xd = (double)x;
    const double InvLn2N = 0x1.71547652b82fep+0 * S2R_EXP2F_N;
    const double Shift = 0x1.8p+52;
    const double z = InvLn2N * xd;
    double kd = z + Shift;
    s2r_expf_partial p;
    p.ki = s2r_d2u(kd);
    kd = kd - Shift;
    // glibc's FMA build contracts `r = z - kd` (z being the product above) into one fma;
    // found by exhaustive comparison with the host expf (2 inputs of 2^32 tell them apart).
    p.r = __builtin_fma(InvLn2N, xd, -kd);
    p.t = T[p.ki % S2R_EXP2F_N];
    return p;
}

S2R_HD float s2r_expf_end(float x, const s2r_expf_partial p) {
    const double C0 = 0x1.c6af84b912394p-5 / S2R_EXP2F_N / S2R_EXP2F_N / S2R_EXP2F_N;
    const double C1 = 0x1.ebfce50fac4f3p-3 / S2R_EXP2F_N / S2R_EXP2F_N;
    const double C2 = 0x1.62e42ff0c52d6p-1 / S2R_EXP2F_N;
    const double r = p.r;
    const uint64_t t = p.t + (p.ki << (52 - 5));
    const double s = s2r_u2d(t);
    const double z = __builtin_fma(C0, r, C1);
    const double r2 = r * r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(z, r2, y);
    y = y * s;
    float res = (float)y;
    // |x| >= 88 or NaN: the value above is meaningless there (no traps, table index is masked);
    // replace it.  Kept after the main path so the common case is straight-line code.
    const uint32_t ux = s2r_f2u(x);
    const uint32_t abstop = (ux >> 20) & 0x7ff;
    if (__builtin_expect(abstop >= 0x42b, 0)) {
        if (ux == 0xff800000u) res = 0.0f;                       // -inf
        else if (abstop >= 0x7f8) res = x + x;                   // +inf / NaN
        else if (x > 0x1.62e42ep6f) res = __builtin_inff();     // overflow
        else if (x < -0x1.9fe368p6f) res = 0.0f;                 // underflow
    }
    return res;
}

S2R_HD float s2r_expf(float x, const uint64_t* T) { return s2r_expf_end(x, s2r_expf_begin(x, T)); }

// ---------------------------------------------------------------------------------------
// glibc powf(2.0f, y) (sysdeps/ieee754/flt-32/e_powf.c, FMA variant).  For x == 2 the
// log2_inline step is exactly 1.0 (table entry {invc=1, logc=0}, r == 0), so
// ylogx == (double)y and the rest is exp2_inline.
// ---------------------------------------------------------------------------------------
S2R_HD float s2r_pow2_libm(float y, const uint64_t* T) {
    const uint32_t iy = s2r_f2u(y);
    if (__builtin_expect(2u * iy - 1u >= 2u * 0x7f800000u - 1u, 0)) {   // y is 0, inf or NaN
        if (2u * iy == 0) return 1.0f;
        if (2u * iy > 2u * 0x7f800000u) return 2.0f + y;                   // NaN
        return (iy & 0x80000000u) ? 0.0f : y * y;                          // -inf -> 0, +inf -> inf
    }
    const double ylogx = (double)y;
    if (__builtin_expect(((s2r_d2u(ylogx) >> 47) & 0xffff) >= (s2r_d2u(126.0) >> 47), 0)) {
        if (ylogx > 0x1.fffffffd1d571p+6) return __builtin_inff();
        if (ylogx <= -150.0) return 0.0f;
    }
    const double Shift = 0x1.8p+52 / S2R_EXP2F_N;
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
    double kd = ylogx + Shift;
    const uint64_t ki = s2r_d2u(kd);
    kd = kd - Shift;
    const double r = ylogx - kd;
    uint64_t t = T[ki % S2R_EXP2F_N];
    t += ki << (52 - 5);
    const double s = s2r_u2d(t);
    const double z = __builtin_fma(C0, r, C1);
    const double r2 = r * r;
    double v = __builtin_fma(C2, r, 1.0);
    v = __builtin_fma(z, r2, v);
    v = v * s;
    return (float)v;
}

// ---------------------------------------------------------------------------------------
// SLEEF u10 powf with x == 2 (sleefsimdsp.c xpowf/logkf/expkf, FMA flavour).
// logkf(2.0) evaluates *exactly* to the double-float constant (ln2_hi, ln2_lo) below
// (m == 1 => x == 0, all correction terms vanish), so pow(2,y) = expkf(df(ln2) * y).
// Valid (bit-exact vs C SLEEF 3.8) wherever the result is a normal float; the patch
// domain |y| <= 10 (Bipolar<10> x Unipolar<1>, static_config.rs:17-20) is far inside.
// ---------------------------------------------------------------------------------------
// core: requires |y| < 150 (the kernels call it with |y| <= 10, guaranteed by the patch ranges
// s2r_set_patch enforces: mod env in [0,1] x Bipolar<10>)
S2R_HD float s2r_pow2_sleef_core(float y) {
    // d = dfmul((ln2_hi, ln2_lo), y)
    const float Lh = 0.69314718246459960938f, Ll = -1.904654323148236017e-09f;
    const float dx = Lh * y;
    const float dy = __builtin_fmaf(Ll, y, __builtin_fmaf(Lh, y, -dx));
    // expkf(d)
    const float R_LN2f = 1.442695040888963407359924681001892137426645954152985934135449406931f;
    const float L2Uf = 0.693145751953125f, L2Lf = 1.428606765330187045e-06f;
    float u = (dx + dy) * R_LN2f;
    const float qf = __builtin_rintf(u);
    const int q = (int)qf;
    // s = dfadd2(d, q * -L2U)
    float a = qf * -L2Uf;
    float sx = dx + a, v = sx - dx;
    float sy = ((dx - (sx - v)) + (a - v)) + dy;
    // s = dfadd2(s, q * -L2L)
    a = qf * -L2Lf;
    float tx = sx + a; v = tx - sx;
    float ty = ((sx - (tx - v)) + (a - v)) + sy;
    // s = dfnormalize(s)
    sx = tx + ty; sy = (tx - sx) + ty;
    u = 0.00136324646882712841033936f;
    u = __builtin_fmaf(u, sx, 0.00836596917361021041870117f);
    u = __builtin_fmaf(u, sx, 0.0416710823774337768554688f);
    u = __builtin_fmaf(u, sx, 0.166665524244308471679688f);
    u = __builtin_fmaf(u, sx, 0.499999850988388061523438f);
    // w = dfsqu(s)
    const float wx = sx * sx;
    const float wy = __builtin_fmaf(sx + sx, sy, __builtin_fmaf(sx, sx, -wx));
    // m = dfmul(w, u)
    const float mx = wx * u;
    const float my = __builtin_fmaf(wy, u, __builtin_fmaf(wx, u, -mx));
    // t = dfadd2(s, m)
    tx = sx + mx; v = tx - sx;
    ty = ((sx - (tx - v)) + (mx - v)) + (sy + my);
    // t = dfadd(1, t)
    const float ox = 1.0f + tx;
    const float oy = ((1.0f - ox) + tx) + ty;
    u = ox + oy;
    // vldexp(u, q): u in (0.7, 1.5), |q| <= 217 => two exact power-of-two scalings
    const int q1 = q >> 1, q2 = q - q1;
    u = u * s2r_u2f((uint32_t)(q1 + 127) << 23) * s2r_u2f((uint32_t)(q2 + 127) << 23);
    if (dx < -104.0f) u = 0.0f;
    return (y == 0.0f) ? 1.0f : u;    // xpowf: "if (y == 0 || x == 1) result = 1"
}

S2R_HD float s2r_pow2_sleef(float y) {
    // Outside |y| < 150 (NaN, inf, certain over/underflow) — never reached from a valid patch.
    if (__builtin_expect(!(__builtin_fabsf(y) < 150.0f), 0)) {
        if (y != y) return y + y;
        return y > 0 ? __builtin_inff() : 0.0f;
    }
    return s2r_pow2_sleef_core(y);
}

// Correctly rounded x / c for a divisor that is constant over a launch; rc = RN(1/c).
// q0 = RN(x*rc); e = x - q0*c exactly (fma); q = RN(q0 + e*rc)  (Markstein's correction).
// Exhaustively verified against true division for the divisors the kernels use it with
// (65535 and the whitelisted sample rates): oracle/xcheck/libm_xcheck.c mode "div".
// Outside a safe exponent window (and for 0, inf, NaN) it falls back to the real quotient.
S2R_HD float s2r_div_const_nocheck(float x, float c, float rc) {   // caller guarantees the window
    const float q0 = x * rc;
    const float e = __builtin_fmaf(-q0, c, x);
    return __builtin_fmaf(e, rc, q0);
}
// v / 65535 for an integer 0 <= v <= 65535 (hashnoise.rs:46, value / u16_max), correctly rounded in two
// operations: 1/65535 = 2^-16 + 2^-32 + 2^-48 + ..., so v/65535 = v*2^-16 (exact) + v*(2^-32 + 2^-48) + v*2^-64*(..),
// and one fma rounds v*(2^-32 + 2^-48) + v*2^-16 — the dropped tail (< 2^-48) never reaches a rounding
// boundary: compared with true division for all 65 536 values (tests/test_host_logic.py, exact in f64;
// tests/test_gpu_parity.py::test_noise_division_all_u16_values on the device).
S2R_HD float s2r_div_u16_by_65535(float v) {
    return __builtin_fmaf(v, 0x1.0001p-32f, v * 0x1p-16f);
}
S2R_HD float s2r_div_const(float x, float c, float rc) {
    const float q0 = x * rc;
    const float e = __builtin_fmaf(-q0, c, x);
    float q = __builtin_fmaf(e, rc, q0);
    const float ax = __builtin_fabsf(x);
    if (__builtin_expect(!(ax > 0x1p-60f && ax < 0x1p60f), 0)) q = x / c;
    return q;
}

// Rust `as u32` / Simd::cast::<u32>() on f32: saturating, NaN -> 0.
S2R_HD uint32_t s2r_f32_as_u32(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    // v_cvt_u32_f32 truncates, clamps to [0, 0xffffffff] and maps NaN to 0: exactly `as u32`
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
#endif
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}

// ---------------------------------------------------------------------------------------
// fmodf specialisations (fmod is exact by definition, so any exact formula is bit-equal).
// ---------------------------------------------------------------------------------------
// fmodf(x, 1.0f): x - trunc(x) is exact for every finite x; sign of a zero result follows x.
S2R_HD float s2r_fmod1(float x) {
    const float r = x - __builtin_truncf(x);
    return __builtin_copysignf(r, x);
}

// fmodf(off, period) at the oscillator call sites (oscillators.rs:66,105,154; lookup.rs:195),
// where off = fma(period, phase, 0.0) with phase in [0,1): almost always 0 <= off < period
// and the result is `off` itself.  Everything else goes to the exact library fmodf
// (device: ocml's __ocml_fmod_f32 via HIP's fmodf; LLVM's own `frem` expansion is NOT exact).
S2R_HD float s2r_fmod_period(float off, float period) {
    // 0 <= off < period, tested on the bit patterns with ONE unsigned compare: a negative or
    // NaN `off` has a larger pattern than any positive finite period (period > 0 is an
    // invariant: pitch > 0, sr > 0, pow2 > 0).
    if (__builtin_expect(s2r_f2u(off) < s2r_f2u(period) && period > 0.0f, 1)) return off;
#if defined(__HIP_DEVICE_COMPILE__)
    return ::fmodf(off, period);
#else
    return __builtin_fmodf(off, period);
#endif
}

// ---------------------------------------------------------------------------------------
// a / b correctly rounded to f32 for a divisor that stays the same over many quotients:
// rb = RN53(1 / (double)b) once, then RN24(RN53((double)a * rb)) per quotient — three plain
// instructions instead of the ~10 of an IEEE f32 division.  Why it is exact: a/b is the ratio of
// two 24-bit significands, so unless it IS an f32 it differs from every f32 rounding boundary
// (a 25-bit midpoint m) by at least 2^-49 relative — a - b*m is a non-zero multiple of the grid
// of the <= 49-bit product b*m, and it cannot be zero because b*m has an odd 25-bit factor — while
// (double)a * rb is within 2^-52 relative of a/b; so both lie on the same side of every boundary.
// (Finite, non-zero b; a*rb inside the double range, which any two floats give.)
// ---------------------------------------------------------------------------------------
S2R_HD double s2r_rcp_f64(float b) { return 1.0 / (double)b; }
S2R_HD float s2r_div_by_rcp64(float a, double rb) { return (float)((double)a * rb); }

// ---------------------------------------------------------------------------------------
// glibc sinf / cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h; FMA ifunc variant):
// double-precision polynomials after a quadrant reduction, rounded once to float.  Used by the
// second-order filters of dsp_filters.rs (Rust f32::sin / f32::cos).  |y| < 120 goes through
// reduce_fast, larger finite arguments through reduce_large (a 192-bit window of 4/pi);
// inf / NaN give NaN (glibc's NaN sign is not reproduced: callers keep arguments finite).
// ---------------------------------------------------------------------------------------
typedef struct s2r_sincos_poly { double c0, c1, c2, c3, c4, s1, s2, s3; } s2r_sincos_poly;

S2R_HD float s2r_sinf_poly(double x, double x2, int negate_cos, int n) {
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5,
                 c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double t1 = __builtin_fma(x2, s3, s2);
        const double x7 = x3 * x2;
        const double s = __builtin_fma(x3, s1, x);
        return (float)__builtin_fma(x7, t1, s);
    } else {
        const double sg = negate_cos ? -1.0 : 1.0;
        const double x4 = x2 * x2;
        const double t2 = __builtin_fma(x2, sg * c4, sg * c3);
        const double t1 = __builtin_fma(x2, sg * c1, sg * c0);
        const double x6 = x4 * x2;
        const double c = __builtin_fma(x4, sg * c2, t1);
        return (float)__builtin_fma(x6, t2, c);
    }
}

// quadrant reduction of reduce_fast (!TOINT_INTRINSICS): hpi_inv is 2/pi * 2^24
S2R_HD double s2r_reduce_fast(double x, int *np) {
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double r = x * hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return __builtin_fma(-(double)n, hpi, x);
}

// reduce_large: the argument's 24-bit mantissa times the 96 bits of 4/pi that matter for its
// exponent, in integer arithmetic; returns the remainder in [-pi/4, pi/4] and the quadrant
S2R_HD double s2r_reduce_large(uint32_t xi, int *np) {
    // __inv_pio4: 4/pi = 0x1.45F306DC9C882A53F84EAFA3EA69BB81B6C52B3278872...p0 in 32-bit windows, 8 bits apart
    static const uint32_t inv_pio4[24] = {                       // (static: a table in constant memory, not an array built on every call's stack)
        0xa2u, 0xa2f9u, 0xa2f983u, 0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u, 0x4e441529u,
        0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u,
        0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};
    const double pi63 = 0x1.921FB54442D18p-62;
    const uint32_t *arr = &inv_pio4[(xi >> 26) & 15];
    const int shift = (xi >> 23) & 7;
    xi = (xi & 0xffffffu) | 0x800000u;
    xi <<= shift;
    uint64_t res0 = (uint64_t)(uint32_t)(xi * arr[0]);
    const uint64_t res1 = (uint64_t)xi * arr[4];
    const uint64_t res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    *np = (int)n;
    return (double)(int64_t)res0 * pi63;
}

S2R_HD float s2r_sinf(float y) {
    const uint32_t top = (s2r_f2u(y) >> 20) & 0x7ff;
    double x = (double)y;
    if (top < 0x3f4) {                                  // |y| < pi/4   (abstop12(0x1.921FB6p-1f) = 0x3f4)
        if (top < 0x398) return y;                      // |y| < 2^-12
        return s2r_sinf_poly(x, x * x, 0, 0);
    }
    int n;
    if (__builtin_expect(top >= 0x42f, 0)) {            // |y| >= 120
        if (top >= 0x7f8) return y - y;                 // inf, NaN
        const uint32_t xi = s2r_f2u(y);
        const int sign = (int)(xi >> 31);
        x = s2r_reduce_large(xi, &n);
        const int q = n + sign;
        const double sgl = ((q & 3) == 1 || (q & 3) == 2) ? -1.0 : 1.0;
        return s2r_sinf_poly(x * sgl, x * x, (q & 2) != 0, n);
    }
    x = s2r_reduce_fast(x, &n);
    const double sg = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;      // sign[n & 3] = {1,-1,-1,1}
    return s2r_sinf_poly(x * sg, x * x, (n & 2) != 0, n);
}

S2R_HD float s2r_cosf(float y) {
    const uint32_t top = (s2r_f2u(y) >> 20) & 0x7ff;
    double x = (double)y;
    if (top < 0x3f4) {
        if (top < 0x398) return 1.0f;
        return s2r_sinf_poly(x, x * x, 0, 1);
    }
    int n;
    if (__builtin_expect(top >= 0x42f, 0)) {            // |y| >= 120
        if (top >= 0x7f8) return y - y;
        const uint32_t xi = s2r_f2u(y);
        const int sign = (int)(xi >> 31);
        x = s2r_reduce_large(xi, &n);
        const int q = n + sign;
        const double sgl = ((q & 3) == 1 || (q & 3) == 2) ? -1.0 : 1.0;
        return s2r_sinf_poly(x * sgl, x * x, (q & 2) != 0, n ^ 1);
    }
    x = s2r_reduce_fast(x, &n);
    const int m = n + 1;
    const double sg = ((m & 3) == 1 || (m & 3) == 2) ? -1.0 : 1.0;
    return s2r_sinf_poly(x * sg, x * x, (m & 2) != 0, n ^ 1);
}

// ---------------------------------------------------------------------------------------
// glibc 2.35 tanf (sysdeps/ieee754/flt-32/s_tanf.c, k_tanf.c, e_rem_pio2f.c): fdlibm's float
// kernel (no FMA variant in that release: every product and sum rounded separately) after the
// sincosf quadrant reduction in double.  Used by the band-pass of dsp_filters.rs:199-230
// ((theta / (2 Q)).tan()).  Every finite argument.
// ---------------------------------------------------------------------------------------

// __kernel_tanf: tan(x + y) on [-pi/4, pi/4] (iy = 1) or -1/tan(x + y) (iy = -1)
S2R_HD float s2r_kernel_tanf(float x, float y, int iy) {
    const float pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;      // 0x3f490fda, 0x33222168
    const float T0 = 3.3333334327e-01f, T1 = 1.3333334029e-01f, T2 = 5.3968254477e-02f, T3 = 2.1869488060e-02f,
                T4 = 8.8632395491e-03f, T5 = 3.5920790397e-03f, T6 = 1.4562094584e-03f, T7 = 5.8804126456e-04f,
                T8 = 2.4646313977e-04f, T9 = 7.8179444245e-05f, T10 = 7.1407252108e-05f, T11 = -1.8558637748e-05f,
                T12 = 2.5907305826e-05f;
    const int32_t hx = (int32_t)s2r_f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    float z, r, v, w, s;
    if (ix < 0x39000000) {                               // |x| < 2^-13
        if ((int)x == 0) {
            if ((ix | (iy + 1)) == 0) return 1.0f / __builtin_fabsf(x);
            else if (iy == 1) return x;
            else return -1.0f / x;
        }
    }
    if (ix >= 0x3f2ca140) {                              // |x| >= 0.6744
        if (hx < 0) { x = -x; y = -y; }
        z = pio4 - x;
        w = pio4lo - y;
        x = z + w; y = 0.0f;
        if (__builtin_fabsf(x) < 0x1p-13f) return (float)(1 - ((hx >> 30) & 2)) * (float)iy * (1.0f - 2.0f * (float)iy * x);
    }
    z = x * x;
    w = z * z;
    r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
    v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
    s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T0 * s;
    w = x + r;
    if (ix >= 0x3f2ca140) {
        v = (float)iy;
        return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
    }
    if (iy == 1) return w;
    // -1 / (x + r), accurately
    float a, t;
    z = s2r_u2f(s2r_f2u(w) & 0xfffff000u);
    v = r - (z - x);
    t = a = -1.0f / w;
    t = s2r_u2f(s2r_f2u(t) & 0xfffff000u);
    s = 1.0f + t * z;
    return t + a * (s + t * v);
}

S2R_HD float s2r_tanf(float x) {
    const uint32_t xi = s2r_f2u(x);
    const int32_t ix = (int32_t)(xi & 0x7fffffffu);
    if (ix <= 0x3f490fda) return s2r_kernel_tanf(x, 0.0f, 1);
    if (ix >= 0x7f800000) return x - x;                  // inf, NaN
    // __ieee754_rem_pio2f (2.35): the sincosf reductions in double, NOT contracted here (this
    // routine has no FMA build: mulsd + subsd), remainder handed on as a float head + tail
    double dx = (double)x;
    int n;
    if (((xi >> 20) & 0x7ff) < 0x42f) {                  // |x| < 120: reduce_fast
        const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
        const double r = dx * hpi_inv;
        n = ((int32_t)r + 0x800000) >> 24;
        const double prod = (double)n * hpi;
        dx = dx - prod;
    } else {
        dx = s2r_reduce_large(xi, &n);
        if (xi >> 31) dx = -dx;
    }
    const float y0 = (float)dx;
    const float y1 = (float)(dx - (double)y0);
    return s2r_kernel_tanf(y0, y1, 1 - ((n & 1) << 1));
}
