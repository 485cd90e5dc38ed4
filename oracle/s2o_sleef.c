/*
 * s2o_sleef.c — TEST INFRASTRUCTURE ONLY (part of oracle/; never linked into the product).
 *
 * Scalar restatement of SLEEF's 1.0-ULP single-precision pow ("xpowf", the
 * function behind `sleef::Sleef::pow` on `f32x16` that the reference calls at
 * components/s2_lib/src/try3/process.rs:3,244).
 *
 * The reference pins the pure-Rust port `sleef 0.3.2` (+ `doubled 0.3.2`)
 * (Cargo.lock:1085-1091, 382-385); its source is NOT under /root/reference, so
 * this file restates the *published* SLEEF algorithm (sleefsimdsp.c: xpowf,
 * logkf, expkf; df.h: double-float helpers) in the FMA flavour
 * (ENABLE_FMA_SP), which is what the reference's build selects
 * (.cargo/config.toml:4-11, target-cpu=skylake-avx512 => target_feature "fma").
 *
 * Pinning: oracle/xcheck/sleef_xcheck.c compares this file bit-for-bit with the
 * compiled C SLEEF 3.8 that ships inside libtorch_cpu.so
 * (Sleef_powf8_u10avx2 / Sleef_powf16_u10avx512f; FMA builds) — see DESIGN.md
 * "Transcendentals".  The Rust port itself cannot be executed here, so parity
 * with *it* remains "unpinned"; parity with C SLEEF is measured.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "s2_oracle.h"

typedef struct { float x, y; } f2;

static inline f2 mk(float x, float y) { f2 r; r.x = x; r.y = y; return r; }

/* vfmapn(x,y,z) = x*y - z ; vfmanp(x,y,z) = -x*y + z ; vfma = x*y + z */
static inline float fmapn(float x, float y, float z) { return fmaf(x, y, -z); }
static inline float fmanp(float x, float y, float z) { return fmaf(-x, y, z); }

static inline f2 dfnormalize(f2 t) {
    float s = t.x + t.y;
    return mk(s, (t.x - s) + t.y);
}
static inline f2 dfscale(f2 d, float s) { return mk(d.x * s, d.y * s); }

static inline f2 dfadd2_f_f(float x, float y) {
    float s = x + y;
    float v = s - x;
    return mk(s, (x - (s - v)) + (y - v));
}
static inline f2 dfadd2_f2_f(f2 x, float y) {
    float s = x.x + y;
    float v = s - x.x;
    float t = (x.x - (s - v)) + (y - v);
    return mk(s, t + x.y);
}
static inline f2 dfadd_f_f2(float x, f2 y) {
    float s = x + y.x;
    return mk(s, ((x - s) + y.x) + y.y);
}
static inline f2 dfadd_f2_f2(f2 x, f2 y) {
    float s = x.x + y.x;
    return mk(s, (((x.x - s) + y.x) + x.y) + y.y);
}
static inline f2 dfadd2_f2_f2(f2 x, f2 y) {
    float s = x.x + y.x;
    float v = s - x.x;
    float t = (x.x - (s - v)) + (y.x - v);
    return mk(s, t + (x.y + y.y));
}
static inline f2 dfdiv(f2 n, f2 d) {
    float t = 1.0f / d.x;
    float s = n.x * t;
    float u = fmapn(t, n.x, s);
    float v = fmanp(d.y, t, fmanp(d.x, t, 1.0f));
    return mk(s, fmaf(s, v, fmaf(n.y, t, u)));
}
static inline f2 dfsqu(f2 x) {
    float s = x.x * x.x;
    return mk(s, fmaf(x.x + x.x, x.y, fmapn(x.x, x.x, s)));
}
static inline f2 dfmul_f2_f2(f2 x, f2 y) {
    float s = x.x * y.x;
    return mk(s, fmaf(x.x, y.y, fmaf(x.y, y.x, fmapn(x.x, y.x, s))));
}
static inline f2 dfmul_f2_f(f2 x, float y) {
    float s = x.x * y;
    return mk(s, fmaf(x.y, y, fmapn(x.x, y, s)));
}

static inline int ilogb2kf(float d) {
    uint32_t u; memcpy(&u, &d, 4);
    return (int)((u >> 23) & 0xff) - 0x7f;
}
static inline float ldexp3kf(float d, int e) {
    uint32_t u; memcpy(&u, &d, 4);
    u += (uint32_t)e << 23;
    float f; memcpy(&f, &u, 4); return f;
}

/* sleefsimdsp.c: logkf (non-getexp path; identical results for normal inputs) */
static f2 logkf(float d) {
    int o = d < 1.17549435e-38f;
    if (o) d = d * (4294967296.0f * 4294967296.0f);
    int e = ilogb2kf(d * (1.0f / 0.75f));
    float m = ldexp3kf(d, -e);
    if (o) e -= 64;

    f2 x = dfdiv(dfadd2_f_f(-1.0f, m), dfadd2_f_f(1.0f, m));
    f2 x2 = dfsqu(x);

    float t = 0.240320354700088500976562f;
    t = fmaf(t, x2.x, 0.285112679004669189453125f);
    t = fmaf(t, x2.x, 0.400007992982864379882812f);
    f2 c = mk(0.66666662693023681640625f, 3.69183861259614332084311e-09f);

    f2 s = dfmul_f2_f(mk(0.69314718246459960938f, -1.904654323148236017e-09f), (float)e);
    s = dfadd_f2_f2(s, dfscale(x, 2.0f));
    s = dfadd_f2_f2(s, dfmul_f2_f2(dfmul_f2_f2(x2, x), dfadd2_f2_f2(dfmul_f2_f(x2, t), c)));
    return s;
}

/* sleefsimdsp.c: vldexp_vf_vf_vi2 — scaling in 5 multiplies (matters only for
 * subnormal / overflowing results, where it is not the same as one ldexpf) */
static float vldexpf(float x, int q) {
    int m = q >> 31;
    m = (((m + q) >> 6) - m) << 4;
    q = q - (m << 2);
    m += 0x7f;
    m = m < 0 ? 0 : m;
    m = m > 0xff ? 0xff : m;
    uint32_t ub = (uint32_t)m << 23; float u; memcpy(&u, &ub, 4);
    x = x * u * u * u * u;
    ub = (uint32_t)(q + 0x7f) << 23; memcpy(&u, &ub, 4);
    return x * u;
}

/* sleefsimdsp.c: expkf */
static float expkf(f2 d) {
    const float R_LN2f = 1.442695040888963407359924681001892137426645954152985934135449406931f;
    const float L2Uf = 0.693145751953125f;
    const float L2Lf = 1.428606765330187045e-06f;
    float u = (d.x + d.y) * R_LN2f;
    int q = (int)rintf(u);
    f2 s, t;

    s = dfadd2_f2_f(d, (float)q * -L2Uf);
    s = dfadd2_f2_f(s, (float)q * -L2Lf);
    s = dfnormalize(s);

    u = 0.00136324646882712841033936f;
    u = fmaf(u, s.x, 0.00836596917361021041870117f);
    u = fmaf(u, s.x, 0.0416710823774337768554688f);
    u = fmaf(u, s.x, 0.166665524244308471679688f);
    u = fmaf(u, s.x, 0.499999850988388061523438f);

    t = dfadd2_f2_f2(s, dfmul_f2_f(dfsqu(s), u));
    t = dfadd_f_f2(1.0f, t);
    u = t.x + t.y;
    u = vldexpf(u, q);

    if (d.x < -104.0f) u = 0.0f;
    return u;
}

float s2o_sleef_powf(float x, float y) {
    float ay = fabsf(y);
    int yisint = (truncf(y) == y) || (ay > 16777216.0f);
    int yisodd = yisint && (ay < 16777216.0f) && (((int)y) & 1);

    float result = expkf(dfmul_f2_f(logkf(fabsf(x)), y));
    if (isnan(result)) result = INFINITY;
    result *= (x > 0) ? 1.0f : (yisint ? (yisodd ? -1.0f : 1.0f) : NAN);

    float efx = copysignf(1.0f, y) * (fabsf(x) - 1.0f);   /* vmulsign(|x|-1, y) */
    if (isinf(y)) result = (efx < 0.0f) ? 0.0f : (efx == 0.0f ? 1.0f : INFINITY);
    if (isinf(x) || x == 0.0f) {
        float sgn = yisodd ? copysignf(1.0f, x) : 1.0f;
        float yy = (x == 0.0f) ? -y : y;
        result = sgn * ((yy < 0.0f) ? 0.0f : INFINITY);
    }
    if (isnan(x) || isnan(y)) result = NAN;
    if (y == 0.0f || x == 1.0f) result = 1.0f;
    return result;
}
