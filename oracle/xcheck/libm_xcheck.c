/*
 * libm_xcheck.c — TEST INFRASTRUCTURE ONLY.
 *
 * Exhaustive bit-for-bit comparison of the PRODUCT's host/device-shared math
 * (synth2_amd/csrc/s2r_math.h, compiled here for the host with gcc) against what the
 * reference actually calls on a Linux host:
 *   expf   : Rust f32::exp  -> glibc expf   (filters.rs:21)
 *   powf2  : Rust 2f32.powf -> glibc powf   (process.rs:227, synth.rs:210)
 *   sleef2 : sleef pow(2,y) -> oracle/s2o_sleef.c (itself pinned to C SLEEF by sleef_xcheck)
 *   sinf/cosf : Rust f32::sin/cos -> glibc sinf/cosf (dsp_filters.rs:30-31,105-109)
 *   tanf   : Rust f32::tan  -> glibc tanf (dsp_filters.rs:205-207)
 *
 * usage: libm_xcheck expf|powf2|sleef2 <lo> <hi>     (every float in [lo,hi])
 *        libm_xcheck expf all | powf2 all            (every one of the 2^32 bit patterns)
 *        libm_xcheck div <c> <stride>                s2r_div_const(x, c) vs x / c for every
 *                                                    stride-th float with 2^-60 < |x| < 2^60
 *        libm_xcheck div65535                        all integers 0..65535 (the noise quotient)
 *        libm_xcheck divrcp <n>                      s2r_div_by_rcp64(a, 1/(double)b) vs a / b for n random (a, b)
 *                                                    plus every dividend against 4096 divisors near all-ones / one
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../synth2_amd/csrc/s2r_math.h"

float s2o_sleef_powf(float x, float y);
static const uint64_t T[32] = S2R_EXP2F_TABLE_INIT;

static uint64_t checked, mism; static int shown;
static int mode;

static inline void one(uint32_t u) {
    float x = s2r_u2f(u), mine, ref;
    if (mode == 0) { mine = s2r_expf(x, T); ref = expf(x); }
    else if (mode == 1) { mine = s2r_pow2_libm(x, T); volatile float two = 2.0f; ref = powf(two, x); }
    else if (mode == 3) { mine = s2r_sinf(x); ref = sinf(x); }
    else if (mode == 4) { mine = s2r_cosf(x); ref = cosf(x); }
    else if (mode == 5) { mine = s2r_tanf(x); ref = tanf(x); }
    else { mine = s2r_pow2_sleef(x); ref = s2o_sleef_powf(2.0f, x); }
    checked++;
    if (mine != mine && ref != ref) return;
    if (s2r_f2u(mine) != s2r_f2u(ref)) {
        mism++;
        if (shown < 10) { shown++; fprintf(stderr, "MISMATCH in=%a mine=%a ref=%a\n", x, mine, ref); }
    }
}

static int div_mode(float c, uint32_t stride) {
    const float rc = 1.0f / c;
    uint64_t n = 0, bad = 0;
    for (int sign = 0; sign < 2; sign++)
        for (uint32_t u = s2r_f2u(0x1p-60f) + 1; u < s2r_f2u(0x1p60f); u += stride) {
            float x = s2r_u2f(u | (sign ? 0x80000000u : 0u));
            float a = s2r_div_const_nocheck(x, c, rc), b = x / c, g = s2r_div_const(x, c, rc);
            n++;
            if (s2r_f2u(a) != s2r_f2u(b) || s2r_f2u(g) != s2r_f2u(b)) { bad++; if (shown < 10) { shown++; fprintf(stderr, "MISMATCH %a / %a: %a vs %a\n", x, c, a, b); } }
        }
    /* the guarded form must also be right outside the window */
    const float edge[] = {0.0f, -0.0f, 0x1p-149f, -0x1p-140f, 0x1p-126f, 0x1p-61f, 0x1p61f, 0x1.fffffep127f, -0x1.fffffep127f, INFINITY, -INFINITY, NAN};
    for (unsigned i = 0; i < sizeof edge / sizeof *edge; i++) {
        float g = s2r_div_const(edge[i], c, rc), b = edge[i] / c;
        n++;
        if (!(g != g && b != b) && s2r_f2u(g) != s2r_f2u(b)) { bad++; fprintf(stderr, "EDGE MISMATCH %a / %a: %a vs %a\n", edge[i], c, g, b); }
    }
    printf("checked=%llu mismatches=%llu\n", (unsigned long long)n, (unsigned long long)bad);
    return bad ? 1 : 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && !strcmp(argv[1], "div65535")) {
        uint64_t bad = 0;
        for (uint32_t v = 0; v <= 65535; v++) {
            float a = s2r_div_const_nocheck((float)v, 65535.0f, 0x1.0001p-16f), b = (float)v / 65535.0f;
            if (s2r_f2u(a) != s2r_f2u(b)) bad++;
        }
        printf("checked=65536 mismatches=%llu\n", (unsigned long long)bad);
        return bad ? 1 : 0;
    }
    if (argc >= 3 && !strcmp(argv[1], "divrcp")) {
        const uint64_t n = strtoull(argv[2], 0, 10);
        uint64_t x = 0x9E3779B97F4A7C15ull, bad = 0, cnt = 0;
        for (uint64_t i = 0; i < n; i++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            /* exponents 2^-40 .. 2^40 (the oscillator's range and far beyond), any significands */
            const uint32_t ua = ((uint32_t)x & 0x007fffffu) | ((87u + (uint32_t)((x >> 24) % 81u)) << 23);
            const uint32_t ub = ((uint32_t)(x >> 32) & 0x007fffffu) | ((87u + (uint32_t)((x >> 56) % 81u)) << 23);
            const float a = s2r_u2f(ua), b = s2r_u2f(ub);
            const float q = s2r_div_by_rcp64(a, s2r_rcp_f64(b)), r = a / b;
            cnt++;
            if (s2r_f2u(q) != s2r_f2u(r)) { bad++; if (shown < 10) { shown++; fprintf(stderr, "MISMATCH %a / %a: %a vs %a\n", a, b, q, r); } }
        }
        /* structured: divisors with significands next to 1.0 and to 2.0 (the hard cases of reciprocal-based division) */
        for (uint32_t k = 0; k < 2048; k++)
            for (int hi = 0; hi < 2; hi++) {
                const float b = s2r_u2f(0x3f800000u | (hi ? 0x007fffffu - k : k));
                const double rb = s2r_rcp_f64(b);
                for (uint32_t m = 0; m < 0x00800000u; m += 37u) {
                    const float a = s2r_u2f(0x3f800000u | m);
                    cnt++;
                    if (s2r_f2u(s2r_div_by_rcp64(a, rb)) != s2r_f2u(a / b)) { bad++; if (shown < 10) { shown++; fprintf(stderr, "MISMATCH %a / %a\n", a, b); } }
                }
            }
        printf("checked=%llu mismatches=%llu\n", (unsigned long long)cnt, (unsigned long long)bad);
        return bad ? 1 : 0;
    }
    if (argc >= 4 && !strcmp(argv[1], "div")) return div_mode(strtof(argv[2], 0), (uint32_t)strtoul(argv[3], 0, 10));
    if (argc < 3) return 2;
    mode = !strcmp(argv[1], "expf") ? 0 : !strcmp(argv[1], "powf2") ? 1 : !strcmp(argv[1], "sinf") ? 3 : !strcmp(argv[1], "cosf") ? 4 : !strcmp(argv[1], "tanf") ? 5 : 2;
    if (!strcmp(argv[2], "all")) {
        uint32_t u = 0; do { one(u); } while (++u != 0);
    } else {
        float lo = strtof(argv[2], 0), hi = strtof(argv[3], 0);
        if (lo < 0) { float top = hi < 0 ? hi : -0.0f; for (uint32_t u = s2r_f2u(top); u <= s2r_f2u(lo); u++) one(u); }
        if (hi >= 0) { float bot = lo > 0 ? lo : 0.0f; for (uint32_t u = s2r_f2u(bot); u <= s2r_f2u(hi); u++) one(u); }
    }
    printf("checked=%llu mismatches=%llu\n", (unsigned long long)checked, (unsigned long long)mism);
    return mism ? 1 : 0;
}
