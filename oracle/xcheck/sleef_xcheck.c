/*
 * sleef_xcheck.c — TEST INFRASTRUCTURE ONLY.
 *
 * Pins oracle/s2o_sleef.c (the restated SLEEF u10 powf) against the compiled
 * C SLEEF 3.8 inside libtorch_cpu.so (FMA build: Sleef_powf8_u10avx2).
 * libtorch is only a *witness* for the published algorithm; nothing of it is
 * shipped or linked into the product.
 *
 * usage: sleef_xcheck <mode> [args]
 *   pow2 lo hi      every float y in [lo,hi] (walks the bit patterns), x = 2.0
 *   grid n seed     n random (x,y) pairs, x in (0, 1e4], y in [-40,40]
 * prints: checked=<n> mismatches=<m> max_ulp=<u>
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__m256 Sleef_powf8_u10avx2(__m256, __m256);
float s2o_sleef_powf(float x, float y);

static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static int64_t ord(float f) { uint32_t u = f2u(f); return (u & 0x80000000u) ? -(int64_t)(u & 0x7fffffffu) : (int64_t)u; }

static uint64_t checked, mism; static int64_t maxulp;
static int shown;

static void check8(const float *x, const float *y, int n) {
    float xx[8], yy[8], rr[8];
    for (int i = 0; i < 8; i++) { xx[i] = x[i < n ? i : 0]; yy[i] = y[i < n ? i : 0]; }
    __m256 r = Sleef_powf8_u10avx2(_mm256_loadu_ps(xx), _mm256_loadu_ps(yy));
    _mm256_storeu_ps(rr, r);
    for (int i = 0; i < n; i++) {
        float mine = s2o_sleef_powf(xx[i], yy[i]);
        checked++;
        if (isnan(mine) && isnan(rr[i])) continue;
        if (f2u(mine) != f2u(rr[i])) {
            mism++;
            int64_t d = llabs(ord(mine) - ord(rr[i]));
            if (d > maxulp) maxulp = d;
            if (shown < 10) { shown++; fprintf(stderr, "MISMATCH x=%a y=%a mine=%a sleef=%a\n", xx[i], yy[i], mine, rr[i]); }
        }
    }
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    if (!strcmp(argv[1], "pow2")) {
        float lo = strtof(argv[2], 0), hi = strtof(argv[3], 0);
        float xs[8], ys[8]; int n = 0;
        for (int i = 0; i < 8; i++) xs[i] = 2.0f;
        /* walk negative range then positive range by bit pattern */
        if (lo < 0) {
            float top = hi < 0 ? hi : -0.0f;
            for (uint32_t u = f2u(top); u <= f2u(lo); u++) { ys[n++] = u2f(u); if (n == 8) { check8(xs, ys, 8); n = 0; } }
        }
        if (hi >= 0) {
            float bot = lo > 0 ? lo : 0.0f;
            for (uint32_t u = f2u(bot); u <= f2u(hi); u++) { ys[n++] = u2f(u); if (n == 8) { check8(xs, ys, 8); n = 0; } }
        }
        if (n) check8(xs, ys, n);
    } else if (!strcmp(argv[1], "grid")) {
        uint64_t n = strtoull(argv[2], 0, 10); uint64_t s = strtoull(argv[3], 0, 10) * 2654435761u + 1;
        float xs[8], ys[8];
        for (uint64_t i = 0; i < n; i += 8) {
            for (int k = 0; k < 8; k++) {
                s = s * 6364136223846793005ull + 1442695040888963407ull;
                double a = (double)(s >> 11) / 9007199254740992.0;
                s = s * 6364136223846793005ull + 1442695040888963407ull;
                double b = (double)(s >> 11) / 9007199254740992.0;
                xs[k] = (float)exp(a * 18.4 - 9.2);     /* 1e-4 .. 1e4, log-uniform */
                ys[k] = (float)(b * 80.0 - 40.0);
            }
            check8(xs, ys, 8);
        }
    } else return 2;
    printf("checked=%llu mismatches=%llu max_ulp=%lld\n", (unsigned long long)checked, (unsigned long long)mism, (long long)maxulp);
    return mism ? 1 : 0;
}
