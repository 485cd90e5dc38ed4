// s2r_stream.cpp — the wire format of the reference's (disabled) websocket audio server:
// one text frame per buffer, `serde_json::to_string(&Vec<f32>)` of BUFFER_SIZE = 4096 mono samples
// rendered at 32 kHz (components/s2_bin/src/threads.rs:6,263,303-305; the consumer is
// www/streamer.js:82-90, `JSON.parse`).  That module is commented out of s2_bin (main.rs:4-5) and does
// not build, so nothing can be run against it: this is the published behaviour of serde_json 1.x
// for f32 — the shortest decimal that reads back as the same f32 (ryu), laid out by ryu's `format32`
// rules, `null` for NaN and the infinities — restated.  PARITY UNPINNED.
// Host-only; no GPU involved.
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include "s2r.h"

namespace {

// one f32 as serde_json prints it; returns the number of chars written (<= 24)
size_t format_f32(float v, char *out) {
    if (!std::isfinite(v)) { std::memcpy(out, "null", 4); return 4; }      // serde_json: non-finite floats -> null
    char *p = out;
    if (std::signbit(v)) { *p++ = '-'; v = -v; }
    if (v == 0.0f) { std::memcpy(p, "0.0", 3); return (size_t)(p + 3 - out); }
    // shortest round-trip digits: d[.ddd]e±XX
    char sci[32];
    const auto res = std::to_chars(sci, sci + sizeof sci, v, std::chars_format::scientific);
    char digits[16]; int n = 0; int exp10 = 0;
    const char *q = sci;
    for (; q < res.ptr && *q != 'e'; ++q) if (*q != '.') digits[n++] = *q;
    if (q < res.ptr) {                                     // exponent
        ++q;
        const bool neg = *q == '-';
        if (*q == '+' || *q == '-') ++q;
        for (; q < res.ptr; ++q) exp10 = exp10 * 10 + (*q - '0');
        if (neg) exp10 = -exp10;
    }
    // value = digits x 10^k with `n` digits; kk = position of the decimal point (ryu pretty::format32)
    const int k = exp10 - (n - 1);
    const int kk = n + k;
    if (0 <= k && kk <= 13) {                              // 1234e7 -> 12340000000.0
        std::memcpy(p, digits, (size_t)n); p += n;
        for (int i = n; i < kk; i++) *p++ = '0';
        *p++ = '.'; *p++ = '0';
    } else if (0 < kk && kk <= 13) {                       // 1234e-2 -> 12.34
        std::memcpy(p, digits, (size_t)kk); p += kk;
        *p++ = '.';
        std::memcpy(p, digits + kk, (size_t)(n - kk)); p += n - kk;
    } else if (-6 < kk && kk <= 0) {                       // 1234e-6 -> 0.001234
        *p++ = '0'; *p++ = '.';
        for (int i = kk; i < 0; i++) *p++ = '0';
        std::memcpy(p, digits, (size_t)n); p += n;
    } else {                                               // 1e30, 1.234e33
        *p++ = digits[0];
        if (n > 1) { *p++ = '.'; std::memcpy(p, digits + 1, (size_t)(n - 1)); p += n - 1; }
        *p++ = 'e';
        int e = kk - 1;
        if (e < 0) { *p++ = '-'; e = -e; }
        if (e >= 10) *p++ = (char)('0' + e / 10);
        *p++ = (char)('0' + e % 10);
    }
    return (size_t)(p - out);
}

}  // namespace

extern "C" size_t s2r_stream_frame_json(const float *samples, size_t n, char *out, size_t cap) {
    // longest element: 16 chars — "-0.00000" + 8 digits (1e-6 <= |v| < 1e-5 needs at most eight), "-0.0000" + 9
    // digits, or sign + 13 integer digits + ".0" — + ',' = 17 (the scientific layout: sign + d + '.' + 8 digits +
    // "e-45" = 15).  The advertised capacity allows 18 per sample; the loop below checks the space it has anyway.
    const size_t need = 2 + n * S2R_STREAM_CHARS_PER_SAMPLE + 1;
    if (!out || cap < need) return need;
    char *p = out;
    *p++ = '[';
    for (size_t i = 0; i < n; i++) {
        if (i) *p++ = ',';
        char one[32];                                      // format_f32 writes at most 24 chars
        const size_t len = format_f32(samples[i], one);
        if ((size_t)(out + cap - p) < len + 3) return need;    // cannot happen with cap >= need; never write past cap
        std::memcpy(p, one, len);
        p += len;
    }
    *p++ = ']';
    *p = '\0';
    return (size_t)(p - out);
}
