// Host-side pieces of libs2r that need no GPU (voice pool, .synth2 parser) under AddressSanitizer and
// UndefinedBehaviorSanitizer: random traffic incl. checkpoint restores, and a mutation fuzz of patch text.
// Built and run by tests/test_sanitizers.py.
#include "s2r_voices.h"
#include "s2r_patch.h"
#include "s2r.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include <string>
int main() {
    std::mt19937 rng(7);
    for (uint32_t total : {1u, 7u, 64u, 65u, 4097u, 300000u}) {
        S2rVoicePool pool(total);
        for (int k = 0; k < 60000; k++) {
            const uint32_t r = rng();
            const uint8_t note = (r >> 8) % 128;
            if ((r & 3) == 0) pool.note_off(note); else if ((r & 3) == 1) pool.advance(r % 5000); else pool.note_on(note, 1.0f);
            if ((k % 50000) == 49999) { for (uint32_t i = 0; i < total; i += 1 + total / 50) pool.set_voice(i, note, (r >> 5) & 1, (r >> 6) & 1, r % 100000, r % 777, 0.5f); pool.rebuild(); }
            (void)pool.next_voice();
        }
    }
    // the shard maps' division (FastDiv): exact for every divisor kind and the whole 32-bit range
    for (uint32_t d : {1u, 2u, 3u, 7u, 8u, 61u, 64u, 100u, 255u, 256u, 1000u, 65536u, 131072u, 1048576u, 3000000001u, 0xffffffffu}) {
        FastDiv f; f.set(d);
        for (uint64_t x = 0; x <= 0xffffffffull; x += (x < 70000 ? 1 : 65521 + rng() % 4096)) {
            if (f.div((uint32_t)x) != (uint32_t)x / d || f.mod((uint32_t)x) != (uint32_t)x % d) { printf("FastDiv wrong: %u / %u\n", (uint32_t)x, d); return 3; }
        }
        for (uint32_t x : {0xffffffffu, 0xfffffffeu, 0x80000000u, 0x7fffffffu})
            if (f.div(x) != x / d) { printf("FastDiv wrong: %u / %u\n", x, d); return 3; }
    }
    const std::string base = "synth lead { osc.kind = sine; osc.gain = 0.25 noise = 0.125, lpf.freq = 1234.5 // c\n lpf.kind = bp2; lpf.q = 2 # x\n amp_env.attack = 1 }";
    int ok = 0, bad = 0;
    for (int k = 0; k < 60000; k++) {
        std::string t = base;
        const int edits = 1 + rng() % 4;
        for (int e = 0; e < edits; e++) {
            const size_t pos = rng() % (t.size() + 1);
            switch (rng() % 3) {
            case 0: t.insert(pos, 1, (char)(rng() % 256)); break;
            case 1: if (pos < t.size()) t.erase(pos, 1 + rng() % 3); break;
            default: if (pos < t.size()) t[pos] = (char)(32 + rng() % 95);
            }
        }
        s2r_patch p; std::string name, err;
        (s2r_parse_patch(t.data(), t.size(), &p, &name, &err) == 0 ? ok : bad)++;
    }
    printf("pool ok; parser: %d accepted, %d rejected\n", ok, bad);
    // s2r_stream_frame_json into a heap buffer of EXACTLY the advertised capacity (ASan sees one byte past it):
    // magnitudes 1e-7 .. 1e13 of both signs, nine-digit values in the "0.00000ddddddddd" window, specials
    size_t longest = 0;
    for (int round = 0; round < 40; round++) {
        const size_t n = round == 0 ? 0 : 1 + rng() % 5000;
        std::vector<float> x(n);
        for (size_t i = 0; i < n; i++) {
            const double mag = std::pow(10.0, -7.0 + 20.0 * (double)(rng() % 100000) / 100000.0);
            float v = (float)(mag * (1.0 + (double)(rng() % 1000003) / 1000003.0));
            if (round % 4 == 1) {          // consecutive floats below 2^-19 = 1.9e-6: eight digits behind "-0.00000", the 16-char layout (with the ',' one more than the 16 per sample the first version allowed)
                uint32_t b; const float hi = 0x1p-19f; std::memcpy(&b, &hi, 4); b -= 1u + (uint32_t)(round * 5000 + i); std::memcpy(&v, &b, 4);
                x[i] = -v; continue;
            }
            if (rng() % 97 == 0) v = (rng() & 1) ? NAN : INFINITY;
            if (rng() % 89 == 0) { uint32_t b = rng(); std::memcpy(&v, &b, 4); }
            x[i] = (rng() & 1) ? -v : v;
        }
        const size_t need = s2r_stream_frame_json(x.data(), n, nullptr, 0);
        if (need != 3 + (size_t)S2R_STREAM_CHARS_PER_SAMPLE * n) { printf("stream: need %zu for n %zu\n", need, n); return 1; }
        char *buf = (char *)std::malloc(need);
        const size_t len = s2r_stream_frame_json(x.data(), n, buf, need);
        if (len >= need || buf[len] != 0 || buf[0] != '[' || buf[len - 1] != ']') { printf("stream: bad frame\n"); return 1; }
        // every element's own length, and the too-small-buffer answer
        size_t start = 1;
        for (size_t i = 1; i <= len; i++) if (buf[i] == ',' || buf[i] == ']') { if (i - start > longest) longest = i - start; start = i + 1; }
        if (n && s2r_stream_frame_json(x.data(), n, buf, need - 1) != need) { printf("stream: short buffer accepted\n"); return 1; }
        std::free(buf);
    }
    printf("stream ok; longest element %zu chars\n", longest);
}
