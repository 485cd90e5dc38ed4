// Host-side pieces of libs2r that need no GPU (voice pool, .synth2 parser) under AddressSanitizer and
// UndefinedBehaviorSanitizer: random traffic incl. checkpoint restores, and a mutation fuzz of patch text.
// Built and run by tests/test_sanitizers.py.
#include "s2r_voices.h"
#include "s2r_patch.h"
#include <cstdio>
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
}
