// The allocation policy's batch form on several threads (S2rVoicePool::resolve_batch, s2r_voices.h) under ThreadSanitizer,
// and held against the event-by-event policy: random batches, stolen voices still held, several boundaries per batch.
// Built and run by tests/test_sanitizers.py.
#include "s2r_voices.h"
#include <cstdio>
#include <random>
#include <vector>
int main() {
    std::mt19937 rng(11);
    for (uint32_t total : {64u, 3000u, 40000u}) {
        S2rVoicePool a(total), b(total);
        b.set_workers(3, 64);
        for (int batch = 0; batch < 40; batch++) {
            const size_t n = 1 + rng() % 6000;
            std::vector<S2rPolicyEvent> ev(n);
            uint16_t f = 0;
            for (size_t k = 0; k < n; k++) {
                if (batch % 2 && rng() % 50 == 0 && f < 1000) f = (uint16_t)(f + 16);
                ev[k].kind = rng() % 100 < 55 ? 1 : 0; ev[k].note = (uint8_t)(40 + rng() % (batch % 3 ? 9 : 80)); ev[k].frame = f;
            }
            std::vector<int64_t> want(n), got(n);
            uint32_t t = 0;
            for (size_t k = 0; k < n; k++) {
                if (ev[k].frame > t) { a.advance(ev[k].frame - t); t = ev[k].frame; }
                want[k] = ev[k].kind ? (int64_t)a.note_on(ev[k].note, 1.0f) : a.note_off(ev[k].note);
            }
            const uint32_t t2 = b.resolve_batch(ev.data(), sizeof(S2rPolicyEvent), n, 0u, got.data());
            if (t2 != t) { printf("clock differs\n"); return 2; }
            for (size_t k = 0; k < n; k++) if (want[k] != got[k]) { printf("pool %u batch %d event %zu: %lld vs %lld\n", total, batch, k, (long long)want[k], (long long)got[k]); return 3; }
            a.advance(1024 - t); b.advance(1024 - t);
            for (uint32_t i = 0; i < total; i += 1 + total / 97) {
                const S2rHostVoice x = a.voice(i), y = b.voice(i);
                if (x.note != y.note || x.started != y.started || x.released != y.released || x.start_clock != y.start_clock || x.release_clock != y.release_clock) {
                    printf("pool %u batch %d voice %u differs\n", total, batch, i); return 4; }
            }
        }
    }
    printf("policy threads ok\n");
    return 0;
}
