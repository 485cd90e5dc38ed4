// development aid: host cost of the allocation policy (s2r_voices.h) on the C3 event schedule, ns per event.
//   g++ -O2 -std=c++17 -pthread -I synth2_amd/csrc tools/ubench/policy_bench.cpp -o tools/ubench/_build/policy_bench && tools/ubench/_build/policy_bench
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <cstdlib>
#include "s2r_voices.h"

struct Ev { uint8_t kind, note; uint16_t frame; };

// usage: policy_bench [worker threads for the batch form, default 3]
int main(int argc, char **argv) {
    const uint32_t n_workers = argc > 1 ? (uint32_t)atoi(argv[1]) : 3u;
    for (uint32_t V : {65536u, 524288u, 1048576u}) {
        const uint32_t period = 64, frames = 1024;
        std::vector<std::vector<Ev>> cyc(period);
        const uint32_t per = V / period;
        for (uint32_t v = 0; v < V; v++) {
            const uint32_t sb = std::min(v / per, period - 1);
            const uint64_t delay = 16ull * (512 + ((1103515245ull * v + 12345ull) % (1ull << 31)) % 2048);
            cyc[sb].push_back({1, (uint8_t)(36 + v % 61), 0});
        }
        std::vector<std::vector<Ev>> offs(period);
        for (uint32_t v = 0; v < V; v++) {
            const uint32_t sb = std::min(v / per, period - 1);
            const uint64_t delay = 16ull * (512 + ((1103515245ull * v + 12345ull) % (1ull << 31)) % 2048);
            offs[(sb + delay / frames) % period].push_back({0, (uint8_t)(36 + v % 61), (uint16_t)(delay % frames)});
        }
        for (uint32_t b = 0; b < period; b++) {
            std::stable_sort(offs[b].begin(), offs[b].end(), [](const Ev &a, const Ev &c) { return a.frame < c.frame; });
            cyc[b].insert(cyc[b].end(), offs[b].begin(), offs[b].end());
        }
        S2rVoicePool pool(V);
        uint64_t sink = 0, n = 0;
        auto run = [&](uint32_t k) {
            uint32_t t = 0;
            for (const Ev &e : cyc[k % period]) {
                if (e.frame != t) { pool.advance(e.frame - t); t = e.frame; }
                if (e.kind) sink += pool.note_on(e.note, 1.0f); else sink += (uint64_t)pool.note_off(e.note);
                n++;
            }
            pool.advance(frames - t);
        };
        for (uint32_t k = 0; k < 2 * period; k++) run(k);
        n = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (uint32_t k = 0; k < 4 * period; k++) run(k);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%8u voices: one event at a time %6.1f ns per event, %7.1f us per buffer (%llu events per buffer)  [%llu]\n", V, dt * 1e9 / n, dt * 1e6 / (4 * period),
               (unsigned long long)(n / (4 * period)), (unsigned long long)sink);
        // the batch form (what s2r_note_events calls): one thread, then the caller's thread + n_workers
        for (uint32_t workers : {0u, n_workers}) {
            S2rVoicePool bp(V);
            bp.set_workers(workers, 1024);
            std::vector<int64_t> out(V);
            uint64_t sink2 = 0, nb = 0;
            auto runb = [&](uint32_t k) {
                const std::vector<Ev> &b = cyc[k % period];
                static_assert(sizeof(Ev) == sizeof(S2rPolicyEvent), "same layout");
                const uint32_t t = bp.resolve_batch(reinterpret_cast<const S2rPolicyEvent *>(b.data()), sizeof(Ev), b.size(), 0u, out.data());
                bp.advance(frames - t);
                for (size_t i = 0; i < b.size(); i += 64) sink2 += (uint64_t)out[i];
                nb += b.size();
            };
            for (uint32_t k = 0; k < 2 * period; k++) runb(k);
            nb = 0;
            auto tb0 = std::chrono::steady_clock::now();
            double best = 1e9;                                  // (the quietest period too: a shared host's noise is one-sided)
            for (uint32_t rep = 0; rep < 8; rep++) {
                auto tp0 = std::chrono::steady_clock::now();
                for (uint32_t k = 0; k < period; k++) runb(k);
                best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count());
            }
            const double dtb = std::chrono::duration<double>(std::chrono::steady_clock::now() - tb0).count();
            printf("%8u voices: batch, %u worker threads    %6.1f ns per event, %7.1f us per buffer (quietest period of 8: %5.1f ns, %6.1f us)  [%llu]\n", V, workers,
                   dtb * 1e9 / nb, dtb * 1e6 / (8 * period), best * 1e9 / (nb / 8), best * 1e6 / period, (unsigned long long)sink2);
        }
    }
    return 0;
}
