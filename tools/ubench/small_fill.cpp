// development aid: wall time of s2r_fill for the reference's own call pattern — 8 voices, 16 frames per call
// (s2_bin/src/main.rs:138-147) — from a C++ caller over the C ABI: a launch per call against the resident kernel
// (s2r_set_low_latency).
//   g++ -O2 -std=c++17 -I include tools/ubench/small_fill.cpp -o tools/ubench/_build/small_fill -L synth2_amd -ls2r -Wl,-rpath,$PWD/synth2_amd
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "s2r.h"

static void die(s2r_synth *s, const char *what, int rc) { std::fprintf(stderr, "%s: %d %s\n", what, rc, s ? s2r_last_error(s) : ""); std::exit(1); }

int main(int argc, char **argv) {
    const bool json = argc > 1 && std::strcmp(argv[1], "--json") == 0;       // bench.py's small_fill leg: one line for 8 voices
    double med[3][2] = {{0, 0}, {0, 0}, {0, 0}}, p99[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    int vi = 0;
    // (1 024 voices — BASELINE config [1]'s pool — are four workgroups: s2r_set_resident's pool-resident kernel, the chain heads
    // and the mix inside it)
    for (uint32_t voices : {8u, 256u, 1024u}) {
        for (int lowlat = 0; lowlat < 2; lowlat++) {
            s2r_config cfg;
            std::memset(&cfg, 0, sizeof cfg);
            cfg.struct_size = sizeof cfg; cfg.total_voices = voices; cfg.max_frames = 2048; cfg.device = -1;
            s2r_synth *s = nullptr;
            int rc = s2r_create(&cfg, &s);
            if (rc) die(s, "s2r_create", rc);
            if ((rc = s2r_set_resident(s, lowlat))) die(s, "s2r_set_resident", rc);
            for (uint8_t n : {57, 64, 69}) s2r_note_on(s, n, 1.0f);
            float buf[16];
            for (int i = 0; i < 500; i++) if ((rc = s2r_fill(s, buf, 16, 48000))) die(s, "s2r_fill", rc);
            std::vector<double> us;
            for (int i = 0; i < 20000; i++) {
                if (i % 16 == 0) { if (i % 32 == 0) s2r_note_on(s, (uint8_t)(60 + (i / 32) % 12), 1.0f); else s2r_note_off(s, (uint8_t)(60 + (i / 32) % 12)); }
                const auto t0 = std::chrono::steady_clock::now();
                if ((rc = s2r_fill(s, buf, 16, 48000))) die(s, "s2r_fill", rc);
                us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
            }
            std::sort(us.begin(), us.end());
            double sum = 0; for (double v : us) sum += v;
            med[vi][lowlat] = us[us.size() / 2]; p99[vi][lowlat] = us[us.size() * 99 / 100];
            if (!json) std::printf("%4u voices, 16-frame s2r_fill from C++, %-16s mean %5.2f us, median %5.2f, p10 %5.2f, p99 %5.2f, max %6.1f   (last sample %g)\n", voices,
                        lowlat ? "resident kernel:" : "launch per call:", sum / us.size(), us[us.size() / 2], us[us.size() / 10], us[us.size() * 99 / 100], us.back(), buf[15]);
            s2r_destroy(s);
        }
        vi++;
    }
    if (json) std::printf("{\"voices\": 8, \"frames\": 16, \"calls\": 20000, \"launch_per_call_us\": %.2f, \"launch_per_call_p99_us\": %.2f, \"resident_kernel_us\": %.2f, "
                          "\"resident_kernel_p99_us\": %.2f, \"voices_256_resident_kernel_us\": %.2f, \"voices_1024_launch_per_call_us\": %.2f, \"voices_1024_resident_kernel_us\": %.2f}\n",
                          med[0][0], p99[0][0], med[0][1], p99[0][1], med[1][1], med[2][0], med[2][1]);
    return 0;
}
