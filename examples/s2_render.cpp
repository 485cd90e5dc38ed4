// s2_render — the synth thread of s2_bin (components/s2_bin/src/main.rs:120-160) as a C++ caller of
// libs2r through the C++ mirror of `Synth` (include/s2_synth.hpp): take device buffers, apply the
// MIDI that is due, call Synth::sample.  Instead of cpal and a MIDI port it reads a note script
// and writes raw little-endian f32 frames, so its output can be compared with anything else.
//
//   s2_render <script.txt> <out.f32> [--buffer N] [--voices N] [--rate HZ] [--patch file.synth2] [--batched] [--launch-per-call]
//
// script lines:  <frame> on <note> [velocity 0..127]   |   <frame> off <note>       (# comments)
//
// default mode: the reference's loop — 16-frame chunks, MIDI applied before each (main.rs:138-147) — with the library's
//               resident render kernel between the calls (Synth::set_low_latency: a command and a polled word per call
//               instead of a kernel launch); --launch-per-call: without it.  Same samples.
// --batched:    one Synth::sample per device buffer; the MIDI of the buffer goes ahead in one
//               s2r_note_events call, stamped with the 16-frame boundary it belongs to.  Same samples.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "s2_synth.hpp"

struct Msg { uint64_t frame; bool on; uint8_t note; uint8_t velocity; };

static std::vector<Msg> read_script(const char *path) {
    std::ifstream in(path);
    if (!in) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(2); }
    std::vector<Msg> out;
    std::string line;
    while (std::getline(in, line)) {
        const size_t h = line.find('#');
        if (h != std::string::npos) line.resize(h);
        std::istringstream ls(line);
        uint64_t frame; std::string kind; int note, vel = 127;
        if (!(ls >> frame >> kind >> note)) continue;
        ls >> vel;
        out.push_back(Msg{frame, kind == "on", (uint8_t)note, (uint8_t)vel});
    }
    std::stable_sort(out.begin(), out.end(), [](const Msg &a, const Msg &b) { return a.frame < b.frame; });
    return out;
}

// main.rs:192-207 apply_midi: velocity byte / 127.0
static void apply(s2::Synth &synth, const Msg &m) {
    if (m.on) synth.note_on(s2::Note{m.note}, s2::Velocity{{(float)m.velocity / 127.0f}});
    else synth.note_off(s2::Note{m.note});
}

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: s2_render <script> <out.f32> [--buffer N] [--voices N] [--rate HZ] [--patch f] [--batched] [--frames N]\n"); return 2; }
    size_t buffer_frames = 2048; uint32_t voices = 8, rate = 48000; bool batched = false, low_latency = true; uint64_t total_frames = 0;
    std::string patch_path;
    for (int i = 3; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--buffer") buffer_frames = std::strtoul(next(), nullptr, 10);
        else if (a == "--voices") voices = (uint32_t)std::strtoul(next(), nullptr, 10);
        else if (a == "--rate") rate = (uint32_t)std::strtoul(next(), nullptr, 10);
        else if (a == "--frames") total_frames = std::strtoull(next(), nullptr, 10);
        else if (a == "--patch") patch_path = next();
        else if (a == "--batched") batched = true;
        else if (a == "--launch-per-call") low_latency = false;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    const std::vector<Msg> script = read_script(argv[1]);
    if (!total_frames) total_frames = (script.empty() ? 0 : script.back().frame) + 4 * buffer_frames;
    try {
        s2::Synth synth(voices, (uint32_t)buffer_frames);
        if (!patch_path.empty()) {
            std::ifstream pf(patch_path);
            std::stringstream ss; ss << pf.rdbuf();
            synth.load_patch(ss.str());
        }
        if (!batched && low_latency) synth.set_low_latency(true);        // (pools of one workgroup; bigger ones render the ordinary way)
        const s2::SampleRateKhz sample_rate{rate};
        std::FILE *out = std::fopen(argv[2], "wb");
        if (!out) { std::fprintf(stderr, "cannot write %s\n", argv[2]); return 2; }
        std::vector<float> buffer(buffer_frames);
        size_t next_msg = 0;
        for (uint64_t pos = 0; pos < total_frames; pos += buffer_frames) {
            const size_t n = (size_t)std::min<uint64_t>(buffer_frames, total_frames - pos);
            if (!batched) {
                // let mut chunks = buffer.array_chunks_mut::<16>(); apply MIDI, sample, ... then the remainder
                for (size_t c = 0; c < n; c += 16) {
                    while (next_msg < script.size() && script[next_msg].frame <= pos + c) apply(synth, script[next_msg++]);
                    synth.sample(buffer.data() + c, std::min<size_t>(16, n - c), sample_rate);
                }
            } else {
                std::vector<s2r_note_event> ev;
                while (next_msg < script.size() && script[next_msg].frame < pos + n) {
                    const Msg &m = script[next_msg++];
                    // the chunk boundary at or after the message: where the reference's loop would apply it
                    const uint64_t at = m.frame <= pos ? 0 : ((m.frame - pos + 15) / 16) * 16;
                    if (at >= n) { next_msg--; break; }
                    ev.push_back(s2r_note_event{(uint8_t)(m.on ? S2R_NOTE_ON : S2R_NOTE_OFF), m.note, (uint16_t)at, (float)m.velocity / 127.0f});
                }
                if (!ev.empty()) {
                    const int rc = s2r_note_events(synth.handle(), ev.data(), ev.size());
                    if (rc != S2R_OK) throw s2::Error(rc, s2r_last_error(synth.handle()));
                }
                synth.sample(buffer.data(), n, sample_rate);
            }
            std::fwrite(buffer.data(), sizeof(float), n, out);
        }
        std::fclose(out);
    } catch (const s2::Error &e) {
        std::fprintf(stderr, "libs2r: %s (status %d)\n", e.what(), e.status);
        return 1;
    }
    return 0;
}
