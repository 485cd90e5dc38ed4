// s2_synth.hpp — header-only C++ mirror of `s2_lib::try3::synth` over the C ABI (s2r.h).
//
// Same names, argument meaning and call pattern as the reference
// (/root/reference/components/s2_lib/src/try3/synth.rs:9-21,53-80,154-156; units.rs:11-14), so a
// C++ caller — or a test — reads like the reference's own call sites
// (components/s2_bin/src/main.rs:132-147,198-205):
//
//     s2::Synth synth;                                   // Synth::new()
//     synth.note_on(s2::Note{69}, s2::Velocity{{1.0f}});
//     synth.sample(buffer, n, s2::SampleRateKhz{48000}); // overwrites buffer
//     synth.note_off(s2::Note{69});
//
// Where the reference panics (offset overflow, process.rs:36) this throws s2::Error.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include "s2r.h"

namespace s2 {

template <unsigned N> struct Unipolar { float v; };   // units.rs:11
struct Note { uint8_t v; };                           // synth.rs:16  Note(pub u8)
struct Velocity { Unipolar<1> v; };                   // synth.rs:18  Velocity(pub Unipolar<1>)
struct SampleRateKhz { uint32_t v; };                 // units.rs:14  (holds Hz, units.rs:21)

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string &msg) : std::runtime_error(msg), status(st) {}
};

class Synth {
  public:
    // Synth::new(): NUM_VOICES = 8 in the reference (synth.rs:7); here a run-time size.
    explicit Synth(uint32_t num_voices = 8, uint32_t max_frames = 2048, int device = -1) {
        s2r_config cfg{};
        cfg.struct_size = sizeof cfg;
        cfg.total_voices = num_voices;
        cfg.max_frames = max_frames;
        cfg.device = device;
        const int rc = s2r_create(&cfg, &h_);
        if (rc != S2R_OK) throw Error(rc, s2r_status_string(rc));
    }
    explicit Synth(const s2r_config &cfg) {
        const int rc = s2r_create(&cfg, &h_);
        if (rc != S2R_OK) throw Error(rc, s2r_status_string(rc));
    }
    // ONE Synth over several GPUs (s2r_config.devices): the pool cut into one shard per device, the allocation policy
    // run once per event, the shards' partial mixes added in shard order on devices[0]
    static Synth with_devices(uint32_t num_voices, const int *devices, uint32_t n_devices, uint32_t max_frames = 2048,
                              uint32_t shard_interleave = 64) {
        s2r_config cfg{};
        cfg.struct_size = sizeof cfg;
        cfg.total_voices = num_voices;
        cfg.max_frames = max_frames;
        cfg.device = -1;
        cfg.shard_interleave = shard_interleave;
        cfg.n_devices = n_devices;
        for (uint32_t k = 0; k < n_devices && k < S2R_MAX_DEVICES; ++k) cfg.devices[k] = devices[k];
        return Synth(cfg);
    }
    ~Synth() { s2r_destroy(h_); }
    Synth(const Synth &) = delete;
    Synth &operator=(const Synth &) = delete;
    Synth(Synth &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }

    void load_patch(const std::string &synth2_text) { check(s2r_load_patch(h_, synth2_text.data(), synth2_text.size())); }
    // multi-timbral extension: a bank of patches, the current program picks the one a note_on uses
    void set_patch_bank(const s2r_patch *patches, uint32_t n) { check(s2r_set_patch_bank(h_, patches, n)); }
    void program_change(uint32_t program) { check(s2r_program_change(h_, program)); }
    void note_on(Note note, Velocity velocity) { check(s2r_note_on(h_, note.v, velocity.v.v)); }     // synth.rs:61-70
    void note_off(Note note) { check(s2r_note_off(h_, note.v)); }                                    // synth.rs:72-80
    // Synth::sample(&mut [f32], SampleRateKhz), synth.rs:154-169
    void sample(float *buffer, size_t len, SampleRateKhz sample_rate) { check(s2r_fill(h_, buffer, len, sample_rate.v)); }
    // the same in two halves for a caller with two buffers in flight (s2_bin: audio_player.rs:56-60, main.rs:135-149)
    void sample_begin(size_t len, SampleRateKhz sample_rate) { check(s2r_fill_begin(h_, len, sample_rate.v)); }
    void sample_end(float *buffer, size_t capacity) { check(s2r_fill_end(h_, buffer, capacity)); }
    // sample() per 16 frames from the audio callback (main.rs:138-147) without a launch per call: a resident render kernel
    // between calls, small pools only (s2r.h: s2r_set_low_latency); same samples either way
    void set_low_latency(bool enabled) { check(s2r_set_low_latency(h_, enabled ? 1 : 0)); }
    // a batch of events, each at frame 0 or at its 16-frame boundary inside the next buffer (main.rs:138-143)
    void note_events(const s2r_note_event *events, size_t n) { check(s2r_note_events(h_, events, n)); }

    s2r_synth *handle() { return h_; }

  private:
    void check(int rc) {
        if (rc != S2R_OK) throw Error(rc, s2r_last_error(h_));
    }
    s2r_synth *h_ = nullptr;
};

}  // namespace s2
