// s2r_device.h — kernel argument blocks and launch entry points shared by the host side
// (s2r_host.cpp) and the gfx950 kernels (s2r_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include "s2r.h"

// voice flag bits (synth.rs:27-28: the two Option discriminants)
#define S2R_VF_STARTED 1u    // current_frame_offset.is_some()
#define S2R_VF_RELEASED 2u   // release_frame_offset.is_some()

// One ADSR resolved for a sample rate.  A, D, R are `Ms::as_samples` (units.rs:44-53);
// the three slopes are the `y_rise / x_run` of simdtest.rs:247-251, hoisted out of the
// per-frame work (a correctly rounded division gives the same bits wherever it runs).
struct S2rEnv {
    float A;          // attack (samples) == decay_offset
    float D;          // decay (samples)
    float S;          // sustain level
    float R;          // release (samples)
    float sus_off;    // A + D                         simdtest.rs:282
    float slope_att;  // 1.0 / A                       simdtest.rs:295-299
    float slope_dec;  // (S - 1.0) / D                 simdtest.rs:303-307
    float slope_rel;  // (-S) / R                      simdtest.rs:313-317
};

// Per-voice state, struct-of-arrays in HBM (one entry per shard voice, padded to a whole
// number of workgroups).  28 B are read and 12 B written per started voice per fill
// (+16 B each way under the second-order filters).
struct S2rVoiceArrays {
    float *pitch;         // note_to_pitch(note)             synth.rs:179,208-212
    uint32_t *offset;     // current_frame_offset            synth.rs:27
    uint32_t *release;    // release_frame_offset            synth.rs:28
    uint32_t *flags;      // S2R_VF_*
    float *phase;         // OscillatorState.phase_accum     oscillators.rs:402-406
    float *lpf_last;      // LowPassFilterState.last         filters.rs:5-7
    uint32_t *seed;       // NoiseState.seed                 state.rs:17-21
    // dsp_filters.rs:12-17,82-89 states; only touched by patches with lpf.kind != onepole
    float *fx1, *fx2, *fy1, *fy2;
    uint32_t *program;    // index into the patch bank the voice was started with (0 without a bank)
};
#define S2R_VOICE_WORDS 12

// One patch of the bank, resolved for the fill's sample rate (what S2rRenderParams carries for
// the single-patch kernels).
struct S2rBankEntry {
    int32_t osc_kind;
    float osc_gain, noise_level, lpf_freq, amt_osc, amt_lpf;
    int32_t lpf_kind;
    float lpf_shape;      // damping_factor (LP2/HP2) or quality_factor (BP2)
    S2rEnv amp, mod;
};

struct S2rTimedEvent;

struct S2rRenderParams {
    // patch (static_config.rs:4-44), shared by every voice
    int32_t osc_kind;
    float osc_gain;
    float noise_level;
    float lpf_freq;
    float amt_osc;        // mod_env_to_osc_freq
    float amt_lpf;        // mod_env_to_lpf_freq
    int32_t lpf_kind;     // s2r_filter_kind; != 0 renders through s2r_render_dspf_kernel
    float lpf_damping;    // damping_factor (LP2/HP2) or quality_factor (BP2)
    S2rEnv amp;
    S2rEnv mod;
    float sr;             // sample_rate as f32 (units.rs:21)
    float rcp_sr;         // RN(1/sr)
    int32_t fast_div_sr;  // 1 => x/sr may use the 3-op exact quotient (rate verified exhaustively)
    int32_t no_flat_shortcut;  // 1 => always recompute the LPF coefficient (measurement knob)
    uint32_t frames;      // this fill
    uint32_t n_voices;    // shard voices
    uint32_t frames_stride; // row stride of block_partials / per_voice (== max_frames or frames)
    uint32_t super_frames;  // frames between two cross-wave combines (set by the launcher)
    S2rVoiceArrays v;
    float *block_partials;   // [n_blocks][frames_stride]
    float *direct_out;       // single-workgroup shard with the root add: the final mix, written by the render
                             // kernel itself ((+0.0) + the block's sum, what s2r_mix_kernel computes for one row)
    int32_t direct_stereo;   // interleaved L,R
    float *per_voice;        // [n_voices][frames] or nullptr (mix-disabled debug/parity path)
    const float *sin_table;  // 1024 floats (tables.rs)
    // coefficient stream: LPF coefficients of the 64-voice groups whose mod
    // envelope moves during this fill, computed ahead of the render kernel by s2r_coeff_kernel
    int32_t use_coeff;          // 0: never look at the stream
    const int32_t *group_slot;  // [n_groups64] stream slot of each 64-voice group, -1 = none
    int32_t *group_slot_w;      // same array, writable (classify kernel)
    uint32_t *slot_group;       // [coeff_capacity] inverse map
    uint32_t *coeff_count;      // [2] slots handed out this fill (index = coeff_parity), next fill's is zeroed
    uint32_t coeff_parity;
    uint32_t coeff_capacity;    // slots the stream buffer holds; more moving groups => in-lane path
    float *coeff;               // [slot][quad][64][4]
    // timed events of this fill (nullptr: none)
    const S2rTimedEvent *tev;
    int32_t *voice_ev_head;     // [padded voices] index of the voice's first timed event, -1 = none
    // patch bank (bank_size > 1: s2r_render_general_kernel<ANY, true>; the fields above then hold patch 0)
    const S2rBankEntry *bank;
    uint32_t bank_size;
};

// Coalesced note events, one record per touched voice per fill (host folds the event
// stream of synth.rs:61-80 between two fills into the voice's final state).
struct S2rVoiceEvent {
    uint32_t voice;       // shard-local index
    uint32_t flags;       // bit0: restart (note_on), bit1: release (note_off after the last on),
                          // bits 16..: the program (patch bank index) a restart carries
    float pitch;          // valid when restart
    uint32_t seed;        // NoiseState.seed for the restarted voice (reference: 0)
};
#define S2R_EV_RESTART 1u
#define S2R_EV_RELEASE 2u
#define S2R_EV_PROGRAM_SHIFT 16

// A note event that takes effect INSIDE a fill, at a 16-frame boundary — what s2_bin does by
// calling sample() 16 frames at a time with MIDI applied in between (main.rs:138-143), here
// inside one launch.  Records live in mapped host memory; a voice's events of one fill form a
// chain (next), its first one is published through voice_ev_head[voice].
struct S2rTimedEvent {
    uint32_t voice;       // shard-local index
    uint32_t frame;       // frame offset inside the fill, multiple of 16
    uint32_t flags;       // S2R_EV_RESTART / S2R_EV_RELEASE, S2R_TEV_FIRST
    float pitch;          // valid when restart
    uint32_t seed;
    int32_t next;         // index of the voice's next event in this fill, -1 = none
    uint32_t program;     // patch bank index of a restart
    uint32_t _pad;
};
#define S2R_TEV_FIRST 0x100u

// Events + classification in one launch: the fill's folded note events ride in
// the kernel arguments (no trip to host memory); every 64-voice group applies the ones that hit
// it and classifies itself on the result.
#define S2R_PREP_MAX_EVENTS 288u
struct S2rPrepParams {
    S2rRenderParams p;
    uint32_t n_events;
    uint32_t _pad;
    uint32_t ev[3 * S2R_PREP_MAX_EVENTS];   // voice, S2rVoiceEvent.flags, pitch bits
};
static_assert(sizeof(S2rPrepParams) <= 4096, "kernel arguments are limited to 4 KiB");

struct S2rMixParams {
    const float *block_partials;  // [n_blocks][frames_stride]
    uint32_t n_blocks;
    uint32_t blocks_per_group;
    uint32_t n_groups;
    uint32_t frames;
    uint32_t frames_stride;
    int32_t root_add;             // 1: out = (+0.0) + total  (synth.rs:176), 0: partial only
    int32_t stereo;               // 1: write interleaved L,R (audio_player.rs:224-228)
    float *out;
};

hipError_t s2r_launch_coeff(const S2rRenderParams &p, hipStream_t stream);   // classify + coefficient pass
hipError_t s2r_launch_prep(const S2rPrepParams &a, hipStream_t stream);      // events + classify in one launch, then the coefficient pass
hipError_t s2r_launch_render(const S2rRenderParams &p, uint32_t block_voices, uint32_t lanes_per_voice, hipStream_t stream);
bool s2r_lane_variants_built();      // the 2- and 4-lanes-per-voice kernels exist only in builds with -DS2R_WITH_LANE_VARIANTS
hipError_t s2r_launch_mix(const S2rMixParams &p, hipStream_t stream);
hipError_t s2r_launch_events(const S2rVoiceArrays &v, const S2rVoiceEvent *dev_events, uint32_t n, hipStream_t stream);
hipError_t s2r_launch_tev_heads(int32_t *heads, const S2rTimedEvent *tev, S2rTimedEvent *tev_copy, uint32_t n, hipStream_t stream);
// build-defined 4x decimator: x = 62 samples of history + 4 * n_out new ones, h = 63 taps (device), and the
// last 62 inputs copied to the front of x afterwards (second launch) for the next call
hipError_t s2r_launch_decimate4(float *x_with_history, const float *taps, uint32_t n_out, float *out, hipStream_t stream);
hipError_t s2r_launch_sum_rows(const float *rows, uint32_t n_rows, uint32_t frames, float *out, hipStream_t stream);
