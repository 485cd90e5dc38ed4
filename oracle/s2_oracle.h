/*
 * s2_oracle.h — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's voice-render path
 * (/root/reference/components/s2_lib/src/try3/{synth,process,oscillators,filters,
 * hashnoise,lookup,envelopes,math,units}.rs and old/simdtest.rs).  It exists so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can CHECK and TIME the
 * algorithm on a CPU.  Nothing under synth2_amd/ (the product) may include, link or call
 * it; the product fails loudly when its HIP library is missing.
 *
 * Pinning status:
 *   - integer / IEEE stages: pinned by the reference's own 4 unit tests and by the
 *     known answers listed in SURVEY.md §8c (tests/test_oracle_known_answers.py);
 *   - expf / powf: the oracle calls the host libm, i.e. the very functions Rust's std
 *     f32::exp / f32::powf resolve to on Linux (filters.rs:21, synth.rs:210, process.rs:227);
 *   - sleef pow: restated (s2o_sleef.c) and pinned bit-for-bit against C SLEEF 3.8;
 *     the Rust port sleef-0.3.2 itself cannot be run here -> that last hop is UNPINNED;
 *   - SIN_TABLE: regenerated, compared with the reference literals when /root/reference
 *     is present, CRC32 committed.
 */
#ifndef S2_ORACLE_H
#define S2_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* static_config.rs:26-32 — enum order as declared there */
enum { S2O_OSC_SQUARE = 0, S2O_OSC_SAW = 1, S2O_OSC_TRIANGLE = 2, S2O_OSC_SINE = 3,
       /* build-defined alias-suppressed shapes: differentiated polynomial waveforms */
       S2O_OSC_DPW_SAW = 4, S2O_OSC_DPW_SQUARE = 5, S2O_OSC_DPW_TRIANGLE = 6 };

/* static_config.rs:38-44 */
typedef struct { float attack_ms, decay_ms, sustain, release_ms; } s2o_adsr_cfg;

/* static_config.rs:4-24 */
typedef struct {
    int32_t osc_kind;
    float osc_gain;
    float noise;
    float lpf_freq;
    s2o_adsr_cfg amp_env;
    s2o_adsr_cfg mod_env;
    float mod_env_to_osc_freq;
    float mod_env_to_lpf_freq;
    /* build-defined wiring of the reference's (unused) dsp_filters.rs into the layer: which
     * filter sample_voice[_x16] runs at the modulated cutoff.  0 = filters.rs one-pole (the
     * reference's live path). */
    int32_t lpf_kind;
    float lpf_damping;           /* SecondOrder{Low,High}PassFilter.damping_factor, Unipolar<10> */
    float lpf_q;                 /* SecondOrderBandPassFilter.quality_factor, Unipolar<10> */
} s2o_layer_cfg;

enum { S2O_FILT_ONEPOLE = 0, S2O_FILT_LP1 = 1, S2O_FILT_HP1 = 2, S2O_FILT_LP2 = 3, S2O_FILT_HP2 = 4, S2O_FILT_BP2 = 5,
       /* build-defined state-variable filter (no counterpart in the reference, notes.md:63 only names
        * it): trapezoidal SVF, low / band / high outputs; `shape` = lpf_q */
       S2O_FILT_SVF_LP = 6, S2O_FILT_SVF_BP = 7, S2O_FILT_SVF_HP = 8 };

/* state.rs:10-21, oscillators.rs:402-406, filters.rs:5-7 */
typedef struct {
    int32_t has_phase;      /* Option<Unipolar<1>> discriminant */
    float phase_accum;
    uint32_t seed;
    float lpf_last;
    float x1, x2, y1, y2;        /* dsp_filters.rs:12-17,82-89 filter states */
    int32_t has_z;               /* DPW oscillators: the differentiator's memory F(s[n-1]) (none before the first frame) */
    float dpw_z;
} s2o_layer_state;

/* synth.rs:23-30 */
typedef struct {
    uint8_t note;
    float velocity;
    int32_t has_current;
    uint32_t current_frame_offset;
    int32_t has_release;
    uint32_t release_frame_offset;
    s2o_layer_state state;
    uint32_t program;       /* patch bank index the voice was started with (build-defined extension) */
} s2o_voice;

/* synth.rs:9-12, with NUM_VOICES (synth.rs:7) made a run-time size */
typedef struct {
    s2o_layer_cfg config;
    uint32_t num_voices;
    s2o_voice *voices;
    int32_t panicked;        /* set where the reference would panic (process.rs:36,71) */
    uint64_t double_release; /* synth.rs:77 warn counter */
    /* patch bank (build-defined): bank_size == 0 -> `config` for every voice;
     * otherwise a voice renders with bank[its program] (bank[0] when the index is past the bank) */
    s2o_layer_cfg *bank;
    uint32_t bank_size;
    uint32_t current_program;
} s2o_synth;

/* GPU mix-tree description: every 16 consecutive voices are added in index
 * order (the reference's order, synth.rs:177-195), the 16-voice sums of a block of
 * `block_voices` in order, blocks in runs of 16 (sequential inside a run, run sums
 * sequential) inside each of `groups` contiguous groups, groups sequentially, and the root is
 * (+0.0f) + total (the `accum = splat(0.0)` of synth.rs:176).  For pools of <= 16 voices
 * this IS the reference's summation order. */
typedef struct { uint32_t block_voices; uint32_t groups; } s2o_tree;

s2o_layer_cfg s2o_default_config(void);                     /* synth.rs:125-152 */
s2o_synth *s2o_synth_new(uint32_t num_voices);              /* synth.rs:54-59 */
void s2o_synth_free(s2o_synth *s);
void s2o_set_bank(s2o_synth *s, const s2o_layer_cfg *cfgs, uint32_t n);   /* n == 0: back to `config` */
void s2o_program_change(s2o_synth *s, uint32_t program);
void s2o_note_on(s2o_synth *s, uint8_t note, float velocity);   /* synth.rs:61-70,101-120 */
void s2o_note_off(s2o_synth *s, uint8_t note);                  /* synth.rs:72-96 */
uint32_t s2o_next_voice_index(const s2o_synth *s);              /* synth.rs:101-120 */

/* Synth::sample (synth.rs:154-203): sequential voice-order mix, exactly as the reference. */
void s2o_sample(s2o_synth *s, float *buffer, size_t frames, uint32_t sample_rate);

/* Same rendering and state advance; every voice's frames go to per_voice[v*frames + i]
 * (rows of never-started voices are +0.0).  Mixing is then a separate step. */
void s2o_render_voices(s2o_synth *s, float *per_voice, size_t frames, uint32_t sample_rate);
/* multi-threaded variant for the cpu_baseline timing leg (voices sharded over threads) */
void s2o_render_voices_mt(s2o_synth *s, float *per_voice, size_t frames, uint32_t sample_rate, int threads);
/* render + sequential mix without materialising per-voice rows, voices sharded over
 * `threads` threads, thread partials summed in thread order (timing leg only). */
void s2o_sample_mt(s2o_synth *s, float *buffer, size_t frames, uint32_t sample_rate, int threads);
/* One buffer the way the reference's caller drives Synth (s2_bin/src/main.rs:138-147): MIDI applied between 16-frame
 * sample() calls.  Events carry the frame (a multiple of 16, non-decreasing) at which they take effect; same layout as
 * libs2r's s2r_note_event.  per_voice ([num_voices][frames]) and / or mix (thread partials in thread order) may be
 * NULL.  Worker threads persist between calls. */
typedef struct { uint8_t kind; uint8_t note; uint16_t frame; float velocity; } s2o_note_event;   /* kind: 0 off, 1 on, 2 program change */
void s2o_render_events_mt(s2o_synth *s, const s2o_note_event *events, size_t n_events, float *per_voice, float *mix,
                          size_t frames, uint32_t sample_rate, int threads);
/* timing legs: pool worker t (t >= 1; the caller is thread 0 and pins itself) runs on CPU cpus[t]; n = 0 unpins */
void s2o_pool_pin(const int *cpus, int n);
/* seconds spent by s2o_render_events_mt [0] in note_on / note_off (the reference's O(V) scans) and [1] rendering */
void s2o_events_seconds(double out[2], int reset);
void s2o_mix_sequential(const float *per_voice, uint32_t voices, size_t frames, float *out);
void s2o_mix_tree(const float *per_voice, uint32_t voices, size_t frames, s2o_tree tree, float *out);
/* partial mix of one shard (what one GPU produces): tree over the shard's voices, no root add */
void s2o_mix_tree_partial(const float *per_voice, uint32_t voices, size_t frames, uint32_t block_voices, float *out);
/* the same two, one frame at a time: the statement of the tree that the row-at-a-time forms above are tested against */
void s2o_mix_tree_scalar(const float *per_voice, uint32_t voices, size_t frames, s2o_tree tree, float *out);
void s2o_mix_tree_partial_scalar(const float *per_voice, uint32_t voices, size_t frames, uint32_t block_voices, float *out);

/* process.rs:14-49 — one voice, caller-owned state */
int s2o_process_layer_buf_simd(const s2o_layer_cfg *cfg, s2o_layer_state *st, float pitch,
                               uint32_t sample_rate, uint32_t offset, int has_release,
                               uint32_t release_offset, float *buf, size_t len);

/* ---- stage-level entry points (known-answer tests) ---- */
float s2o_note_to_pitch(uint8_t note);                                          /* synth.rs:208-212 */
float s2o_ms_as_samples(float ms, uint32_t sample_rate);                        /* units.rs:44-53 */
void s2o_adsr_x16(float attack, float decay, float sustain, float release,      /* simdtest.rs:270-331 */
                  const uint32_t offset[16], int has_release, uint32_t release_offset, float out[16]);
float s2o_adsr_scalar(float attack, float decay, float sustain, float release,  /* envelopes.rs:21-150 */
                      uint32_t offset, int has_release, uint32_t release_offset);
uint32_t s2o_hash_word(uint32_t start, uint32_t word);                          /* hashnoise.rs:53-55 */
void s2o_hash_word_x16(const uint32_t start[16], const uint32_t word[16], uint32_t out[16]); /* :57-68 */
void s2o_hash_noise_x16(uint32_t seed, const float offset[16], float out[16]);  /* hashnoise.rs:33-51 */
float s2o_hash_noise(uint32_t seed, float offset);                              /* hashnoise.rs:14-27 */
float s2o_table_lookup_exclusive(const float *table, uint32_t len, float value, float range, int *panicked);
float s2o_table_lookup_inclusive(const float *table, uint32_t len, float value, float range, int *panicked);
void s2o_table_lookup_exclusive_x16(const float *table, uint32_t len, const float value[16], const float range[16], float out[16]);
void s2o_table_lookup_inclusive_x16(const float *table, uint32_t len, const float value[16], const float range[16], float out[16]);
void s2o_table_lookup_periodic_x16(const float *table, uint32_t len, const float value[16], const float range[16], float out[16]);
const float *s2o_sin_table(void);                                               /* tables.rs */
float s2o_lpf_process(float *last, uint32_t sample_rate, float freq, float input); /* filters.rs:16-34 */
void s2o_modulate_freq_unipolar_x16(float freq, const float mod[16], float amount, float out[16]); /* process.rs:231-250 */
float s2o_modulate_freq_unipolar(float freq, float mod, float amount);          /* process.rs:221-229 */
float s2o_sleef_powf(float x, float y);                                         /* sleef::Sleef::pow */
/* BUILD-DEFINED 4x oversampling (no counterpart in the reference; BASELINE config [4]): the path is rendered at
 * 4 x the output rate and decimated by a 63-tap Blackman-windowed sinc (cutoff 0.115 cycles per input sample),
 * taps computed in double and rounded to f32, each output = the taps applied in index order, product and sum
 * rounded separately, over input samples 4n-62 .. 4n (62 samples of history precede the block). */
#define S2O_DECIM_TAPS 63
void s2o_decim4_taps(float *h63);
void s2o_decimate4(const float *x_with_history, size_t n_out, const float *h63, float *out);

/* dsp_filters.rs:25-230: one step of the first/second-order filters (kind = S2O_FILT_*);
 * `shape` is damping_factor for LP2/HP2, quality_factor for BP2, unused by LP1/HP1 */
float s2o_dsp_filter_process(int kind, float *x1, float *x2, float *y1, float *y2,
                             uint32_t sample_rate, float cutoff, float shape, float input);

#ifdef __cplusplus
}
#endif
#endif
