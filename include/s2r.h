/*
 * s2r.h — C ABI of libs2r: the MI355X (gfx950) voice-render path behind s2_lib's
 * buffer-fill API.
 *
 * This is the drop-in boundary.  Each entry point names the reference interface it
 * replaces (paths relative to /root/reference/components/s2_lib/src/try3/).  The
 * reference's public surface is `synth::Synth::{new, note_on, note_off, sample}`
 * (synth.rs:53-80,154-169) as used by s2_bin (components/s2_bin/src/main.rs:132-147,
 * 198-205); rust/s2_lib_gpu/src/lib.rs wraps this header back into exactly those
 * signatures, include/s2_synth.hpp does the same for C++.
 *
 * Conventions
 *   - every function returns an s2r_status (0 = ok, < 0 = error); nothing unwinds or
 *     aborts across the boundary (the reference panics instead: process.rs:36,71);
 *   - a handle is NOT thread-safe: one caller thread at a time, like `&mut Synth`;
 *   - s2r_fill OVERWRITES the caller's buffer (synth.rs:201-202) and returns when it is
 *     complete; no allocation happens inside fill;
 *   - note events take effect at the next fill boundary (s2_bin applies MIDI between
 *     `sample` calls: main.rs:140-147);
 *   - the library needs a gfx950 device: s2r_create fails with S2R_ERR_NO_DEVICE
 *     otherwise.  There is no CPU fallback.
 */
#ifndef S2R_H
#define S2R_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Changes when a struct layout or an existing signature changes.  Entry points added since 4.0 without touching either:
 * s2r_set_low_latency, s2r_low_latency_active, s2r_build_id, s2r_set_resident, s2r_resident_active, s2r_quiesce,
 * s2r_exchange_create, s2r_exchange_attach, s2r_voice_pool_set_threads, s2r_voice_pool_resolve. */
#define S2R_ABI_VERSION 4

typedef enum {
    S2R_OK = 0,
    S2R_ERR_INVALID = -1,          /* bad argument / null handle */
    S2R_ERR_NO_DEVICE = -2,        /* no usable gfx950 device */
    S2R_ERR_HIP = -3,              /* a HIP runtime call failed; see s2r_last_error */
    S2R_ERR_PATCH_SYNTAX = -4,     /* .synth2 text rejected */
    S2R_ERR_PATCH_RANGE = -5,      /* value outside its unit's range (units.rs:55-65) */
    S2R_ERR_TOO_MANY_FRAMES = -6,  /* frames > max_frames given at create */
    S2R_ERR_OFFSET_OVERFLOW = -7,  /* a voice's frame offset would pass u32::MAX: the
                                      reference panics here (process.rs:36 "overflow") */
    S2R_ERR_OUT_OF_MEMORY = -8
} s2r_status;

/* static_config.rs:26-32 (declaration order) */
typedef enum { S2R_OSC_SQUARE = 0, S2R_OSC_SAW = 1, S2R_OSC_TRIANGLE = 2, S2R_OSC_SINE = 3,
               /* build-defined alias-suppressed shapes (the reference has only the naive ones above and links to the
                * literature, notes.md:32,79): differentiated polynomial waveforms, DESIGN.md 4.10 gives the op sequence */
               S2R_OSC_DPW_SAW = 4, S2R_OSC_DPW_SQUARE = 5, S2R_OSC_DPW_TRIANGLE = 6 } s2r_osc_kind;

/* static_config.rs:38-44  sc::Adsr  (Ms, Ms, Unipolar<1>, Ms) */
typedef struct { float attack_ms, decay_ms, sustain, release_ms; } s2r_adsr;

/* static_config.rs:4-24  sc::Layer — the patch.  ONE patch is shared by all voices, as in
 * the reference (synth.rs:10). */
typedef struct {
    int32_t osc_kind;              /* s2r_osc_kind          sc::Oscillator.kind */
    float osc_gain;                /* Unipolar<1>           sc::Oscillator.gain */
    float noise;                   /* Unipolar<1>           sc::Layer.noise */
    float lpf_freq;                /* Hz                    sc::LowPassFilter.freq */
    s2r_adsr amp_env;
    s2r_adsr mod_env;
    float mod_env_to_osc_freq;     /* Bipolar<10>           sc::Modulations */
    float mod_env_to_lpf_freq;     /* Bipolar<10> */
    /* Which filter the layer runs at the modulated cutoff.  The reference's live path is the
     * one-pole of filters.rs (S2R_FILT_ONEPOLE, the default); dsp_filters.rs:25-180 holds four
     * more that nothing in the reference calls yet — here they are selectable per patch. */
    int32_t lpf_kind;              /* s2r_filter_kind */
    float lpf_damping;             /* Unipolar<10>          SecondOrder*Filter.damping_factor
                                      (dsp_filters.rs:95 "sqrt(2) is neutral"), LP2/HP2 only */
    float lpf_q;                   /* Unipolar<10>          SecondOrderBandPassFilter.quality_factor
                                      (dsp_filters.rs:194 "3 is neutral"), BP2 only; must be > 0 */
} s2r_patch;

typedef enum {
    S2R_FILT_ONEPOLE = 0,          /* filters.rs:16-34          LowPassFilter */
    S2R_FILT_LP1 = 1,              /* dsp_filters.rs:25-45      FirstOrderLowPassFilter */
    S2R_FILT_HP1 = 2,              /* dsp_filters.rs:60-80      FirstOrderHighPassFilter */
    S2R_FILT_LP2 = 3,              /* dsp_filters.rs:99-130     SecondOrderLowPassFilter */
    S2R_FILT_HP2 = 4,              /* dsp_filters.rs:149-180    SecondOrderHighPassFilter */
    S2R_FILT_BP2 = 5,              /* dsp_filters.rs:199-230    SecondOrderBandPassFilter (center = the cutoff) */
    /* build-defined (the reference only names an SVF, notes.md:63): trapezoidal state-variable
     * filter at the modulated cutoff, resonance lpf_q (> 0); DESIGN.md 4.6 gives the op sequence */
    S2R_FILT_SVF_LP = 6, S2R_FILT_SVF_BP = 7, S2R_FILT_SVF_HP = 8
} s2r_filter_kind;

#define S2R_MAX_DEVICES 16
typedef struct {
    uint32_t struct_size;          /* = sizeof(s2r_config) */
    uint32_t total_voices;         /* size of the voice pool; the reference fixes it at
                                      NUM_VOICES = 8 (synth.rs:7) */
    uint32_t shard_begin;          /* first pool index rendered by THIS handle */
    uint32_t shard_voices;         /* voices rendered by this handle; 0 => total_voices.
                                      Voice allocation always runs over the whole pool, so
                                      every shard's handle must see the same event stream */
    uint32_t max_frames;           /* largest `frames` a fill will be asked for */
    int32_t device;                /* HIP device ordinal; -1 => current device */
    uint32_t block_voices;         /* voices per workgroup: 64..1024, multiple of 64;
                                      0 => 256.  Part of the mix-tree spec */
    uint32_t mix_groups;           /* >= 1: second-level grouping of the block partials so a
                                      1-GPU run reproduces the G-GPU summation order; 0 => 1 */
    uint32_t reserved0;            /* must be 0 */
    /* Round-robin sharding (0 => off: this handle renders the contiguous range above).  G > 0: the
     * pool is dealt out in runs of G consecutive voices to shard_count handles, and this one
     * (shard_index) renders every shard_count-th run — its local voice l is pool voice
     * ((l / G) * shard_count + shard_index) * G + l % G; shard_begin is ignored and shard_voices is
     * total_voices / shard_count.  G is a multiple of 16 that divides block_voices, and
     * total_voices a multiple of G * shard_count.  The allocation policy sweeps the pool in index
     * order (synth.rs:101-120 picks the oldest voice, first index on ties), so contiguous shards take
     * a burst of note-ons one GPU at a time; dealt-out shards share it. */
    uint32_t shard_interleave;
    uint32_t shard_index;
    uint32_t shard_count;
    /* Device list.  n_devices <= 1: the handle renders on `device`.  n_devices = N > 1: ONE handle
     * renders the pool on devices[0..N) — the pool is cut into N shards (contiguous ranges when shard_interleave is
     * 0, else dealt out in runs of shard_interleave voices), shard k lives on devices[k] (an ordinal may repeat), the
     * allocation policy of synth.rs:61-120 runs ONCE per event on the calling thread and the event is routed to the
     * shard that holds the chosen voice (the shards' kernels are launched by the calling thread too, shard after
     * shard, each on its device's stream); every fill each shard leaves its partial mix in row k of a buffer on
     * devices[0] (a peer-to-peer write over xGMI, 4 KiB) and devices[0] adds the rows in shard order rooted at +0.0
     * (synth.rs:176,195) — with contiguous shards the very association one device produces with mix_groups = N.
     * shard_begin / shard_voices / shard_index / shard_count must be 0 (or shard_count 1) then; total_voices must be
     * a multiple of N * block_voices. */
    uint32_t n_devices;
    int32_t devices[S2R_MAX_DEVICES];
} s2r_config;

/* One voice's complete state, for checkpoint/resume and tests.
 * synth.rs:23-30 (Voice) + state.rs:10-21 (st::Layer). */
typedef struct {
    uint8_t note;                  /* Voice.note */
    uint8_t started;               /* current_frame_offset.is_some() */
    uint8_t released;              /* release_frame_offset.is_some() */
    uint8_t program;               /* patch bank index the voice was started with */
    uint32_t current_frame_offset;
    uint32_t release_frame_offset;
    float pitch_hz;                /* note_to_pitch(note), synth.rs:208-212 */
    float phase_accum;             /* OscillatorState (None == 0.0, oscillators.rs:483) */
    float lpf_last;                /* LowPassFilterState.last */
    uint32_t noise_seed;           /* NoiseState.seed (always 0 in the reference, synth.rs:68) */
    float velocity;                /* stored, never used in rendering (synth.rs:18,26) */
    float filt_x1, filt_x2, filt_y1, filt_y2;   /* dsp_filters.rs:12-17,82-89 filter states */
    float osc_z;                   /* DPW oscillators: the differentiator's memory; NaN = none yet (a fresh voice) */
} s2r_voice_state;

typedef struct s2r_synth s2r_synth;

/* Synth::new (synth.rs:54-59): default_config() patch, all voices idle. */
int s2r_create(const s2r_config *cfg, s2r_synth **out);
void s2r_destroy(s2r_synth *s);

/* The `.synth2` patch text (example.synth2:1-3 sketches `synth <ident> { }`; the reference
 * ships no loader — grammar in DESIGN.md).  An empty body == Synth::default_config()
 * (synth.rs:125-152). */
int s2r_load_patch(s2r_synth *s, const char *text, size_t len);
int s2r_set_patch(s2r_synth *s, const s2r_patch *patch);
int s2r_get_patch(const s2r_synth *s, s2r_patch *out);

/* Patch bank ("per-voice patches"; build-defined, the reference has one patch per
 * Synth).  A bank of 1..S2R_MAX_BANK patches; the current program (MIDI program change) selects
 * the patch a note_on gives its voice, and the voice keeps it until it is restarted.
 * s2r_set_patch / s2r_load_patch / s2r_get_patch address patch 0; a fresh handle has a bank of
 * one.  A voice whose program lies past a later, smaller bank renders with patch 0. */
#define S2R_MAX_BANK 256u
int s2r_set_patch_bank(s2r_synth *s, const s2r_patch *patches, uint32_t n);
uint32_t s2r_patch_bank_size(const s2r_synth *s);
int s2r_program_change(s2r_synth *s, uint32_t program);      /* S2R_ERR_INVALID if program >= bank size */
void s2r_default_patch(s2r_patch *out);                      /* synth.rs:125-152 */

/* Synth::note_on(Note, Velocity) (synth.rs:61-70) incl. next_voice (synth.rs:101-120).
 * Optionally reports the chosen pool index. */
int s2r_note_on(s2r_synth *s, uint8_t note, float velocity);
int s2r_note_on_ex(s2r_synth *s, uint8_t note, float velocity, uint32_t *voice_index_out);
/* Synth::note_off(Note) (synth.rs:72-96): last active voice holding `note`. */
int s2r_note_off(s2r_synth *s, uint8_t note);

/* A batch of note_on / note_off calls applied in order — what s2_bin's
 * apply_all_midi_messages loop does between two sample() calls (main.rs:170-187), in one
 * crossing of the boundary.
 *
 * `frame` = 0: the event takes effect before the next fill (like s2r_note_on/off).
 * `frame` = a multiple of 16 below the next fill's length: the event takes effect INSIDE the
 * next fill at that frame, exactly as if the caller had split the fill there — s2_bin's
 * apply-MIDI-every-16-frames loop (main.rs:138-143) reproduced inside one launch.  Events
 * must be submitted in non-decreasing frame order. */
typedef enum { S2R_NOTE_OFF = 0, S2R_NOTE_ON = 1,
               S2R_PROGRAM_CHANGE = 2   /* `note` = patch bank index for the note_ons that follow */
} s2r_note_kind;
typedef struct { uint8_t kind; uint8_t note; uint16_t frame; float velocity; } s2r_note_event;
int s2r_note_events(s2r_synth *s, const s2r_note_event *events, size_t n);

/* Synth::sample(&mut [f32], SampleRateKhz) (synth.rs:154-169).  `sample_rate_hz` is what
 * the reference calls SampleRateKhz but holds Hz (units.rs:14).  Full 16-frame chunks take
 * the x16 code path, a tail of frames % 16 the scalar path, restarting per call
 * (synth.rs:158, process.rs:25-48).  Output = the mix of this handle's shard, root-added to
 * +0.0 (synth.rs:176). */
int s2r_fill(s2r_synth *s, float *mono_out, size_t frames, uint32_t sample_rate_hz);
/* s2r_fill in two halves, for a caller that keeps TWO buffers in flight the way s2_bin does (its synth thread fills one
 * buffer while the audio thread plays the other: audio_player.rs:56-60 pre-sends two, main.rs:135-149 refills whichever
 * comes back).  s2r_fill_begin applies the events handed over so far and queues the fill; s2r_fill_end waits for the
 * OLDEST fill begun and not yet ended and copies its `frames` samples to mono_out.  At most two fills may be in flight;
 * between a begin and its end the caller may hand over the next buffer's events and begin that fill.  s2r_fill is
 * begin + end.  (How a begun fill is launched is the library's business and bit-neutral: ONE handle per device and process —
 * the first created whose grid is at most one workgroup per compute unit — runs its render kernels on a stream of their own
 * beside a second stream's chain heads and mixes, which wait for each other's workgroups inside the kernels and therefore
 * need the device's compute units to themselves; every other handle takes one launch per fill in which no kernel waits for
 * another.  A fill that fails on the device comes back from s2r_fill_end as an error ONCE, its buffer zeroed, and the handle
 * refuses events and fills from then on: DESIGN.md 4.2.) */
int s2r_fill_begin(s2r_synth *s, size_t frames, uint32_t sample_rate_hz);
/* `capacity` = floats `mono_out` can take: S2R_ERR_INVALID (and nothing is consumed) when it is smaller than the
 * `frames` the oldest fill was begun with — s2r_fill_pending_frames says how many that is (0: none in flight). */
int s2r_fill_end(s2r_synth *s, float *mono_out, size_t capacity);
size_t s2r_fill_pending_frames(const s2r_synth *s);
uint32_t s2r_fills_in_flight(const s2r_synth *s);
/* The audio callback's mono -> every channel copy (s2_bin/src/audio_player.rs:224-228),
 * done on device: interleaved L,R with L == R. */
int s2r_fill_stereo(s2r_synth *s, float *interleaved_lr_out, size_t frames, uint32_t sample_rate_hz);

/* BUILD-DEFINED 4x oversampling (the reference has none; BASELINE config [4]): renders 4 * frames at
 * 4 * sample_rate_hz through the same path and decimates the mix by a 63-tap windowed sinc whose history
 * carries over from call to call (DESIGN.md 4.9 gives taps and arithmetic).  4 * frames must not exceed
 * max_frames.  Voices' offsets, envelopes and filters all run at the oversampled rate. */
#define S2R_OVERSAMPLE 4u
int s2r_fill_oversampled(s2r_synth *s, float *mono_out, size_t frames, uint32_t sample_rate_hz);

/* Multi-GPU building block: renders this shard and leaves its PARTIAL mix (no root add) in
 * `dev_partial_out` (device memory, `frames` floats) on `hip_stream` (a hipStream_t, may
 * be NULL) without synchronising.  Partials of all shards are then combined in rank order
 * by s2r_sum_partials_device. */
int s2r_fill_device(s2r_synth *s, float *dev_partial_out, size_t frames, uint32_t sample_rate_hz, void *hip_stream);
/* Single-shard form of the above: the FINAL mix (root-added, exactly what s2r_fill returns)
 * left in device memory on `hip_stream`, no synchronisation. */
int s2r_fill_device_root(s2r_synth *s, float *dev_out, size_t frames, uint32_t sample_rate_hz, void *hip_stream);
/* out[i] = ((+0.0 + rows[0][i]) + rows[1][i]) + ...   rows is [n_rows][frames] on device. */
int s2r_sum_partials_device(const float *dev_rows, uint32_t n_rows, size_t frames, float *dev_out, void *hip_stream);

/* Mix disabled: every shard voice's frames, host array [shard_voices][frames] (rows of idle
 * voices are +0.0).  Advances state exactly like s2r_fill.  process::process_layer_buf_simd
 * per voice (process.rs:14-49). */
int s2r_render_voices(s2r_synth *s, float *per_voice_out, size_t frames, uint32_t sample_rate_hz);

/* The reference's lower-level public entry for callers that keep their own st::Layer:
 *   process::process_layer_buf_simd(&sc::Layer, &mut st::Layer, Hz, SampleRateKhz, offset: u32, release_offset: Option<u32>, &mut [f32])
 * (process.rs:14-49, pub through try3/mod.rs) — whole 16-frame chunks through process_layer_x16, the remainder through the
 * scalar path, the state advanced in place, the offset passed by value — for n_layers independent layers side by side.
 * static_config = the handle's patch (s2r_set_patch / s2r_load_patch; `program` picks a bank entry); bufs is [n_layers][frames].
 * The handle is the workspace: its first n_layers voices are replaced by the layers (at offset + frames afterwards), the
 * rest go idle — give the calls a handle of their own, sized to the batch (one device, n_layers <= total_voices).
 * S2R_ERR_OFFSET_OVERFLOW where the reference panics (process.rs:36). */
typedef struct {
    float pitch_hz;                /* Hz */
    uint32_t offset;               /* frames since the layer's note_on; by value: the caller adds `frames` (synth.rs:197) */
    uint32_t release_offset;       /* valid when has_release */
    uint8_t has_release;           /* Option<u32>::is_some() */
    uint8_t program;               /* patch bank index (0 without a bank) */
    uint8_t _pad[2];
    /* st::Layer (state.rs:10-21), updated in place */
    float phase_accum;             /* OscillatorState */
    float lpf_last;                /* LowPassFilterState.last */
    uint32_t noise_seed;           /* NoiseState.seed */
    float filt_x1, filt_x2, filt_y1, filt_y2;   /* dsp_filters.rs kinds / SVF */
    float osc_z;                   /* DPW oscillators (NaN: none yet) */
} s2r_layer_call;
int s2r_process_layers(s2r_synth *s, s2r_layer_call *layers, uint32_t n_layers, float *bufs, size_t frames, uint32_t sample_rate_hz);

/* Checkpoint / resume and test access; `voices` has shard_voices entries. */
int s2r_export_state(s2r_synth *s, s2r_voice_state *voices);
int s2r_import_state(s2r_synth *s, const s2r_voice_state *voices);
/* NoiseState.seed of one pool voice ("todo don't default this", state.rs:19). */
int s2r_set_noise_seed(s2r_synth *s, uint32_t voice_index, uint32_t seed);

/* Synth::next_voice's `log::debug!("using new voice index {} for note {}", …)` (synth.rs:118) as a callback: called once
 * per note_on (s2r_note_on, s2r_note_on_ex, every note_on of s2r_note_events, in event order) with the voice the allocation
 * policy chose.  NULL switches it off (the default).  Not on the shards of a device list (their parent's pool decides). */
typedef void (*s2r_voice_log_fn)(void *user, uint32_t voice_index, uint8_t note);
int s2r_set_voice_log(s2r_synth *s, s2r_voice_log_fn fn, void *user);

/* Introspection */
uint32_t s2r_abi_version(void);
/* Which sources this binary was built from: "<sha256 over synth2_amd/csrc, 16 hex>-<include/s2r.h + compiler flags, 8 hex>",
 * compiled in by synth2_amd/build.py.  A loader that has the sources at hand (synth2_amd.load_library, bench.py) compares it
 * with their hash and refuses — or rebuilds — a binary that does not match, whatever the files' times say; bench.py prints it
 * beside the profile's source hash.  (No counterpart in the reference: cargo rebuilds s2_lib from source.) */
const char *s2r_build_id(void);
uint32_t s2r_shard_voices(const s2r_synth *s);
uint32_t s2r_block_voices(const s2r_synth *s);
uint32_t s2r_device_count(const s2r_synth *s);                  /* 1, or the N of a device list */
/* note_offs that found no active (started, unreleased) voice holding the note and therefore did nothing.  The
 * reference ignores them silently: its `log::warn!("note {} released twice")` (synth.rs:77) sits behind
 * find_active_voice, which only returns unreleased voices (synth.rs:36-38,82-90), so it can never fire — this
 * counter is the diagnostic that warning was meant to be. */
uint64_t s2r_double_release_count(const s2r_synth *s);
/* Low-latency fills for the reference's own call pattern — a small pool rendered 16 frames at a time from the audio
 * callback (synth.rs:154-203 called per 16 frames, s2_bin/src/main.rs:138-147, audio_player.rs:56-60).  Enabled, a handle
 * whose shard is ONE workgroup (at most 256 voices, block_voices permitting) with a single one-pole patch keeps a resident
 * render kernel on the device between s2r_fill / s2r_fill_stereo calls: a fill is then a command written to mapped host
 * memory and a completion word (for fills of up to 64 frames: the tags of the frames themselves) polled, not a kernel
 * launch.  The kernel leaves by itself after 1 ms
 * without a fill (the next fill starts it again) and is stopped by every other entry point that touches the device or
 * the patch; fills it cannot take (timed events, more than 9 note events since the last fill, seed overrides) go the
 * ordinary way.  Same bits either way.  While it runs it occupies one compute unit and the handle's stream.
 * s2r_low_latency_active: 1 while the resident kernel is believed to be on the device. */
int s2r_set_low_latency(s2r_synth *s, int enabled);
int s2r_low_latency_active(const s2r_synth *s);
/* The same idea for the THROUGHPUT path: keep the shard's whole render grid on the device between fills.
 * Enabled, a handle (or every shard of a device-list handle) whose grid is at most one workgroup of at most 256 voices per
 * compute unit — any patch, any patch bank — renders the fills of s2r_fill / s2r_fill_stereo / s2r_fill_begin through a
 * POOL-RESIDENT kernel: a fill is a 64-byte command plus the fill's note events grouped by workgroup, written to memory the
 * kernel polls — no launch; every workgroup builds its own voices' event chains, renders, and the last ones to finish add
 * the rows up and end the fill (one workgroup per fill, for a device list, also adds the shards' rows).  With two fills in
 * flight a workgroup that is done early starts the next fill while the slowest still finish this one.  The kernel leaves by
 * itself after 2 ms without a fill (the next fill starts it again) and is stopped by every entry point that touches the
 * device, the patch or a knob, and by s2r_quiesce; fills it cannot take (per-voice rows, caller-owned streams, timing
 * on, the 4x-oversampled fill) go the ordinary way.  Same bits either way.  While it runs it holds the handle's stream
 * and one workgroup slot per 256 voices: enable it for ONE handle per device, on a device the caller does not share.
 * A single-workgroup handle gets s2r_set_low_latency's kernel.  (The reference's Synth is a value on the audio thread's
 * stack, synth.rs:9-12: it is always "resident".)
 * s2r_resident_active: 1 while a resident kernel of either kind is believed to be on the device.
 * s2r_quiesce: stops any resident kernel of the handle and waits for it — what a caller does before it synchronises the
 * whole device (hipDeviceSynchronize would otherwise wait out the kernel's patience). */
/* One process per GPU WITHOUT a collective library in the step (SURVEY 8e's preferred shape): the ranks' partial rows are
 * written into one block of the root's device memory — mapped by the other ranks through an IPC handle; a peer-to-peer store
 * over xGMI where the ranks' devices differ — and added in rank order from +0.0 (synth.rs:176,195) by the root's last
 * workgroup, all inside the render kernels.  Rank 0 calls s2r_exchange_create on its shard's handle and hands the 64 handle
 * bytes to the other ranks by whatever channel the caller has (once, at start-up); they call s2r_exchange_attach.  From
 * then on every rank drives its handle with the SAME events and the same s2r_fill / s2r_fill_begin / s2r_fill_end calls;
 * the root's buffers receive the mix, the other ranks' buffers silence.  Bit for bit what one device returns with
 * mix_groups = n_ranks (contiguous shards).  Shards of more than one workgroup; any patch or patch bank. */
#define S2R_EXCHANGE_HANDLE_BYTES 64
int s2r_exchange_create(s2r_synth *s, uint32_t n_ranks, void *handle_out, size_t handle_bytes);
int s2r_exchange_attach(s2r_synth *s, uint32_t rank, uint32_t n_ranks, const void *handle, size_t handle_bytes);
int s2r_set_resident(s2r_synth *s, int enabled);
int s2r_resident_active(const s2r_synth *s);
int s2r_quiesce(s2r_synth *s);
/* device time of the most recent fill's render kernel in milliseconds (HIP events recorded
 * on the library's stream around the launch); < 0 if timing is off.  Enable with
 * s2r_set_timing(s, 1): adds two event records per fill. */
int s2r_set_timing(s2r_synth *s, int enabled);
/* Measurement knob (default on): while the mod envelope of every voice of a wavefront is in a
 * flat stage the LPF coefficient is reused instead of recomputed — same bits either way.
 * Turning it off makes every frame pay the full pow/exp chain (bench.py's
 * `value_all_voices_modulating` leg). */
int s2r_set_flat_shortcut(s2r_synth *s, int enabled);
/* Measurement knob (default 1): where a patch's filter coefficients come from while a mod envelope moves.
 * 0: computed in-lane per frame; 1: read from the patch's coefficient tables, and a fill with few
 * untimed events carries them in the render kernel's arguments; 2: tables, but note events always through their
 * own launch; 3 / 4: synonyms of 1 / 2 (earlier rounds' test matrices).  Same bits in every mode. */
int s2r_set_coeff_stream(s2r_synth *s, int enabled);
float s2r_last_render_ms(s2r_synth *s);
const char *s2r_last_error(const s2r_synth *s);             /* never NULL */
const char *s2r_status_string(int status);

/* ---- host-only helpers: usable without a device (front-ends that route events to shards,
 * CPU-only tests of the host logic) ---- */

/* The wire format of the reference's (disabled) websocket audio server, one text frame per buffer:
 * serde_json::to_string(&Vec<f32>) (threads.rs:303-305; BUFFER_SIZE = 4096 frames, 32 kHz mono,
 * threads.rs:6,263; consumer www/streamer.js:82-90).  Writes a NUL-terminated JSON array into `out`
 * and returns its length; when `out` is NULL or `cap` is below the worst case (3 + S2R_STREAM_CHARS_PER_SAMPLE * n:
 * the longest element, e.g. "-0.0000012345678", is 16 chars + ','; one spare) it writes nothing and returns the capacity to provide.  Host-only. */
#define S2R_STREAM_CHARS_PER_SAMPLE 18u
#define S2R_STREAM_FRAMES 4096u
#define S2R_STREAM_RATE_HZ 32000u
size_t s2r_stream_frame_json(const float *samples, size_t n, char *out, size_t cap);

/* The .synth2 parser on its own; err_buf (may be NULL) receives a message on failure. */
int s2r_parse_patch_text(const char *text, size_t len, s2r_patch *out, char *err_buf, size_t err_cap);

/* The voice-allocation / release policy of Synth (synth.rs:61-120) for a pool of any size,
 * O(1) per event, without rendering.  Offsets advance by s2r_voice_pool_advance. */
typedef struct s2r_voice_pool s2r_voice_pool;
s2r_voice_pool *s2r_voice_pool_create(uint32_t total_voices);
void s2r_voice_pool_destroy(s2r_voice_pool *p);
uint32_t s2r_voice_pool_note_on(s2r_voice_pool *p, uint8_t note, float velocity);  /* returns the chosen index */
int64_t s2r_voice_pool_note_off(s2r_voice_pool *p, uint8_t note);                  /* released index or -1 */
void s2r_voice_pool_advance(s2r_voice_pool *p, uint64_t frames);
uint32_t s2r_voice_pool_next_voice(const s2r_voice_pool *p);                       /* synth.rs:101-120 */
/* The batch form s2r_note_events runs: the whole array at once — what a loop of s2r_voice_pool_advance (to each event's
 * frame) / _note_on / _note_off computes, with the same results, voice_out[k] = the voice event k takes or releases (-1: a
 * note_off that found none, a program change).  `frames_moved`: how far the clock already is inside the fill the events belong
 * to; returns the last event's frame.  s2r_voice_pool_set_threads: batches of at least `batch_threshold` events are resolved
 * by the calling thread (the queue: synth.rs:101-120 depends on earlier note_ons alone) plus `worker_threads` threads that
 * share the notes among them (synth.rs:82-90 depends on one note's events alone); 0 = the calling thread alone. */
void s2r_voice_pool_set_threads(s2r_voice_pool *p, uint32_t worker_threads, size_t batch_threshold);
uint32_t s2r_voice_pool_resolve(s2r_voice_pool *p, const s2r_note_event *events, size_t n, uint32_t frames_moved, int64_t *voice_out);
int s2r_voice_pool_query(const s2r_voice_pool *p, uint32_t voice_index, s2r_voice_state *out);

#ifdef __cplusplus
}
#endif
#endif /* S2R_H */
