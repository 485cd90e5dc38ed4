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
    float *osc_z;         // DPW oscillators: the differentiator's memory F(s[n-1]); NaN in a voice that has not rendered a frame
};
#define S2R_VOICE_WORDS 13
#define S2R_OSC_Z_NONE 0x7fc00000u   // the bits of that NaN

// One patch of the bank, resolved for the fill's sample rate (what S2rRenderParams carries for
// the single-patch kernels).
struct S2rBankEntry {
    int32_t osc_kind;
    float osc_gain, noise_level, lpf_freq, amt_osc, amt_lpf;
    int32_t lpf_kind;
    float lpf_shape;      // damping_factor (LP2/HP2) or quality_factor (BP2)
    S2rEnv amp, mod;
    // this patch's coefficient tables (S2rTabRef with `base` = the bank's table buffer + tab_off floats); always four
    // planes: c0, c1, c2 (x, 1 - x, - for the one-pole; alpha, beta, gamma otherwise), pow(2, mod * amt_osc)
    uint32_t tab_valid;   // 0: envelopes too long to tabulate — the voices of this patch compute in-lane
    uint32_t tab_off, tab_plane;
    int32_t tab_ad, tab_rc;
    uint32_t tab_rc_t0;
    int32_t tab_ru, tab_sus, tab_end, tab_dead;
};

struct S2rTimedEvent;

// How the LAST kernel of a fill tells the host that the output (in mapped host memory) is complete: the last workgroup
// to finish (a counter in device memory, reset by that workgroup) stores `value` into a word of mapped host memory the
// host polls — a few hundred nanoseconds after the data, where waiting on a HIP event costs the caller tens of
// microseconds between the GPU finishing and the wait returning.  flag == nullptr: nothing is signalled.
struct S2rDone {
    uint32_t *flag;          // mapped host memory (device view)
    uint32_t value;
    uint32_t *counter;       // device memory, 0 between launches
};

// Coefficient tables of ONE patch at one sample rate.  The x16 ADSR (old/simdtest.rs:270-331) makes
// the mod envelope's value a function of the frame offset t alone while the voice is in its attack or decay stage
// (the release offset is clamped to attack + decay, simdtest.rs:283), of t alone in a release that starts at that
// clamp, and of d = t - release_offset alone in a later release; in sustain and after the end it is a constant.  So
// the chain behind it — sleef pow -> * freq -> filter coefficients (process.rs:146-152,231-250; filters.rs:20-24 or
// dsp_filters.rs) — is tabulated once per (patch, sample rate) by s2r_table_kernel with the very routines the render
// kernels would run per voice and frame, and a voice reads its coefficients at base[idx + (offset & mask)].
// Planes (each `plane` floats apart): one-pole: x = exp(-2 pi f / sr), 1 - x; dsp_filters.rs kinds / SVF: alpha,
// beta, gamma; then, under oscillator FM, pow(2, mod * mod_env_to_osc_freq).
// Valid for offsets below 2^24 (where (offset as f32) is exact); older voices compute in-lane.
struct S2rTabRef {
    const float *base;    // plane 0, entry 0; nullptr: no tables (too long an envelope, or switched off)
    uint32_t plane;       // floats per plane
    int32_t ad;           // attack + decay:      entry ad + t,              0 <= t < attack + decay
    int32_t rc;           // release at the clamp: entry rc + (t - rc_t0),   rc_t0 = ceil(attack + decay)
    uint32_t rc_t0;
    int32_t ru;           // later release:       entry ru + (t - release_frame_offset)
    int32_t sus;          // 16 equal entries: the sustain stage's constant
    int32_t end;          // 16 equal entries: after the release
    int32_t dead;         // 16 equal entries: x = 1, 1 - x = 0 — what makes a lane without a started voice put +0.0
                          // into the mix without a select (one-pole kernel)
    uint32_t fm_plane;    // index of the pow(2, mod * mod_env_to_osc_freq) plane (2 or 3), 0 = none
};
#define S2R_TAB_PAD 16u          // entries a 16-frame chunk may read past a region's last used one
#define S2R_TAB_MAX_ENTRIES (1u << 22)

// ONE launch per fill.  The render kernel does the two small jobs that used to be launches of their own:
//   * chain heads: the host hands over the fill's timed-event records GROUPED BY WORKGROUP (mapped host memory) with the
//     groups' bounds; every workgroup copies its own slice into HBM and publishes its own voices' chain heads before it
//     loads its state — nobody waits for anybody;
//   * the mix ("ticket mix"): a workgroup that has written its partial row counts itself in on `arrive`; the LAST
//     `n_mixers` to arrive wait for the very last one (bounded: they are the fill's slowest workgroups, and every workgroup
//     they wait for is already running) and add the rows up in the documented order, 16-frame blocks dealt
//     out among them; the last mixer to finish ends the fill: the completion word, or — a shard of a device list / of a
//     group of processes — one more count on `rows_done` in the root's memory: the shard that counts in LAST adds the
//     shards' rows in shard order from +0.0 (synth.rs:176,195) and ends the fill.  No wait anywhere in that.
struct S2rMixTail {
    uint32_t n_mixers;            // 0: no in-kernel mix (the rows are left for s2r_mix_kernel)
    uint32_t n_blocks, blocks_per_group, n_groups;
    int32_t root_add, stereo;
    float *out;                   // the fill's output (root_add) or this shard's partial row
    S2rDone done;                 // counter: the mixers' own arrival; flag/value: the completion word (no exchange)
    // exchange of partial rows between shards (device list, one process per GPU): nullptr = none
    uint32_t *rows_done;          // counter in the root's memory (system scope)
    uint32_t rows_target;         // value it has when every shard has counted in for this fill
    int32_t xmode;                // 0: the shard that counts in last adds the rows; 1: this shard is the ROOT of a group of
                                  // processes: it waits for every rank and adds the rows; 2: a rank other than the root
    uint32_t n_rows, row_stride;
    const float *rows;            // [n_rows][row_stride], the root's
    float *final_out;
    S2rDone final_done;
    int32_t final_stereo;
    unsigned long long *granules; // != nullptr: the fill's output goes out as tagged 8-byte words instead (mono; no completion word)
    uint32_t granule_tag;
};

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
    int32_t no_flat_shortcut;  // 1 => always recompute the filter coefficient per frame, in-lane (measurement knob)
    uint32_t frames;      // this fill
    uint32_t n_voices;    // shard voices
    uint32_t frames_stride; // row stride of block_partials / per_voice (== max_frames or frames)
    uint32_t super_frames;  // frames between two cross-wave combines (set by the launcher)
    S2rVoiceArrays v;
    float *block_partials;   // [n_blocks][frames_stride]
    float *direct_out;       // single-workgroup shard with the root add: the final mix, written by the render
                             // kernel itself ((+0.0) + the block's sum, what s2r_mix_kernel computes for one row)
    int32_t direct_stereo;   // interleaved L,R
    S2rDone done;            // with direct_out: told to the host when the output is written
    float *per_voice;        // [n_voices][frames] or nullptr (mix-disabled debug/parity path)
    const float *sin_table;  // 1024 floats (tables.rs)
    const float *noise_tab;  // 65536 floats: the x16 noise (hashnoise.rs:33-51) of hashed word x, seed and offset folded into x
                             // (s2r_noise_table_kernel); nullptr: computed per frame
    // coefficient tables: everything of a frame that depends on the mod envelope alone, as a
    // function of the envelope's own clock — shared by every voice that plays this patch
    S2rTabRef tab;
    // timed events of this fill (nullptr: none)
    const S2rTimedEvent *tev;
    int32_t *voice_ev_head;     // [padded voices] index of the voice's first timed event, -1 = none
    // diagnostic builds only (-DS2R_STAMPS, tools/stamps.py): [waves][16] s_memtime stamps of the render kernel's phases
    unsigned long long *stamps;
    // ... and (tools/gpu_timeline.py) [launches][2] s_memrealtime of this launch's first entry and last exit, slot `tl_slot`
    unsigned long long *timeline;
    uint32_t tl_slot;
    // patch bank (bank_size > 1: s2r_render_general_kernel<ANY, true>; the fields above then hold patch 0)
    const S2rBankEntry *bank;
    uint32_t bank_size;
    // Fills in flight on two streams (S2rOverlap below; the one-pole kernel): this launch waits for its chain heads, built
    // by a kernel on the other stream, and counts its workgroups in for the mix that waits there.  nullptr: not used.
    const uint32_t *ov_heads_counter;
    uint32_t ov_heads_target;
    uint32_t *ov_render_counter;
    uint32_t *ov_fail;
    // One launch per fill (S2rMixTail above): arrive != nullptr
    const S2rTimedEvent *tev_src;    // the fill's records, grouped by workgroup (mapped host memory, device view)
    S2rTimedEvent *tev_copy;         // ... and where the workgroups put them in HBM (== tev)
    const uint32_t *slices;          // workgroup b's records are [slices[b], slices[b + 1]); nullptr: S2rRenderArgs.ev holds the bounds
    uint32_t *arrive;                // rows counter of this fill's parity (device memory; only grows)
    uint32_t arrive_target;          // its value when every workgroup has written its row
    S2rMixTail mt;
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

// The fill's folded note events ride in the render kernel's own arguments when there are few of them (no trip to
// host memory, no launch of their own): every wave picks the records that hit its 64 voices and applies them before it
// loads its state.
#define S2R_ARG_MAX_EVENTS 288u
struct S2rRenderArgs {
    S2rRenderParams p;
    uint32_t n_events;
    uint32_t _pad;
    uint32_t ev[3 * S2R_ARG_MAX_EVENTS];   // voice, S2rVoiceEvent.flags, pitch bits
};
static_assert(sizeof(S2rRenderArgs) <= 4096, "kernel arguments are limited to 4 KiB");

// The resident render kernel (s2r_resident_kernel; s2r_set_low_latency): ONE workgroup that stays on the device between
// fills, polls a command in mapped host memory, renders the fill it describes (the one-pole kernel's fill, state in HBM
// between fills as ever) straight into mapped host memory and reports through the fill's completion word — a 16-frame
// fill then costs two trips over the host link and a few microseconds of kernel, not a launch.
//   command: 32 words in two 64-byte lines, read by one wave with one load per lane.  The host writes the payload, then
//   word 31, then word 0 (x86 stores become visible in program order): a reader that finds word 0 == word 31 == a new
//   sequence number has the whole command.
#define S2R_RES_CMD_WORDS 32u
#define S2R_RES_MAX_EVENTS 9u          // 4 header words + 9 * 3 + the closing sequence word
#define S2R_RES_FLAG_EXIT 1u
#define S2R_RES_GRANULE_FRAMES 64u     // fills up to this length come back as granules (FillCtl, s2r_kern_common.h)
struct S2rResident {
    const uint32_t *cmd;         // [32], device address of the mapped host command:
                                 //   [0] seq  [1] frames | flags << 16  [2] n_events  [3] the completion word's value
                                 //   [4 .. 30] events (voice, flags, pitch bits)  [31] seq
    uint32_t *exited;            // mapped host word: the kernel stores `launch_id` there as its last act
    uint32_t launch_id;
    uint32_t first_seq;          // the first command's sequence number (everything before it is stale)
    uint32_t idle_ticks;         // leaves after this many 100 MHz ticks without a command ...
    uint32_t max_polls;          // ... or this many polls, whichever comes first: the grid always drains
    uint32_t *done_flag, *done_counter;   // the fills' completion word (S2rDone; the command carries the value)
    unsigned long long *granules;         // [S2R_RES_GRANULE_FRAMES] mapped host memory: short fills' output (tag = the command's completion value)
};

// The POOL-RESIDENT render kernel (s2r_pool_kernel; s2r_set_resident): the whole grid of a shard — at most one workgroup
// per compute unit — stays on the device between fills.  The host posts a fill as a COMMAND (one 64-byte line in a ring of
// S2R_POOL_CMD_SLOTS, in fine-grained device memory behind the BAR or in mapped host memory), the fill's records grouped by
// workgroup and the groups' bounds; every workgroup renders the fill with the code a launch per fill runs (one launch per
// fill's form: S2rMixTail), then looks for the next command — a workgroup that is done early starts the next fill while
// the slowest ones still finish this one, so a fill lasts as long as the AVERAGE workgroup, not the slowest, once two are in
// flight.  No launch, no kernel boundary, no launch latency per fill.
//   command words: [0] seq  [1] frames | flags << 16  [2] records in this fill  [3] the completion word's value
//                  [4] output: 0, 1 the ring slots, 2 the synchronous buffer; | 256: stereo  [5] event slot (tev_src index)
//                  [6] parity  [7] arrive_target  [8] n_mixers  [9] rows_target  [10] rows slot  [11] heads target (two streams)
//                  [12] 1: the output as granules (tag = [3])  [15] seq
//   Whether command `s` runs is decided ONCE for the whole grid, by whoever gets there first: word `decided` in device
//   memory holds 2 * seq + bail of the last command decided; a workgroup that sees command s = last + 1 complete moves it
//   2 * last -> 2 * s (run) and one whose patience has run out moves it -> 2 * s + 1 (everybody leaves: the host finds
//   `exited` == launch_id, waits for the stream and starts the kernel again in front of the commands that were not run).
//   Every wait is bounded in ticks and in polls: the grid always drains.
#define S2R_POOL_CMD_SLOTS 4u
#define S2R_POOL_CMD_WORDS 16u
#define S2R_POOL_FLAG_EXIT 1u
#define S2R_POOL_FLAG_TWO_STREAMS 2u   // the fill's chain heads and mix are a launch on the other stream (S2rOverlapWords); [11] the heads' target
struct S2rPool {
    const uint32_t *cmd;              // [S2R_POOL_CMD_SLOTS][16]
    const uint32_t *slices;           // [S2R_POOL_CMD_SLOTS][n_blocks + 1], mapped host memory
    uint32_t slices_stride;
    const S2rTimedEvent *tev_src[4];  // the event slots' records (mapped host memory, device view)
    S2rTimedEvent *tev_copy[2];       // by parity
    int32_t *heads[2];
    float *partials[2];
    uint32_t *arrive;                 // [2] by parity
    const uint32_t *ov_heads;         // S2rOverlapWords.heads_done[2], .render_done[2] (two-stream fills)
    uint32_t *ov_render;
    float *out[3];                    // ring slot 0, ring slot 1, the synchronous buffer (mapped host memory) — or this shard's rows
    uint32_t *done_flag, *done_counter;   // [3] each, as `out`
    uint32_t *decided;                // device memory
    uint32_t *exited;                 // mapped host memory
    uint32_t *fail;
    uint32_t launch_id, first_seq, idle_ticks, max_polls;
    S2rMixTail mt;                    // what does not change from fill to fill (n_blocks, groups, exchange geometry)
    float *rows[2], *rows_mine[2];    // exchange: the root's rows by rows slot, and this shard's row among them; the ROOT's outputs:
    float *final_out[3];
    uint32_t *final_flag, *final_counter;
    unsigned long long *granules;     // [S2R_RES_GRANULE_FRAMES] mapped host memory: synchronous fills of up to that many frames
};

// what s2r_table_kernel needs: the patch resolved for a sample rate and where the planes go
struct S2rTabBuild {
    S2rEnv mod;
    float lpf_freq, amt_lpf, amt_osc, sr, rcp_sr;
    int32_t fast_div_sr;
    int32_t lpf_kind;
    float lpf_damping;
    float *base;
    uint32_t plane, n_ad, n_rel, n_entries;   // region lengths incl. padding; entries per plane
    uint32_t rc_t0;
    uint32_t fm_plane;
};

// Two fills in flight on TWO streams (s2r_fill_begin / s2r_fill_end).  Between the render kernels of
// consecutive fills the stream used to run the previous fill's mix and this fill's chain heads (3 us and two kernel
// boundaries per step).  Both now run on a second stream, beside the render kernels, and what orders them is in device
// memory: a counter of workgroups per buffer parity that have finished the chain heads (the render kernel of that fill
// waits for it before it looks at its voices' chains) and one of render workgroups that have written their partial row
// (the mix waits for it).  Counters only grow; the host hands every waiter the value to wait for.  Every wait is
// bounded (50 ms of the device clock): a waiter that gives up raises `fail` (mapped host memory) and goes on, and the
// host reports the fill as failed — nothing spins for ever.  Heads, event copies and partial rows are double-buffered by
// the fill's parity (at most two fills are in flight).
struct S2rOverlapWords { uint32_t render_done[2], heads_done[2]; };

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
    S2rDone done;                 // told to the host when `out` (mapped host memory) is complete
    const uint32_t *ov_render_counter;   // S2rOverlap: wait until *counter reaches ov_render_target before reading the rows
    uint32_t ov_render_target;
    uint32_t *ov_heads_counter;   // ... and (s2r_mix_and_heads_kernel) every chain-heads workgroup counts itself in here
    uint32_t *ov_fail;
    unsigned long long *timeline; // diagnostic builds only: as S2rRenderParams.timeline
    uint32_t tl_slot;
    unsigned long long *granules; // short fills of the pool-resident kernel: the output as tagged 8-byte words (FillCtl.granules)
    uint32_t granule_tag;
};

hipError_t s2r_launch_tables(const S2rTabBuild &b, hipStream_t stream);
hipError_t s2r_launch_noise_table(float *table_65536, hipStream_t stream);
// one-pole single-patch handles of one workgroup only (a.p.direct_out set, a.p.frames = the longest fill): false otherwise
hipError_t s2r_launch_resident(const S2rRenderArgs &a, const S2rResident &rs, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_render(const S2rRenderArgs &a, uint32_t block_voices, hipStream_t stream);
// the pool-resident kernel of a one-pole single-patch shard (a.p.frames = the longest fill)
hipError_t s2r_launch_pool(const S2rRenderArgs &a, const S2rPool &pl, uint32_t block_voices, hipStream_t stream);
hipError_t s2r_launch_mix(const S2rMixParams &p, hipStream_t stream);
hipError_t s2r_launch_events(const S2rVoiceArrays &v, const S2rVoiceEvent *dev_events, uint32_t n, hipStream_t stream);
hipError_t s2r_launch_mix_and_heads(const S2rMixParams &m, int32_t *heads, const S2rTimedEvent *tev, S2rTimedEvent *tev_copy, uint32_t n,
                                    hipStream_t stream);
hipError_t s2r_launch_tev_heads(int32_t *heads, const S2rTimedEvent *tev, S2rTimedEvent *tev_copy, uint32_t n, hipStream_t stream,
                                uint32_t *ov_heads_counter = nullptr);
// build-defined 4x decimator: x = 62 samples of history + 4 * n_out new ones, h = 63 taps (device), and the
// last 62 inputs copied to the front of x afterwards (second launch) for the next call
hipError_t s2r_launch_decimate4(float *x_with_history, const float *taps, uint32_t n_out, float *out, hipStream_t stream);
// out[i] = ((+0.0 + rows[0][i]) + rows[1][i]) + ...; rows are `stride` floats apart; stereo: interleaved L, R with L == R
hipError_t s2r_launch_sum_rows(const float *rows, uint32_t n_rows, uint32_t frames, uint32_t stride, int stereo, float *out, hipStream_t stream,
                               const S2rDone *done);
