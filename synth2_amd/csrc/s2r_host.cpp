// s2r_host.cpp — the C ABI of libs2r (include/s2r.h): handle, device memory, voice pool,
// event folding, launches.  Compiled with hipcc, -ffp-contract=off.
//
// Reference boundary being replaced: s2_lib::try3::synth::Synth
// (/root/reference/components/s2_lib/src/try3/synth.rs:9-203).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <atomic>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "s2r.h"
#include "s2r_device.h"
#include "s2r_math.h"
#include "s2r_patch.h"
#include "s2r_voices.h"

namespace {

constexpr int kEventSlots = 4;
constexpr size_t kVoiceWords = S2R_VOICE_WORDS;    // 32-bit words of per-voice state (S2rVoiceArrays)

// components/s2_bin/src/tables.rs:6-10 regenerated; see oracle/s2_oracle.c for the note on
// the four entries the reference's literal table (tables.rs) rounds one ULP away from zero.
// CRC-32 of the 4096 table bytes is checked at create so a libm surprise fails loudly.
constexpr uint32_t kSinTableCrc = 0x55293b66u;

uint32_t crc32_bytes(const void *data, size_t n) {
    const uint8_t *p = (const uint8_t *)data;
    uint32_t c = 0xffffffffu;
    for (size_t i = 0; i < n; i++) {
        c ^= p[i];
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
    }
    return ~c;
}

void build_sin_table(float *t) {
    for (int k = 0; k < 1024; k++) {
        float i = (float)k / 1024.0f;
        i = i * 3.14159274101257324f * 2.0f;
        t[k] = (float)std::sin((double)i);
    }
    static const int bump[4] = {395, 399, 610, 627};
    for (int j = 0; j < 4; j++) t[bump[j]] = s2r_u2f(s2r_f2u(t[bump[j]]) + 1u);
}

// synth.rs:208-212: 440.0 * 2_f32.powf((note - 69.0) / 12.0), through the host libm's powf —
// the very call the reference makes.
void build_pitch_table(float *t) {
    volatile float two = 2.0f;     // keep it a real powf call
    for (int n = 0; n < 256; n++) {
        const float note = (float)n;
        t[n] = 440.0f * powf(two, (note - 69.0f) / 12.0f);
    }
}

// units.rs:44-53
float ms_as_samples(float ms, uint32_t sample_rate) {
    const float sr = (float)sample_rate;
    const float seconds = ms / 1000.0f;
    return sr * seconds;
}

// one turn of a spin loop
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::atomic_signal_fence(std::memory_order_seq_cst);
#endif
}
inline void store_fence() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_sfence();
#else
    std::atomic_thread_fence(std::memory_order_seq_cst);
#endif
}

S2rEnv resolve_env(const s2r_adsr &a, uint32_t sample_rate) {
    S2rEnv e;
    e.A = ms_as_samples(a.attack_ms, sample_rate);
    e.D = ms_as_samples(a.decay_ms, sample_rate);
    e.S = a.sustain;
    e.R = ms_as_samples(a.release_ms, sample_rate);
    e.sus_off = e.A + e.D;
    e.slope_att = 1.0f / e.A;
    e.slope_dec = (e.S - 1.0f) / e.D;
    e.slope_rel = (-e.S) / e.R;
    return e;
}

// Sample rates for which s2r_div_const(x, sr) was compared with x / sr over every float in
// its window (oracle/xcheck/libm_xcheck.c, mode "div"; tests/test_transcendentals_pinned.py).
bool fast_div_rate(uint32_t sr) {
    static const uint32_t ok[] = {8000, 11025, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000, 176400, 192000};
    for (uint32_t r : ok) if (r == sr) return true;
    return false;
}

struct EventSlot {
    S2rVoiceEvent *host = nullptr;   // pinned
    S2rVoiceEvent *dev = nullptr;
    S2rTimedEvent *thost = nullptr;  // pinned, mapped: timed events of one fill
    S2rTimedEvent *tdev = nullptr;
    // Who still reads the slot's pinned records: nobody (0); the kernels in front of `done`, a HIP event recorded behind
    // them (1); or a fill whose completion word — index `idx`, value `seq` — the host can look at (2).  An event record
    // between two kernels of a stream costs the GPU ~3 us of idle time at that boundary (measured, tools/gpu_timeline.py:
    // boundaries of 4.5-5.0 us with the records, 1.7-2.4 without), so the fills that are tracked by a completion word
    // record none.
    hipEvent_t done = nullptr;
    int state = 0;
    volatile uint32_t *word = nullptr;   // state 2: the completion word (host view) ...
    uint32_t seq = 0;                    // ... and the value it shows once the fill is done
};

// mapped host memory the host polls, or reads behind a flag the device sets: coherent (fine-grained) whatever the
// runtime's default or HIP_HOST_COHERENT say, and visible to every device of a device list
constexpr unsigned kHostPolled = hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable;

}  // namespace

// Kernels that wait for each other's workgroups inside the kernels — the two streams of s2r_fill_begin — need those workgroups
// resident TOGETHER; a second handle doing the same on the same device (or several sharing a hardware queue) could hold a
// producer out until the bounded wait gives up.  So one handle per device and process gets the two streams (the first to ask);
// the others take the one-launch form, in which no kernel waits for another.  (Other processes on the device are the
// caller's to know about: s2r.h.)
static std::atomic<int> g_two_stream_handles[64];

// a workgroup's partial row starts on a 16-byte boundary (combine_groups stores four frames at a time)
static inline uint32_t partials_stride(uint32_t max_frames) { return (max_frames + 3u) & ~3u; }

struct s2r_synth {
    s2r_config cfg{};
    int device = 0;
    uint32_t shard_begin = 0, shard_voices = 0, padded_voices = 0, block_voices = 256, n_blocks = 0, mix_groups = 1;
    uint32_t interleave = 0, shard_index = 0, shard_count = 1;   // round-robin sharding (s2r_config.shard_interleave)
    FastDiv div_interleave, div_count, div_per_kid;              // ... and its divisions (interleave, shard count, voices per contiguous shard)
    std::vector<s2r_patch> bank;                 // bank[0] is "the" patch of the reference's Synth
    uint32_t program = 0;                        // current program: the patch the next note_on gives its voice
    S2rBankEntry *bank_dev = nullptr;            // S2R_MAX_BANK entries, resolved for bank_rate
    bool bank_dirty = true; uint32_t bank_rate = 0;
    std::shared_ptr<S2rVoicePool> pool;          // one per Synth: the shards of a device-list handle share their parent's
    // Device list (s2r_config.n_devices > 1): this handle is the PARENT — it owns the pool, runs the allocation policy
    // once per event and routes the event to the shard (`kids[k]`, an ordinary single-device handle on devices[k])
    // that holds the voice; per fill every shard writes its partial mix into row k of `rows_dev` on the parent's
    // device (peer-to-peer where the devices differ) and the parent adds the rows in shard order rooted at +0.0.
    s2r_synth *parent = nullptr;
    std::vector<s2r_synth *> kids;
    float *rows_dev[2] = {nullptr, nullptr};     // [n kids][max_frames], one per fill in flight
    std::vector<float *> kid_stage;              // per kid: a row on ITS device when it cannot write the parent's rows directly
    std::vector<hipEvent_t> kid_done[2];         // per slot, per kid: its partial row is in rows_dev[slot]
    uint32_t rows_slot = 0;
    std::vector<uint32_t> seed_override;         // per pool voice; 0 = reference behaviour
    // event folding (one record per touched shard voice between two fills)
    std::vector<S2rVoiceEvent> pending;
    std::vector<int32_t> pending_slot;           // shard-local voice -> index in pending, -1
    EventSlot slots[kEventSlots];
    int next_slot = 0;
    // timed events (take effect inside the next fill at a 16-frame boundary)
    std::vector<S2rTimedEvent> tpending;
    std::vector<int32_t> tlast;                  // shard-local voice -> its last timed event this fill, -1
    std::vector<int32_t> tfirst;                 // scratch of flush_events: shard-local voice -> its first timed event, -1
    uint32_t fill_time = 0;                      // frames of the next fill the pool clock has already moved
    uint32_t tev_capacity = 0;
    int32_t *voice_ev_head = nullptr;
    S2rTimedEvent *tev_copy = nullptr;           // the fill's timed events in HBM (copied from the mapped slot by the heads kernel)
    // device
    hipStream_t stream = nullptr;
    S2rVoiceArrays v{};
    void *voice_mem = nullptr;
    float *block_partials = nullptr;
    float *out_dev = nullptr;
    float *out_host = nullptr;                   // pinned and device-mapped, 2*max_frames
    float *out_host_dev = nullptr;               // the device's view of out_host
    // Completion words (S2rDone): mapped host memory the last kernel of a fill writes its sequence number to — [0], [1]
    // the two ring slots of s2r_fill_begin / s2r_fill_end, [2] the synchronous fills — and the arrival counter in HBM
    uint32_t *done_host = nullptr, *done_dev = nullptr, *done_counter = nullptr;
    uint32_t done_seq = 0;                       // last value handed out
    uint32_t ring_seq[2] = {0, 0};               // what done_host[slot] must show before s2r_fill_end copies slot's buffer (0: wait on the event)
    // s2r_fill_begin / s2r_fill_end: two more mapped output buffers, the fills in flight (oldest first)
    float *ring_host[2] = {nullptr, nullptr}, *ring_dev[2] = {nullptr, nullptr};
    hipEvent_t ring_done[2] = {nullptr, nullptr};
    size_t ring_frames[2] = {0, 0};
    uint32_t ring_head = 0, ring_count = 0;
    // s2r_fill_begin leaves its fill's mix to the NEXT s2r_fill_begin, which launches it together with its own chain
    // heads (s2r_mix_and_heads_kernel: one launch boundary less per buffer), or to s2r_fill_end, whichever comes first
    struct DeferredMix { bool active = false; bool overlap = false; S2rMixParams m{}; int ring_slot = -1; } dmix;
    // Two streams for the fills of s2r_fill_begin (S2rOverlapWords, s2r_device.h): the mixes and the chain heads run on
    // stream_b beside the render kernels on `stream`; [1] of each pair is the second buffer (by the fill's parity)
    hipStream_t stream_b = nullptr;
    bool ov_enabled = false;                     // stream_b and the second buffers exist (S2R_OVERLAP=0 leaves them out)
    bool ov_registered = false;                  // ... and this handle holds its device's two-stream slot (g_two_stream_handles)
    bool ov_busy = false;                        // kernels of an overlapped fill may still be running on stream_b
    float *partials2[2] = {nullptr, nullptr};
    int32_t *heads2[2] = {nullptr, nullptr};
    S2rTimedEvent *tevcopy2[2] = {nullptr, nullptr};
    S2rOverlapWords *ov_words = nullptr;         // device memory
    uint32_t ov_render_target[2] = {0, 0}, ov_heads_target[2] = {0, 0};
    uint32_t ov_fill = 0;                        // overlapped fills begun so far: the next one's parity is its low bit
    // One launch per fill (S2rMixTail, s2r_device.h): the render kernel builds its own chain heads from the fill's records grouped
    // by workgroup and the last workgroups to finish add the rows up.  S2R_FUSED=0: the three-launch form (heads, render, mix)
    int fused_mode = 1;                          // 0 never; 1 everything but the two-stream fills of s2r_fill_begin; 2 those too
    uint32_t *fz_arrive = nullptr;               // device memory [2]: rows written, by parity (launches per fill use [0])
    uint32_t fz_target[2] = {0, 0};
    FastDiv div_block;                           // local voice -> workgroup
    std::vector<S2rTimedEvent> tsorted;          // the fill's records grouped by workgroup ...
    std::vector<uint32_t> tperm, tbounds;        // ... arrival index -> grouped index; workgroup b's records are [tbounds[b], tbounds[b + 1])
    // exchange of partial rows between the shards of a device list: a counter per rows slot in the parent's device memory
    uint32_t *rows_done = nullptr;
    uint32_t rows_target[2] = {0, 0};
    // exchange between PROCESSES (one process per GPU; s2r_exchange_create / _attach): the ranks' rows and the counters live in
    // one block of the root's device memory that the other ranks map through an IPC handle
    bool xg_on = false;
    uint32_t xg_rank = 0, xg_n = 0, xg_target[2] = {0, 0};
    float *xg_rows = nullptr;                    // [2][xg_n][max_frames]
    uint32_t *xg_done = nullptr;                 // [2] counters behind the rows, then [2] words `consumed` (the root's: fused_tail)
    void *xg_block = nullptr; bool xg_owner = false;
    bool force_stage = false, force_peer = false;   // S2R_FORCE_STAGE / S2R_FORCE_PEER: the multi-device branches on one device (tests)
    // The pool-resident render kernel (S2rPool, s2r_device.h; s2r_set_resident): running on `stream` between fills while
    // pool_running; stopped by every entry point that touches the device or what the kernel's arguments were built from
    bool resident = false, pool_running = false;
    uint32_t pool_seq = 0, pool_launch_id = 0, pool_rate = 0, pool_fills = 0;
    uint32_t *pool_cmd = nullptr, *pool_cmd_dev = nullptr; bool pool_cmd_vram = false;
    uint32_t *pool_slices = nullptr, *pool_slices_dev = nullptr;   // mapped host memory [S2R_POOL_CMD_SLOTS][n_blocks + 1]
    uint32_t *pool_host = nullptr, *pool_host_dev = nullptr;       // mapped host memory: [0] the kernel's "exited" word
    uint32_t *pool_decided = nullptr;                              // device memory
    bool pool_gran_pending = false;                                // the fill just posted comes back as granules (fill_host reads them)
    EventSlot *pool_gran_slot = nullptr;                           // ... and this event slot's records are free again once it has
    volatile uint32_t *pool_staged = nullptr; uint32_t pool_staged_seq = 0;   // a command whose payload is written and whose sequence words
                                                                   // the caller (a device list's parent) writes with its other shards'
    uint32_t pool_idle_ticks = 200000u;                            // 2 ms without a command
    int n_cu = 0;
    float *os_buf = nullptr, *os_taps = nullptr; // 4x oversampling: [62 history + max_frames] mix at 4x rate, 63 taps
    float *sin_dev = nullptr;
    float *noise_dev = nullptr;                  // the noise table (S2rRenderParams.noise_tab), 65 536 floats
    float *per_voice_dev = nullptr; size_t per_voice_cap = 0;
    // coefficient tables of the patch (S2rTabRef): rebuilt on the device when the patch or the sample
    // rate changes
    float *tab_dev = nullptr; size_t tab_cap = 0;        // floats
    S2rTabRef tab{};
    bool tab_dirty = true; uint32_t tab_rate = 0;
    float *bank_tab_dev = nullptr; size_t bank_tab_cap = 0;      // the patch bank's coefficient tables (floats)
    unsigned long long *stamps_dev = nullptr;            // diagnostic builds (-DS2R_STAMPS): per-wave phase stamps of the last fill
    unsigned long long *timeline_dev = nullptr; uint32_t timeline_n = 0, timeline_cap = 0;   // ... and the launches' timeline
    bool use_tab = true, use_arg_events = true;
    // s2r_set_low_latency: the resident kernel (S2rResident, s2r_device.h) — running on `stream` between fills while
    // res_running; every entry point that touches the device or what the kernel's arguments were built from stops it first
    bool low_latency = false, res_running = false, res_stereo = false;
    uint32_t res_rate = 0, res_seq = 0, res_launch_id = 0;
    uint32_t *res_host = nullptr, *res_dev = nullptr;    // mapped host memory: [32] the kernel's "exited" word, and, where the CPU cannot
                                                         // write device memory, [0 .. 31] the command
    unsigned long long *res_gran = nullptr, *res_gran_dev = nullptr;   // mapped: the granules of fills of up to S2R_RES_GRANULE_FRAMES frames
    uint32_t *res_cmd = nullptr;                         // the command as the CPU writes it: res_host, or 32 words of fine-grained
    uint32_t *res_cmd_dev = nullptr;                     // DEVICE memory (large BAR) that the kernel polls without crossing the link
    bool res_cmd_vram = false;
    float pitch_table[256];
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool timing = false, timed = false, no_flat_shortcut = false;
    uint64_t double_release = 0;
    s2r_voice_log_fn voice_log = nullptr;    // synth.rs:118's log::debug!, as a callback
    void *voice_log_user = nullptr;
    // A kernel that gave up a bounded wait for another kernel's (or workgroup's) work left the fill unrendered — it touches
    // neither the voices' state nor the chain heads then — while the host's pool clock and event bookkeeping had moved on:
    // the handle says so from then on instead of rendering something else than what its caller believes (import a checkpoint
    // into a new handle to go on).
    bool broken = false;
    std::string err = "";
};


namespace {

int set_err(s2r_synth *s, int code, const char *fmt, ...) {
    if (s) {
        char buf[512];
        va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        s->err = buf;
    }
    return code;
}

// every entry point that touches the device, or anything a running resident kernel's arguments were built from, first
#define S2R_QUIESCE(s) do { const int rc_q_ = quiesce(s); if (rc_q_ != S2R_OK) return rc_q_; } while (0)
int resident_stop(s2r_synth *s);
int pool_stop(s2r_synth *s);
int quiesce(s2r_synth *s) { int rc = resident_stop(s); return rc != S2R_OK ? rc : pool_stop(s); }

#define S2R_HIP(s, call)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return set_err((s), S2R_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// pool index -> this handle's local voice index, or -1 when another handle renders it
inline int64_t to_local(const s2r_synth *s, uint32_t pool_index) {
    if (s->interleave == 0) {
        if (pool_index < s->shard_begin || pool_index >= s->shard_begin + s->shard_voices) return -1;
        return (int64_t)(pool_index - s->shard_begin);
    }
    const uint32_t run = s->div_interleave.div(pool_index), within = pool_index - run * s->interleave;
    const uint32_t lrun = s->div_count.div(run);
    if (run - lrun * s->shard_count != s->shard_index) return -1;
    return (int64_t)(lrun * s->interleave + within);
}
inline uint32_t to_pool(const s2r_synth *s, uint32_t local) {
    if (s->interleave == 0) return s->shard_begin + local;
    const uint32_t lrun = s->div_interleave.div(local);
    return (lrun * s->shard_count + s->shard_index) * s->interleave + (local - lrun * s->interleave);
}

// the handle that renders pool voice `pool_index` — this one, or for a device-list handle the shard that holds it — and
// the voice's index there; nullptr when another process' handle renders it
inline s2r_synth *shard_of(s2r_synth *s, uint32_t pool_index, uint32_t *local) {
    if (!s->kids.empty()) {
        const uint32_t k = s->interleave ? s->div_count.mod(s->div_interleave.div(pool_index)) : s->div_per_kid.div(pool_index);
        s = s->kids[k];
    }
    const int64_t mine = to_local(s, pool_index);
    if (mine < 0) return nullptr;
    *local = (uint32_t)mine;
    return s;
}

void push_event(s2r_synth *top, uint32_t pool_index, uint32_t flags, float pitch, uint32_t seed, uint32_t program = 0) {
    uint32_t local = 0;
    s2r_synth *s = shard_of(top, pool_index, &local);
    if (!s) return;
    int32_t slot = s->pending_slot[local];
    if (slot < 0) {
        slot = (int32_t)s->pending.size();
        s->pending_slot[local] = slot;
        s->pending.push_back(S2rVoiceEvent{local, 0u, 0.0f, 0u});
    }
    S2rVoiceEvent &e = s->pending[(size_t)slot];
    if (flags & S2R_EV_RESTART) {                 // wipes an earlier release
        e.flags = S2R_EV_RESTART | (program << S2R_EV_PROGRAM_SHIFT); e.pitch = pitch; e.seed = seed;
    }
    if (flags & S2R_EV_RELEASE) e.flags |= S2R_EV_RELEASE;
}

// An untimed event of a voice whose shard already holds frame-0 RECORDS for the next fill (a big batch's, s2r_note_events) joins
// them, behind them: folded, it would be applied in front.  false: no records are waiting (the caller folds it).
bool append_frame0_record(s2r_synth *top, uint32_t pool_index, uint32_t flags, float pitch, uint32_t seed, uint32_t program) {
    uint32_t local = 0;
    s2r_synth *sh = shard_of(top, pool_index, &local);
    if (!sh || sh->tpending.empty()) return false;
    const int32_t idx = (int32_t)sh->tpending.size();
    S2rTimedEvent te{};
    te.voice = local; te.frame = 0; te.flags = flags; te.pitch = pitch; te.seed = seed; te.next = -1; te.program = program;
    if (sh->tlast[local] >= 0) sh->tpending[(size_t)sh->tlast[local]].next = idx;
    else te.flags |= S2R_TEV_FIRST;
    sh->tlast[local] = idx;
    sh->tpending.push_back(te);
    return true;
}

// The other way round, for the entry points that take no chains (per-voice rows, checkpoints, seeds): while no event has a frame
// inside the fill (fill_time == 0: every record waiting is a frame-0 one) the records are folded back, in their order, behind
// whatever was folded before them.
void fold_frame0_records(s2r_synth *top) {
    for (s2r_synth *kid : top->kids) fold_frame0_records(kid);
    s2r_synth *root = top->parent ? top->parent : top;
    if (top->tpending.empty() || root->fill_time != 0) return;
    for (const S2rTimedEvent &te : top->tpending) {
        int32_t slot = top->pending_slot[te.voice];
        if (slot < 0) {
            slot = (int32_t)top->pending.size();
            top->pending_slot[te.voice] = slot;
            top->pending.push_back(S2rVoiceEvent{te.voice, 0u, 0.0f, 0u});
        }
        S2rVoiceEvent &e = top->pending[(size_t)slot];
        if (te.flags & S2R_EV_RESTART) { e.flags = S2R_EV_RESTART | (te.program << S2R_EV_PROGRAM_SHIFT); e.pitch = te.pitch; e.seed = te.seed; }
        if (te.flags & S2R_EV_RELEASE) e.flags |= S2R_EV_RELEASE;
    }
    for (const S2rTimedEvent &te : top->tpending) top->tlast[te.voice] = -1;
    top->tpending.clear();
}

// upload the folded events and apply them on `stream`; publish the timed ones.  Returns the slot
// whose `done` event the caller must record AFTER the render kernel when timed events exist
// (the kernel reads them from the slot's mapped memory), or nullptr.
// the mix a fill left behind, on its own (no chain heads to go with)
int launch_deferred_mix(s2r_synth *s, hipStream_t stream) {
    if (!s->dmix.active) return S2R_OK;
    s->dmix.active = false;
    // (an overlapped fill's mix waits in the kernel for its rows: it goes to the second stream wherever it is launched from)
    S2R_HIP(s, s2r_launch_mix(s->dmix.m, s->dmix.overlap ? s->stream_b : stream));
    return S2R_OK;
}

// Leaves the two-stream mode: the last overlapped fill's mix is launched and both streams are waited for, so that whatever
// follows on `stream` alone finds the partial rows, the chain heads and the event copies free.
int overlap_drain(s2r_synth *s) {
    if (!s->ov_busy) return S2R_OK;
    if (s->dmix.active && s->dmix.overlap) { int rc = launch_deferred_mix(s, s->stream); if (rc != S2R_OK) return rc; }
    // (a pool-resident kernel never ends a wait for its stream; the last mix on stream_b waits in the kernel for the rows of
    // every workgroup of its fill, so it says what the render side's stream would)
    if (!s->pool_running) S2R_HIP(s, hipStreamSynchronize(s->stream));
    S2R_HIP(s, hipStreamSynchronize(s->stream_b));
    s->ov_busy = false;
    return S2R_OK;
}

// waits until nothing on the device reads the slot's pinned records any more
int slot_release(s2r_synth *s, EventSlot &sl) {
    if (sl.state == 1) S2R_HIP(s, hipEventSynchronize(sl.done));
    else if (sl.state == 2 && (int32_t)(*sl.word - sl.seq) < 0) {
        // (four slots rotate and at most two fills are in flight, so this is the rare path: the fill's last kernel — its
        // mix, possibly still deferred — has not reported yet)
        int rc = launch_deferred_mix(s, s->stream);
        if (rc != S2R_OK) return rc;
        if (!s->pool_running) S2R_HIP(s, hipStreamSynchronize(s->stream));
        if (s->ov_busy) S2R_HIP(s, hipStreamSynchronize(s->stream_b));
        // (a fill of the pool-resident kernel: only its completion word tells)
        for (int spin = 0; s->pool_running && (int32_t)(*sl.word - sl.seq) < 0 && spin < 2000000; spin++) cpu_relax();
    }
    sl.state = 0;
    return S2R_OK;
}

// `done`: the completion word of the fill these events belong to (nullptr: none; the slot is then guarded by an event)
// `ov_parity` >= 0: the fill is an overlapped one (two streams): its chain heads and event copy go to that parity's buffers,
// built on stream_b, and count themselves in for the render kernel that waits for them
int flush_events(s2r_synth *s, hipStream_t stream, EventSlot **timed_slot, const S2rTimedEvent **tev_dev, const S2rDone *done = nullptr,
                 int ov_parity = -1) {
    *timed_slot = nullptr; *tev_dev = nullptr;
    if (s->pending.empty() && s->tpending.empty()) return S2R_OK;
    EventSlot &sl = s->slots[s->next_slot];
    s->next_slot = (s->next_slot + 1) % kEventSlots;
    { int rc = slot_release(s, sl); if (rc != S2R_OK) return rc; }
    if (!s->pending.empty() && !s->tpending.empty()) {
        // The fill has timed events anyway (the render kernel will walk per-voice event chains): the events of its first
        // frame join them as frame-0 records at the head of their voice's chain — one launch (the chain heads) instead
        // of two.  A folded record is "restart, then maybe release" or "release" (push_event), which is what one timed
        // record expresses too.
        std::vector<int32_t> &first = s->tfirst;
        for (size_t i = 0; i < s->tpending.size(); i++)
            if (s->tpending[i].flags & S2R_TEV_FIRST) first[s->tpending[i].voice] = (int32_t)i;
        for (const S2rVoiceEvent &e : s->pending) {
            S2rTimedEvent te{};
            te.voice = e.voice; te.frame = 0; te.flags = (e.flags & (S2R_EV_RESTART | S2R_EV_RELEASE)) | S2R_TEV_FIRST;
            te.pitch = e.pitch; te.seed = e.seed; te.program = e.flags >> S2R_EV_PROGRAM_SHIFT;
            te.next = first[e.voice];
            if (te.next >= 0) s->tpending[(size_t)te.next].flags &= ~S2R_TEV_FIRST;
            s->tpending.push_back(te);
        }
        for (const S2rTimedEvent &e : s->tpending) first[e.voice] = -1;
        for (const S2rVoiceEvent &e : s->pending) s->pending_slot[e.voice] = -1;
        s->pending.clear();
    }
    const uint32_t n = (uint32_t)s->pending.size();
    if (n) {
        std::memcpy(sl.host, s->pending.data(), n * sizeof(S2rVoiceEvent));
        // the kernel reads the pinned (device-mapped) host buffer directly: a few KB over PCIe inside
        // the kernel instead of a separate copy node in front of it
        S2R_HIP(s, s2r_launch_events(s->v, sl.dev, n, stream));
    }
    const uint32_t nt = (uint32_t)s->tpending.size();
    if (nt > s->tev_capacity) {
        // more timed events in one fill than the buffers hold: grow all of them (rare; every earlier user of the
        // old buffers is waited for first)
        S2R_HIP(s, hipStreamSynchronize(stream));
        S2R_HIP(s, hipStreamSynchronize(s->stream));
        if (s->stream_b) S2R_HIP(s, hipStreamSynchronize(s->stream_b));
        uint32_t cap = s->tev_capacity;
        while (cap < nt) cap *= 2u;
        for (EventSlot &e : s->slots) {
            e.state = 0;                                       // (both streams were just waited for)
            S2R_HIP(s, hipHostFree(e.thost)); e.thost = nullptr; e.tdev = nullptr;
            S2R_HIP(s, hipHostMalloc((void **)&e.thost, (size_t)cap * sizeof(S2rTimedEvent), kHostPolled));
            S2R_HIP(s, hipHostGetDevicePointer((void **)&e.tdev, e.thost, 0));
        }
        S2R_HIP(s, hipFree(s->tev_copy)); s->tev_copy = nullptr;
        S2R_HIP(s, hipMalloc((void **)&s->tev_copy, (size_t)cap * sizeof(S2rTimedEvent)));
        s->tevcopy2[0] = s->tev_copy;
        if (s->ov_enabled) {
            S2R_HIP(s, hipFree(s->tevcopy2[1])); s->tevcopy2[1] = nullptr;
            S2R_HIP(s, hipMalloc((void **)&s->tevcopy2[1], (size_t)cap * sizeof(S2rTimedEvent)));
        }
        s->tev_capacity = cap;
    }
    if (nt) {
        std::memcpy(sl.thost, s->tpending.data(), nt * sizeof(S2rTimedEvent));
        if (ov_parity >= 0) {
            // two streams: the chain heads (and, with them, the previous overlapped fill's mix, which waits for its rows in
            // the kernel) on stream_b, beside whatever render kernel is running
            uint32_t *hc = &s->ov_words->heads_done[ov_parity];
            s->ov_heads_target[ov_parity] += (nt + 255u) / 256u;
            // (test hook: S2R_DEBUG_WITHHOLD_HEADS=k leaves out the k-th such launch — a producer that never runs; the render
            // kernel's bounded wait, the fill's error and the handle's refusal afterwards are what tests/test_gpu_parity.py holds)
            static const long withhold = [] { const char *e = std::getenv("S2R_DEBUG_WITHHOLD_HEADS"); return e ? std::atol(e) : 0L; }();
            static long n_heads_launches = 0;
            const bool skip_heads = withhold > 0 && ++n_heads_launches == withhold;
            if (skip_heads) {
                if (s->dmix.active && s->dmix.overlap) { int rc = launch_deferred_mix(s, s->stream); if (rc != S2R_OK) return rc; }
            } else
            if (s->dmix.active && s->dmix.overlap) {
                s->dmix.active = false;
                S2rMixParams m = s->dmix.m;
                m.ov_heads_counter = hc;
                // (a launch that fails counts nobody in: the target goes back, and the handle is done for — ADVICE r3)
                if (s2r_launch_mix_and_heads(m, s->heads2[ov_parity], sl.tdev, s->tevcopy2[ov_parity], nt, s->stream_b) != hipSuccess) {
                    s->ov_heads_target[ov_parity] -= (nt + 255u) / 256u; s->broken = true;
                    return set_err(s, S2R_ERR_HIP, "the launch of the chain heads and the mix failed");
                }
            } else {
                if (s2r_launch_tev_heads(s->heads2[ov_parity], sl.tdev, s->tevcopy2[ov_parity], nt, s->stream_b, hc) != hipSuccess) {
                    s->ov_heads_target[ov_parity] -= (nt + 255u) / 256u; s->broken = true;
                    return set_err(s, S2R_ERR_HIP, "the launch of the chain heads failed");
                }
            }
            *timed_slot = &sl; *tev_dev = s->tevcopy2[ov_parity];
        } else if (s->dmix.active && stream == s->stream) {    // the previous fill's mix rides with this fill's chain heads
            s->dmix.active = false;
            S2R_HIP(s, s2r_launch_mix_and_heads(s->dmix.m, s->voice_ev_head, sl.tdev, s->tev_copy, nt, stream));
            *timed_slot = &sl; *tev_dev = s->tev_copy;         // the kernels read the HBM copy
        } else {
            S2R_HIP(s, s2r_launch_tev_heads(s->voice_ev_head, sl.tdev, s->tev_copy, nt, stream));
            *timed_slot = &sl; *tev_dev = s->tev_copy;
        }
        for (const S2rTimedEvent &e : s->tpending) s->tlast[e.voice] = -1;
        s->tpending.clear();
    } else if (done) {
        sl.state = 2; sl.word = s->done_host + (done->flag - s->done_dev); sl.seq = done->value;
    } else {
        S2R_HIP(s, hipEventRecord(sl.done, stream));
        sl.state = 1;
    }
    for (const S2rVoiceEvent &e : s->pending) s->pending_slot[e.voice] = -1;
    s->pending.clear();
    return S2R_OK;
}

#define S2R_REFUSE_BROKEN(s) do { if ((s)->broken) return set_err((s), S2R_ERR_HIP, "an earlier fill failed on the device (a bounded wait between kernels ran out): the device's voices no longer match the host's bookkeeping; destroy the handle"); } while (0)

int check_fill(s2r_synth *s, size_t frames, uint32_t sample_rate) {
    if (!s) return S2R_ERR_INVALID;
    S2R_REFUSE_BROKEN(s);
    if (frames > s->cfg.max_frames) return set_err(s, S2R_ERR_TOO_MANY_FRAMES, "frames %zu > max_frames %u", frames, s->cfg.max_frames);
    if (sample_rate == 0) return set_err(s, S2R_ERR_INVALID, "sample_rate_hz must be > 0");
    // process.rs:36,71: offset.checked_add(..).expect("overflow") — the reference panics once a
    // voice's offset would pass u32::MAX; report it instead of rendering garbage.
    if (s->fill_time && s->fill_time >= frames)
        return set_err(s, S2R_ERR_INVALID, "a timed event at frame %u does not fall inside this %zu-frame fill", s->fill_time, frames);
    if (s->pool->oldest_offset() + (frames - s->fill_time) > 0xffffffffull)
        return set_err(s, S2R_ERR_OFFSET_OVERFLOW, "a voice's frame offset would overflow u32 (the reference panics here)");
    for (size_t k = 0; k < s->bank.size(); k++) {
        const s2r_patch &pt = s->bank[k];
        if (pt.lpf_kind == S2R_FILT_ONEPOLE) continue;
        // dsp_filters.rs evaluates sin/cos of theta = 2 pi f / sr.  The device restatement of the libm
        // routines is exact for every finite argument; only an overflowing 2 pi f (inf -> NaN, whose
        // sign bit differs between x86 and the GPU) is refused.
        const double amt = pt.mod_env_to_lpf_freq > 0.0f ? (double)pt.mod_env_to_lpf_freq : 0.0;
        const double num_max = 2.0 * 3.14159265358979323846 * (double)pt.lpf_freq * std::exp2(amt) * 1.000001;
        if (!(num_max < 3.4028234e38))
            return set_err(s, S2R_ERR_PATCH_RANGE, "patch %zu, lpf.kind %d: 2 pi * lpf.freq * 2^mod_env_to_lpf_freq overflows f32",
                           k, pt.lpf_kind);
        // dsp_filters.rs:205-207: tan(theta / (2 Q)); Q = 0 makes that tan(inf) = NaN (same sign caveat)
        if (pt.lpf_kind == S2R_FILT_BP2 &&
            !(pt.lpf_q > 0.0f && num_max / (double)sample_rate / (2.0 * (double)pt.lpf_q) < 3.4028234e38))
            return set_err(s, S2R_ERR_PATCH_RANGE, "patch %zu: lpf.kind bp2 needs lpf.q > 0 (tan(theta / (2 q)) must stay finite)", k);
        if (pt.lpf_kind >= S2R_FILT_SVF_LP && !(pt.lpf_q >= 0x1p-100f))
            return set_err(s, S2R_ERR_PATCH_RANGE, "patch %zu: the state-variable filter needs lpf.q > 0 (k = 1 / q)", k);
    }
    return S2R_OK;
}

// Coefficient tables exist for a single patch (the one-pole kernel, and the dsp_filters.rs kinds with workgroups of up
// to 256 voices) while the flat-envelope logic is enabled.
bool tables_wanted(const s2r_synth *s) {
    return s->use_tab && !s->no_flat_shortcut && s->bank.size() == 1 && s->bank[0].osc_kind <= S2R_OSC_SINE &&
           (s->bank[0].lpf_kind == S2R_FILT_ONEPOLE || s->block_voices <= 256u);
}

// region lengths of one patch's tables; false: an envelope too long to tabulate
bool plan_tables(const S2rEnv &e, S2rTabBuild &b) {
    if (!(e.sus_off >= 0.0f && e.sus_off < (float)S2R_TAB_MAX_ENTRIES && e.R >= 0.0f && e.R < (float)S2R_TAB_MAX_ENTRIES)) return false;
    b.mod = e;
    b.rc_t0 = (uint32_t)std::ceil((double)e.sus_off);
    b.n_ad = b.rc_t0 + 1u + S2R_TAB_PAD;
    b.n_rel = (uint32_t)std::ceil((double)e.R) + 2u + S2R_TAB_PAD;
    b.n_entries = b.n_ad + 2u * b.n_rel + 48u;
    b.plane = (b.n_entries + 3u) & ~3u;
    return true;
}

// (Re)builds the patch's coefficient tables for this sample rate on `stream` when the patch or the rate changed.
// Envelopes too long to tabulate (attack + decay or release beyond S2R_TAB_MAX_ENTRIES frames) leave tab.base null:
// such a patch computes in-lane.
int ensure_tables(s2r_synth *s, const S2rRenderParams &p, uint32_t sample_rate, hipStream_t stream) {
    if (!tables_wanted(s)) { return S2R_OK; }
    if (!s->tab_dirty && s->tab_rate == sample_rate) return S2R_OK;
    s->tab = S2rTabRef{};
    s->tab_dirty = false; s->tab_rate = sample_rate;
    S2rTabBuild b{};
    if (!plan_tables(p.mod, b)) return S2R_OK;
    b.lpf_freq = p.lpf_freq; b.amt_lpf = p.amt_lpf; b.amt_osc = p.amt_osc; b.sr = p.sr; b.rcp_sr = p.rcp_sr;
    b.fast_div_sr = p.fast_div_sr; b.lpf_kind = p.lpf_kind; b.lpf_damping = p.lpf_damping;
    const bool onepole = p.lpf_kind == S2R_FILT_ONEPOLE;
    const uint32_t n_planes = (onepole ? 2u : 3u) + (p.amt_osc != 0.0f ? 1u : 0u);
    b.fm_plane = p.amt_osc != 0.0f ? (onepole ? 2u : 3u) : 0u;
    const size_t need = (size_t)b.plane * n_planes;
    if (need > s->tab_cap) {
        // earlier fills (other streams included) may still read the old planes
        S2R_HIP(s, hipStreamSynchronize(stream));
        S2R_HIP(s, hipStreamSynchronize(s->stream));
        if (s->tab_dev) { S2R_HIP(s, hipFree(s->tab_dev)); s->tab_dev = nullptr; s->tab_cap = 0; }
        S2R_HIP(s, hipMalloc((void **)&s->tab_dev, need * sizeof(float)));
        s->tab_cap = need;
    }
    b.base = s->tab_dev;
    S2R_HIP(s, s2r_launch_tables(b, stream));
    S2rTabRef &t = s->tab;
    t.base = s->tab_dev; t.plane = b.plane;
    t.ad = 0; t.rc = (int32_t)b.n_ad; t.rc_t0 = b.rc_t0; t.ru = (int32_t)(b.n_ad + b.n_rel);
    t.sus = (int32_t)(b.n_ad + 2u * b.n_rel); t.end = t.sus + 16; t.dead = t.sus + 32;
    t.fm_plane = b.fm_plane;
    return S2R_OK;
}

S2rRenderParams make_params(s2r_synth *s, size_t frames, uint32_t sample_rate) {
    S2rRenderParams p{};
    p.osc_kind = s->bank[0].osc_kind;
    p.osc_gain = s->bank[0].osc_gain;
    p.noise_level = s->bank[0].noise;
    p.lpf_freq = s->bank[0].lpf_freq;
    p.amt_osc = s->bank[0].mod_env_to_osc_freq;
    p.amt_lpf = s->bank[0].mod_env_to_lpf_freq;
    p.lpf_kind = s->bank[0].lpf_kind;
    p.lpf_damping = s->bank[0].lpf_kind >= S2R_FILT_BP2 ? s->bank[0].lpf_q : s->bank[0].lpf_damping;
    p.amp = resolve_env(s->bank[0].amp_env, sample_rate);
    p.mod = resolve_env(s->bank[0].mod_env, sample_rate);
    p.sr = (float)sample_rate;
    p.rcp_sr = 1.0f / p.sr;
    // 2^-10 <= pow2 <= 2^10 (|mod * amount| <= 10), so lpf_freq in [2^-30, 2^30] keeps the
    // dividend inside the window the 3-op quotient was verified for
    p.fast_div_sr = (fast_div_rate(sample_rate) && s->bank[0].lpf_freq >= 0x1p-30f && s->bank[0].lpf_freq <= 0x1p30f) ? 1 : 0;
    p.no_flat_shortcut = s->no_flat_shortcut ? 1 : 0;
    p.frames = (uint32_t)frames;
    p.n_voices = s->shard_voices;
    p.frames_stride = partials_stride(s->cfg.max_frames);
    p.v = s->v;
    p.block_partials = s->block_partials;
    p.per_voice = nullptr;
    p.sin_table = s->sin_dev;
    { static const bool off = [] { const char *e = std::getenv("S2R_NOISE_TAB"); return e && e[0] == '0'; }();   // (measurement aid: per-frame noise)
      p.noise_tab = off ? nullptr : s->noise_dev; }
    p.bank = s->bank_dev;
    p.bank_size = (uint32_t)s->bank.size();
    return p;
}

// The kernel that renders patch banks also renders single patches with a DPW oscillator: it reads its patch from the device copy
// of the bank.  Resolves every patch for this sample rate (Ms::as_samples, units.rs:44-53), builds the bank's coefficient tables
// and replaces the device copy when the bank or the rate changed; rare (bank edits, rate changes), so a synchronous hand-over.
int ensure_bank(s2r_synth *s, uint32_t sample_rate, hipStream_t stream, bool *bank_kernel_out) {
    bool bank_kernel = s->bank.size() > 1;
    for (const s2r_patch &pt : s->bank) if (pt.osc_kind > S2R_OSC_SINE) bank_kernel = true;
    if (bank_kernel && (s->bank_dirty || s->bank_rate != sample_rate)) {
        // resolve every patch for this sample rate (Ms::as_samples, units.rs:44-53) and replace the
        // device copy; rare (bank edits, rate changes), so a synchronous hand-over is fine
        std::vector<S2rBankEntry> host(s->bank.size());
        std::vector<S2rTabBuild> builds(s->bank.size());
        size_t tab_floats = 0;
        const float srf = (float)sample_rate;
        for (size_t k = 0; k < s->bank.size(); k++) {
            const s2r_patch &pt = s->bank[k];
            S2rBankEntry &e = host[k];
            std::memset(&e, 0, sizeof e);
            e.osc_kind = pt.osc_kind; e.osc_gain = pt.osc_gain; e.noise_level = pt.noise; e.lpf_freq = pt.lpf_freq;
            e.amt_osc = pt.mod_env_to_osc_freq; e.amt_lpf = pt.mod_env_to_lpf_freq; e.lpf_kind = pt.lpf_kind;
            e.lpf_shape = pt.lpf_kind >= S2R_FILT_BP2 ? pt.lpf_q : pt.lpf_damping;
            e.amp = resolve_env(pt.amp_env, sample_rate);
            e.mod = resolve_env(pt.mod_env, sample_rate);
            // this patch's coefficient tables, four planes in the bank's table buffer
            S2rTabBuild &b = builds[k];
            b = S2rTabBuild{};
            // (the whole bank's tables stay under 2^28 floats = 1 GiB, so tab_off cannot wrap its 32 bits: a patch that
            // would pass the budget computes in-lane, tab_valid = 0)
            if (s->use_tab && !s->no_flat_shortcut && plan_tables(e.mod, b) && tab_floats + (size_t)b.plane * 4u <= ((size_t)1 << 28)) {
                b.lpf_freq = e.lpf_freq; b.amt_lpf = e.amt_lpf; b.amt_osc = e.amt_osc; b.sr = srf; b.rcp_sr = 1.0f / srf;
                b.fast_div_sr = 0;                   // (the per-lane-patch kernel divides by the sample rate with a true division)
                b.lpf_kind = e.lpf_kind; b.lpf_damping = e.lpf_shape; b.fm_plane = 3u;
                e.tab_valid = 1u; e.tab_off = (uint32_t)tab_floats; e.tab_plane = b.plane;
                e.tab_ad = 0; e.tab_rc = (int32_t)b.n_ad; e.tab_rc_t0 = b.rc_t0; e.tab_ru = (int32_t)(b.n_ad + b.n_rel);
                e.tab_sus = (int32_t)(b.n_ad + 2u * b.n_rel); e.tab_end = e.tab_sus + 16; e.tab_dead = e.tab_sus + 32;
                tab_floats += (size_t)b.plane * 4u;
            }
        }
        if (tab_floats > s->bank_tab_cap) {
            S2R_HIP(s, hipStreamSynchronize(stream));
            S2R_HIP(s, hipStreamSynchronize(s->stream));
            if (s->bank_tab_dev) { S2R_HIP(s, hipFree(s->bank_tab_dev)); s->bank_tab_dev = nullptr; s->bank_tab_cap = 0; }
            S2R_HIP(s, hipMalloc((void **)&s->bank_tab_dev, tab_floats * sizeof(float)));
            s->bank_tab_cap = tab_floats;
        }
        for (size_t k = 0; k < s->bank.size(); k++)
            if (host[k].tab_valid) { builds[k].base = s->bank_tab_dev + host[k].tab_off; S2R_HIP(s, s2r_launch_tables(builds[k], stream)); }
        S2R_HIP(s, hipMemcpyAsync(s->bank_dev, host.data(), host.size() * sizeof(S2rBankEntry), hipMemcpyHostToDevice, stream));
        S2R_HIP(s, hipStreamSynchronize(stream));
        s->bank_dirty = false; s->bank_rate = sample_rate;
    }
    *bank_kernel_out = bank_kernel;
    return S2R_OK;
}

// ---- one launch per fill (S2rMixTail, s2r_device.h) and the pool-resident kernel (S2rPool) ----

// what a shard's last mixer does with its row when the fill's output is the sum of several shards' rows (a device list)
struct Exchange {
    uint32_t *rows_done = nullptr;       // counter in the root's memory, this rows slot's
    uint32_t rows_target = 0;
    uint32_t n_rows = 0, row_stride = 0;
    const float *rows = nullptr;
    float *final_out = nullptr;
    S2rDone final_done{nullptr, 0u, nullptr};
    bool final_stereo = false;
    volatile uint32_t *slot_word = nullptr;   // the completion word (host view) that says the fill's records are no longer read
    int xmode = 0;                            // S2rMixTail.xmode
};

bool onepole_single_patch(const s2r_synth *s) {
    return s->bank.size() == 1 && s->bank[0].osc_kind <= S2R_OSC_SINE && s->bank[0].lpf_kind == S2R_FILT_ONEPOLE;
}

// The fill's folded frame-0 events become frame-0 records at the head of their voices' chains: a folded record is "restart,
// then maybe release" or "release" (push_event), which is what one timed record expresses too.
void merge_pending_into_chains(s2r_synth *s) {
    if (s->pending.empty()) return;
    std::vector<int32_t> &first = s->tfirst;
    for (size_t i = 0; i < s->tpending.size(); i++)
        if (s->tpending[i].flags & S2R_TEV_FIRST) first[s->tpending[i].voice] = (int32_t)i;
    for (const S2rVoiceEvent &e : s->pending) {
        S2rTimedEvent te{};
        te.voice = e.voice; te.frame = 0; te.flags = (e.flags & (S2R_EV_RESTART | S2R_EV_RELEASE)) | S2R_TEV_FIRST;
        te.pitch = e.pitch; te.seed = e.seed; te.program = e.flags >> S2R_EV_PROGRAM_SHIFT;
        te.next = first[e.voice];
        if (te.next >= 0) s->tpending[(size_t)te.next].flags &= ~S2R_TEV_FIRST;
        s->tpending.push_back(te);
    }
    for (const S2rTimedEvent &e : s->tpending) first[e.voice] = -1;
    for (const S2rVoiceEvent &e : s->pending) s->pending_slot[e.voice] = -1;
    s->pending.clear();
}

// more timed events in one fill than the buffers hold: grow all of them (rare; every user of the old buffers is waited for)
int ensure_tev_capacity(s2r_synth *s, uint32_t nt, hipStream_t stream) {
    if (nt <= s->tev_capacity) return S2R_OK;
    S2R_HIP(s, hipStreamSynchronize(stream));
    S2R_HIP(s, hipStreamSynchronize(s->stream));
    if (s->stream_b) S2R_HIP(s, hipStreamSynchronize(s->stream_b));
    uint32_t cap = s->tev_capacity;
    while (cap < nt) cap *= 2u;
    for (EventSlot &e : s->slots) {
        e.state = 0;                                       // (both streams were just waited for)
        S2R_HIP(s, hipHostFree(e.thost)); e.thost = nullptr; e.tdev = nullptr;
        S2R_HIP(s, hipHostMalloc((void **)&e.thost, (size_t)cap * sizeof(S2rTimedEvent), kHostPolled));
        S2R_HIP(s, hipHostGetDevicePointer((void **)&e.tdev, e.thost, 0));
    }
    S2R_HIP(s, hipFree(s->tev_copy)); s->tev_copy = nullptr;
    S2R_HIP(s, hipMalloc((void **)&s->tev_copy, (size_t)cap * sizeof(S2rTimedEvent)));
    s->tevcopy2[0] = s->tev_copy;
    if (s->tevcopy2[1]) {
        S2R_HIP(s, hipFree(s->tevcopy2[1])); s->tevcopy2[1] = nullptr;
        S2R_HIP(s, hipMalloc((void **)&s->tevcopy2[1], (size_t)cap * sizeof(S2rTimedEvent)));
    }
    s->tev_capacity = cap;
    return S2R_OK;
}

// The fill's events — all of them, the folded frame-0 ones as records too — grouped by workgroup in an event slot's mapped
// buffer, the groups' bounds in s->tbounds.  *slot_out: the slot (nullptr: the fill has no events).
int fused_events(s2r_synth *s, hipStream_t stream, EventSlot **slot_out, uint32_t *nt_out) {
    *slot_out = nullptr; *nt_out = 0;
    merge_pending_into_chains(s);
    const uint32_t nt = (uint32_t)s->tpending.size();
    s->tbounds.assign((size_t)s->n_blocks + 1u, 0u);
    if (nt == 0) return S2R_OK;
    { int rc = ensure_tev_capacity(s, nt, stream); if (rc != S2R_OK) return rc; }
    EventSlot &sl = s->slots[s->next_slot];
    s->next_slot = (s->next_slot + 1) % kEventSlots;
    { int rc = slot_release(s, sl); if (rc != S2R_OK) return rc; }
    // counting sort by workgroup; a voice's chain keeps its order (the links are carried over to the new indices)
    uint32_t *b = s->tbounds.data();
    for (const S2rTimedEvent &e : s->tpending) b[s->div_block.div(e.voice) + 1u]++;
    for (uint32_t k = 0; k < s->n_blocks; k++) b[k + 1u] += b[k];
    s->tperm.resize(nt);
    {
        static thread_local std::vector<uint32_t> cur;
        cur.assign(b, b + s->n_blocks);
        for (uint32_t i = 0; i < nt; i++) s->tperm[i] = cur[s->div_block.div(s->tpending[i].voice)]++;
    }
    s->tsorted.resize(nt);
    for (uint32_t i = 0; i < nt; i++) {
        S2rTimedEvent e = s->tpending[i];
        if (e.next >= 0) e.next = (int32_t)s->tperm[(size_t)e.next];
        s->tsorted[s->tperm[i]] = e;
    }
    std::memcpy(sl.thost, s->tsorted.data(), (size_t)nt * sizeof(S2rTimedEvent));
    for (const S2rTimedEvent &e : s->tpending) s->tlast[e.voice] = -1;
    s->tpending.clear();
    *slot_out = &sl; *nt_out = nt;
    return S2R_OK;
}

// How many of the fill's last workgroups add the rows up: one per 16-frame block of the fill where the grid is big enough to
// spare them (what s2r_mix_kernel's grid was: the mix is then ~3 us behind the last row), never more than half the grid.
uint32_t pick_mixers(const s2r_synth *s, size_t frames) {
    uint32_t m = (uint32_t)((frames + 15u) / 16u);
    const uint32_t half = s->n_blocks > 1u ? s->n_blocks / 2u : 1u;
    if (m > half) m = half;
    if (m > 64u) m = 64u;
    return m ? m : 1u;
}

// the part of S2rMixTail that a shard's geometry fixes
S2rMixTail mix_tail_of(const s2r_synth *s, bool root_add) {
    S2rMixTail mt{};
    mt.n_blocks = s->n_blocks;
    mt.n_groups = root_add ? s->mix_groups : 1u;
    mt.blocks_per_group = (s->n_blocks + mt.n_groups - 1u) / mt.n_groups;
    mt.root_add = root_add ? 1 : 0;
    return mt;
}

// Can this shard's fills take the one-launch form?  More than one workgroup (a single one writes the output itself), the bounds
// in the kernel arguments, the mix's run sums in the staging the render kernel has anyway (both render kernels take the form).
bool fused_shape_ok(const s2r_synth *s) {
    if (!s->fused_mode || s->n_blocks < 2u) return false;
    if (!onepole_single_patch(s) && s->block_voices > 256u) return false;   // (the general kernel's staging for bigger workgroups is the smallest)
    if ((size_t)s->n_blocks + 1u > 3u * S2R_ARG_MAX_EVENTS) return false;
    const uint32_t groups = s->mix_groups ? s->mix_groups : 1u;
    const uint32_t runs = ((s->n_blocks + groups - 1u) / groups + 15u) / 16u * groups;
    return runs * 16u <= 16u * 32u;                              // (the smallest group-sum buffer: 16 groups x 32 frames)
}

// events -> ONE render launch on `stream`; the mix (root-added or this shard's partial row) lands in `dev_out`
int enqueue_fused(s2r_synth *s, size_t frames, uint32_t sample_rate, hipStream_t stream, float *dev_out, bool root_add, bool stereo,
                  const S2rDone *done, const Exchange *xc) {
    { int rc = overlap_drain(s); if (rc != S2R_OK) return rc; }
    { int rc = launch_deferred_mix(s, s->stream); if (rc != S2R_OK) return rc; }
    bool arg_events = s->use_arg_events && s->tpending.empty() && s->pending.size() <= S2R_ARG_MAX_EVENTS;
    if (arg_events)
        for (const S2rVoiceEvent &e : s->pending) if (e.seed != 0u) { arg_events = false; break; }
    EventSlot *slot = nullptr;
    uint32_t nt = 0;
    if (!arg_events) { int rc = fused_events(s, stream, &slot, &nt); if (rc != S2R_OK) return rc; }
    static thread_local S2rRenderArgs a;
    S2rRenderParams &p = a.p;
    bool bank_kernel = false;
    { int rc = ensure_bank(s, sample_rate, stream, &bank_kernel); if (rc != S2R_OK) return rc; }
    p = make_params(s, frames, sample_rate);
    { int rc = ensure_tables(s, p, sample_rate, stream); if (rc != S2R_OK) return rc; }
    if (tables_wanted(s)) p.tab = s->tab;
    else if (bank_kernel) { p.tab = S2rTabRef{}; p.tab.base = s->bank_tab_dev; }      // the per-lane-patch kernel adds each entry's tab_off
    p.stamps = s->stamps_dev;
    if (s->timeline_dev && s->timeline_n < s->timeline_cap) { p.timeline = s->timeline_dev; p.tl_slot = s->timeline_n++; }
    p.voice_ev_head = s->voice_ev_head;
    p.ov_fail = s->done_dev + 3;
    a.n_events = 0;
    if (arg_events) {
        a.n_events = (uint32_t)s->pending.size();
        for (uint32_t i = 0; i < a.n_events; i++) {
            const S2rVoiceEvent &e = s->pending[i];
            a.ev[3u * i] = e.voice; a.ev[3u * i + 1u] = e.flags; a.ev[3u * i + 2u] = s2r_f2u(e.pitch);
        }
        for (const S2rVoiceEvent &e : s->pending) s->pending_slot[e.voice] = -1;
        s->pending.clear();
    } else if (slot) {
        p.tev = s->tev_copy; p.tev_copy = s->tev_copy; p.tev_src = slot->tdev; p.slices = nullptr;
        std::memcpy(a.ev, s->tbounds.data(), ((size_t)s->n_blocks + 1u) * sizeof(uint32_t));
    }
    p.arrive = s->fz_arrive;
    s->fz_target[0] += s->n_blocks;
    p.arrive_target = s->fz_target[0];
    S2rMixTail &mt = p.mt;
    mt = mix_tail_of(s, root_add);
    mt.n_mixers = pick_mixers(s, frames);
    mt.stereo = stereo ? 1 : 0;
    mt.out = dev_out;
    mt.done = done ? *done : S2rDone{nullptr, 0u, s->done_counter + 3};
    if (xc) {
        if (xc->xmode != 2) mt.done.flag = nullptr;               // (a rank other than the root reports its own completion)
        mt.xmode = xc->xmode;
        mt.rows_done = xc->rows_done; mt.rows_target = xc->rows_target;
        mt.n_rows = xc->n_rows; mt.row_stride = xc->row_stride; mt.rows = xc->rows;
        mt.final_out = xc->final_out; mt.final_done = xc->final_done; mt.final_stereo = xc->final_stereo ? 1 : 0;
    }
    if (s->timing) S2R_HIP(s, hipEventRecord(s->t0, stream));
    S2R_HIP(s, s2r_launch_render(a, s->block_voices, stream));
    if (s->timing) { S2R_HIP(s, hipEventRecord(s->t1, stream)); s->timed = true; }
    if (slot) {                                   // the render kernel is the only reader of the slot's records
        if (xc && xc->slot_word) { slot->state = 2; slot->word = xc->slot_word; slot->seq = xc->final_done.value; }
        else if (done && done->flag) { slot->state = 2; slot->word = s->done_host + (done->flag - s->done_dev); slot->seq = done->value; }
        else { S2R_HIP(s, hipEventRecord(slot->done, stream)); slot->state = 1; }
    }
    if (!s->parent) {                             // (a device-list handle moves the shared clock once, after its shards)
        s->pool->advance(frames - s->fill_time);
        s->fill_time = 0;
    }
    return S2R_OK;
}

// ---- the pool-resident kernel (S2rPool, s2r_device.h; s2r_set_resident) ----

bool pool_exited(const s2r_synth *s) { return __atomic_load_n(&s->pool_host[0], __ATOMIC_ACQUIRE) == s->pool_launch_id; }

// a fill the pool-resident kernel can take: the one-launch form's shape, a grid that is resident as a whole (at most one
// workgroup of at most 256 threads per compute unit), nothing that brackets or watches single launches
bool pool_eligible(const s2r_synth *s, size_t frames) {
    return s->resident && s->kids.empty() && s->pool_cmd != nullptr && fused_shape_ok(s) && (int)s->n_blocks <= s->n_cu && s->block_voices <= 256u &&
           !s->timing && s->timeline_dev == nullptr && frames <= 0xffffu && frames <= s->cfg.max_frames;
}

// what the kernel needs besides the handle's ordinary buffers: the command ring, the slices, the decision word, and the second
// parity's rows, heads and event copy (a workgroup may be a fill ahead of its neighbours)
int pool_setup(s2r_synth *s) {
    if (s->pool_cmd) return S2R_OK;
    S2R_HIP(s, hipSetDevice(s->device));
    S2R_HIP(s, hipHostMalloc((void **)&s->pool_host, 16 * sizeof(uint32_t), kHostPolled));
    std::memset(s->pool_host, 0, 16 * sizeof(uint32_t));
    S2R_HIP(s, hipHostGetDevicePointer((void **)&s->pool_host_dev, s->pool_host, 0));
    const size_t sl_words = (size_t)S2R_POOL_CMD_SLOTS * (s->n_blocks + 1u);
    S2R_HIP(s, hipHostMalloc((void **)&s->pool_slices, sl_words * sizeof(uint32_t), kHostPolled));
    std::memset(s->pool_slices, 0, sl_words * sizeof(uint32_t));
    S2R_HIP(s, hipHostGetDevicePointer((void **)&s->pool_slices_dev, s->pool_slices, 0));
    S2R_HIP(s, hipMalloc((void **)&s->pool_decided, sizeof(uint32_t)));
    const size_t pv = s->padded_voices;
    if (!s->partials2[1]) S2R_HIP(s, hipMalloc((void **)&s->partials2[1], (size_t)s->n_blocks * partials_stride(s->cfg.max_frames) * sizeof(float)));
    if (!s->heads2[1]) {
        S2R_HIP(s, hipMalloc((void **)&s->heads2[1], pv * sizeof(int32_t)));
        S2R_HIP(s, hipMemset(s->heads2[1], 0xff, pv * sizeof(int32_t)));
    }
    if (!s->tevcopy2[1]) S2R_HIP(s, hipMalloc((void **)&s->tevcopy2[1], (size_t)s->tev_capacity * sizeof(S2rTimedEvent)));
    if (!s->res_gran) {                                          // short synchronous fills come back as tagged 8-byte words (FillCtl.granules)
        S2R_HIP(s, hipHostMalloc((void **)&s->res_gran, S2R_RES_GRANULE_FRAMES * sizeof(unsigned long long), kHostPolled));
        std::memset(s->res_gran, 0, S2R_RES_GRANULE_FRAMES * sizeof(unsigned long long));
        S2R_HIP(s, hipHostGetDevicePointer((void **)&s->res_gran_dev, s->res_gran, 0));
    }
    // The command where the CPU's write is one posted trip and the kernel's polls none: fine-grained device memory behind the
    // BAR (as the one-workgroup resident kernel's); else mapped host memory.  S2R_RES_CMD_HOST=1 keeps it in host memory.
    const size_t cmd_words = (size_t)S2R_POOL_CMD_SLOTS * S2R_POOL_CMD_WORDS;
    int large_bar = 0;
    const char *force_host = std::getenv("S2R_RES_CMD_HOST");
    if (!(force_host && force_host[0] == '1') &&
        hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, s->device) == hipSuccess && large_bar) {
        uint32_t *v = nullptr;
        if (hipExtMallocWithFlags((void **)&v, cmd_words * sizeof(uint32_t), hipDeviceMallocFinegrained) == hipSuccess && v) {
            S2R_HIP(s, hipMemset(v, 0, cmd_words * sizeof(uint32_t)));
            S2R_HIP(s, hipDeviceSynchronize());
            s->pool_cmd = v; s->pool_cmd_dev = v; s->pool_cmd_vram = true;
        } else (void)hipGetLastError();
    }
    if (!s->pool_cmd) {
        uint32_t *h = nullptr;
        S2R_HIP(s, hipHostMalloc((void **)&h, cmd_words * sizeof(uint32_t), kHostPolled));
        std::memset(h, 0, cmd_words * sizeof(uint32_t));
        S2R_HIP(s, hipHostGetDevicePointer((void **)&s->pool_cmd_dev, h, 0));
        s->pool_cmd = h; s->pool_cmd_vram = false;
    }
    return S2R_OK;
}

// Device memory that kernels on OTHER devices (or in other processes) write and this device's kernels read while they run —
// the shards' rows and the counters of an exchange: fine-grained, so that no L2 holds a line of it across a peer's store (a
// coarse-grained allocation is only coherent at kernel boundaries); plain device memory where the runtime has none to give.
hipError_t malloc_exchange(void **p, size_t bytes) {
    if (hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained) == hipSuccess && *p) return hipSuccess;
    (void)hipGetLastError();
    return hipMalloc(p, bytes);
}

// payload first, then word 15, then word 0 (device memory behind the BAR is write-combining: the fences order the stages).
// The three stages apart, so that the shards of a device list share the fences: every shard's payload, one fence, every
// shard's word 15, one fence, every shard's word 0 (a fence flushes the CPU's write-combining buffers whatever they hold).
volatile uint32_t *pool_post_payload(s2r_synth *s, const uint32_t *w, uint32_t *seq_out) {
    const uint32_t seq = ++s->pool_seq;
    volatile uint32_t *c = s->pool_cmd + (size_t)(seq % S2R_POOL_CMD_SLOTS) * S2R_POOL_CMD_WORDS;
    for (uint32_t i = 1; i < 15; i++) c[i] = w[i];
    *seq_out = seq;
    return c;
}
uint32_t pool_post(s2r_synth *s, const uint32_t *w) {
    uint32_t seq = 0;
    volatile uint32_t *c = pool_post_payload(s, w, &seq);
    if (s->pool_cmd_vram) store_fence();
    __atomic_store_n(&c[15], seq, __ATOMIC_RELEASE);
    if (s->pool_cmd_vram) store_fence();
    __atomic_store_n(&c[0], seq, __ATOMIC_RELEASE);
    if (s->pool_cmd_vram) store_fence();
    return seq;
}

// Ends the pool-resident kernel, if there is one, and waits for it: the stream is the caller's again.
int pool_stop(s2r_synth *s) {
    if (!s) return S2R_OK;
    for (s2r_synth *kid : s->kids) { int rc = pool_stop(kid); if (rc != S2R_OK) { s->err = kid->err; return rc; } }
    if (!s->pool_running) return S2R_OK;
    uint32_t w[16] = {0};
    w[1] = S2R_POOL_FLAG_EXIT << 16;
    (void)pool_post(s, w);
    s->pool_running = false;
    S2R_HIP(s, hipSetDevice(s->device));
    S2R_HIP(s, hipStreamSynchronize(s->stream));
    return S2R_OK;
}

int pool_launch(s2r_synth *s, uint32_t sample_rate, uint32_t first_seq) {
    S2R_HIP(s, hipSetDevice(s->device));
    static thread_local S2rRenderArgs a;
    S2rRenderParams &p = a.p;
    bool bank_kernel = false;
    { int rc = ensure_bank(s, sample_rate, s->stream, &bank_kernel); if (rc != S2R_OK) return rc; }
    p = make_params(s, s->cfg.max_frames, sample_rate);          // (p.frames: the longest fill, sizes the staging)
    { int rc = ensure_tables(s, p, sample_rate, s->stream); if (rc != S2R_OK) return rc; }
    if (tables_wanted(s)) p.tab = s->tab;
    else if (bank_kernel) { p.tab = S2rTabRef{}; p.tab.base = s->bank_tab_dev; }      // the per-lane-patch kernel adds each entry's tab_off
    p.voice_ev_head = s->voice_ev_head;
    p.stamps = s->stamps_dev;                                    // (diagnostic builds: tools/stamps_pool.py)
    p.tev = s->tev_copy;                                         // (MODE 2: a fill's events come as chains)
    a.n_events = 0;
    S2rPool pl{};
    pl.cmd = s->pool_cmd_dev; pl.slices = s->pool_slices_dev; pl.slices_stride = s->n_blocks + 1u;
    for (int k = 0; k < 4; k++) pl.tev_src[k] = s->slots[k].tdev;
    for (int b = 0; b < 2; b++) { pl.tev_copy[b] = s->tevcopy2[b]; pl.heads[b] = s->heads2[b]; pl.partials[b] = s->partials2[b]; }
    pl.arrive = s->fz_arrive;
    if (s->ov_words) { pl.ov_heads = s->ov_words->heads_done; pl.ov_render = s->ov_words->render_done; }
    pl.out[0] = s->ring_dev[0]; pl.out[1] = s->ring_dev[1]; pl.out[2] = s->out_host_dev;
    pl.done_flag = s->done_dev; pl.done_counter = s->done_counter;
    pl.decided = s->pool_decided; pl.exited = s->pool_host_dev; pl.fail = s->done_dev + 3;
    pl.granules = s->res_gran_dev;
    pl.launch_id = ++s->pool_launch_id; pl.first_seq = first_seq;
    pl.idle_ticks = s->pool_idle_ticks; pl.max_polls = 1u << 24;
    s2r_synth *par = s->parent;
    pl.mt = mix_tail_of(s, par == nullptr);
    if (par) {                                                   // a shard of a device list: its row, the count, and — the first shard — the sum
        uint32_t k = 0;
        while (k < par->kids.size() && par->kids[k] != s) k++;
        pl.mt.rows_done = par->rows_done;
        pl.mt.n_rows = (uint32_t)par->kids.size(); pl.mt.row_stride = par->cfg.max_frames;
        for (int b = 0; b < 2; b++) { pl.rows[b] = par->rows_dev[b]; pl.rows_mine[b] = par->rows_dev[b] + (size_t)k * par->cfg.max_frames; }
        pl.final_out[0] = par->ring_dev[0]; pl.final_out[1] = par->ring_dev[1]; pl.final_out[2] = par->out_host_dev;
        pl.final_flag = par->done_dev; pl.final_counter = nullptr;
    }
    if (s->xg_on) {                                               // a rank of a group of processes
        pl.mt = mix_tail_of(s, false);
        pl.mt.xmode = s->xg_rank == 0 ? 1 : 2;
        pl.mt.rows_done = s->xg_done; pl.mt.n_rows = s->xg_n; pl.mt.row_stride = s->cfg.max_frames;
        for (int b = 0; b < 2; b++) { pl.rows[b] = s->xg_rows + (size_t)b * s->xg_n * s->cfg.max_frames; pl.rows_mine[b] = pl.rows[b] + (size_t)s->xg_rank * s->cfg.max_frames; }
        pl.final_out[0] = s->ring_dev[0]; pl.final_out[1] = s->ring_dev[1]; pl.final_out[2] = s->out_host_dev;
        pl.final_flag = s->done_dev; pl.final_counter = nullptr;
    }
    S2R_HIP(s, hipMemsetD32Async((hipDeviceptr_t)s->pool_decided, (int)((first_seq - 1u) << 1), 1, s->stream));
    S2R_HIP(s, s2r_launch_pool(a, pl, s->block_voices, s->stream));
    s->pool_running = true; s->pool_rate = sample_rate;
    return S2R_OK;
}

// The kernel has left with commands unexecuted (it ran out of patience just as one was posted): start it again in front of them.
int pool_recover(s2r_synth *s) {
    if (!s->pool_running || !pool_exited(s)) return S2R_OK;
    s->pool_running = false;
    const bool dbg = std::getenv("S2R_DEBUG_STUCK") != nullptr;
    if (dbg) std::fprintf(stderr, "[s2r rank %u] pool_recover: kernel %u has left; pool_seq %u\n", s->xg_rank, s->pool_launch_id, s->pool_seq);
    S2R_HIP(s, hipSetDevice(s->device));
    S2R_HIP(s, hipStreamSynchronize(s->stream));
    uint32_t d = 0;
    S2R_HIP(s, hipMemcpy(&d, s->pool_decided, sizeof d, hipMemcpyDeviceToHost));
    const uint32_t first = (d & 1u) ? (d >> 1) : (d >> 1) + 1u;  // (left at command d >> 1, or — no bail recorded — after it)
    if (dbg) std::fprintf(stderr, "[s2r rank %u] pool_recover: decided %u -> first pending %u\n", s->xg_rank, d, first);
    if ((int32_t)(s->pool_seq - first) < 0) return S2R_OK;        // nothing was pending
    const int rc = pool_launch(s, s->pool_rate, first);
    if (dbg) std::fprintf(stderr, "[s2r rank %u] pool_recover: launched again (%d)\n", s->xg_rank, rc);
    return rc;
}

// One fill through the pool-resident kernel: the events grouped by workgroup, the bounds, the command.  `sel`: where the output
// goes (0, 1 the ring slots, 2 the synchronous buffer); xc: the shard's part in a device list's exchange.
int pool_fill(s2r_synth *s, size_t frames, uint32_t sample_rate, uint32_t sel, bool stereo, uint32_t done_value, const Exchange *xc, uint32_t rows_slot,
              bool stage_only = false) {
    { int rc = overlap_drain(s); if (rc != S2R_OK) return rc; }
    { int rc = launch_deferred_mix(s, s->stream); if (rc != S2R_OK) return rc; }
    if (s->pool_running && (s->pool_rate != sample_rate || pool_exited(s) || s->pending.size() + s->tpending.size() > s->tev_capacity)) {
        int rc = pool_exited(s) ? pool_recover(s) : S2R_OK;
        if (rc == S2R_OK) rc = pool_stop(s);
        if (rc != S2R_OK) return rc;
    }
    EventSlot *slot = nullptr;
    uint32_t nt = 0;
    { int rc = fused_events(s, s->stream, &slot, &nt); if (rc != S2R_OK) return rc; }
    const uint32_t par = s->pool_fills++ & 1u;
    s->fz_target[par] += s->n_blocks;
    uint32_t w[16] = {0};
    w[1] = (uint32_t)frames; w[2] = nt; w[3] = done_value; w[4] = sel | (stereo ? 256u : 0u);
    w[5] = slot ? (uint32_t)(slot - s->slots) : 0u; w[6] = par; w[7] = s->fz_target[par]; w[8] = pick_mixers(s, frames);
    w[9] = xc ? xc->rows_target : 0u; w[10] = rows_slot;
    // a short synchronous fill of a single device: every frame comes back as one tagged 8-byte word — no completion word, no wait
    // for a store acknowledged across the link (as the one-workgroup resident kernel's)
    s->pool_gran_pending = sel == 2u && !xc && frames <= S2R_RES_GRANULE_FRAMES && s->res_gran != nullptr;
    w[12] = s->pool_gran_pending ? 1u : 0u;
    const uint32_t next_seq = s->pool_seq + 1u;
    std::memcpy(s->pool_slices + (size_t)(next_seq % S2R_POOL_CMD_SLOTS) * (s->n_blocks + 1u), s->tbounds.data(), ((size_t)s->n_blocks + 1u) * sizeof(uint32_t));
    std::atomic_thread_fence(std::memory_order_release);
    uint32_t seq = 0;
    if (stage_only && s->pool_running) { s->pool_staged = pool_post_payload(s, w, &seq); s->pool_staged_seq = seq; }
    else seq = pool_post(s, w);
    if (!s->pool_running) { int rc = pool_launch(s, sample_rate, seq); if (rc != S2R_OK) return rc; }
    s->pool_gran_slot = nullptr;
    if (slot) {
        slot->state = 2; slot->seq = done_value;
        slot->word = (xc && xc->slot_word) ? xc->slot_word : s->done_host + sel;
        // (a fill that comes back as granules writes no completion word: fill_host frees the slot when the frames are in)
        if (s->pool_gran_pending) s->pool_gran_slot = slot;
    }
    if (!s->parent) {
        s->pool->advance(frames - s->fill_time);
        s->fill_time = 0;
    }
    return S2R_OK;
}

// events -> render -> (mix) on `stream`; the partial or final mix lands in `dev_out`
// `defer_ring_slot` >= 0 (s2r_fill_begin): the fill's mix is left to the next fill_begin / fill_end (DeferredMix)
int enqueue_fill(s2r_synth *s, size_t frames, uint32_t sample_rate, hipStream_t stream, float *dev_out,
                 bool root_add, bool stereo, float *per_voice_dev, int defer_ring_slot = -1, const S2rDone *done = nullptr,
                 const Exchange *xc = nullptr, bool via_pool = false) {
    // one launch per fill where the shard's shape allows it (the fills of s2r_fill_begin keep the two streams unless asked)
    if (dev_out != nullptr && per_voice_dev == nullptr && fused_shape_ok(s) &&
        (xc != nullptr || s->fused_mode >= 2 || !(s->ov_enabled && defer_ring_slot >= 0 && stream == s->stream && root_add && done != nullptr)))
        return enqueue_fused(s, frames, sample_rate, stream, dev_out, root_add, stereo, done, xc);
    if (s->dmix.active && stream != s->stream) {
        // a fill on a caller's stream behind one in flight on ours: that one's mix reads the partial rows this fill
        // is about to overwrite
        int rc = launch_deferred_mix(s, s->stream);
        if (rc != S2R_OK) return rc;
        S2R_HIP(s, hipStreamSynchronize(s->stream));
    }
    // Two streams (S2rOverlapWords): the fills of s2r_fill_begin on our own stream, rendered into partial rows that a mix
    // will add up.  Anything else first waits for the overlapped fills before it.
    const bool overlap = s->ov_enabled && defer_ring_slot >= 0 && stream == s->stream && dev_out != nullptr && root_add && s->n_blocks > 1 &&
                         per_voice_dev == nullptr && done != nullptr;
    const int ov_parity = overlap ? (int)(s->ov_fill & 1u) : -1;
    if (!overlap) { int rc = overlap_drain(s); if (rc != S2R_OK) return rc; }
    else if (s->dmix.active && !s->dmix.overlap) { int rc = launch_deferred_mix(s, stream); if (rc != S2R_OK) return rc; }
    // The common case — a handful of untimed events without seed overrides — needs no launch for them: they ride in
    // the render kernel's arguments and every wave applies the ones that hit its voices before it loads its state.
    bool arg_events = s->use_arg_events && s->tpending.empty() && s->pending.size() <= S2R_ARG_MAX_EVENTS;
    if (arg_events)
        for (const S2rVoiceEvent &e : s->pending) if (e.seed != 0u) { arg_events = false; break; }
    via_pool = via_pool && overlap;
    if (via_pool) {
        // The fill goes to the pool-resident kernel in its two-stream form: the render launch becomes a posted command, the
        // chain heads and the mix stay the other stream's launch.  Every event comes as a chain (no kernel arguments to ride in).
        if (s->pool_running && (s->pool_rate != sample_rate || pool_exited(s) || s->pending.size() + s->tpending.size() > s->tev_capacity)) {
            int rc = pool_exited(s) ? pool_recover(s) : S2R_OK;
            if (rc == S2R_OK) rc = pool_stop(s);
            if (rc != S2R_OK) return rc;
        }
        arg_events = false;
        merge_pending_into_chains(s);
    }
    EventSlot *timed_slot = nullptr;
    const S2rTimedEvent *tev_dev = nullptr;
    if (!arg_events) {
        int rc = flush_events(s, stream, &timed_slot, &tev_dev, done, ov_parity);
        if (rc != S2R_OK) return rc;
    }
    {   // (no chain heads in this fill to take the previous fill's mix along: it goes alone, before the render kernel
        // overwrites the partial rows)
        int rc = launch_deferred_mix(s, stream);
        if (rc != S2R_OK) return rc;
    }
    bool bank_kernel = false;
    { int rc = ensure_bank(s, sample_rate, stream, &bank_kernel); if (rc != S2R_OK) return rc; }
    static thread_local S2rRenderArgs a;             // 4 KiB of kernel arguments, copied by the launch
    S2rRenderParams &p = a.p;
    p = make_params(s, frames, sample_rate);
    {
        int rc = ensure_tables(s, p, sample_rate, stream);
        if (rc != S2R_OK) return rc;
    }
    if (tables_wanted(s)) p.tab = s->tab;
    else if (bank_kernel) { p.tab = S2rTabRef{}; p.tab.base = s->bank_tab_dev; }      // the per-lane-patch kernel adds each entry's tab_off
    p.stamps = s->stamps_dev;
    if (s->timeline_dev && s->timeline_n < s->timeline_cap) { p.timeline = s->timeline_dev; p.tl_slot = s->timeline_n++; }
    p.per_voice = per_voice_dev;
    p.tev = tev_dev;
    p.voice_ev_head = s->voice_ev_head;
    if (overlap) {
        p.block_partials = s->partials2[ov_parity];
        p.voice_ev_head = s->heads2[ov_parity];
        if (tev_dev) { p.ov_heads_counter = &s->ov_words->heads_done[ov_parity]; p.ov_heads_target = s->ov_heads_target[ov_parity]; }
        p.ov_render_counter = &s->ov_words->render_done[ov_parity];
        p.ov_fail = s->done_dev + 3;
        s->ov_render_target[ov_parity] += s->n_blocks;       // (taken back below if the render launch fails)
    }
    a.n_events = 0;
    if (arg_events) {
        a.n_events = (uint32_t)s->pending.size();
        for (uint32_t i = 0; i < a.n_events; i++) {
            const S2rVoiceEvent &e = s->pending[i];
            a.ev[3u * i] = e.voice; a.ev[3u * i + 1u] = e.flags; a.ev[3u * i + 2u] = s2r_f2u(e.pitch);
        }
        for (const S2rVoiceEvent &e : s->pending) s->pending_slot[e.voice] = -1;
        s->pending.clear();
    }
    // a shard of one workgroup needs no mix launch: its only partial row, root-added, is the output
    const bool direct = dev_out != nullptr && root_add && s->n_blocks == 1;
    if (direct) { p.direct_out = dev_out; p.direct_stereo = stereo ? 1 : 0; if (done) p.done = *done; }
    if (via_pool) {
        uint32_t w[16] = {0};
        w[1] = (uint32_t)frames | (S2R_POOL_FLAG_TWO_STREAMS << 16); w[2] = tev_dev ? 1u : 0u; w[6] = (uint32_t)ov_parity;
        w[11] = s->ov_heads_target[ov_parity];
        const uint32_t seq = pool_post(s, w);
        if (!s->pool_running) { int rc = pool_launch(s, sample_rate, seq); if (rc != S2R_OK) return rc; }
    } else {
    if (s->timing) S2R_HIP(s, hipEventRecord(s->t0, stream));      // brackets the render kernel alone
    if (s2r_launch_render(a, s->block_voices, stream) != hipSuccess) {
        if (overlap) { s->ov_render_target[ov_parity] -= s->n_blocks; s->broken = true; }
        return set_err(s, S2R_ERR_HIP, "the render kernel's launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    if (s->timing) { S2R_HIP(s, hipEventRecord(s->t1, stream)); s->timed = true; }
    }
    if (timed_slot) {                     // the render kernel was the last reader of the slot's records
        if (done) { timed_slot->state = 2; timed_slot->word = s->done_host + (done->flag - s->done_dev); timed_slot->seq = done->value; }
        else { S2R_HIP(s, hipEventRecord(timed_slot->done, stream)); timed_slot->state = 1; }
    }
    if (dev_out && !direct) {
        S2rMixParams m{};
        m.block_partials = s->block_partials;
        m.n_blocks = s->n_blocks;
        m.n_groups = root_add ? s->mix_groups : 1u;
        m.blocks_per_group = (s->n_blocks + m.n_groups - 1) / m.n_groups;
        m.frames = (uint32_t)frames;
        m.frames_stride = partials_stride(s->cfg.max_frames);
        m.root_add = root_add ? 1 : 0;
        m.stereo = stereo ? 1 : 0;
        m.out = dev_out;
        if (done) m.done = *done;
        if (s->timeline_dev && s->timeline_n < s->timeline_cap) { m.timeline = s->timeline_dev; m.tl_slot = s->timeline_n++; }
        if (overlap) {
            m.block_partials = s->partials2[ov_parity];
            m.ov_render_counter = &s->ov_words->render_done[ov_parity];
            m.ov_render_target = s->ov_render_target[ov_parity];
            m.ov_fail = s->done_dev + 3;
            s->ov_fill++;
            s->ov_busy = true;
        }
        if (defer_ring_slot >= 0 && stream == s->stream) { s->dmix.active = true; s->dmix.overlap = overlap; s->dmix.m = m; s->dmix.ring_slot = defer_ring_slot; }
        else S2R_HIP(s, s2r_launch_mix(m, stream));
    }
    if (!s->parent) {                             // (a device-list handle moves the shared clock once, after its shards)
        s->pool->advance(frames - s->fill_time);
        s->fill_time = 0;
    }
    return S2R_OK;
}


// A device-list handle's fill: every shard renders on its own device and stream and leaves its partial mix in row k of rows_dev[slot] on the parent's device; the parent's
// stream waits for the rows and adds them in shard order rooted at +0.0 (synth.rs:176,195) into `dev_out`.
int enqueue_multi(s2r_synth *s, size_t frames, uint32_t sample_rate, float *dev_out, bool stereo, float *per_voice_host = nullptr,
                  const S2rDone *done = nullptr, int rows_slot_in = -1) {
    const uint32_t n = (uint32_t)s->kids.size();
    // (the rows of a fill of s2r_fill_begin belong to its ring slot; every other fill finds no fill in flight — fill_host and
    // the other synchronous calls refuse a device-list handle otherwise — and takes the slot the next ring fill will not)
    const uint32_t slot = rows_slot_in >= 0 ? (uint32_t)rows_slot_in : ((s->ring_head + s->ring_count) & 1u) ^ 1u;
    // One launch per shard and nothing else (S2rMixTail's exchange): every shard's last mixer counts in on a word in the
    // parent's memory, and the shard that counts in last adds the rows.  Needs a
    // completion word to end the fill with (a consumer ordered by the parent's STREAM takes the launches below) and shards
    // that write their rows where they are read.
    bool exchange = done != nullptr && done->flag != nullptr && !per_voice_host && s->rows_done != nullptr;
    for (uint32_t k = 0; k < n && exchange; k++) exchange = fused_shape_ok(s->kids[k]) && s->kid_stage[k] == nullptr;
    if (exchange) s->rows_target[slot] += n;
    else { int rc = pool_stop(s); if (rc != S2R_OK) return rc; }
    auto shard_job = [s, frames, sample_rate, slot, per_voice_host, exchange, n, dev_out, stereo, done](uint32_t k) -> int {
        s2r_synth *kid = s->kids[k];
        if (hipSetDevice(kid->device) != hipSuccess) return set_err(kid, S2R_ERR_HIP, "hipSetDevice(%d) failed", kid->device);
        if (per_voice_host) {                     // (mix disabled: the shard's rows, scattered to pool order by the caller)
            const size_t need = (size_t)kid->shard_voices * frames;
            if (need > kid->per_voice_cap) {
                if (kid->per_voice_dev) { S2R_HIP(kid, hipFree(kid->per_voice_dev)); kid->per_voice_dev = nullptr; kid->per_voice_cap = 0; }
                S2R_HIP(kid, hipMalloc((void **)&kid->per_voice_dev, need * sizeof(float)));
                kid->per_voice_cap = need;
            }
            int rc = enqueue_fill(kid, frames, sample_rate, kid->stream, nullptr, false, false, kid->per_voice_dev);
            if (rc != S2R_OK) return rc;
            return S2R_OK;
        }
        float *row = s->rows_dev[slot] + (size_t)k * s->cfg.max_frames;
        if (exchange) {
            Exchange xc;
            xc.rows_done = s->rows_done + slot; xc.rows_target = s->rows_target[slot];
            xc.n_rows = n; xc.row_stride = s->cfg.max_frames; xc.rows = s->rows_dev[slot];
            xc.final_out = dev_out; xc.final_done = *done; xc.final_stereo = stereo;
            xc.slot_word = s->done_host + (done->flag - s->done_dev);
            if (pool_eligible(kid, frames))
                return pool_fill(kid, frames, sample_rate, (uint32_t)(done->flag - s->done_dev), stereo, done->value, &xc, slot, true);
            { int rc = pool_stop(kid); if (rc != S2R_OK) return rc; }
            const S2rDone kd{nullptr, 0u, kid->done_counter + slot};
            return enqueue_fill(kid, frames, sample_rate, kid->stream, row, false, false, nullptr, -1, &kd, &xc);
        }
        float *dst = s->kid_stage[k] ? s->kid_stage[k] : row;
        // (the shard's own completion word: its event slots are then tracked without an event record per fill)
        const S2rDone kd{kid->done_dev + slot, ++kid->done_seq, kid->done_counter + slot};
        int rc = enqueue_fill(kid, frames, sample_rate, kid->stream, dst, false, false, nullptr, -1, &kd);
        if (rc != S2R_OK) return rc;
        if (s->kid_stage[k]) S2R_HIP(kid, hipMemcpyPeerAsync(row, s->device, dst, kid->device, frames * sizeof(float), kid->stream));
        S2R_HIP(kid, hipEventRecord(s->kid_done[slot][k], kid->stream));
        return S2R_OK;
    };
    // The shards are launched by the calling thread, one after the other.  (One host thread per shard was built and
    // measured: HIP serialises the launches of a process — fill_begin 32 / 50 / 80 us on threads against 40 / 76 / 130 us
    // in sequence at 2 / 4 / 8 shards — while the threads cost the caller's own event processing more than that:
    // tools/devlist_host_cost.py.)
    int rc = S2R_OK;
    for (uint32_t k = 0; k < n && rc == S2R_OK; k++) { rc = shard_job(k); if (rc != S2R_OK) s->err = "shard " + std::to_string(k) + ": " + s->kids[k]->err; }
    if (rc != S2R_OK) return rc;
    s->pool->advance(frames - s->fill_time);      // the shared clock, once
    s->fill_time = 0;
    if (exchange) {
        // the staged commands of the resident shards: their sequence words behind shared fences (pool_post_payload)
        bool any = false, vram = false;
        for (s2r_synth *kid : s->kids) if (kid->pool_staged) { any = true; vram = vram || kid->pool_cmd_vram; }
        if (any) {
            if (vram) store_fence();
            for (s2r_synth *kid : s->kids) if (kid->pool_staged) __atomic_store_n(&kid->pool_staged[15], kid->pool_staged_seq, __ATOMIC_RELEASE);
            if (vram) store_fence();
            for (s2r_synth *kid : s->kids) if (kid->pool_staged) { __atomic_store_n(&kid->pool_staged[0], kid->pool_staged_seq, __ATOMIC_RELEASE); kid->pool_staged = nullptr; }
            if (vram) store_fence();
        }
    }
    if (per_voice_host || exchange) return S2R_OK;
    S2R_HIP(s, hipSetDevice(s->device));
    for (uint32_t k = 0; k < n; k++) S2R_HIP(s, hipStreamWaitEvent(s->stream, s->kid_done[slot][k], 0));
    S2R_HIP(s, s2r_launch_sum_rows(s->rows_dev[slot], n, (uint32_t)frames, s->cfg.max_frames, stereo ? 1 : 0, dev_out, s->stream, done));
    return S2R_OK;
}

// Waits for the fill whose last kernel stores `seq` into completion word `idx`: polls the word (mapped host memory) for
// a bounded time, then falls back to the stream — an error on the device never leaves the caller spinning.
int wait_done(s2r_synth *s, uint32_t idx, uint32_t seq) {
    volatile uint32_t *f = s->done_host + idx;
    // the spin is bounded by wall time (a fill of a big pool lasts milliseconds: no core is burnt for that long), then the
    // stream decides
    timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        for (int i = 0; i < 2000; i++) {
            if ((int32_t)(*f - seq) >= 0) { std::atomic_thread_fence(std::memory_order_acquire); return S2R_OK; }
            cpu_relax();
        }
        // a pool-resident kernel that left just as its command was posted is started again in front of it
        if (s->pool_running && pool_exited(s)) { int rc = pool_recover(s); if (rc != S2R_OK) return rc; }
        for (s2r_synth *kid : s->kids) if (kid->pool_running && pool_exited(kid)) { int rc = pool_recover(kid); if (rc != S2R_OK) { s->err = kid->err; return rc; } }
        timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
        const double us = (double)(t1.tv_sec - t0.tv_sec) * 1e6 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-3;
        const bool pool = s->pool_running || (!s->kids.empty() && s->kids[0]->pool_running);
        if (us > (pool ? 200000.0 : 300.0)) {                    // (a resident kernel never ends a stream wait: only the word tells)
            if (std::getenv("S2R_DEBUG_STUCK"))
                std::fprintf(stderr, "[s2r rank %u] wait_done(idx %u, seq %u): word %u after %.0f us; pool_running %d exited %d launch_id %u pool_host[0] %u pool_seq %u "
                             "done_host %u %u %u %u ring_count %u\n", s->xg_rank, idx, seq, *f, us, (int)s->pool_running, (int)(s->pool_running && pool_exited(s)),
                             s->pool_launch_id, s->pool_host ? s->pool_host[0] : 0u, s->pool_seq, s->done_host[0], s->done_host[1], s->done_host[2], s->done_host[3], s->ring_count);
            break;
        }
    }
    if (s->pool_running || (!s->kids.empty() && s->kids[0]->pool_running)) {
        if (std::getenv("S2R_DEBUG_STUCK")) std::fprintf(stderr, "[s2r rank %u] wait_done: stopping the pool-resident kernel\n", s->xg_rank);
        (void)pool_stop(s);
        if (std::getenv("S2R_DEBUG_STUCK")) std::fprintf(stderr, "[s2r rank %u] wait_done: stopped; word %u\n", s->xg_rank, *f);
        if ((int32_t)(*f - seq) < 0) return set_err(s, S2R_ERR_HIP, "the pool-resident kernel did not report the fill");
        return S2R_OK;
    }
    for (s2r_synth *kid : s->kids) { S2R_HIP(s, hipSetDevice(kid->device)); S2R_HIP(s, hipStreamSynchronize(kid->stream)); }
    S2R_HIP(s, hipSetDevice(s->device));
    S2R_HIP(s, hipStreamSynchronize(s->stream));
    if (s->stream_b) S2R_HIP(s, hipStreamSynchronize(s->stream_b));
    if ((int32_t)(*f - seq) < 0) return set_err(s, S2R_ERR_HIP, "the fill's last kernel finished without signalling completion");
    return S2R_OK;
}

// (two streams) a kernel that gave up waiting for the other stream's work says so in done_host[3]
int overlap_check(s2r_synth *s) {
    uint32_t who = *(volatile uint32_t *)(s->done_host + 3);
    s->done_host[3] = 0u;
    for (s2r_synth *kid : s->kids) { const uint32_t w = *(volatile uint32_t *)(kid->done_host + 3); kid->done_host[3] = 0u; if (w) who = w; }
    if (who == 0u) return S2R_OK;
    s->broken = true;
    return set_err(s, S2R_ERR_HIP, "%s gave up waiting for another kernel's or workgroup's work: the fill's output is not valid",
                   who == 1u ? "a render kernel (for its chain heads)" : who == 2u ? "a mix (for its partial rows)" : "the sum of the shards' rows (for a shard's row)");
}

// the fill of any handle on ITS stream: the final mix (root-added) lands in `dev_out`
int enqueue_root(s2r_synth *s, size_t frames, uint32_t sample_rate, float *dev_out, bool stereo, int defer_ring_slot = -1,
                 const S2rDone *done = nullptr) {
    if (!s->kids.empty()) return enqueue_multi(s, frames, sample_rate, dev_out, stereo, nullptr, done, defer_ring_slot);
    if (s->xg_on) {
        // One process per GPU: this rank's partial row goes into the root's block and the rows are added by the root's last
        // mixer, all inside the render kernels (S2rMixTail.xmode 1 / 2) — no collective, no call into another library per step.
        if (!(done && done->flag) || !fused_shape_ok(s))
            return set_err(s, S2R_ERR_INVALID, "a handle in a process group fills through s2r_fill / s2r_fill_begin, with more than one workgroup (of at most 256 voices unless the patch is a single one-pole one)");
        const uint32_t slot = defer_ring_slot >= 0 ? (uint32_t)defer_ring_slot : ((s->ring_head + s->ring_count) & 1u) ^ 1u;
        Exchange xc;
        s->xg_target[slot] += s->xg_n;
        xc.xmode = s->xg_rank == 0 ? 1 : 2;
        xc.rows_done = s->xg_done + slot; xc.rows_target = s->xg_target[slot];
        xc.n_rows = s->xg_n; xc.row_stride = s->cfg.max_frames; xc.rows = s->xg_rows + (size_t)slot * s->xg_n * s->cfg.max_frames;
        xc.final_out = dev_out; xc.final_done = *done; xc.final_stereo = stereo;
        xc.slot_word = s->done_host + (done->flag - s->done_dev);
        float *row = s->xg_rows + ((size_t)slot * s->xg_n + s->xg_rank) * s->cfg.max_frames;
        if (pool_eligible(s, frames)) return pool_fill(s, frames, sample_rate, (uint32_t)(done->flag - s->done_dev), stereo, done->value, &xc, slot);
        { int rc = pool_stop(s); if (rc != S2R_OK) return rc; }
        return enqueue_fill(s, frames, sample_rate, s->stream, row, false, false, nullptr, -1, done, &xc);
    }
    if (done && done->flag && pool_eligible(s, frames)) {
        // (S2R_POOL_FORM=fused: every fill in the one-launch form, the chain heads and the mix in the render kernel itself)
        static const bool fused_only = [] { const char *e = std::getenv("S2R_POOL_FORM"); return e && e[0] == 'f'; }();
        if (s->ov_enabled && defer_ring_slot >= 0 && !stereo && !fused_only)
            return enqueue_fill(s, frames, sample_rate, s->stream, dev_out, true, stereo, nullptr, defer_ring_slot, done, nullptr, true);
        return pool_fill(s, frames, sample_rate, (uint32_t)(done->flag - s->done_dev), stereo, done->value, nullptr, 0u);
    }
    { int rc = pool_stop(s); if (rc != S2R_OK) return rc; }
    return enqueue_fill(s, frames, sample_rate, s->stream, dev_out, true, stereo, nullptr, defer_ring_slot, done);
}

// ---- the resident kernel (s2r_set_low_latency) ----

void resident_post(s2r_synth *s, uint32_t seq) {                // payload first, then word 31, then word 0 (S2rResident)
    volatile uint32_t *c = s->res_cmd;
    // (device memory behind the BAR is write-combining: the fences are what orders the three stages on the link)
    if (s->res_cmd_vram) store_fence();
    __atomic_store_n(&c[31], seq, __ATOMIC_RELEASE);
    if (s->res_cmd_vram) store_fence();
    __atomic_store_n(&c[0], seq, __ATOMIC_RELEASE);
    if (s->res_cmd_vram) store_fence();
}

bool resident_exited(const s2r_synth *s) { return __atomic_load_n(&s->res_host[32], __ATOMIC_ACQUIRE) == s->res_launch_id; }

// Ends the resident kernel, if there is one, and waits for it: the stream is the caller's again.
int resident_stop(s2r_synth *s) {
    if (!s || !s->res_running) return S2R_OK;
    volatile uint32_t *c = s->res_cmd;
    c[1] = S2R_RES_FLAG_EXIT << 16; c[2] = 0; c[3] = 0;
    resident_post(s, ++s->res_seq);
    s->res_running = false;
    S2R_HIP(s, hipSetDevice(s->device));
    S2R_HIP(s, hipStreamSynchronize(s->stream));
    return S2R_OK;
}

// A fill the resident kernel can take: one workgroup of the one-pole kernel, a handful of untimed events, nothing else
// in flight on the stream.
bool resident_eligible(const s2r_synth *s, size_t frames) {
    if (!s->low_latency || s->parent || !s->kids.empty() || s->n_blocks != 1 || s->block_voices > 256u) return false;
    if (s->bank.size() != 1 || s->bank[0].osc_kind > S2R_OSC_SINE || s->bank[0].lpf_kind != S2R_FILT_ONEPOLE) return false;
    if (!s->use_arg_events || !s->tpending.empty() || s->fill_time != 0 || s->pending.size() > S2R_RES_MAX_EVENTS) return false;
    if (s->timing || s->dmix.active || s->ring_count != 0 || s->timeline_dev || frames > 0xffffu) return false;
    for (const S2rVoiceEvent &e : s->pending) if (e.seed != 0u) return false;
    return true;
}

int resident_launch(s2r_synth *s, uint32_t sample_rate, bool stereo) {
    S2R_HIP(s, hipSetDevice(s->device));
    static thread_local S2rRenderArgs a;
    S2rRenderParams &p = a.p;
    p = make_params(s, s->cfg.max_frames, sample_rate);          // (p.frames: the longest fill, sizes the staging)
    { int rc = ensure_tables(s, p, sample_rate, s->stream); if (rc != S2R_OK) return rc; }
    if (tables_wanted(s)) p.tab = s->tab;
    p.voice_ev_head = s->voice_ev_head;
    p.direct_out = s->out_host_dev; p.direct_stereo = stereo ? 1 : 0;
    p.stamps = s->stamps_dev;                                    // (diagnostic builds: tools/stamps_small.py)
    a.n_events = 0;
    S2rResident rs{};
    rs.cmd = s->res_cmd_dev; rs.exited = s->res_dev + 32;
    rs.launch_id = ++s->res_launch_id;
    rs.first_seq = s->res_seq + 1u;
    rs.idle_ticks = 100000u;                                     // 1 ms without a command (real-time callers come every 0.33 ms)
    rs.max_polls = 1u << 20;
    rs.done_flag = s->done_dev + 2; rs.done_counter = s->done_counter + 2;
    rs.granules = s->res_gran_dev;
    S2R_HIP(s, s2r_launch_resident(a, rs, s->block_voices, s->stream));
    s->res_running = true; s->res_rate = sample_rate; s->res_stereo = stereo;
    return S2R_OK;
}

int resident_fill(s2r_synth *s, float *out, size_t frames, uint32_t sample_rate, bool stereo) {
    if (s->res_running && (s->res_rate != sample_rate || s->res_stereo != stereo || resident_exited(s))) {
        int rc = resident_stop(s);
        if (rc != S2R_OK) return rc;
    }
    if (!s->res_running) { int rc = resident_launch(s, sample_rate, stereo); if (rc != S2R_OK) return rc; }
    volatile uint32_t *c = s->res_cmd;
    const uint32_t n = (uint32_t)s->pending.size();
    const uint32_t done_value = ++s->done_seq;
    c[1] = (uint32_t)frames; c[2] = n; c[3] = done_value;
    for (uint32_t i = 0; i < n; i++) {
        const S2rVoiceEvent &e = s->pending[i];
        c[4u + 3u * i] = e.voice; c[5u + 3u * i] = e.flags; c[6u + 3u * i] = s2r_f2u(e.pitch);
    }
    for (const S2rVoiceEvent &e : s->pending) s->pending_slot[e.voice] = -1;
    s->pending.clear();
    resident_post(s, ++s->res_seq);
    s->pool->advance(frames);
    // The fill's end: the completion word — or, for a short fill, the tags of its granules (every frame's 8-byte word
    // carries the fill's value above the sample).  A kernel that left (idle for too long) just before the command reached
    // it is started again.
    volatile uint32_t *f = s->done_host + 2;
    const bool gran = frames <= S2R_RES_GRANULE_FRAMES && s->res_gran != nullptr;
    volatile unsigned long long *g = s->res_gran;
    auto arrived = [&]() -> bool {
        if (!gran) return (int32_t)(*f - done_value) >= 0;
        for (size_t i = frames; i-- > 0;) if ((uint32_t)(g[i] >> 32) != done_value) return false;     // (the last frame is the last to come)
        return true;
    };
    bool done = false;
    for (int attempt = 0; attempt < 3 && !done; attempt++) {
        for (int round = 0; round < 4000 && !done; round++) {
            for (int i = 0; i < 1000; i++) {
                if (arrived()) { done = true; break; }
                cpu_relax();
            }
            if (!done && resident_exited(s)) break;
        }
        if (done || arrived()) { done = true; break; }
        if (!resident_exited(s)) break;                          // neither finished nor gone: the stream decides below
        s->res_running = false;
        S2R_HIP(s, hipSetDevice(s->device));
        S2R_HIP(s, hipStreamSynchronize(s->stream));
        if (arrived()) { done = true; break; }
        int rc = resident_launch(s, sample_rate, stereo);        // (first_seq = the command's successor: post it again)
        if (rc != S2R_OK) return rc;
        resident_post(s, ++s->res_seq);
    }
    if (!done) {
        (void)resident_stop(s);
        if (!arrived()) return set_err(s, S2R_ERR_HIP, "the resident kernel did not report the fill");
    }
    if (gran) {
        for (size_t i = 0; i < frames; i++) {
            const float v = s2r_u2f((uint32_t)g[i]);
            if (stereo) { out[2 * i] = v; out[2 * i + 1] = v; } else out[i] = v;
        }
        return S2R_OK;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    std::memcpy(out, s->out_host, frames * (stereo ? 2 : 1) * sizeof(float));
    return S2R_OK;
}

int fill_host(s2r_synth *s, float *out, size_t frames, uint32_t sample_rate, bool stereo) {
    int rc = check_fill(s, frames, sample_rate);
    if (rc != S2R_OK) return rc;
    if (frames == 0) return S2R_OK;
    if (!out) return set_err(s, S2R_ERR_INVALID, "null output buffer");
    if (!s->kids.empty() && s->ring_count)
        return set_err(s, S2R_ERR_INVALID, "a device-list handle takes no synchronous fill while fills of s2r_fill_begin are in flight (their rows are the shards'): s2r_fill_end first");
    // (the resident kernel's fills make no HIP call while it runs)
    if (resident_eligible(s, frames)) return resident_fill(s, out, frames, sample_rate, stereo);
    S2R_HIP(s, hipSetDevice(s->device));
    rc = resident_stop(s);
    if (rc != S2R_OK) return rc;
    // the last kernel of the fill writes the few KiB of output straight into mapped host memory: no copy
    // command between the launch and the wait
    const S2rDone done{s->done_dev + 2, ++s->done_seq, s->done_counter + 2};
    rc = enqueue_root(s, frames, sample_rate, s->out_host_dev, stereo, -1, &done);
    if (rc != S2R_OK) return rc;
    const size_t n = frames * (stereo ? 2 : 1);
    if (s->pool_gran_pending) {
        // the frames of a short fill of the pool-resident kernel, each with the fill's tag above the sample
        s->pool_gran_pending = false;
        volatile unsigned long long *g = s->res_gran;
        auto arrived = [&]() -> bool {
            for (size_t i = frames; i-- > 0;) if ((uint32_t)(g[i] >> 32) != done.value) return false;
            return true;
        };
        timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
        bool ok = false;
        for (;;) {
            for (int i = 0; i < 2000 && !ok; i++) { ok = arrived(); if (!ok) cpu_relax(); }
            if (ok) break;
            if (s->pool_running && pool_exited(s)) { int rc2 = pool_recover(s); if (rc2 != S2R_OK) return rc2; }
            timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((double)(t1.tv_sec - t0.tv_sec) * 1e6 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-3 > 200000.0) break;
        }
        if (!ok) { (void)pool_stop(s); if (!arrived()) return set_err(s, S2R_ERR_HIP, "the pool-resident kernel did not report the fill"); }
        if (s->pool_gran_slot) { s->pool_gran_slot->state = 0; s->pool_gran_slot = nullptr; }
        { int rc2 = overlap_check(s); if (rc2 != S2R_OK) return rc2; }
        for (size_t i = 0; i < frames; i++) {
            const float v = s2r_u2f((uint32_t)g[i]);
            if (stereo) { out[2 * i] = v; out[2 * i + 1] = v; } else out[i] = v;
        }
        return S2R_OK;
    }
    rc = wait_done(s, 2, done.value);
    if (rc != S2R_OK) return rc;
    { int rc2 = overlap_check(s); if (rc2 != S2R_OK) return rc2; }
    if (s->xg_on && s->xg_rank != 0) std::memset(out, 0, n * sizeof(float));        // (the mix is the root's)
    else std::memcpy(out, s->out_host, n * sizeof(float));
    return S2R_OK;
}

void release_all(s2r_synth *s) {
    if (!s) return;
    (void)quiesce(s);
    for (s2r_synth *kid : s->kids) release_all(kid);
    s->kids.clear();
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->stream_b) (void)hipStreamSynchronize(s->stream_b);
    for (int b = 0; b < 2; b++) {
        if (s->rows_dev[b]) (void)hipFree(s->rows_dev[b]);
        for (hipEvent_t e : s->kid_done[b]) if (e) (void)hipEventDestroy(e);
    }
    for (float *st : s->kid_stage) if (st) (void)hipFree(st);
    for (EventSlot &sl : s->slots) {
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.thost) (void)hipHostFree(sl.thost);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (s->xg_block) { if (s->xg_owner) (void)hipFree(s->xg_block); else (void)hipIpcCloseMemHandle(s->xg_block); }
    if (s->fz_arrive) (void)hipFree(s->fz_arrive);
    if (s->rows_done) (void)hipFree(s->rows_done);
    if (s->pool_cmd_vram && s->pool_cmd) (void)hipFree(s->pool_cmd);
    if (s->pool_host) (void)hipHostFree(s->pool_host);
    if (s->pool_slices) (void)hipHostFree(s->pool_slices);
    if (s->pool_decided) (void)hipFree(s->pool_decided);
    if (s->voice_mem) (void)hipFree(s->voice_mem);
    if (s->bank_dev) (void)hipFree(s->bank_dev);
    if (s->os_buf) (void)hipFree(s->os_buf);
    if (s->os_taps) (void)hipFree(s->os_taps);
    if (s->block_partials) (void)hipFree(s->block_partials);
    if (s->out_dev) (void)hipFree(s->out_dev);
    if (s->out_host) (void)hipHostFree(s->out_host);
    if (s->done_host) (void)hipHostFree(s->done_host);
    if (s->done_counter) (void)hipFree(s->done_counter);
    for (int k = 0; k < 2; k++) {
        if (s->ring_host[k]) (void)hipHostFree(s->ring_host[k]);
        if (s->ring_done[k]) (void)hipEventDestroy(s->ring_done[k]);
    }
    if (s->sin_dev) (void)hipFree(s->sin_dev);
    if (s->noise_dev) (void)hipFree(s->noise_dev);
    if (s->res_cmd_vram && s->res_cmd) (void)hipFree(s->res_cmd);
    if (s->res_host) (void)hipHostFree(s->res_host);
    if (s->res_gran) (void)hipHostFree(s->res_gran);
    if (s->per_voice_dev) (void)hipFree(s->per_voice_dev);
    if (s->voice_ev_head) (void)hipFree(s->voice_ev_head);
    if (s->tev_copy) (void)hipFree(s->tev_copy);
    if (s->partials2[1]) (void)hipFree(s->partials2[1]);
    if (s->heads2[1]) (void)hipFree(s->heads2[1]);
    if (s->tevcopy2[1]) (void)hipFree(s->tevcopy2[1]);
    if (s->ov_words) (void)hipFree(s->ov_words);
    if (s->ov_registered && s->device < 64) g_two_stream_handles[s->device].store(0);
    if (s->stream_b) (void)hipStreamDestroy(s->stream_b);
    if (s->tab_dev) (void)hipFree(s->tab_dev);
    if (s->stamps_dev) (void)hipFree(s->stamps_dev);
    if (s->timeline_dev) (void)hipFree(s->timeline_dev);
    if (s->bank_tab_dev) (void)hipFree(s->bank_tab_dev);
    if (s->t0) (void)hipEventDestroy(s->t0);
    if (s->t1) (void)hipEventDestroy(s->t1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

}  // namespace

extern "C" {

uint32_t s2r_abi_version(void) { return S2R_ABI_VERSION; }

// synth2_amd/build.py passes -DS2R_BUILD_ID="<hash of the sources>"; the marker in front is what lets a loader read the id
// out of the file without mapping it (build.py embedded_build_id)
#ifndef S2R_BUILD_ID
#define S2R_BUILD_ID "0000000000000000-00000000"
#endif
static const char kBuildIdMarked[] = "S2R_BUILD_ID=" S2R_BUILD_ID;
const char *s2r_build_id(void) { return kBuildIdMarked + 13; }

const char *s2r_status_string(int status) {
    switch (status) {
    case S2R_OK: return "ok";
    case S2R_ERR_INVALID: return "invalid argument";
    case S2R_ERR_NO_DEVICE: return "no usable gfx950 device";
    case S2R_ERR_HIP: return "HIP runtime error";
    case S2R_ERR_PATCH_SYNTAX: return ".synth2 syntax error";
    case S2R_ERR_PATCH_RANGE: return "patch value out of range";
    case S2R_ERR_TOO_MANY_FRAMES: return "frames exceed max_frames";
    case S2R_ERR_OFFSET_OVERFLOW: return "voice frame offset overflow";
    case S2R_ERR_OUT_OF_MEMORY: return "out of memory";
    default: return "unknown status";
    }
}

// Worker threads of the allocation policy's batch form (S2rVoicePool::resolve_batch): S2R_POLICY_THREADS, default NONE.
// Built for the pools of multi-GPU runs (every rank resolves the WHOLE pool's events: 16 384 per buffer at 8 x 65 536 voices),
// bit-identical (tests/native/tsan_policy.cpp) and measured SLOWER than one thread on the bench host at every pool size
// (profiles/r04/ngpu_host_cost.txt: 6.7 ns per event alone, 11.5 with three workers in the caller's core complex, 40 across
// complexes): what one thread spends per event is less than one cache line's trip between two cores, and the hand-over
// from the queue to the notes' owners is a line per eight events per worker.
static void configure_policy_threads(S2rVoicePool *pool, uint32_t total_voices) {
    (void)total_voices;
    const char *e = std::getenv("S2R_POLICY_THREADS");
    uint32_t n = 0u;
    if (e && e[0] >= '0' && e[0] <= '9') n = (uint32_t)std::atoi(e);
    pool->set_workers(n, 4096);
}

// one shard on one device (s2r_create without a device list, and each shard of one with).  `pool` != null: a shard of
// the device-list handle `parent`, which owns the pool and runs the allocation policy.
static int create_single(const s2r_config *cfg, std::shared_ptr<S2rVoicePool> pool, s2r_synth *parent, s2r_synth **out) {
    *out = nullptr;
    if (cfg->total_voices == 0 || cfg->max_frames == 0) return S2R_ERR_INVALID;
    const uint32_t bv = cfg->block_voices ? cfg->block_voices : 256u;
    if (bv < 64 || bv > 1024 || (bv & 63u)) return S2R_ERR_INVALID;
    uint32_t shard_voices;
    if (cfg->shard_interleave) {
        const uint32_t g = cfg->shard_interleave, n = cfg->shard_count;
        if ((g & 15u) || bv % g || n == 0 || cfg->shard_index >= n || cfg->total_voices % ((uint64_t)g * n)) return S2R_ERR_INVALID;
        shard_voices = cfg->total_voices / n;
        if (cfg->shard_voices && cfg->shard_voices != shard_voices) return S2R_ERR_INVALID;
    } else {
        shard_voices = cfg->shard_voices ? cfg->shard_voices : cfg->total_voices - cfg->shard_begin;
        if ((uint64_t)cfg->shard_begin + shard_voices > cfg->total_voices || shard_voices == 0) return S2R_ERR_INVALID;
    }

    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return S2R_ERR_NO_DEVICE;
    int dev = cfg->device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) return S2R_ERR_NO_DEVICE; }
    if (dev >= n_dev) return S2R_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return S2R_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return S2R_ERR_NO_DEVICE;   // code objects are gfx950 only

    s2r_synth *s = new (std::nothrow) s2r_synth();
    if (!s) return S2R_ERR_OUT_OF_MEMORY;
    s->cfg = *cfg;
    s->device = dev;
    s->shard_begin = cfg->shard_interleave ? 0u : cfg->shard_begin;
    s->interleave = cfg->shard_interleave; s->shard_index = cfg->shard_index; s->shard_count = cfg->shard_interleave ? cfg->shard_count : 1u;
    s->shard_voices = shard_voices;
    s->div_interleave.set(s->interleave); s->div_count.set(s->shard_count);
    s->block_voices = bv;
    s->n_blocks = (shard_voices + bv - 1) / bv;
    s->padded_voices = s->n_blocks * bv;
    s->mix_groups = cfg->mix_groups ? cfg->mix_groups : 1u;
    if (cfg->reserved0 != 0) { delete s; return S2R_ERR_INVALID; }
    s->parent = parent;
    s->bank.resize(1);
    s2r_default_patch(&s->bank[0]);
    if (pool) s->pool = pool;
    else { s->pool.reset(new S2rVoicePool(cfg->total_voices)); s->seed_override.assign(cfg->total_voices, 0u); configure_policy_threads(s->pool.get(), cfg->total_voices); }
    s->pending_slot.assign(shard_voices, -1);
    build_pitch_table(s->pitch_table);

#define CREATE_HIP(call)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "libs2r: %s failed: %s\n", #call, hipGetErrorString(e_));      \
            release_all(s);                                                                \
            return e_ == hipErrorOutOfMemory ? S2R_ERR_OUT_OF_MEMORY : S2R_ERR_HIP;        \
        }                                                                                  \
    } while (0)

    CREATE_HIP(hipSetDevice(dev));
    CREATE_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    const size_t pv = s->padded_voices;
    CREATE_HIP(hipMalloc(&s->voice_mem, pv * kVoiceWords * sizeof(uint32_t)));
    CREATE_HIP(hipMemsetAsync(s->voice_mem, 0, pv * kVoiceWords * sizeof(uint32_t), s->stream));
    uint32_t *base = (uint32_t *)s->voice_mem;
    s->v.pitch = (float *)(base + 0 * pv);
    s->v.offset = base + 1 * pv;
    s->v.release = base + 2 * pv;
    s->v.flags = base + 3 * pv;
    s->v.phase = (float *)(base + 4 * pv);
    s->v.lpf_last = (float *)(base + 5 * pv);
    s->v.seed = base + 6 * pv;
    s->v.fx1 = (float *)(base + 7 * pv);
    s->v.fx2 = (float *)(base + 8 * pv);
    s->v.fy1 = (float *)(base + 9 * pv);
    s->v.fy2 = (float *)(base + 10 * pv);
    s->v.program = base + 11 * pv;
    s->v.osc_z = (float *)(base + 12 * pv);
    CREATE_HIP(hipMalloc((void **)&s->bank_dev, S2R_MAX_BANK * sizeof(S2rBankEntry)));
    CREATE_HIP(hipMalloc((void **)&s->block_partials, (size_t)s->n_blocks * partials_stride(cfg->max_frames) * sizeof(float)));
    CREATE_HIP(hipMalloc((void **)&s->out_dev, (size_t)2 * cfg->max_frames * sizeof(float)));
    CREATE_HIP(hipHostMalloc((void **)&s->out_host, (size_t)2 * cfg->max_frames * sizeof(float), kHostPolled));
    CREATE_HIP(hipHostGetDevicePointer((void **)&s->out_host_dev, s->out_host, 0));
    for (int k = 0; k < 2; k++) {
        CREATE_HIP(hipHostMalloc((void **)&s->ring_host[k], (size_t)cfg->max_frames * sizeof(float), kHostPolled));
        CREATE_HIP(hipHostGetDevicePointer((void **)&s->ring_dev[k], s->ring_host[k], 0));
        CREATE_HIP(hipEventCreateWithFlags(&s->ring_done[k], hipEventDisableTiming));
    }
    CREATE_HIP(hipHostMalloc((void **)&s->done_host, 16 * sizeof(uint32_t), kHostPolled));
    std::memset(s->done_host, 0, 16 * sizeof(uint32_t));
    CREATE_HIP(hipHostGetDevicePointer((void **)&s->done_dev, s->done_host, 0));
    CREATE_HIP(hipMalloc((void **)&s->done_counter, 4 * sizeof(uint32_t)));
    CREATE_HIP(hipMemsetAsync(s->done_counter, 0, 4 * sizeof(uint32_t), s->stream));
    CREATE_HIP(hipMalloc((void **)&s->fz_arrive, 2 * sizeof(uint32_t)));
    CREATE_HIP(hipMemsetAsync(s->fz_arrive, 0, 2 * sizeof(uint32_t), s->stream));
    s->div_block.set(bv);
    { const char *e = std::getenv("S2R_FUSED"); if (e && e[0] >= '0' && e[0] <= '2') s->fused_mode = e[0] - '0'; }
    if (hipDeviceGetAttribute(&s->n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) s->n_cu = 0;
    s->tev_capacity = shard_voices < 4096u ? 4096u : shard_voices;
    s->tlast.assign(shard_voices, -1);
    s->tfirst.assign(shard_voices, -1);
    for (EventSlot &sl : s->slots) {
        CREATE_HIP(hipHostMalloc((void **)&sl.host, (size_t)shard_voices * sizeof(S2rVoiceEvent), kHostPolled));
        CREATE_HIP(hipHostGetDevicePointer((void **)&sl.dev, sl.host, 0));
        CREATE_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        CREATE_HIP(hipHostMalloc((void **)&sl.thost, (size_t)s->tev_capacity * sizeof(S2rTimedEvent), kHostPolled));
        CREATE_HIP(hipHostGetDevicePointer((void **)&sl.tdev, sl.thost, 0));
    }
    CREATE_HIP(hipMalloc((void **)&s->voice_ev_head, pv * sizeof(int32_t)));
    CREATE_HIP(hipMalloc((void **)&s->tev_copy, (size_t)s->tev_capacity * sizeof(S2rTimedEvent)));
    CREATE_HIP(hipMemsetAsync(s->voice_ev_head, 0xff, pv * sizeof(int32_t), s->stream));
    s->partials2[0] = s->block_partials; s->heads2[0] = s->voice_ev_head; s->tevcopy2[0] = s->tev_copy;
    {   // two streams for the fills of s2r_fill_begin (S2rOverlapWords): shards of more than one workgroup; S2R_OVERLAP=0: off
        // Only where the kernels that wait for each other can always be resident TOGETHER: a render grid of at most one
        // workgroup of at most 256 threads per compute unit takes at most half of every SIMD's registers and of the LDS, so the
        // chain-heads workgroups it waits for always find room beside it.  (A grid of two workgroups per CU holds every
        // register: its waiting workgroups would keep out the ones they wait for — seen at 131 072 voices, as the bounded
        // wait's error.)
        // (a shard of a device list never begins a fill of its own: no second stream, no second buffers for it)
        const char *e = std::getenv("S2R_OVERLAP");
        const int n_cu = s->n_cu;
        bool mine = false;
        if (!parent && s->n_blocks > 1 && (int)s->n_blocks <= n_cu && s->block_voices <= 256u && !(e && e[0] == '0') && dev < 64) {
            int expected = 0;
            mine = g_two_stream_handles[dev].compare_exchange_strong(expected, 1);
            s->ov_registered = mine;
        }
        if (mine) {
            CREATE_HIP(hipStreamCreateWithFlags(&s->stream_b, hipStreamNonBlocking));
            CREATE_HIP(hipMalloc((void **)&s->partials2[1], (size_t)s->n_blocks * partials_stride(cfg->max_frames) * sizeof(float)));
            CREATE_HIP(hipMalloc((void **)&s->heads2[1], pv * sizeof(int32_t)));
            CREATE_HIP(hipMemsetAsync(s->heads2[1], 0xff, pv * sizeof(int32_t), s->stream));
            CREATE_HIP(hipMalloc((void **)&s->tevcopy2[1], (size_t)s->tev_capacity * sizeof(S2rTimedEvent)));
            CREATE_HIP(hipMalloc((void **)&s->ov_words, sizeof(S2rOverlapWords)));
            CREATE_HIP(hipMemsetAsync(s->ov_words, 0, sizeof(S2rOverlapWords), s->stream));
            s->ov_enabled = true;
        }
    }
    CREATE_HIP(hipEventCreate(&s->t0));
    CREATE_HIP(hipEventCreate(&s->t1));
    {
        float table[1024];
        build_sin_table(table);
        if (crc32_bytes(table, sizeof table) != kSinTableCrc) {
            fprintf(stderr, "libs2r: regenerated SIN_TABLE does not match the reference table (CRC mismatch)\n");
            release_all(s);
            return S2R_ERR_INVALID;
        }
        CREATE_HIP(hipMalloc((void **)&s->sin_dev, sizeof table));
        CREATE_HIP(hipMemcpy(s->sin_dev, table, sizeof table, hipMemcpyHostToDevice));
        CREATE_HIP(hipMalloc((void **)&s->noise_dev, 65536 * sizeof(float)));
        CREATE_HIP(s2r_launch_noise_table(s->noise_dev, s->stream));
    }
    CREATE_HIP(hipStreamSynchronize(s->stream));
#undef CREATE_HIP
    *out = s;
    return S2R_OK;
}


int s2r_create(const s2r_config *cfg, s2r_synth **out) {
    if (!cfg || !out || cfg->struct_size != sizeof(s2r_config)) return S2R_ERR_INVALID;
    *out = nullptr;
    if (cfg->n_devices <= 1) {
        s2r_config one = *cfg;
        if (cfg->n_devices == 1) one.device = cfg->devices[0];
        one.n_devices = 0;
        return create_single(&one, nullptr, nullptr, out);
    }
    // ---- a device list: one parent, one shard handle per device ----
    const uint32_t n = cfg->n_devices;
    const uint32_t bv = cfg->block_voices ? cfg->block_voices : 256u;
    if (n > S2R_MAX_DEVICES || cfg->total_voices == 0 || cfg->max_frames == 0 || cfg->reserved0 != 0) return S2R_ERR_INVALID;
    if (cfg->shard_begin || cfg->shard_voices || cfg->shard_index || cfg->shard_count > 1 || cfg->mix_groups > 1) return S2R_ERR_INVALID;
    if (bv < 64 || bv > 1024 || (bv & 63u) || cfg->total_voices % ((uint64_t)n * bv)) return S2R_ERR_INVALID;
    if (cfg->shard_interleave && ((cfg->shard_interleave & 15u) || bv % cfg->shard_interleave)) return S2R_ERR_INVALID;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return S2R_ERR_NO_DEVICE;
    for (uint32_t k = 0; k < n; k++) if (cfg->devices[k] < 0 || cfg->devices[k] >= n_dev) return S2R_ERR_NO_DEVICE;
    s2r_synth *s = new (std::nothrow) s2r_synth();
    if (!s) return S2R_ERR_OUT_OF_MEMORY;
    s->cfg = *cfg;
    s->device = cfg->devices[0];
    s->shard_voices = cfg->total_voices;
    s->block_voices = bv;
    s->interleave = cfg->shard_interleave; s->shard_count = n;
    s->div_interleave.set(s->interleave); s->div_count.set(n); s->div_per_kid.set(cfg->total_voices / n);
    s->bank.resize(1);
    s2r_default_patch(&s->bank[0]);
    s->pool.reset(new S2rVoicePool(cfg->total_voices));
    configure_policy_threads(s->pool.get(), cfg->total_voices);
    s->seed_override.assign(cfg->total_voices, 0u);
    build_pitch_table(s->pitch_table);
    for (uint32_t k = 0; k < n; k++) {
        s2r_config kc = *cfg;
        kc.n_devices = 0; kc.device = cfg->devices[k]; kc.mix_groups = 1; kc.block_voices = bv;
        if (cfg->shard_interleave) { kc.shard_index = k; kc.shard_count = n; kc.shard_begin = 0; kc.shard_voices = 0; }
        else { kc.shard_begin = k * (cfg->total_voices / n); kc.shard_voices = cfg->total_voices / n; kc.shard_index = 0; kc.shard_count = 1; }
        s2r_synth *kid = nullptr;
        const int rc = create_single(&kc, s->pool, s, &kid);
        if (rc != S2R_OK) { release_all(s); return rc; }
        s->kids.push_back(kid);
    }
#define CREATE_HIP(call)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "libs2r: %s failed: %s\n", #call, hipGetErrorString(e_));      \
            release_all(s);                                                                \
            return e_ == hipErrorOutOfMemory ? S2R_ERR_OUT_OF_MEMORY : S2R_ERR_HIP;        \
        }                                                                                  \
    } while (0)
    CREATE_HIP(hipSetDevice(s->device));
    CREATE_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; b++) {
        CREATE_HIP(malloc_exchange((void **)&s->rows_dev[b], (size_t)n * cfg->max_frames * sizeof(float)));
        CREATE_HIP(hipMemsetAsync(s->rows_dev[b], 0, (size_t)n * cfg->max_frames * sizeof(float), s->stream));
        s->kid_done[b].assign(n, nullptr);
    }
    CREATE_HIP(hipMalloc((void **)&s->out_dev, (size_t)2 * cfg->max_frames * sizeof(float)));
    CREATE_HIP(hipHostMalloc((void **)&s->out_host, (size_t)2 * cfg->max_frames * sizeof(float), kHostPolled));
    CREATE_HIP(hipHostGetDevicePointer((void **)&s->out_host_dev, s->out_host, 0));
    for (int k = 0; k < 2; k++) {
        CREATE_HIP(hipHostMalloc((void **)&s->ring_host[k], (size_t)cfg->max_frames * sizeof(float), kHostPolled));
        CREATE_HIP(hipHostGetDevicePointer((void **)&s->ring_dev[k], s->ring_host[k], 0));
        CREATE_HIP(hipEventCreateWithFlags(&s->ring_done[k], hipEventDisableTiming));
    }
    CREATE_HIP(hipHostMalloc((void **)&s->done_host, 16 * sizeof(uint32_t), kHostPolled));
    std::memset(s->done_host, 0, 16 * sizeof(uint32_t));
    CREATE_HIP(hipHostGetDevicePointer((void **)&s->done_dev, s->done_host, 0));
    CREATE_HIP(hipMalloc((void **)&s->done_counter, 4 * sizeof(uint32_t)));
    CREATE_HIP(hipMemsetAsync(s->done_counter, 0, 4 * sizeof(uint32_t), s->stream));
    CREATE_HIP(malloc_exchange((void **)&s->rows_done, 2 * sizeof(uint32_t)));
    CREATE_HIP(hipMemsetAsync(s->rows_done, 0, 2 * sizeof(uint32_t), s->stream));
    CREATE_HIP(hipStreamSynchronize(s->stream));
    // A shard on another device writes its row straight into the parent's buffer when the devices are peers (one
    // 4 KiB write over xGMI by its mixers: SURVEY 5's preferred shape); otherwise into a row of its own that a
    // peer copy moves.  On a box with ONE device the two multi-device branches are still reachable (tests/test_gpu_device_list.py
    // runs under both): S2R_FORCE_PEER=1 sends a shard on the parent's own device through the peer set-up (the queries and
    // hipDeviceEnablePeerAccess, whose refusal of a device as its own peer is the one error tolerated), S2R_FORCE_STAGE=1
    // gives every shard the staging row and the peer copy.
    { const char *e = std::getenv("S2R_FORCE_STAGE"); s->force_stage = e && e[0] == '1'; }
    { const char *e = std::getenv("S2R_FORCE_PEER"); s->force_peer = e && e[0] == '1'; }
    s->kid_stage.assign(n, nullptr);
    for (uint32_t k = 0; k < n; k++) {
        s2r_synth *kid = s->kids[k];
        CREATE_HIP(hipSetDevice(kid->device));
        for (int b = 0; b < 2; b++) CREATE_HIP(hipEventCreateWithFlags(&s->kid_done[b][k], hipEventDisableTiming));
        const bool same = kid->device == s->device;
        bool direct = same;
        if (!same || s->force_peer) {
            int can = 0;
            const hipError_t q = hipDeviceCanAccessPeer(&can, kid->device, s->device);
            (void)hipGetLastError();
            if (same) can = 1;                                   // (a device reaches its own memory whatever the query says of it as a peer)
            direct = false;
            if ((q == hipSuccess || same) && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(s->device, 0);
                direct = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled || same;
                (void)hipGetLastError();
            }
        }
        if (s->force_stage) direct = false;
        if (!direct) CREATE_HIP(hipMalloc((void **)&s->kid_stage[k], (size_t)cfg->max_frames * sizeof(float)));
    }
#undef CREATE_HIP
    *out = s;
    return S2R_OK;
}

void s2r_destroy(s2r_synth *s) { release_all(s); }

int s2r_set_patch(s2r_synth *s, const s2r_patch *patch) {
    if (!s || !patch) return S2R_ERR_INVALID;
    std::string err;
    int rc = s2r_validate_patch(patch, &err);
    if (rc != S2R_OK) return set_err(s, rc, "%s", err.c_str());
    S2R_QUIESCE(s);
    s->bank[0] = *patch;
    s->bank_dirty = true; s->tab_dirty = true;
    for (s2r_synth *kid : s->kids) { kid->bank[0] = *patch; kid->bank_dirty = true; kid->tab_dirty = true; }
    return S2R_OK;
}

int s2r_set_patch_bank(s2r_synth *s, const s2r_patch *patches, uint32_t n) {
    if (!s || !patches) return S2R_ERR_INVALID;
    if (n == 0 || n > S2R_MAX_BANK) return set_err(s, S2R_ERR_INVALID, "a patch bank holds 1..%u patches, not %u", S2R_MAX_BANK, n);
    for (uint32_t k = 0; k < n; k++) {
        std::string err;
        int rc = s2r_validate_patch(&patches[k], &err);
        if (rc != S2R_OK) return set_err(s, rc, "patch %u: %s", k, err.c_str());
    }
    S2R_QUIESCE(s);
    s->bank.assign(patches, patches + n);
    if (s->program >= n) s->program = 0;
    s->bank_dirty = true; s->tab_dirty = true;
    for (s2r_synth *kid : s->kids) { kid->bank = s->bank; kid->bank_dirty = true; kid->tab_dirty = true; }
    return S2R_OK;
}

uint32_t s2r_patch_bank_size(const s2r_synth *s) { return s ? (uint32_t)s->bank.size() : 0u; }

int s2r_program_change(s2r_synth *s, uint32_t program) {
    if (!s) return S2R_ERR_INVALID;
    if (program >= s->bank.size()) return set_err(s, S2R_ERR_INVALID, "program %u: the bank holds %zu patches", program, s->bank.size());
    S2R_QUIESCE(s);
    s->program = program;
    return S2R_OK;
}

int s2r_get_patch(const s2r_synth *s, s2r_patch *out) {
    if (!s || !out) return S2R_ERR_INVALID;
    *out = s->bank[0];
    return S2R_OK;
}

int s2r_load_patch(s2r_synth *s, const char *text, size_t len) {
    if (!s || (!text && len)) return S2R_ERR_INVALID;
    s2r_patch p;
    std::string err;
    int rc = s2r_parse_patch(text ? text : "", len, &p, nullptr, &err);
    if (rc != S2R_OK) return set_err(s, rc, "%s", err.c_str());
    S2R_QUIESCE(s);
    s->bank[0] = p;
    s->bank_dirty = true; s->tab_dirty = true;
    for (s2r_synth *kid : s->kids) { kid->bank[0] = p; kid->bank_dirty = true; kid->tab_dirty = true; }
    return S2R_OK;
}

int s2r_parse_patch_text(const char *text, size_t len, s2r_patch *out, char *err_buf, size_t err_cap) {
    if ((!text && len) || !out) return S2R_ERR_INVALID;
    std::string err;
    int rc = s2r_parse_patch(text ? text : "", len, out, nullptr, &err);
    if (err_buf && err_cap) { std::snprintf(err_buf, err_cap, "%s", err.c_str()); }
    return rc;
}

int s2r_note_on_ex(s2r_synth *s, uint8_t note, float velocity, uint32_t *voice_index_out) {
    if (!s) return S2R_ERR_INVALID;
    if (s->fill_time) return set_err(s, S2R_ERR_INVALID, "untimed note_on after timed events: fill first or give it a frame");
    const uint32_t i = s->pool->note_on(note, velocity);
    if (voice_index_out) *voice_index_out = i;
    if (s->voice_log) s->voice_log(s->voice_log_user, i, note);
    if (!append_frame0_record(s, i, S2R_EV_RESTART, s->pitch_table[note], s->seed_override[i], s->program))
        push_event(s, i, S2R_EV_RESTART, s->pitch_table[note], s->seed_override[i], s->program);
    return S2R_OK;
}

int s2r_note_on(s2r_synth *s, uint8_t note, float velocity) { return s2r_note_on_ex(s, note, velocity, nullptr); }

int s2r_note_off(s2r_synth *s, uint8_t note) {
    if (!s) return S2R_ERR_INVALID;
    if (s->fill_time) return set_err(s, S2R_ERR_INVALID, "untimed note_off after timed events: fill first or give it a frame");
    const int64_t i = s->pool->note_off(note);
    if (i >= 0) { if (!append_frame0_record(s, (uint32_t)i, S2R_EV_RELEASE, 0.0f, 0u, 0u)) push_event(s, (uint32_t)i, S2R_EV_RELEASE, 0.0f, 0u); }
    else s->double_release++;                 // synth.rs:77: `log::warn!("double release")`, nothing else happens
    return S2R_OK;
}

int s2r_note_events(s2r_synth *s, const s2r_note_event *events, size_t n) {
    if (!s || (!events && n)) return S2R_ERR_INVALID;
    S2R_REFUSE_BROKEN(s);
    // The whole batch is checked before the first event touches the pool, the pool clock or the pending lists: a
    // rejected batch leaves the handle exactly as it was (the reference's note_on / note_off cannot fail at all,
    // synth.rs:61-80, so every error here is a malformed batch, not a state of the synth).
    {
        uint32_t t = s->fill_time;
        size_t bank_n = s->bank.size();
        for (size_t k = 0; k < n; k++) {
            const s2r_note_event &e = events[k];
            if (e.kind == S2R_PROGRAM_CHANGE) {
                if (e.note >= bank_n) return set_err(s, S2R_ERR_INVALID, "event %zu: program %u, the bank holds %zu patches", k, (unsigned)e.note, bank_n);
                continue;
            }
            if (e.kind != S2R_NOTE_ON && e.kind != S2R_NOTE_OFF)
                return set_err(s, S2R_ERR_INVALID, "event %zu: unknown kind %u", k, (unsigned)e.kind);
            const uint32_t frame = e.frame;
            if (frame % 16u) return set_err(s, S2R_ERR_INVALID, "event %zu: frame %u is not a multiple of 16 (events land between 16-frame chunks, main.rs:138-143)", k, frame);
            if (frame < t) return set_err(s, S2R_ERR_INVALID, "event %zu: frame %u precedes an earlier event at %u", k, frame, t);
            // an event must fall inside the next fill, and no fill is longer than max_frames: a later frame could
            // never be rendered and would leave the handle refusing every fill
            if (frame >= s->cfg.max_frames) return set_err(s, S2R_ERR_INVALID, "event %zu: frame %u is not inside any fill (max_frames %u)", k, frame, s->cfg.max_frames);
            t = frame;
        }
    }
    // The allocation policy for the whole batch at once (S2rVoicePool::resolve_batch: what the loop of pool clock moves, note_on
    // and note_off over the events computes — on several threads for a multi-GPU-sized batch): the voice every event takes or
    // releases.  An event inside the next fill first moves the pool's clock to its frame (the policy sees the offsets every
    // voice has AT that frame, like the reference between two 16-frame calls); frame-0 events take effect before the fill.
    static thread_local std::vector<int64_t> chosen;
    if (chosen.size() < n) chosen.resize(n);
    static_assert(sizeof(S2rPolicyEvent) == 4 && S2R_NOTE_ON == S2R_POLICY_NOTE_ON && S2R_NOTE_OFF == S2R_POLICY_NOTE_OFF, "s2r_note_event's first four bytes");
    // The policy runs over the whole batch first (its note_on runs and note_off runs are loops of their own), the records are
    // built behind it.
    if (n) s->fill_time = s->pool->resolve_batch(reinterpret_cast<const S2rPolicyEvent *>(events), sizeof(s2r_note_event), n, s->fill_time,
                                                 chosen.data(), &events[0].velocity, sizeof(s2r_note_event));
    // An event at frame 0 takes effect before the next fill.  A small batch's are FOLDED per voice (push_event: one record per
    // touched voice, which can ride in the render kernel's arguments); a big batch's — and whatever follows records already
    // waiting on the shard — go straight into the voices' chains as frame-0 records, which is where the folded ones of a fill
    // with chains end up anyway (merge_pending_into_chains), without the fold's lookup per event and the merge's pass per fill.
    if (s->voice_log)
        for (size_t k = 0; k < n; k++) if (events[k].kind == S2R_NOTE_ON) s->voice_log(s->voice_log_user, (uint32_t)chosen[k], events[k].note);
    const bool may_fold = n <= S2R_ARG_MAX_EVENTS;
    const int64_t *vi_of = chosen.data();
    const float *pitch_of = s->pitch_table;
    const uint32_t *seed_of = s->seed_override.data();
    s2r_synth *one = s->kids.empty() ? s : nullptr;             // (not a device list: every record is this handle's own)
    if (one) one->tpending.reserve(one->tpending.size() + n);
    uint32_t double_release = 0;
    for (size_t k = 0; k < n; k++) {
        const s2r_note_event &e = events[k];
        if (e.kind == S2R_PROGRAM_CHANGE) {      // host-side state: which patch the following note_ons get
            s->program = e.note;
            continue;
        }
        const int64_t vi = vi_of[k];
        if (vi < 0) { double_release++; continue; }              // synth.rs:77 logs "double release" and carries on
        uint32_t local = 0;
        s2r_synth *sh;
        if (one) {
            const int64_t mine = to_local(one, (uint32_t)vi);
            if (mine < 0) continue;                              // (another rank's voice)
            local = (uint32_t)mine; sh = one;
        } else {
            sh = shard_of(s, (uint32_t)vi, &local);
            if (!sh) continue;
        }
        const uint32_t frame = e.frame;
        const bool on = e.kind == S2R_NOTE_ON;
        if (frame == 0 && may_fold && sh->tpending.empty()) {
            if (on) push_event(s, (uint32_t)vi, S2R_EV_RESTART, pitch_of[e.note], seed_of[(size_t)vi], s->program);
            else push_event(s, (uint32_t)vi, S2R_EV_RELEASE, 0.0f, 0u);
            continue;
        }
        // (no capacity limit here: the device-side buffers grow in flush_events when a fill brings more timed
        // events than they hold)
        std::vector<S2rTimedEvent> &tp = sh->tpending;
        const int32_t idx = (int32_t)tp.size();
        int32_t *last = &sh->tlast[local];
        uint32_t fl = on ? S2R_EV_RESTART : S2R_EV_RELEASE;
        if (*last >= 0) tp[(size_t)*last].next = idx;
        else fl |= S2R_TEV_FIRST;
        *last = idx;
        tp.push_back(S2rTimedEvent{local, frame, fl, on ? pitch_of[e.note] : 0.0f, on ? seed_of[(size_t)vi] : 0u, -1, s->program, 0u});
    }
    s->double_release += double_release;
    return S2R_OK;
}

int s2r_fill(s2r_synth *s, float *mono_out, size_t frames, uint32_t sample_rate_hz) {
    return fill_host(s, mono_out, frames, sample_rate_hz, false);
}

int s2r_fill_begin(s2r_synth *s, size_t frames, uint32_t sample_rate_hz) {
    int rc = check_fill(s, frames, sample_rate_hz);
    if (rc != S2R_OK) return rc;
    if (s->ring_count >= 2) return set_err(s, S2R_ERR_INVALID, "two fills are already in flight: s2r_fill_end first");
    { const int rc_r = resident_stop(s); if (rc_r != S2R_OK) return rc_r; }      // (the pool-resident kernel stays: enqueue_root decides)
    S2R_HIP(s, hipSetDevice(s->device));
    const uint32_t slot = (s->ring_head + s->ring_count) & 1u;
    if (frames) {
        // the last kernel of the fill writes the mix straight into this slot's mapped host buffer
        const S2rDone done{s->done_dev + slot, ++s->done_seq, s->done_counter + slot};
        rc = enqueue_root(s, frames, sample_rate_hz, s->ring_dev[slot], false, (int)slot, &done);
        if (rc != S2R_OK) return rc;
        s->ring_seq[slot] = done.value;
    } else s->ring_seq[slot] = 0;
    // (a fill is tracked by its completion word; an empty one by the event)
    if (!frames) S2R_HIP(s, hipEventRecord(s->ring_done[slot], s->stream));
    s->ring_frames[slot] = frames;
    s->ring_count++;
    return S2R_OK;
}

size_t s2r_fill_pending_frames(const s2r_synth *s) { return (s && s->ring_count) ? s->ring_frames[s->ring_head] : 0; }
uint32_t s2r_fills_in_flight(const s2r_synth *s) { return s ? s->ring_count : 0; }

int s2r_fill_end(s2r_synth *s, float *mono_out, size_t capacity) {
    if (!s) return S2R_ERR_INVALID;
    if (s->ring_count == 0) return set_err(s, S2R_ERR_INVALID, "no fill in flight");
    const uint32_t slot = s->ring_head;
    if (s->ring_frames[slot] && !mono_out) return set_err(s, S2R_ERR_INVALID, "null output buffer");
    if (capacity < s->ring_frames[slot])
        return set_err(s, S2R_ERR_INVALID, "the oldest fill in flight has %zu frames, the buffer takes %zu", s->ring_frames[slot], capacity);
    if (s->dmix.active && s->dmix.ring_slot == (int)slot) {      // nobody began another fill in the meantime
        S2R_HIP(s, hipSetDevice(s->device));
        int rc = launch_deferred_mix(s, s->stream);
        if (rc != S2R_OK) return rc;
    }
    // A fill that failed on the device is reported ONCE, with its buffer zeroed, and leaves the ring like any other: the next
    // s2r_fill_end is the next fill's (which, the handle being broken by then, reports that).
    int rc_fill = S2R_OK;
    if (std::getenv("S2R_DEBUG_STUCK") && s->xg_on) { static thread_local unsigned n_calls = 0; if ((++n_calls & 15u) == 0u) std::fprintf(stderr, "[s2r rank %u] fill_end %u: waits for word %u (now %u)\n", s->xg_rank, n_calls, s->ring_seq[slot], s->done_host[slot]); }
    if (s->ring_seq[slot]) rc_fill = wait_done(s, slot, s->ring_seq[slot]);
    else if (hipEventSynchronize(s->ring_done[slot]) != hipSuccess) rc_fill = set_err(s, S2R_ERR_HIP, "hipEventSynchronize failed");
    if (rc_fill == S2R_OK) rc_fill = overlap_check(s);
    if (rc_fill == S2R_OK && s->broken)          // (begun before the failure was known: rendered from voices a fill behind the host's clock)
        rc_fill = set_err(s, S2R_ERR_HIP, "an earlier fill failed on the device: this one was rendered from voices that no longer match the host's bookkeeping");
    if (s->ring_frames[slot]) {
        if (rc_fill != S2R_OK || (s->xg_on && s->xg_rank != 0)) std::memset(mono_out, 0, s->ring_frames[slot] * sizeof(float));    // (or: the mix is the root's)
        else std::memcpy(mono_out, s->ring_host[slot], s->ring_frames[slot] * sizeof(float));
    }
    s->ring_head ^= 1u;
    s->ring_count--;
    return rc_fill;
}

int s2r_fill_stereo(s2r_synth *s, float *interleaved_lr_out, size_t frames, uint32_t sample_rate_hz) {
    return fill_host(s, interleaved_lr_out, frames, sample_rate_hz, true);
}

int s2r_fill_oversampled(s2r_synth *s, float *mono_out, size_t frames, uint32_t sample_rate_hz) {
    if (!s) return S2R_ERR_INVALID;
    if (sample_rate_hz > 0xffffffffu / S2R_OVERSAMPLE) return set_err(s, S2R_ERR_INVALID, "sample rate too high to oversample");
    S2R_QUIESCE(s);
    if (!s->kids.empty() && s->ring_count) return set_err(s, S2R_ERR_INVALID, "a device-list handle takes no synchronous fill while fills are in flight: s2r_fill_end first");
    const size_t os_frames = frames * S2R_OVERSAMPLE;
    int rc = check_fill(s, os_frames, sample_rate_hz * S2R_OVERSAMPLE);
    if (rc != S2R_OK) return rc;
    if (frames == 0) return S2R_OK;
    if (!mono_out) return set_err(s, S2R_ERR_INVALID, "null output buffer");
    S2R_HIP(s, hipSetDevice(s->device));
    constexpr uint32_t kTaps = 63;
    if (!s->os_buf) {
        // Blackman-windowed sinc, cutoff 0.115 cycles per input sample, unit DC gain; double arithmetic, rounded once
        double d[kTaps], sum = 0.0;
        const double PI = 3.14159265358979323846, fc = 0.115;
        for (uint32_t k = 0; k < kTaps; k++) {
            const double t = (double)((int)k - (int)(kTaps - 1) / 2);
            const double ideal = t == 0.0 ? 2.0 * fc : std::sin(2.0 * PI * fc * t) / (PI * t);
            const double w = 0.42 - 0.5 * std::cos(2.0 * PI * k / (kTaps - 1)) + 0.08 * std::cos(4.0 * PI * k / (kTaps - 1));
            d[k] = ideal * w; sum += d[k];
        }
        float h[kTaps];
        for (uint32_t k = 0; k < kTaps; k++) h[k] = (float)(d[k] / sum);
        S2R_HIP(s, hipMalloc((void **)&s->os_taps, sizeof h));
        S2R_HIP(s, hipMemcpy(s->os_taps, h, sizeof h, hipMemcpyHostToDevice));
        S2R_HIP(s, hipMalloc((void **)&s->os_buf, ((size_t)(kTaps - 1) + s->cfg.max_frames) * sizeof(float)));
        S2R_HIP(s, hipMemsetAsync(s->os_buf, 0, ((size_t)(kTaps - 1) + s->cfg.max_frames) * sizeof(float), s->stream));
    }
    rc = enqueue_root(s, os_frames, sample_rate_hz * S2R_OVERSAMPLE, s->os_buf + (kTaps - 1), false);
    if (rc != S2R_OK) return rc;
    S2R_HIP(s, s2r_launch_decimate4(s->os_buf, s->os_taps, (uint32_t)frames, s->out_host_dev, s->stream));
    S2R_HIP(s, hipStreamSynchronize(s->stream));
    std::memcpy(mono_out, s->out_host, frames * sizeof(float));
    return S2R_OK;
}

int s2r_fill_device(s2r_synth *s, float *dev_partial_out, size_t frames, uint32_t sample_rate_hz, void *hip_stream) {
    int rc = check_fill(s, frames, sample_rate_hz);
    if (rc != S2R_OK) return rc;
    if (frames == 0) return S2R_OK;
    if (!dev_partial_out) return set_err(s, S2R_ERR_INVALID, "null device output buffer");
    if (!s->kids.empty()) return set_err(s, S2R_ERR_INVALID, "s2r_fill_device is the per-shard building block: a device-list handle combines its shards itself (s2r_fill)");
    S2R_QUIESCE(s);
    S2R_HIP(s, hipSetDevice(s->device));
    return enqueue_fill(s, frames, sample_rate_hz, (hipStream_t)hip_stream, dev_partial_out, false, false, nullptr);
}

int s2r_fill_device_root(s2r_synth *s, float *dev_out, size_t frames, uint32_t sample_rate_hz, void *hip_stream) {
    int rc = check_fill(s, frames, sample_rate_hz);
    if (rc != S2R_OK) return rc;
    if (frames == 0) return S2R_OK;
    if (!dev_out) return set_err(s, S2R_ERR_INVALID, "null device output buffer");
    if (!s->kids.empty()) return set_err(s, S2R_ERR_INVALID, "s2r_fill_device_root takes a single-device handle");
    S2R_QUIESCE(s);
    S2R_HIP(s, hipSetDevice(s->device));
    return enqueue_fill(s, frames, sample_rate_hz, (hipStream_t)hip_stream, dev_out, true, false, nullptr);
}

int s2r_sum_partials_device(const float *dev_rows, uint32_t n_rows, size_t frames, float *dev_out, void *hip_stream) {
    if (!dev_rows || !dev_out || n_rows == 0) return S2R_ERR_INVALID;
    return s2r_launch_sum_rows(dev_rows, n_rows, (uint32_t)frames, (uint32_t)frames, 0, dev_out, (hipStream_t)hip_stream, nullptr) == hipSuccess ? S2R_OK : S2R_ERR_HIP;
}

int s2r_render_voices(s2r_synth *s, float *per_voice_out, size_t frames, uint32_t sample_rate_hz) {
    int rc = check_fill(s, frames, sample_rate_hz);
    if (rc != S2R_OK) return rc;
    if (frames == 0) return S2R_OK;
    if (!per_voice_out) return set_err(s, S2R_ERR_INVALID, "null output buffer");
    fold_frame0_records(s);
    if (!s->tpending.empty()) return set_err(s, S2R_ERR_INVALID, "s2r_render_voices does not take timed events; use s2r_fill");
    S2R_QUIESCE(s);
    if (!s->kids.empty()) {                       // every shard's rows, put back into pool order
        if (s->ring_count) return set_err(s, S2R_ERR_INVALID, "a device-list handle takes no synchronous fill while fills are in flight: s2r_fill_end first");
        for (s2r_synth *kid : s->kids) if (!kid->tpending.empty()) return set_err(s, S2R_ERR_INVALID, "s2r_render_voices does not take timed events; use s2r_fill");
        rc = enqueue_multi(s, frames, sample_rate_hz, nullptr, false, per_voice_out);
        if (rc != S2R_OK) return rc;
        std::vector<float> tmp;
        for (s2r_synth *kid : s->kids) {
            S2R_HIP(s, hipSetDevice(kid->device));
            tmp.resize((size_t)kid->shard_voices * frames);
            S2R_HIP(s, hipMemcpyAsync(tmp.data(), kid->per_voice_dev, tmp.size() * sizeof(float), hipMemcpyDeviceToHost, kid->stream));
            S2R_HIP(s, hipStreamSynchronize(kid->stream));
            for (uint32_t l = 0; l < kid->shard_voices; l++)
                std::memcpy(per_voice_out + (size_t)to_pool(kid, l) * frames, tmp.data() + (size_t)l * frames, frames * sizeof(float));
        }
        return S2R_OK;
    }
    S2R_HIP(s, hipSetDevice(s->device));
    const size_t need = (size_t)s->shard_voices * frames;
    if (need > s->per_voice_cap) {
        if (s->per_voice_dev) { S2R_HIP(s, hipFree(s->per_voice_dev)); s->per_voice_dev = nullptr; s->per_voice_cap = 0; }
        S2R_HIP(s, hipMalloc((void **)&s->per_voice_dev, need * sizeof(float)));
        s->per_voice_cap = need;
    }
    rc = enqueue_fill(s, frames, sample_rate_hz, s->stream, nullptr, false, false, s->per_voice_dev);
    if (rc != S2R_OK) return rc;
    S2R_HIP(s, hipMemcpyAsync(per_voice_out, s->per_voice_dev, need * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    S2R_HIP(s, hipStreamSynchronize(s->stream));
    return S2R_OK;
}

int s2r_export_state(s2r_synth *s, s2r_voice_state *voices) {
    if (!s || !voices) return S2R_ERR_INVALID;
    S2R_QUIESCE(s);
    if (!s->parent) fold_frame0_records(s);
    if (!s->kids.empty()) {                       // pool order: every shard's voices put back where the pool has them
        std::vector<s2r_voice_state> tmp;
        for (s2r_synth *kid : s->kids) {
            tmp.resize(kid->shard_voices);
            const int rc = s2r_export_state(kid, tmp.data());
            if (rc != S2R_OK) { s->err = kid->err; return rc; }
            for (uint32_t l = 0; l < kid->shard_voices; l++) voices[to_pool(kid, l)] = tmp[l];
        }
        return S2R_OK;
    }
    S2R_HIP(s, hipSetDevice(s->device));
    if (!s->tpending.empty()) return set_err(s, S2R_ERR_INVALID, "export_state with timed events pending: fill first");
    EventSlot *ts = nullptr; const S2rTimedEvent *td = nullptr;
    int rc = flush_events(s, s->stream, &ts, &td);
    if (rc != S2R_OK) return rc;
    const size_t pv = s->padded_voices;
    std::vector<uint32_t> h(pv * kVoiceWords);
    S2R_HIP(s, hipMemcpyAsync(h.data(), s->voice_mem, pv * kVoiceWords * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    S2R_HIP(s, hipStreamSynchronize(s->stream));
    for (uint32_t i = 0; i < s->shard_voices; i++) {
        const S2rHostVoice &hv = s->pool->voice(to_pool(s, i));
        s2r_voice_state &o = voices[i];
        std::memset(&o, 0, sizeof o);
        const uint32_t fl = h[3 * pv + i];
        o.note = hv.note;
        o.started = (fl & S2R_VF_STARTED) ? 1 : 0;
        o.released = (fl & S2R_VF_RELEASED) ? 1 : 0;
        o.current_frame_offset = h[1 * pv + i];
        o.release_frame_offset = h[2 * pv + i];
        o.pitch_hz = s2r_u2f(h[0 * pv + i]);
        o.phase_accum = s2r_u2f(h[4 * pv + i]);
        o.lpf_last = s2r_u2f(h[5 * pv + i]);
        o.noise_seed = h[6 * pv + i];
        o.filt_x1 = s2r_u2f(h[7 * pv + i]); o.filt_x2 = s2r_u2f(h[8 * pv + i]);
        o.filt_y1 = s2r_u2f(h[9 * pv + i]); o.filt_y2 = s2r_u2f(h[10 * pv + i]);
        o.program = (uint8_t)h[11 * pv + i];
        o.osc_z = s2r_u2f(h[12 * pv + i]);
        o.velocity = hv.velocity;
    }
    return S2R_OK;
}

int s2r_import_state(s2r_synth *s, const s2r_voice_state *voices) {
    if (!s || !voices) return S2R_ERR_INVALID;
    S2R_QUIESCE(s);
    if (!s->parent) fold_frame0_records(s);
    if (!s->kids.empty()) {
        std::vector<s2r_voice_state> tmp;
        for (s2r_synth *kid : s->kids) {
            tmp.resize(kid->shard_voices);
            for (uint32_t l = 0; l < kid->shard_voices; l++) tmp[l] = voices[to_pool(kid, l)];
            const int rc = s2r_import_state(kid, tmp.data());
            if (rc != S2R_OK) { s->err = kid->err; return rc; }
        }
        return S2R_OK;
    }
    S2R_HIP(s, hipSetDevice(s->device));
    // pending events refer to the state being replaced
    for (const S2rVoiceEvent &e : s->pending) s->pending_slot[e.voice] = -1;
    s->pending.clear();
    if (!s->tpending.empty()) return set_err(s, S2R_ERR_INVALID, "import_state with timed events pending");
    const size_t pv = s->padded_voices;
    std::vector<uint32_t> h(pv * kVoiceWords, 0u);
    for (uint32_t i = 0; i < s->shard_voices; i++) {
        const s2r_voice_state &in = voices[i];
        h[0 * pv + i] = s2r_f2u(in.started ? in.pitch_hz : 0.0f);
        h[1 * pv + i] = in.current_frame_offset;
        h[2 * pv + i] = in.release_frame_offset;
        h[3 * pv + i] = (in.started ? S2R_VF_STARTED : 0u) | ((in.started && in.released) ? S2R_VF_RELEASED : 0u);
        h[4 * pv + i] = s2r_f2u(in.phase_accum);
        h[5 * pv + i] = s2r_f2u(in.lpf_last);
        h[6 * pv + i] = in.noise_seed;
        h[7 * pv + i] = s2r_f2u(in.filt_x1); h[8 * pv + i] = s2r_f2u(in.filt_x2);
        h[9 * pv + i] = s2r_f2u(in.filt_y1); h[10 * pv + i] = s2r_f2u(in.filt_y2);
        h[11 * pv + i] = in.program;
        h[12 * pv + i] = s2r_f2u(in.osc_z);
        s->pool->set_voice(to_pool(s, i), in.note, in.started != 0, in.released != 0,
                           in.current_frame_offset, in.release_frame_offset, in.velocity);
    }
    s->pool->rebuild();
    S2R_HIP(s, hipMemcpyAsync(s->voice_mem, h.data(), pv * kVoiceWords * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    S2R_HIP(s, hipStreamSynchronize(s->stream));
    return S2R_OK;
}

int s2r_set_voice_log(s2r_synth *s, s2r_voice_log_fn fn, void *user) {
    if (!s || s->parent) return S2R_ERR_INVALID;
    s->voice_log = fn; s->voice_log_user = user;
    return S2R_OK;
}

// process::process_layer_buf_simd (process.rs:14-49) for layers the caller keeps itself: the handle's first n voices become the
// layers (checkpoint in), one fill of per-voice rows, the layers' states come back (checkpoint out).
int s2r_process_layers(s2r_synth *s, s2r_layer_call *layers, uint32_t n_layers, float *bufs, size_t frames, uint32_t sample_rate_hz) {
    if (!s || s->parent || (!layers && n_layers)) return S2R_ERR_INVALID;
    const uint32_t total = s->pool->size();
    if (!s->kids.empty() || s->shard_voices != total) return set_err(s, S2R_ERR_INVALID, "s2r_process_layers takes a handle that holds its whole pool on one device");
    if (n_layers > total) return set_err(s, S2R_ERR_INVALID, "%u layers, the handle has %u voices", n_layers, total);
    if (n_layers && frames && !bufs) return set_err(s, S2R_ERR_INVALID, "null output buffer");
    for (uint32_t i = 0; i < n_layers; i++)                     // process.rs:36: `offset.checked_add(16).expect("overflow")`
        if ((uint64_t)layers[i].offset + frames > 0xffffffffull) return set_err(s, S2R_ERR_OFFSET_OVERFLOW, "layer %u: offset %u + %zu frames overflows u32 (process.rs:36)", i, layers[i].offset, frames);
    std::vector<s2r_voice_state> st(total);
    std::memset(st.data(), 0, st.size() * sizeof(s2r_voice_state));
    for (uint32_t i = 0; i < n_layers; i++) {
        const s2r_layer_call &c = layers[i];
        s2r_voice_state &v = st[i];
        v.started = 1; v.released = c.has_release ? 1 : 0; v.program = c.program;
        v.current_frame_offset = c.offset; v.release_frame_offset = c.has_release ? c.release_offset : 0u;
        v.pitch_hz = c.pitch_hz; v.phase_accum = c.phase_accum; v.lpf_last = c.lpf_last; v.noise_seed = c.noise_seed; v.velocity = 1.0f;
        v.filt_x1 = c.filt_x1; v.filt_x2 = c.filt_x2; v.filt_y1 = c.filt_y1; v.filt_y2 = c.filt_y2; v.osc_z = c.osc_z;
    }
    int rc = s2r_import_state(s, st.data());
    if (rc != S2R_OK) return rc;
    if (frames && n_layers) {
        std::vector<float> rows((size_t)total * frames);
        rc = s2r_render_voices(s, rows.data(), frames, sample_rate_hz);
        if (rc != S2R_OK) return rc;
        std::memcpy(bufs, rows.data(), (size_t)n_layers * frames * sizeof(float));
    }
    rc = s2r_export_state(s, st.data());
    if (rc != S2R_OK) return rc;
    for (uint32_t i = 0; i < n_layers; i++) {
        s2r_layer_call &c = layers[i];
        const s2r_voice_state &v = st[i];
        c.phase_accum = v.phase_accum; c.lpf_last = v.lpf_last; c.noise_seed = v.noise_seed;
        c.filt_x1 = v.filt_x1; c.filt_x2 = v.filt_x2; c.filt_y1 = v.filt_y1; c.filt_y2 = v.filt_y2; c.osc_z = v.osc_z;
    }
    return S2R_OK;
}

int s2r_set_noise_seed(s2r_synth *s, uint32_t voice_index, uint32_t seed) {
    if (!s || s->parent || voice_index >= s->pool->size()) return S2R_ERR_INVALID;
    s->seed_override[voice_index] = seed;
    S2R_QUIESCE(s);
    fold_frame0_records(s);
    uint32_t local = 0;
    s2r_synth *sh = shard_of(s, voice_index, &local);
    if (sh) {
        S2R_HIP(s, hipSetDevice(sh->device));
        if (!sh->tpending.empty()) return set_err(s, S2R_ERR_INVALID, "set_noise_seed with timed events pending: fill first");
        EventSlot *ts = nullptr; const S2rTimedEvent *td = nullptr;
        int rc = flush_events(sh, sh->stream, &ts, &td);
        if (rc != S2R_OK) { s->err = sh->err; return rc; }
        S2R_HIP(s, hipMemcpyAsync(sh->v.seed + local, &s->seed_override[voice_index], sizeof(uint32_t), hipMemcpyHostToDevice, sh->stream));
        S2R_HIP(s, hipStreamSynchronize(sh->stream));
    }
    return S2R_OK;
}

uint32_t s2r_shard_voices(const s2r_synth *s) { return s ? s->shard_voices : 0; }
uint32_t s2r_block_voices(const s2r_synth *s) { return s ? s->block_voices : 0; }
uint32_t s2r_device_count(const s2r_synth *s) { return s ? (s->kids.empty() ? 1u : (uint32_t)s->kids.size()) : 0; }
uint64_t s2r_double_release_count(const s2r_synth *s) { return s ? s->double_release : 0; }

int s2r_set_flat_shortcut(s2r_synth *s, int enabled) {
    if (!s) return S2R_ERR_INVALID;
    S2R_QUIESCE(s);
    s->no_flat_shortcut = enabled == 0;
    for (s2r_synth *kid : s->kids) kid->no_flat_shortcut = s->no_flat_shortcut;
    return S2R_OK;
}

int s2r_set_coeff_stream(s2r_synth *s, int enabled) {
    if (!s) return S2R_ERR_INVALID;
    S2R_QUIESCE(s);
    s->use_tab = enabled != 0;                    // 0: coefficients in-lane; else from the patch's tables
    s->use_arg_events = enabled != 2 && enabled != 4;   // 2, 4: note events through their own launch, never in the kernel arguments
    s->bank_dirty = true;                         // (a resolved bank carries the choice in its entries' tab_valid)
    for (s2r_synth *kid : s->kids) { kid->use_tab = s->use_tab; kid->use_arg_events = s->use_arg_events; kid->bank_dirty = true; }
    return S2R_OK;
}

int s2r_set_low_latency(s2r_synth *s, int enabled) {
    if (!s) return S2R_ERR_INVALID;
    if (s->parent || !s->kids.empty()) return set_err(s, S2R_ERR_INVALID, "s2r_set_low_latency takes a single-device handle");
    S2R_QUIESCE(s);
    if (enabled && !s->res_host) {
        S2R_HIP(s, hipSetDevice(s->device));
        S2R_HIP(s, hipHostMalloc((void **)&s->res_host, 64 * sizeof(uint32_t), kHostPolled));
        std::memset(s->res_host, 0, 64 * sizeof(uint32_t));
        S2R_HIP(s, hipHostGetDevicePointer((void **)&s->res_dev, s->res_host, 0));
        s->res_cmd = s->res_host; s->res_cmd_dev = s->res_dev; s->res_cmd_vram = false;
        S2R_HIP(s, hipHostMalloc((void **)&s->res_gran, S2R_RES_GRANULE_FRAMES * sizeof(unsigned long long), kHostPolled));
        std::memset(s->res_gran, 0, S2R_RES_GRANULE_FRAMES * sizeof(unsigned long long));
        S2R_HIP(s, hipHostGetDevicePointer((void **)&s->res_gran_dev, s->res_gran, 0));
        // Where the whole of device memory is visible to the CPU (large BAR: hipDeviceAttributeIsLargeBar), the command lives
        // in fine-grained device memory: the CPU's stores cross the link once, posted, and the kernel's polls stay on the
        // device (a poll of host memory is a round trip over the link, and the command is seen a trip later).
        // S2R_RES_CMD_HOST=1 keeps it in host memory (measurement aid).
        int large_bar = 0;
        const char *force_host = std::getenv("S2R_RES_CMD_HOST");
        if (!(force_host && force_host[0] == '1') &&
            hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, s->device) == hipSuccess && large_bar) {
            uint32_t *v = nullptr;
            if (hipExtMallocWithFlags((void **)&v, 64 * sizeof(uint32_t), hipDeviceMallocFinegrained) == hipSuccess && v) {
                S2R_HIP(s, hipMemset(v, 0, 64 * sizeof(uint32_t)));
                S2R_HIP(s, hipDeviceSynchronize());
                s->res_cmd = v; s->res_cmd_dev = v; s->res_cmd_vram = true;
            } else (void)hipGetLastError();
        }
    }
    s->low_latency = enabled != 0;
    return S2R_OK;
}

int s2r_low_latency_active(const s2r_synth *s) { return (s && s->res_running) ? 1 : 0; }

int s2r_set_resident(s2r_synth *s, int enabled) {
    if (!s || s->parent) return S2R_ERR_INVALID;
    S2R_QUIESCE(s);
    std::vector<s2r_synth *> shards(s->kids.begin(), s->kids.end());
    if (shards.empty()) shards.push_back(s);
    // Resident kernels of one process on ONE device each hold a stream, and streams share the runtime's hardware queues (4
    // unless GPU_MAX_HW_QUEUES says otherwise): two resident kernels in one queue would run one after the other, each waiting
    // out the other's patience.  Shards of a device list that share a device stay resident only while every one of them (and
    // the parent's stream) has a queue to itself — a rehearsal of N shards on one device sets GPU_MAX_HW_QUEUES >= N + 1
    // before the first HIP call; N devices need nothing.
    int hw_queues = 4;
    { const char *e = std::getenv("GPU_MAX_HW_QUEUES"); if (e && std::atoi(e) > 0) hw_queues = std::atoi(e); }
    for (s2r_synth *t : shards) {
        int same_device = 0, grid_total = 0;
        for (s2r_synth *u : shards) if (u->device == t->device) { same_device++; grid_total += (int)u->n_blocks; }
        const bool queues_ok = shards.size() == 1 || (same_device + 1 <= hw_queues && grid_total <= t->n_cu);
        // (a shard whose grid cannot be resident as a whole — more workgroups than compute units, workgroups of more than 256
        // threads — or of a single workgroup keeps its launches; a single-workgroup handle has s2r_set_low_latency's kernel)
        if (enabled && queues_ok && t->n_blocks > 1u && (int)t->n_blocks <= t->n_cu && t->block_voices <= 256u) {
            const int rc = pool_setup(t);
            if (rc != S2R_OK) { s->err = t->err; return rc; }
        }
        t->resident = enabled != 0;
    }
    s->resident = enabled != 0;
    if (s->kids.empty() && s->n_blocks == 1u) return s2r_set_low_latency(s, enabled);
    return S2R_OK;
}

int s2r_resident_active(const s2r_synth *s) {
    if (!s) return 0;
    if (s->res_running || s->pool_running) return 1;
    for (const s2r_synth *kid : s->kids) if (kid->pool_running) return 1;
    return 0;
}

static size_t xg_block_bytes(const s2r_synth *s, uint32_t n_ranks) { return ((size_t)2 * n_ranks * s->cfg.max_frames) * sizeof(float) + 64; }

static int xg_bind(s2r_synth *s, void *block, bool owner, uint32_t rank, uint32_t n_ranks) {
    s->xg_block = block; s->xg_owner = owner; s->xg_rank = rank; s->xg_n = n_ranks;
    s->xg_rows = (float *)block;
    s->xg_done = (uint32_t *)((char *)block + (size_t)2 * n_ranks * s->cfg.max_frames * sizeof(float));
    s->xg_target[0] = s->xg_target[1] = 0;
    s->xg_on = true;
    return S2R_OK;
}

int s2r_exchange_create(s2r_synth *s, uint32_t n_ranks, void *handle_out, size_t handle_bytes) {
    if (!s || s->parent || !s->kids.empty() || n_ranks < 1 || n_ranks > 64 || !handle_out || handle_bytes < sizeof(hipIpcMemHandle_t)) return S2R_ERR_INVALID;
    if (s->xg_on) return set_err(s, S2R_ERR_INVALID, "the handle is in a process group already");
    if (!fused_shape_ok(s)) return set_err(s, S2R_ERR_INVALID, "a process group takes shards of more than one workgroup (of at most 256 voices unless the patch is a single one-pole one)");
    S2R_QUIESCE(s);
    S2R_HIP(s, hipSetDevice(s->device));
    void *block = nullptr;
    S2R_HIP(s, malloc_exchange(&block, xg_block_bytes(s, n_ranks)));
    S2R_HIP(s, hipMemset(block, 0, xg_block_bytes(s, n_ranks)));
    S2R_HIP(s, hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, block) != hipSuccess) { (void)hipFree(block); return set_err(s, S2R_ERR_HIP, "hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment?)"); }
    std::memcpy(handle_out, &h, sizeof h);
    return xg_bind(s, block, true, 0u, n_ranks);
}

int s2r_exchange_attach(s2r_synth *s, uint32_t rank, uint32_t n_ranks, const void *handle, size_t handle_bytes) {
    if (!s || s->parent || !s->kids.empty() || rank == 0 || rank >= n_ranks || n_ranks > 64 || !handle || handle_bytes < sizeof(hipIpcMemHandle_t)) return S2R_ERR_INVALID;
    if (s->xg_on) return set_err(s, S2R_ERR_INVALID, "the handle is in a process group already");
    if (!fused_shape_ok(s)) return set_err(s, S2R_ERR_INVALID, "a process group takes shards of more than one workgroup (of at most 256 voices unless the patch is a single one-pole one)");
    S2R_QUIESCE(s);
    S2R_HIP(s, hipSetDevice(s->device));
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof h);
    void *block = nullptr;
    S2R_HIP(s, hipIpcOpenMemHandle(&block, h, hipIpcMemLazyEnablePeerAccess));
    return xg_bind(s, block, false, rank, n_ranks);
}

int s2r_quiesce(s2r_synth *s) {
    if (!s) return S2R_ERR_INVALID;
    S2R_QUIESCE(s);
    return S2R_OK;
}

int s2r_set_timing(s2r_synth *s, int enabled) {
    if (!s) return S2R_ERR_INVALID;
    S2R_QUIESCE(s);
    s->timing = enabled != 0;
    s->timed = false;
    for (s2r_synth *kid : s->kids) { kid->timing = s->timing; kid->timed = false; }
    return S2R_OK;
}

float s2r_last_render_ms(s2r_synth *s) {
    if (s) (void)quiesce(s);
    if (s && !s->kids.empty()) {                  // the longest of the shards' render kernels
        float worst = -1.0f;
        for (s2r_synth *kid : s->kids) { const float ms = s2r_last_render_ms(kid); if (ms < 0.0f) return -1.0f; if (ms > worst) worst = ms; }
        return worst;
    }
    if (!s || !s->timing || !s->timed) return -1.0f;
    if (hipEventSynchronize(s->t1) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, s->t0, s->t1) != hipSuccess) return -1.0f;
    return ms;
}

const char *s2r_last_error(const s2r_synth *s) { return s ? s->err.c_str() : "null handle"; }

// Diagnostic builds only (-DS2R_STAMPS; tools/stamps.py): copies the last fill's per-wave phase stamps ([waves][16]
// s_memtime ticks) to `out`; returns the number of waves, 0 in a product build.  Not declared in s2r.h.
extern "C" uint32_t s2r_debug_read_stamps(s2r_synth *s, unsigned long long *out, uint32_t max_waves) {
#if defined(S2R_STAMPS)
    if (!s) return 0;
    (void)quiesce(s);
    const uint32_t waves = s->padded_voices / 64u;
    if (hipSetDevice(s->device) != hipSuccess) return 0;
    if (!s->stamps_dev) {
        if (hipMalloc((void **)&s->stamps_dev, (size_t)waves * 16 * sizeof(unsigned long long)) != hipSuccess) return 0;
        (void)hipMemset(s->stamps_dev, 0, (size_t)waves * 16 * sizeof(unsigned long long));
        return waves;                             // armed: the next fill writes them
    }
    if (!out) return waves;
    const uint32_t n = waves < max_waves ? waves : max_waves;
    (void)hipStreamSynchronize(s->stream);
    if (hipMemcpy(out, s->stamps_dev, (size_t)n * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
#else
    (void)s; (void)out; (void)max_waves;
    return 0;
#endif
}

// Diagnostic builds only (tools/gpu_timeline.py): arm (`out` == nullptr: room for max_launches render and mix launches from
// now on) or read back [n][2] = {first entry, last exit} in ticks of the GPU's 100 MHz clock, in launch order (a render
// launch and the mix that follows it take consecutive slots); returns the number of launches recorded.
extern "C" uint32_t s2r_debug_timeline(s2r_synth *s, unsigned long long *out, uint32_t max_launches) {
#if defined(S2R_STAMPS)
    if (!s) return 0;
    (void)quiesce(s);
    if (hipSetDevice(s->device) != hipSuccess) return 0;
    if (!out) {
        if (s->timeline_dev) { (void)hipFree(s->timeline_dev); s->timeline_dev = nullptr; }
        std::vector<unsigned long long> init((size_t)max_launches * 2);
        for (uint32_t i = 0; i < max_launches; i++) { init[2 * i] = ~0ull; init[2 * i + 1] = 0ull; }
        if (hipMalloc((void **)&s->timeline_dev, init.size() * sizeof(unsigned long long)) != hipSuccess) return 0;
        (void)hipMemcpy(s->timeline_dev, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice);
        s->timeline_n = 0; s->timeline_cap = max_launches;
        return max_launches;
    }
    (void)hipStreamSynchronize(s->stream);
    const uint32_t n = s->timeline_n < max_launches ? s->timeline_n : max_launches;
    if (hipMemcpy(out, s->timeline_dev, (size_t)n * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
#else
    (void)s; (void)out; (void)max_launches;
    return 0;
#endif
}

// ---- host-only helpers: the voice-allocation policy without a device (tests, front-ends
// that route events to shards) ----
struct s2r_voice_pool { S2rVoicePool pool; explicit s2r_voice_pool(uint32_t n) : pool(n) {} };

s2r_voice_pool *s2r_voice_pool_create(uint32_t total_voices) {
    if (total_voices == 0) return nullptr;
    return new (std::nothrow) s2r_voice_pool(total_voices);
}
void s2r_voice_pool_destroy(s2r_voice_pool *p) { delete p; }
uint32_t s2r_voice_pool_note_on(s2r_voice_pool *p, uint8_t note, float velocity) { return p->pool.note_on(note, velocity); }
int64_t s2r_voice_pool_note_off(s2r_voice_pool *p, uint8_t note) { return p->pool.note_off(note); }
void s2r_voice_pool_advance(s2r_voice_pool *p, uint64_t frames) { p->pool.advance(frames); }
uint32_t s2r_voice_pool_next_voice(const s2r_voice_pool *p) { return p->pool.next_voice(); }
void s2r_voice_pool_set_threads(s2r_voice_pool *p, uint32_t worker_threads, size_t batch_threshold) { if (p) p->pool.set_workers(worker_threads, batch_threshold); }
uint32_t s2r_voice_pool_resolve(s2r_voice_pool *p, const s2r_note_event *events, size_t n, uint32_t frames_moved, int64_t *voice_out) {
    if (!p || (!events && n) || !voice_out) return frames_moved;
    return p->pool.resolve_batch(reinterpret_cast<const S2rPolicyEvent *>(events), sizeof(s2r_note_event), n, frames_moved, voice_out,
                                 n ? &events[0].velocity : nullptr, sizeof(s2r_note_event));
}
int s2r_voice_pool_query(const s2r_voice_pool *p, uint32_t voice_index, s2r_voice_state *out) {
    if (!p || !out || voice_index >= p->pool.size()) return S2R_ERR_INVALID;
    const S2rHostVoice &v = p->pool.voice(voice_index);
    std::memset(out, 0, sizeof *out);
    out->note = v.note; out->started = v.started; out->released = v.released; out->velocity = v.velocity;
    if (v.started) {
        out->current_frame_offset = p->pool.offset_of(voice_index);
        if (v.released) out->release_frame_offset = p->pool.release_offset_of(voice_index);
    }
    return S2R_OK;
}

}  // extern "C"
