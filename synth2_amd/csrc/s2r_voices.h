// s2r_voices.h — host-side voice pool: the allocation and release policy of
// s2_lib::try3::synth::Synth (synth.rs:61-120) for a pool of any size, in O(1) per event
// instead of the reference's O(V) scans (a bucket queue and per-note bitmaps), with identical choices:
//
//   next_voice (synth.rs:101-120): the voice with the greatest current_frame_offset, an idle
//     voice counting as u32::MAX, first index on ties (strict `>`).  All started voices
//     advance in lockstep (every fill adds `frames` to each), so "greatest offset" ==
//     "earliest start"; we keep voices ordered by (start_clock, index) and idle voices by
//     index.  (u32 saturation of very old offsets is unreachable: the reference panics at
//     that point, process.rs:36, and s2r_fill reports S2R_ERR_OFFSET_OVERFLOW before it.)
//   note_off (synth.rs:72-96): the LAST index whose note matches and which is active
//     (started and not yet released).
//
// State is struct-of-arrays (a note per voice is one byte: the whole pool's notes stay in the cache), "released" is not
// stored at all — a started voice is released exactly when it is not in its note's set of active voices — and a BATCH of
// events (s2r_note_events' whole array) can be resolved by several threads (resolve_batch): every rank of an N-GPU run
// simulates the WHOLE pool's events, 16 384 per buffer at 8 x 65 536 voices under bench.py's C3 schedule, and the
// reference's policy is sequential in them — but only through ONE queue: which voice a note_on takes depends on the
// earlier note_ons alone (the oldest start; whether a voice is released does not matter, synth.rs:101-120), and which
// voice a note_off(n) releases depends on the events of note n alone plus the note_ons that took a voice away from n.
// So the caller's thread runs the queue (phase A: the chosen voice and the note it held, per note_on) and P workers, each
// owning the notes n with n % P == its number, replay the batch behind it and keep their notes' sets (phase B).
//
// Pure C++ (no HIP) so it is unit-testable on a CPU-only machine.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <utility>
#include <vector>

// x / d and x % d for a divisor fixed at create (shard geometry): every note event of a sharded handle is mapped from its
// pool index to (shard, local voice), and a hardware division per step of that map is what the event path cannot afford
// (2 048 events per buffer, six divisions each: tens of microseconds).  Round-up multiply-shift (Granlund-Montgomery):
// exact for every 32-bit x.
struct FastDiv {
    uint32_t d = 1, mul = 0, sh1 = 0, sh2 = 0;
    void set(uint32_t div) {
        d = div ? div : 1u;
        uint32_t l = 0; while ((1ull << l) < d) l++;
        mul = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1ull);
        sh1 = l ? 1u : 0u; sh2 = l ? l - 1u : 0u;
    }
    inline uint32_t div(uint32_t x) const { const uint32_t t = (uint32_t)(((uint64_t)mul * x) >> 32); return (t + ((x - t) >> sh1)) >> sh2; }
    inline uint32_t mod(uint32_t x) const { return x - div(x) * d; }
};

// one voice as the entry points that ask about single voices see it (s2r_export_state, s2r_voice_pool_query)
struct S2rHostVoice {
    uint8_t note = 0;
    bool started = false;
    bool released = false;
    float velocity = 0.0f;
    uint64_t start_clock = 0;      // pool clock (frames) at note_on
    uint64_t release_clock = 0;    // pool clock at note_off
};

// The sets of ACTIVE voices (started, not released), one per note, with "greatest member" in a few word operations: per note a
// bitmap and two levels of summary bits (one bit per 64-bit word below), all 256 notes in ONE zero-initialised block — a set is
// reached by address arithmetic, not through a container per note (three pointers to fetch per operation, and again after
// every byte the policy stores: a uint8_t store may alias anything), and pages no note ever touches are never backed.
// The summaries are kept LAZILY: set() raises them, clear() touches the bitmap alone, and a summary bit over a word that has
// gone empty is taken down by the next search that runs into it — every stale bit is cleaned once per set() that raised it, so
// the searches stay O(1) amortised, and a voice taken over by a note_on (the common event of a full pool) costs its old
// note's set one word.
class S2rNoteSets {
  public:
    S2rNoteSets() = default;
    ~S2rNoteSets() { std::free(mem_); }
    S2rNoteSets(const S2rNoteSets &) = delete;
    S2rNoteSets &operator=(const S2rNoteSets &) = delete;
    // empties every set (fresh zero pages rather than a memset over all of them)
    void init(uint32_t n_voices) {
        std::free(mem_);
        w0_ = ((size_t)n_voices + 63) / 64; w1_ = (w0_ + 63) / 64; w2_ = (w1_ + 63) / 64;
        stride_ = (w2_ + w1_ + w0_ + 7) & ~(size_t)7;
        mem_ = static_cast<uint64_t *>(std::calloc(stride_ * 256, sizeof(uint64_t)));
        if (!mem_) throw std::bad_alloc();
    }
    bool test(uint32_t note, uint32_t i) const { return ((l0(note)[i >> 6] >> (i & 63)) & 1ull) != 0; }
    void set(uint32_t note, uint32_t i) {
        uint64_t *m = mem_ + note * stride_;
        m[i >> 18] |= 1ull << ((i >> 12) & 63);
        m[w2_ + (i >> 12)] |= 1ull << ((i >> 6) & 63);
        m[w2_ + w1_ + (i >> 6)] |= 1ull << (i & 63);
    }
    // removes i if it is a member (a voice taken over by a note_on may or may not still be held)
    void clear(uint32_t note, uint32_t i) { mem_[note * stride_ + w2_ + w1_ + (i >> 6)] &= ~(1ull << (i & 63)); }
    // removes and returns the greatest member, or -1 (cleans the stale summary bits it meets)
    int64_t take_last(uint32_t note) {
        uint64_t *l2 = mem_ + note * stride_, *l1 = l2 + w2_, *l0 = l1 + w1_;
        for (size_t w2 = w2_; w2-- > 0;) {
            while (l2[w2]) {
                const size_t b2 = 63 - (size_t)__builtin_clzll(l2[w2]), w1 = (w2 << 6) + b2;
                if (!l1[w1]) { l2[w2] &= ~(1ull << b2); continue; }
                const size_t b1 = 63 - (size_t)__builtin_clzll(l1[w1]), w0 = (w1 << 6) + b1;
                if (!l0[w0]) { l1[w1] &= ~(1ull << b1); continue; }
                const size_t b0 = 63 - (size_t)__builtin_clzll(l0[w0]);
                // (its own word gone empty is cleaned here, branch-free: the next search of this note would meet it first)
                const uint64_t left0 = l0[w0] & ~(1ull << b0);
                l0[w0] = left0;
                const uint64_t left1 = l1[w1] & ~((uint64_t)(left0 == 0) << b1);
                l1[w1] = left1;
                l2[w2] &= ~((uint64_t)(left1 == 0) << b2);
                return (int64_t)((w0 << 6) + b0);
            }
        }
        return -1;
    }
  private:
    const uint64_t *l0(uint32_t note) const { return mem_ + note * stride_ + w2_ + w1_; }
    uint64_t *mem_ = nullptr;
    size_t w0_ = 0, w1_ = 0, w2_ = 0, stride_ = 0;
};

// one event of a batch as resolve_batch sees it (s2r_note_event's first four bytes)
struct S2rPolicyEvent { uint8_t kind, note; uint16_t frame; };
#define S2R_POLICY_NOTE_OFF 0
#define S2R_POLICY_NOTE_ON 1

class S2rVoicePool {
  public:
    explicit S2rVoicePool(uint32_t total) : n_(total), note_(total, 0), started_(total, 0), velocity_(total, 0.0f), start_(total, 0), release_(total, 0) { rebuild(); }
    ~S2rVoicePool() { stop_workers(); }
    S2rVoicePool(const S2rVoicePool &) = delete;
    S2rVoicePool &operator=(const S2rVoicePool &) = delete;
    uint32_t size() const { return n_; }
    uint64_t clock() const { return now_; }
    S2rHostVoice voice(uint32_t i) const {
        S2rHostVoice v;
        v.note = note_[i]; v.started = started_[i] != 0; v.released = v.started && !active_.test(v.note, i);
        v.velocity = velocity_[i]; v.start_clock = start_[i]; v.release_clock = v.released ? release_[i] : 0;
        return v;
    }

    // synth.rs:101-120
    uint32_t next_voice() const {
        if (idle_head_ < idle_.size()) return idle_[idle_head_];
        return front();
    }

    // synth.rs:61-70; returns the chosen index
    uint32_t note_on(uint8_t note, float velocity) {
        uint8_t old_note; bool was_started;
        const uint32_t i = take_voice(note, velocity, &old_note, &was_started);
        if (was_started) active_.clear(old_note, i);            // stolen, possibly while still held
        active_.set(note, i);
        return i;
    }

    // synth.rs:72-96; returns the released index or -1 when no active voice holds `note`
    int64_t note_off(uint8_t note) {
        const int64_t i = active_.take_last(note);
        if (i < 0) return -1;
        release_[(size_t)i] = now_;
        return i;
    }

    // every started voice's offset grows by `frames` (synth.rs:197)
    void advance(uint64_t frames) { now_ += frames; }

    // A whole batch of events at once — what a loop of advance / note_on / note_off over `ev` computes, with the same
    // results: voice_out[k] = the voice event k takes (note_on) or releases (note_off; -1: none).  `t0`: the frames the
    // clock has already moved inside the fill the events belong to (an event at frame f first moves it by f - t); returns
    // the last event's frame (>= t0).  Batches of at least `mt_threshold` events are resolved by the worker threads.
    uint32_t resolve_batch(const S2rPolicyEvent *ev, size_t stride_bytes, size_t n, uint32_t t0, int64_t *voice_out, const float *velocity = nullptr,
                           size_t velocity_stride = 0) {
        if (n >= mt_threshold_ && workers_wanted_ > 0 && n_ <= (1u << kHandoverVoiceBits)) return resolve_mt(ev, stride_bytes, n, t0, voice_out, velocity, velocity_stride);
        // One thread: the note_ons that follow one another at one clock are resolved as a RUN (note_on_run: the queue handed
        // out by the bucket instead of popped and pushed per event — most of what an event cost), the rest one by one.
        const char *base = reinterpret_cast<const char *>(ev);
        const char *vbase = reinterpret_cast<const char *>(velocity);
        uint32_t t = t0;
        size_t k = 0;
        while (k < n) {
            const S2rPolicyEvent &e = *reinterpret_cast<const S2rPolicyEvent *>(base + k * stride_bytes);
            if (e.frame > t) { now_ += e.frame - t; t = e.frame; }
            if (e.kind == S2R_POLICY_NOTE_ON) {
                size_t r = k + 1;
                while (r < n) {
                    const S2rPolicyEvent &f = *reinterpret_cast<const S2rPolicyEvent *>(base + r * stride_bytes);
                    if (f.kind != S2R_POLICY_NOTE_ON || f.frame > t) break;
                    r++;
                }
                note_on_run(base, stride_bytes, vbase, velocity_stride, k, r, voice_out);
                k = r;
            } else if (e.kind == S2R_POLICY_NOTE_OFF) k = note_off_run(base, stride_bytes, k, n, t, voice_out);
            else voice_out[k++] = -1;
        }
        return t;
    }
    // worker threads for resolve_batch (0: always the calling thread alone) and the batch size from which they are used
    void set_workers(uint32_t n_workers, size_t threshold = 4096) {
        if (n_workers != workers_wanted_) stop_workers();
        workers_wanted_ = n_workers > 16u ? 16u : n_workers; mt_threshold_ = threshold;
        for (uint32_t n = 0; n < 256; n++) owner_[n] = (uint8_t)(n % (workers_wanted_ + 1u));     // (partition 0 is the caller's)
    }
    uint32_t workers() const { return workers_wanted_; }

    // current_frame_offset of a started voice (saturating like synth.rs:197)
    uint32_t offset_of(uint32_t i) const {
        const uint64_t d = now_ - start_[i];
        return d > 0xffffffffull ? 0xffffffffu : (uint32_t)d;
    }
    uint32_t release_offset_of(uint32_t i) const {
        const uint64_t d = release_[i] - start_[i];
        return d > 0xffffffffull ? 0xffffffffu : (uint32_t)d;
    }
    // greatest current_frame_offset among started voices (0 if none)
    uint64_t oldest_offset() const {
        if (buckets_.empty()) return 0;
        return now_ - buckets_.front().clock;
    }

    // restore one voice (import_state); call rebuild() afterwards
    void set_voice(uint32_t i, uint8_t note, bool started, bool released, uint32_t offset,
                   uint32_t release_offset, float velocity) {
        note_[i] = note; started_[i] = started ? 1 : 0; velocity_[i] = velocity;
        if (started && now_ < offset) pending_min_clock_ = std::max<uint64_t>(pending_min_clock_, offset);
        pending_.push_back({i, offset, release_offset, started && released});
    }
    void rebuild() {
        // (a voice set_voice did not touch keeps its state: released or not is read off the sets before they are rebuilt)
        std::vector<uint8_t> rel(n_, 0);
        for (uint32_t i = 0; i < n_; i++) rel[i] = (started_[i] && !active_.test(note_[i], i)) ? 1 : 0;
        for (const auto &p : pending_) rel[p.i] = p.released ? 1 : 0;
        if (now_ < pending_min_clock_) {
            // shift the whole time base forward so no start_clock underflows; offsets are unchanged
            const uint64_t shift = pending_min_clock_ - now_;
            for (uint32_t i = 0; i < n_; i++) { start_[i] += shift; release_[i] += shift; }
            now_ += shift;
        }
        for (const auto &p : pending_) {
            if (started_[p.i]) {
                start_[p.i] = now_ - p.offset;
                release_[p.i] = p.released ? start_[p.i] + p.release_offset : 0;
            }
        }
        pending_.clear(); pending_min_clock_ = 0;
        idle_.clear(); idle_head_ = 0;
        buckets_.clear(); ring_.clear(); ring_head_ = 0;
        std::vector<std::pair<uint64_t, uint32_t>> all;
        active_.init(n_);
        for (uint32_t i = 0; i < n_; i++) {
            if (!started_[i]) { idle_.push_back(i); continue; }
            all.push_back({start_[i], i});
            if (!rel[i]) active_.set(note_[i], i);
        }
        std::sort(all.begin(), all.end());
        for (const auto &e : all) push(e.first, e.second);
    }

  private:
    struct Pending { uint32_t i, offset, release_offset; bool released; };

    // Started voices ordered by (start_clock, index): every one of them sits exactly once in `ring_` (the live part is
    // [ring_head_, ring_.size()); compacted when the dead part outgrows it), cut into per-clock BUCKETS.  The pool clock
    // never goes back, so a push goes to the last bucket (or opens one); inside a bucket the indices are sorted lazily
    // when it reaches the front.  pop-min and push are O(1) amortised.
    struct Bucket { uint64_t clock; size_t begin, end; bool sorted; };
    uint32_t front() const {
        Bucket &b = buckets_.front();
        if (!b.sorted) { std::sort(ring_.begin() + (std::ptrdiff_t)b.begin, ring_.begin() + (std::ptrdiff_t)b.end); b.sorted = true; }
        return ring_[b.begin];
    }
    void pop() {
        Bucket &b = buckets_.front();
        ring_head_ = ++b.begin;
        if (b.begin == b.end) buckets_.pop_front();
    }
    void push(uint64_t clock, uint32_t i) {
        if (buckets_.empty() || buckets_.back().clock != clock) {
            if (ring_head_ > ring_.size() / 2 + 1024 && ring_.size() >= 4096) compact();
            buckets_.push_back(Bucket{clock, ring_.size(), ring_.size(), true});
        }
        Bucket &b = buckets_.back();
        if (buckets_.size() == 1 && b.sorted && b.begin > 0 && b.begin < b.end && b.begin == ring_head_ && i <= ring_[b.begin]) {
            ring_[--b.begin] = i; ring_head_ = b.begin;       // re-queued at its own clock: still the minimum
            return;
        }
        if (b.end > b.begin && i < ring_[b.end - 1]) b.sorted = false;
        ring_.push_back(i);
        b.end = ring_.size();
    }
    void compact() {
        const size_t dead = ring_head_;
        ring_.erase(ring_.begin(), ring_.begin() + (std::ptrdiff_t)dead);
        for (Bucket &b : buckets_) { b.begin -= dead; b.end -= dead; }
        ring_head_ = 0;
    }

    // the queue's half of a note_on: which voice, what it held
    uint32_t take_voice(uint8_t note, float velocity, uint8_t *old_note, bool *was_started) {
        uint32_t i;
        if (idle_head_ < idle_.size()) { i = idle_[idle_head_++]; *was_started = false; *old_note = 0; started_[i] = 1; }
        else { i = front(); pop(); *was_started = true; *old_note = note_[i]; }   // (the chosen voice is always the queue's minimum)
        note_[i] = note; velocity_[i] = velocity; start_[i] = now_;
        push(now_, i);
        return i;
    }

    // Events [k, r) of a batch: note_ons at the present clock.  The same choices as r - k calls of note_on: idle voices in index
    // order, then the queue's front — whole stretches of the front buckets, each sorted once — and every taken voice goes to the
    // back of the queue at the present clock.  (Once the front bucket IS the present clock's — every voice of the pool taken at
    // this very clock — the rest of the run goes through note_on, whose re-queueing rule is a case of its own.)
    // SETS = false (phase A of the threaded form): the sets are left alone; h[k] = the voice | the note it held << 23 (256: none).
    template <bool SETS = true>
    void note_on_run(const char *ev, size_t stride, const char *vel, size_t vstride, size_t k, size_t r, int64_t *voice_out, uint32_t *h = nullptr) {
        while (k < r) {
            const bool stolen = idle_head_ >= idle_.size();
            if (stolen && (buckets_.empty() || buckets_.front().clock == now_)) break;
            // where the stretch goes: the back of the queue, at the present clock (push() for a stretch; before the stretch is
            // looked at, because making room may move the ring)
            if (buckets_.empty() || buckets_.back().clock != now_) {
                if (ring_head_ > ring_.size() / 2 + 1024 && ring_.size() >= 4096) compact();
                buckets_.push_back(Bucket{now_, ring_.size(), ring_.size(), true});
            }
            size_t c, from;
            if (!stolen) { from = idle_head_; c = std::min(r - k, idle_.size() - idle_head_); }
            else { front(); const Bucket &b = buckets_.front(); from = b.begin; c = std::min(r - k, b.end - b.begin); }   // (front() sorts the bucket if it has to)
            const size_t at = ring_.size();
            ring_.resize(at + c);
            const uint32_t *src = stolen ? ring_.data() + from : idle_.data() + from;
            // (locals: the bytes stored below may alias anything the compiler has to fetch through `this`)
            uint8_t *note_of = note_.data(), *started = started_.data();
            float *velocity = velocity_.data();
            uint64_t *start = start_.data();
            const uint64_t now = now_;
            const char *e = ev + k * stride, *v = vel ? vel + k * vstride : nullptr;
            int64_t *out = voice_out + k;
            for (size_t q = 0; q < c; q++, e += stride) {
                const uint32_t i = src[q];
                const uint8_t note = reinterpret_cast<const S2rPolicyEvent *>(e)->note;
                if (!SETS) h[k + q] = i | (stolen ? (uint32_t)note_of[i] : 256u) << kHandoverVoiceBits;
                if (!stolen) started[i] = 1;
                else if (SETS) active_.clear(note_of[i], i);               // taken over, possibly while still held
                note_of[i] = note; start[i] = now;
                if (v) { velocity[i] = *reinterpret_cast<const float *>(v); v += vstride; } else velocity[i] = 1.0f;
                if (SETS) active_.set(note, i);
                out[q] = i;
            }
            // a stretch is ascending (idle voices by index, a sorted bucket's part); the bucket it joins may end above its start
            Bucket &back = buckets_.back();
            if (back.end > back.begin && src[0] < ring_[back.end - 1]) back.sorted = false;
            std::memcpy(ring_.data() + at, src, c * sizeof(uint32_t));
            back.end = at + c;
            if (!stolen) idle_head_ += c;
            else {
                Bucket &b = buckets_.front();
                ring_head_ = (b.begin += c);
                if (b.begin == b.end) buckets_.pop_front();
            }
            k += c;
        }
        for (; k < r; k++) {
            const S2rPolicyEvent &e = *reinterpret_cast<const S2rPolicyEvent *>(ev + k * stride);
            const float velocity = vel ? *reinterpret_cast<const float *>(vel + k * vstride) : 1.0f;
            if (SETS) voice_out[k] = note_on(e.note, velocity);
            else {
                uint8_t old; bool was;
                const uint32_t i = take_voice(e.note, velocity, &old, &was);
                voice_out[k] = i; h[k] = i | (was ? (uint32_t)old : 256u) << kHandoverVoiceBits;
            }
        }
    }
    // Events [k, ...) of a batch while they are note_offs: returns the first that is not (or n).  `t`: the frames the clock has
    // moved inside the fill.
    size_t note_off_run(const char *ev, size_t stride, size_t k, size_t n, uint32_t &t, int64_t *voice_out) {
        uint64_t now = now_;
        uint32_t tt = t;
        uint64_t *release = release_.data();
        const char *e = ev + k * stride;
        for (; k < n; k++, e += stride) {
            const S2rPolicyEvent &x = *reinterpret_cast<const S2rPolicyEvent *>(e);
            if (x.kind != S2R_POLICY_NOTE_OFF) break;
            if (x.frame > tt) { now += x.frame - tt; tt = x.frame; }
            const int64_t i = active_.take_last(x.note);
            voice_out[k] = i;
            if (i >= 0) release[(size_t)i] = now;
        }
        now_ = now; t = tt;
        return k;
    }

    // ---- resolve_batch on several threads ----
    // FORK-JOIN, in two phases.  Phase A, the caller alone: the queue — every note_on's voice, the note that voice held (what
    // note_on_run computes, without the sets) — for the whole batch.  Phase B, the caller and the workers side by side, each
    // owning the notes n with owner_[n] == its number: the batch replayed against its own notes' sets (a note_on's voice into
    // the new note's set and out of the old note's, a note_off's search).  Phase A's hand-over is one 32-bit word per event,
    // complete before anybody reads it: the first threaded form streamed it to the workers WHILE phase A wrote it, and every
    // line made the trip between cores while still being written (measured: 2-5x slower than one thread).  The workers'
    // results — the voices their note_offs released — go to lists of their own and are put in place by the caller afterwards
    // (written straight into the caller's array, a worker's note_offs next to another's, they were bouncing lines again).
    static constexpr uint32_t kHandoverVoiceBits = 23;         // (larger pools: one thread)
    struct Job {
        const char *ev = nullptr; size_t stride = 0, n = 0;
        uint32_t t0 = 0; uint64_t now0 = 0;
        std::vector<uint32_t> h;                               // voice | old note << 23 (old note 256: the voice was idle)
    };
    uint32_t resolve_mt(const S2rPolicyEvent *ev, size_t stride, size_t n, uint32_t t0, int64_t *voice_out, const float *velocity, size_t velocity_stride) {
        start_workers();
        Job &j = job_;
        j.ev = reinterpret_cast<const char *>(ev); j.stride = stride; j.n = n;
        if (j.h.size() < n) j.h.resize(n);
        j.t0 = t0; j.now0 = now_;
        // phase A
        const char *vbase = reinterpret_cast<const char *>(velocity);
        uint32_t t = t0;
        size_t k = 0;
        while (k < n) {
            const S2rPolicyEvent &e = *reinterpret_cast<const S2rPolicyEvent *>(j.ev + k * stride);
            if (e.frame > t) { now_ += e.frame - t; t = e.frame; }
            if (e.kind == S2R_POLICY_NOTE_ON) {
                size_t r = k + 1;
                while (r < n) {
                    const S2rPolicyEvent &f = *reinterpret_cast<const S2rPolicyEvent *>(j.ev + r * stride);
                    if (f.kind != S2R_POLICY_NOTE_ON || f.frame > t) break;
                    r++;
                }
                note_on_run<false>(j.ev, stride, vbase, velocity_stride, k, r, voice_out, j.h.data());
                k = r;
            } else voice_out[k++] = -1;
        }
        // phase B
        generation_.fetch_add(1, std::memory_order_release);
        wake_workers();
        replay_sets(0);
        for (auto &w : workers_) while (w->done.load(std::memory_order_acquire) != generation_.load(std::memory_order_relaxed)) cpu_pause();
        // the note_offs' voices, in event order per owner, back into the caller's array
        const int64_t *from[17];
        from[0] = released0_.data();
        for (size_t w = 0; w < workers_.size(); w++) from[w + 1] = workers_[w]->released.data();
        const uint8_t *owner = owner_;
        const char *e = j.ev;
        for (size_t q = 0; q < n; q++, e += stride) {
            const S2rPolicyEvent &x = *reinterpret_cast<const S2rPolicyEvent *>(e);
            if (x.kind == S2R_POLICY_NOTE_OFF) voice_out[q] = *from[owner[x.note]]++;
        }
        return t;
    }
    // phase B, partition `me` (0: the caller): the sets of the notes it owns, event by event
    void replay_sets(uint32_t me) {
        const Job &j = job_;
        std::vector<int64_t> &rel = me ? workers_[me - 1]->released : released0_;
        rel.clear();
        const uint8_t *owner = owner_;
        const uint32_t *h = j.h.data();
        uint32_t t = j.t0; uint64_t now = j.now0;
        const char *e = j.ev;
        for (size_t k = 0; k < j.n; k++, e += j.stride) {
            const S2rPolicyEvent &x = *reinterpret_cast<const S2rPolicyEvent *>(e);
            if (x.frame > t) { now += x.frame - t; t = x.frame; }            // (the clock, as phase A moved it)
            if (x.kind == S2R_POLICY_NOTE_ON) {
                const uint32_t voice = h[k] & ((1u << kHandoverVoiceBits) - 1u), old = h[k] >> kHandoverVoiceBits;
                if (old < 256u && owner[old] == me) active_.clear(old, voice);
                if (owner[x.note] == me) active_.set(x.note, voice);
            } else if (x.kind == S2R_POLICY_NOTE_OFF && owner[x.note] == me) {
                const int64_t i = active_.take_last(x.note);
                rel.push_back(i);
                if (i >= 0) {
                    // The clock of the voice's LATEST release.  A voice released under this partition's note, taken over by a
                    // later note_on and released again under another partition's note is written by both, in either order:
                    // clocks only grow with the events, so the greater value is the later release's.
                    uint64_t *r = &release_[(size_t)i];
                    uint64_t cur = __atomic_load_n(r, __ATOMIC_RELAXED);
                    while (cur < now && !__atomic_compare_exchange_n(r, &cur, now, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
                }
            }
        }
    }
    struct alignas(128) Worker { std::thread th; std::atomic<uint64_t> done{0}; std::vector<int64_t> released; };
    static void cpu_pause() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        asm volatile("yield" ::: "memory");
#endif
    }
    void start_workers() {
        if (!workers_.empty()) return;
        quit_.store(false);
        const uint32_t P = workers_wanted_;
        for (uint32_t w = 0; w < P; w++) {
            workers_.emplace_back(new Worker());
            Worker *me = workers_.back().get();
            me->done.store(generation_.load());
            me->th = std::thread([this, me, w] {
                uint64_t seen = me->done.load();
                for (;;) {
                    // spin for a while (the next batch of a caller in a loop comes within tens of microseconds), then sleep
                    uint64_t g = seen; uint32_t spins = 0;
                    while ((g = generation_.load(std::memory_order_acquire)) == seen && !quit_.load(std::memory_order_relaxed)) {
                        if (++spins < 200000u) cpu_pause();
                        else {
                            std::unique_lock<std::mutex> lk(mu_);
                            sleepers_++;
                            cv_.wait(lk, [&] { return generation_.load(std::memory_order_acquire) != seen || quit_.load(); });
                            sleepers_--;
                        }
                    }
                    if (quit_.load()) return;
                    replay_sets(w + 1);
                    seen = g;
                    me->done.store(g, std::memory_order_release);
                }
            });
        }
    }
    void wake_workers() {
        if (sleepers_ > 0) { std::lock_guard<std::mutex> lk(mu_); cv_.notify_all(); }
    }
    void stop_workers() {
        if (workers_.empty()) return;
        { std::lock_guard<std::mutex> lk(mu_); quit_.store(true); cv_.notify_all(); }
        for (auto &w : workers_) if (w->th.joinable()) w->th.join();
        workers_.clear();
    }

    uint32_t n_;
    std::vector<uint8_t> note_, started_;
    std::vector<float> velocity_;
    std::vector<uint64_t> start_, release_;
    std::vector<uint32_t> idle_;        // ascending; consumed from idle_head_ (voices never go idle again)
    size_t idle_head_ = 0;
    mutable std::vector<uint32_t> ring_;
    size_t ring_head_ = 0;
    mutable std::deque<Bucket> buckets_;
    S2rNoteSets active_;
    std::vector<Pending> pending_;
    uint64_t pending_min_clock_ = 0;
    uint64_t now_ = 0;
    // worker threads
    uint32_t workers_wanted_ = 0;
    size_t mt_threshold_ = 4096;
    std::vector<std::unique_ptr<Worker>> workers_;
    std::atomic<uint64_t> generation_{0};
    std::atomic<bool> quit_{false};
    std::atomic<int> sleepers_{0};
    std::mutex mu_;
    std::condition_variable cv_;
    Job job_;
    uint8_t owner_[256] = {0};           // which partition of the threaded form keeps a note's set
    std::vector<int64_t> released0_;     // partition 0's note_offs
};
