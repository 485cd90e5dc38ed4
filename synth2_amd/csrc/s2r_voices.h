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
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
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

// A set of voice indices with "greatest member" in three word operations: a bitmap and two
// levels of summary bits (one bit per 64-bit word below).
class S2rIndexSet {
  public:
    void init(uint32_t n) {
        l0_.assign(((size_t)n + 63) / 64, 0);
        l1_.assign((l0_.size() + 63) / 64, 0);
        l2_.assign((l1_.size() + 63) / 64, 0);
    }
    bool ready() const { return !l0_.empty(); }
    bool test(uint32_t i) const { return ready() && ((l0_[i >> 6] >> (i & 63)) & 1ull) != 0; }
    void set(uint32_t i) {
        uint64_t &w = l0_[i >> 6];
        const bool was_empty = w == 0;
        w |= 1ull << (i & 63);
        if (!was_empty) return;                   // (the summaries already say so)
        l1_[i >> 12] |= 1ull << ((i >> 6) & 63);
        l2_[i >> 18] |= 1ull << ((i >> 12) & 63);
    }
    // removes i if it is a member (a voice taken over by a note_on may or may not still be held)
    void clear(uint32_t i) {
        if (!ready()) return;
        uint64_t &w = l0_[i >> 6];
        const uint64_t bit = 1ull << (i & 63);
        if (!(w & bit)) return;
        if ((w &= ~bit) != 0) return;
        if ((l1_[i >> 12] &= ~(1ull << ((i >> 6) & 63))) != 0) return;
        l2_[i >> 18] &= ~(1ull << ((i >> 12) & 63));
    }
    // greatest member or -1
    int64_t last() const {
        for (size_t w2 = l2_.size(); w2-- > 0;) {
            if (!l2_[w2]) continue;
            const size_t w1 = (w2 << 6) + (63 - (size_t)__builtin_clzll(l2_[w2]));
            const size_t w0 = (w1 << 6) + (63 - (size_t)__builtin_clzll(l1_[w1]));
            return (int64_t)((w0 << 6) + (63 - (size_t)__builtin_clzll(l0_[w0])));
        }
        return -1;
    }
  private:
    std::vector<uint64_t> l0_, l1_, l2_;
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
        v.note = note_[i]; v.started = started_[i] != 0; v.released = v.started && !active_[v.note].test(i);
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
        if (was_started) active_[old_note].clear(i);           // stolen, possibly while still held
        mark_active(note, i);
        return i;
    }

    // synth.rs:72-96; returns the released index or -1 when no active voice holds `note`
    int64_t note_off(uint8_t note) {
        S2rIndexSet &a = active_[note];
        if (!a.ready()) return -1;
        const int64_t i = a.last();
        if (i < 0) return -1;
        a.clear((uint32_t)i);
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
        if (n >= mt_threshold_ && workers_wanted_ > 0) return resolve_mt(ev, stride_bytes, n, t0, voice_out, velocity, velocity_stride);
        uint32_t t = t0;
        for (size_t k = 0; k < n; k++) {
            const S2rPolicyEvent &e = *reinterpret_cast<const S2rPolicyEvent *>(reinterpret_cast<const char *>(ev) + k * stride_bytes);
            if (e.frame > t) { now_ += e.frame - t; t = e.frame; }
            if (e.kind == S2R_POLICY_NOTE_ON) {
                const float vel = velocity ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(velocity) + k * velocity_stride) : 1.0f;
                voice_out[k] = note_on(e.note, vel);
            } else if (e.kind == S2R_POLICY_NOTE_OFF) voice_out[k] = note_off(e.note);
            else voice_out[k] = -1;
        }
        return t;
    }
    // worker threads for resolve_batch (0: always the calling thread alone) and the batch size from which they are used
    void set_workers(uint32_t n_workers, size_t threshold = 4096) {
        if (n_workers != workers_wanted_) stop_workers();
        workers_wanted_ = n_workers > 16u ? 16u : n_workers; mt_threshold_ = threshold;
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
        for (uint32_t i = 0; i < n_; i++) rel[i] = (started_[i] && !active_[note_[i]].test(i)) ? 1 : 0;
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
        for (int n = 0; n < 256; n++) active_[n] = S2rIndexSet();
        for (uint32_t i = 0; i < n_; i++) {
            if (!started_[i]) { idle_.push_back(i); continue; }
            all.push_back({start_[i], i});
            if (!rel[i]) mark_active(note_[i], i);
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

    // active = started and not yet released; one index set per note, created on first use
    void mark_active(uint8_t note, uint32_t i) {
        S2rIndexSet &a = active_[note];
        if (!a.ready()) a.init(n_);
        a.set(i);
    }

    // ---- resolve_batch on several threads ----
    // What phase A hands to the workers, one 8-byte word per event, written once by the caller's thread and only read by the
    // others (no line is written from two cores): the voice a note_on took, the note it held (256: none).  The workers' own
    // results — the voices their note_offs released — go to lists of their own and are put in place by the caller afterwards:
    // results written straight into the caller's array, a worker's note_offs next to the caller's note_ons, made every event a
    // cache line bouncing between cores (measured: the threaded form five times SLOWER than one thread).
    struct Handover { uint32_t voice; uint16_t old_note; uint16_t pad; };
    struct Job {
        const char *ev = nullptr; size_t stride = 0, n = 0;
        uint32_t t0 = 0; uint64_t now0 = 0;
        std::vector<Handover> h;
    };
    uint32_t resolve_mt(const S2rPolicyEvent *ev, size_t stride, size_t n, uint32_t t0, int64_t *voice_out, const float *velocity, size_t velocity_stride) {
        start_workers();
        Job &j = job_;
        j.ev = reinterpret_cast<const char *>(ev); j.stride = stride; j.n = n;
        for (size_t k = 0; k < n; k++) {                          // (no set is created while the workers run)
            const S2rPolicyEvent &e = *reinterpret_cast<const S2rPolicyEvent *>(j.ev + k * stride);
            if (e.kind == S2R_POLICY_NOTE_ON && !active_[e.note].ready()) active_[e.note].init(n_);
        }
        if (j.h.size() < n) j.h.resize(n);
        j.t0 = t0; j.now0 = now_;
        a_done_.store(0, std::memory_order_relaxed);
        generation_.fetch_add(1, std::memory_order_release);
        wake_workers();
        // phase A, this thread: the queue.  Published to the workers every 128 events.
        uint32_t t = t0;
        for (size_t k = 0; k < n; k++) {
            const S2rPolicyEvent &e = *reinterpret_cast<const S2rPolicyEvent *>(j.ev + k * stride);
            if (e.frame > t) { now_ += e.frame - t; t = e.frame; }
            if (e.kind == S2R_POLICY_NOTE_ON) {
                const float vel = velocity ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(velocity) + k * velocity_stride) : 1.0f;
                uint8_t old; bool was;
                const uint32_t i = take_voice(e.note, vel, &old, &was);
                voice_out[k] = i;
                j.h[k] = Handover{i, (uint16_t)(was ? old : 256u), 0};
            } else voice_out[k] = -1;
            if ((k & 255u) == 255u) a_done_.store(k + 1, std::memory_order_release);
        }
        a_done_.store(n, std::memory_order_release);
        for (auto &w : workers_) while (w->done.load(std::memory_order_acquire) != generation_.load(std::memory_order_relaxed)) cpu_pause();
        // the workers' note_offs, in event order per worker, back into the caller's array
        const uint32_t P = (uint32_t)workers_.size();
        size_t cur[16] = {0};
        for (size_t k = 0; k < n; k++) {
            const S2rPolicyEvent &e = *reinterpret_cast<const S2rPolicyEvent *>(j.ev + k * stride);
            if (e.kind == S2R_POLICY_NOTE_OFF) { const uint32_t w = e.note % P; voice_out[k] = workers_[w]->released[cur[w]++]; }
        }
        return t;
    }
    // phase B, worker `me` of `P`: the sets of the notes it owns, event by event behind phase A
    void worker_batch(uint32_t me, uint32_t P) {
        const Job &j = job_;
        std::vector<int64_t> &rel = workers_[me]->released;
        rel.clear();
        size_t avail = 0;
        uint32_t t = j.t0; uint64_t now = j.now0;
        for (size_t k = 0; k < j.n; k++) {
            while (k >= avail) { avail = a_done_.load(std::memory_order_acquire); if (k >= avail) cpu_pause(); }
            const S2rPolicyEvent &e = *reinterpret_cast<const S2rPolicyEvent *>(j.ev + k * j.stride);
            if (e.frame > t) { now += e.frame - t; t = e.frame; }            // (the clock, as phase A moves it)
            if (e.kind == S2R_POLICY_NOTE_ON) {
                const Handover h = j.h[k];
                if (h.old_note < 256u && h.old_note % P == me) active_[h.old_note].clear(h.voice);
                if (e.note % P == me) active_[e.note].set(h.voice);
            } else if (e.kind == S2R_POLICY_NOTE_OFF && e.note % P == me) {
                S2rIndexSet &a = active_[e.note];
                const int64_t i = a.ready() ? a.last() : -1;
                rel.push_back(i);
                if (i >= 0) {
                    a.clear((uint32_t)i);
                    // The clock of the voice's LATEST release.  A voice released under this worker's note, taken over by a
                    // later note_on and released again under another worker's note is written by both, in either order:
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
            me->th = std::thread([this, me, w, P] {
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
                    worker_batch(w, P);
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
    S2rIndexSet active_[256];
    std::vector<Pending> pending_;
    uint64_t pending_min_clock_ = 0;
    uint64_t now_ = 0;
    // worker threads
    uint32_t workers_wanted_ = 0;
    size_t mt_threshold_ = 4096;
    std::vector<std::unique_ptr<Worker>> workers_;
    std::atomic<uint64_t> generation_{0};
    std::atomic<size_t> a_done_{0};
    std::atomic<bool> quit_{false};
    std::atomic<int> sleepers_{0};
    std::mutex mu_;
    std::condition_variable cv_;
    Job job_;
};
