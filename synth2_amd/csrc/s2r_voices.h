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
// Pure C++ (no HIP) so it is unit-testable on a CPU-only machine.
#pragma once
#include <algorithm>
#include <cstdint>
#include <deque>
#include <functional>
#include <queue>
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

struct S2rHostVoice {
    uint8_t note = 0;
    bool started = false;
    bool released = false;
    float velocity = 0.0f;
    uint64_t start_clock = 0;      // pool clock (frames) at note_on
    uint64_t release_clock = 0;    // pool clock at note_off
};

// A set of voice indices with "greatest member" in three word operations: a bitmap and two
// levels of summary bits (one bit per 64-bit word below).  Every rank of an N-GPU run simulates the
// WHOLE pool's events, so note_off's "last active voice holding this note" is on its critical path.
class S2rIndexSet {
  public:
    void init(uint32_t n) {
        l0_.assign(((size_t)n + 63) / 64, 0);
        l1_.assign((l0_.size() + 63) / 64, 0);
        l2_.assign((l1_.size() + 63) / 64, 0);
    }
    bool ready() const { return !l0_.empty(); }
    void set(uint32_t i) {
        l0_[i >> 6] |= 1ull << (i & 63);
        l1_[i >> 12] |= 1ull << ((i >> 6) & 63);
        l2_[i >> 18] |= 1ull << ((i >> 12) & 63);
    }
    void clear(uint32_t i) {
        if ((l0_[i >> 6] &= ~(1ull << (i & 63))) != 0) return;
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

class S2rVoicePool {
  public:
    explicit S2rVoicePool(uint32_t total) : voices_(total) { rebuild(); }
    uint32_t size() const { return (uint32_t)voices_.size(); }
    uint64_t clock() const { return now_; }
    const S2rHostVoice &voice(uint32_t i) const { return voices_[i]; }

    // synth.rs:101-120
    uint32_t next_voice() const {
        if (idle_head_ < idle_.size()) return idle_[idle_head_];
        return started_front().second;
    }

    // synth.rs:61-70; returns the chosen index
    uint32_t note_on(uint8_t note, float velocity) {
        uint32_t i;
        if (idle_head_ < idle_.size()) {
            i = idle_[idle_head_++];
        } else {
            // the chosen voice is always the queue's minimum, so it never holds stale entries
            i = started_front().second;
            started_pop();
            const S2rHostVoice &old = voices_[i];
            if (!old.released) active_[old.note].clear(i);     // stolen while still held
        }
        S2rHostVoice &v = voices_[i];
        v.note = note; v.velocity = velocity;
        v.started = true; v.released = false;
        v.start_clock = now_; v.release_clock = 0;
        started_push(now_, i);
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
        S2rHostVoice &v = voices_[(size_t)i];
        v.released = true;
        v.release_clock = now_;
        return i;
    }

    // every started voice's offset grows by `frames` (synth.rs:197)
    void advance(uint64_t frames) { now_ += frames; }

    // current_frame_offset of a started voice (saturating like synth.rs:197)
    uint32_t offset_of(uint32_t i) const {
        const uint64_t d = now_ - voices_[i].start_clock;
        return d > 0xffffffffull ? 0xffffffffu : (uint32_t)d;
    }
    uint32_t release_offset_of(uint32_t i) const {
        const uint64_t d = voices_[i].release_clock - voices_[i].start_clock;
        return d > 0xffffffffull ? 0xffffffffu : (uint32_t)d;
    }
    // greatest current_frame_offset among started voices (0 if none)
    uint64_t oldest_offset() const {
        if (buckets_.empty()) return 0;
        return now_ - started_front().first;
    }

    // restore one voice (import_state); call rebuild() afterwards
    void set_voice(uint32_t i, uint8_t note, bool started, bool released, uint32_t offset,
                   uint32_t release_offset, float velocity) {
        S2rHostVoice &v = voices_[i];
        v.note = note; v.started = started; v.released = started && released; v.velocity = velocity;
        if (started && now_ < offset) pending_min_clock_ = std::max<uint64_t>(pending_min_clock_, offset);
        pending_.push_back({i, offset, release_offset});
    }
    void rebuild() {
        if (now_ < pending_min_clock_) {
            // shift the whole time base forward so no start_clock underflows; offsets are unchanged
            const uint64_t shift = pending_min_clock_ - now_;
            for (auto &v : voices_) { v.start_clock += shift; v.release_clock += shift; }
            now_ += shift;
        }
        for (const auto &p : pending_) {
            S2rHostVoice &v = voices_[p.i];
            if (v.started) {
                v.start_clock = now_ - p.offset;
                v.release_clock = v.released ? v.start_clock + p.release_offset : 0;
            }
        }
        pending_.clear(); pending_min_clock_ = 0;
        idle_.clear(); idle_head_ = 0;
        buckets_.clear();
        std::vector<std::pair<uint64_t, uint32_t>> all;
        for (int n = 0; n < 256; n++) active_[n] = S2rIndexSet();
        for (uint32_t i = 0; i < voices_.size(); i++) {
            const S2rHostVoice &v = voices_[i];
            if (!v.started) { idle_.push_back(i); continue; }
            all.push_back({v.start_clock, i});
            if (!v.released) mark_active(v.note, i);
        }
        std::sort(all.begin(), all.end());
        for (const auto &e : all) started_push(e.first, e.second);
    }

  private:
    struct Pending { uint32_t i, offset, release_offset; };

    // Started voices ordered by (start_clock, index).  The pool clock never goes back, so keys
    // arrive in non-decreasing clock order: a deque of per-clock buckets, each an index list that
    // is sorted lazily when it reaches the front.  pop-min and push are O(1) amortised (one
    // std::sort per bucket), against two ~19-level heap operations on a half-million-entry heap.
    struct Bucket {
        uint64_t clock;
        std::vector<uint32_t> idx;
        size_t head = 0;
        bool sorted = true;
    };
    std::pair<uint64_t, uint32_t> started_front() const {     // (lazy sort: buckets_ is mutable)
        Bucket &b = buckets_.front();
        if (!b.sorted) { std::sort(b.idx.begin() + (std::ptrdiff_t)b.head, b.idx.end()); b.sorted = true; }
        return {b.clock, b.idx[b.head]};
    }
    void started_pop() {
        Bucket &b = buckets_.front();
        if (++b.head == b.idx.size()) buckets_.pop_front();
    }
    void started_push(uint64_t clock, uint32_t i) {
        if (buckets_.empty() || buckets_.back().clock != clock) {
            buckets_.emplace_back();
            buckets_.back().clock = clock;
        }
        Bucket &b = buckets_.back();
        if (b.head > 0 && b.head < b.idx.size() && b.sorted && i <= b.idx[b.head]) {
            b.idx[--b.head] = i;                  // re-queued at its own clock: still the minimum
            return;
        }
        if (b.idx.size() > b.head && i < b.idx.back()) b.sorted = false;
        b.idx.push_back(i);
    }

    // active = started and not yet released; one index set per note, created on first use
    void mark_active(uint8_t note, uint32_t i) {
        S2rIndexSet &a = active_[note];
        if (!a.ready()) a.init((uint32_t)voices_.size());
        a.set(i);
    }

    std::vector<S2rHostVoice> voices_;
    std::vector<uint32_t> idle_;        // ascending; consumed from idle_head_ (voices never go idle again)
    size_t idle_head_ = 0;
    mutable std::deque<Bucket> buckets_;   // one entry per started voice
    S2rIndexSet active_[256];
    std::vector<Pending> pending_;
    uint64_t pending_min_clock_ = 0;
    uint64_t now_ = 0;
};
