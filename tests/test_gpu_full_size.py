"""GPU parity at BASELINE.json's full sizes (VERDICT r2 "next round" item 1): exactly the path bench.py times, config [2]'s
SVF leg, and config [3]'s 1 048 576-voice pool as eight shards on one device — each against the CPU oracle, bit for bit
(per-voice rows) and through the documented mix tree (the oracle's own restatement of it, oracle/s2_oracle.c)."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import assert_bits_equal, make_patch, oracle_cfg_from_patch
from oracle import s2o
import synth2_amd as s2

pytestmark = pytest.mark.gpu

SR = 48000


def _threads():
    return max(1, min(32, len(os.sched_getaffinity(0))))


def test_bench_c3_path_at_65536_voices_against_the_oracle():
    """What `bench.py` times, at its size: `bench.make_c3_events(65536)` — 1 024 note-ons at frame 0 and ~1 024 note-offs
    on their own 16-frame boundaries per buffer — through s2r_note_events + s2r_fill_begin / s2r_fill_end with two
    buffers in flight (the MODE-2 render kernel, releases booked ahead, frame-0 events merged into the chains, the mix
    deferred to the next fill's chain-heads launch), for one whole period + 3 buffers.  Every buffer is compared bit for
    bit with the oracle driven the way the reference's caller drives Synth: MIDI applied between 16-frame sample() calls
    (s2_bin/src/main.rs:138-147; synth.rs:154-203), mixed through the documented tree."""
    import bench
    V, period = 65536, bench.PERIOD
    cyc = bench.make_c3_events(V, period)
    gpu = s2.Synth(V, max_frames=bench.FRAMES)
    gpu.load_patch("synth mySynth {\n\n}\n")               # example.synth2
    ora = s2o.OracleSynth(V)
    threads = _threads()
    queue = []
    n_buffers = period + 3
    n_timed = 0
    for k in range(n_buffers):
        ev = cyc[k % period]
        n_timed += int(np.count_nonzero(ev["frame"]))
        gpu.note_events(ev)
        gpu.sample_begin(bench.FRAMES, SR)
        pv = ora.render_events(ev, bench.FRAMES, SR, threads=threads)
        queue.append((k, s2o.mix_tree(pv, gpu.block_voices, 1)))
        del pv
        if len(queue) == 2:
            kk, want = queue.pop(0)
            assert_bits_equal(gpu.sample_end(np.empty(bench.FRAMES, dtype=np.float32)), want, "C3 bench path, buffer %d" % kk)
    kk, want = queue.pop(0)
    assert_bits_equal(gpu.sample_end(np.empty(bench.FRAMES, dtype=np.float32)), want, "C3 bench path, buffer %d" % kk)
    assert n_timed > 60000 and not ora.panicked
    # the state the run leaves behind: every voice's control words against the oracle's
    st = gpu.export_state()
    vs = np.ctypeslib.as_array(C.cast(ora.p.contents.voices, C.POINTER(C.c_uint8)), shape=(V, C.sizeof(s2o.Voice)))
    vo = np.frombuffer(vs.tobytes(), dtype=np.dtype({"names": ["has_current", "cfo", "has_release", "rfo"],
                                                     "formats": [np.int32, np.uint32, np.int32, np.uint32],
                                                     "offsets": [s2o.Voice.has_current.offset, s2o.Voice.current_frame_offset.offset,
                                                                 s2o.Voice.has_release.offset, s2o.Voice.release_frame_offset.offset],
                                                     "itemsize": C.sizeof(s2o.Voice)}))
    assert np.array_equal(st["started"] != 0, vo["has_current"] != 0)
    live = vo["has_current"] != 0
    assert np.array_equal(st["current_frame_offset"][live], vo["cfo"][live])
    assert np.array_equal((st["released"] != 0)[live], (vo["has_release"] != 0)[live])
    rel = live & (vo["has_release"] != 0)
    assert np.array_equal(st["release_frame_offset"][rel], vo["rfo"][rel])


def _leg_patch(text):
    """a bench.py config leg's patch text -> (synth2_amd.Patch, oracle LayerCfg)"""
    patch = s2.parse_patch(text)
    return patch, oracle_cfg_from_patch(patch)


def _whole_pool_on(voices):
    ev = np.zeros(voices, dtype=s2.NOTE_EVENT_DTYPE)
    ev["kind"] = 1; ev["note"] = 36 + np.arange(voices) % 61; ev["velocity"] = 1.0
    return ev


def test_bench_config2_leg_at_65536_voices_against_the_oracle():
    """What `bench.py`'s config_legs[0] times, as it times it: BASELINE config [2] as written — 65 536
    voices, saw + ADSR + SVF, the leg's own patch text — on `bench.make_c3_events(65536)` through s2r_note_events +
    s2r_fill_begin / s2r_fill_end with two buffers in flight (the general render kernel in MODE 2 on two streams, timed
    note-offs, frame-0 events merged into the chains).  The pool is filled first (one buffer), so that every note-on of the
    schedule steals the oldest voice as it does in the aged population the leg is timed on; then 22 buffers of the schedule.
    EVERY buffer is compared bit for bit with the oracle driven as the reference's caller drives Synth — MIDI applied
    between 16-frame sample() calls (s2_bin/src/main.rs:138-147), process.rs:306-379 with the filter call swapped — and
    mixed through the documented tree; no window, no self-comparison."""
    import bench
    V = 65536
    text = "synth c2 { lpf.kind = svf_lp; lpf.q = 1.4 }"           # bench.py main(): config_legs[0]
    assert text in open(bench.__file__).read()
    patch, cfg = _leg_patch(text)
    cyc = bench.make_c3_events(V, bench.PERIOD)
    gpu = s2.Synth(V, max_frames=bench.FRAMES)
    gpu.load_patch(text)
    ora = s2o.OracleSynth(V)
    ora.config = cfg
    threads = _threads()
    queue = []
    batches = [_whole_pool_on(V)] + [cyc[k % bench.PERIOD] for k in range(22)]
    n_timed = 0
    for k, ev in enumerate(batches):
        n_timed += int(np.count_nonzero(ev["frame"]))
        gpu.note_events(ev)
        gpu.sample_begin(bench.FRAMES, SR)
        pv = ora.render_events(ev, bench.FRAMES, SR, threads=threads)
        queue.append((k, s2o.mix_tree(pv, gpu.block_voices, 1)))
        del pv
        if len(queue) == 2:
            kk, want = queue.pop(0)
            assert_bits_equal(gpu.sample_end(np.empty(bench.FRAMES, dtype=np.float32)), want, "config [2] leg, buffer %d" % kk)
    kk, want = queue.pop(0)
    assert_bits_equal(gpu.sample_end(np.empty(bench.FRAMES, dtype=np.float32)), want, "config [2] leg, buffer %d" % kk)
    assert n_timed > 20000 and not ora.panicked


def test_bench_config4_leg_at_32768_voices_against_the_oracle():
    """What `bench.py`'s config_legs[1] times, as it times it: config [4]'s per-GPU share — 32 768 voices, DPW saw + SVF,
    4x oversampled — on `bench.make_c3_events(32768, PERIOD, 4096)` (note-offs on the internal rate's 16-frame boundaries)
    through s2r_note_events + s2r_fill_oversampled, one buffer at a time: the whole pool started, then five buffers of the
    schedule, each compared bit for bit with the oracle's 4 096 internal frames at 192 kHz (events between 16-frame calls),
    its tree and its decimator with the history carried from buffer to buffer."""
    import bench
    V, F4 = 32768, 4 * bench.FRAMES
    text = "synth c4 { osc.kind = dpw_saw; lpf.kind = svf_lp; lpf.q = 1.4 }"     # bench.py main(): config_legs[1]
    assert text in open(bench.__file__).read()
    patch, cfg = _leg_patch(text)
    cyc = bench.make_c3_events(V, bench.PERIOD, F4)
    gpu = s2.Synth(V, max_frames=F4)
    gpu.load_patch(text)
    ora = s2o.OracleSynth(V)
    ora.config = cfg
    threads = _threads()
    hist = np.zeros(62, dtype=np.float32)
    n_timed = 0
    for k, ev in enumerate([_whole_pool_on(V)] + [cyc[k % bench.PERIOD] for k in range(5)]):
        n_timed += int(np.count_nonzero(ev["frame"]))
        gpu.note_events(ev)
        got = gpu.sample_oversampled(bench.FRAMES, SR)
        pv = ora.render_events(ev, F4, 4 * SR, threads=threads)
        x = np.concatenate([hist, s2o.mix_tree(pv, gpu.block_voices, 1)])
        del pv
        assert_bits_equal(got, s2o.decimate4(x, bench.FRAMES), "config [4] leg, buffer %d" % k)
        hist = x[-62:]
    assert n_timed > 2000 and not ora.panicked


def test_config2_svf_at_65536_voices():
    """BASELINE config [2] as written — 65 536 voices, saw + ADSR + SVF (the build-defined state-variable filter at the
    modulated cutoff): a 2 048-voice window of per-voice rows bit for bit against the oracle, and the mix equal to the
    documented tree over the GPU's own rows (the association does not depend on the size), over the attack, a churned
    buffer and a buffer with timed note-offs."""
    voices = 65536
    patch = make_patch(lpf_kind=s2.FILT_SVF_LP, lpf_freq=900.0, lpf_q=1.4)
    a = s2.Synth(voices, max_frames=1024)
    b = s2.Synth(voices, max_frames=1024)
    a.set_patch(patch); b.set_patch(patch)
    ev = np.zeros(voices, dtype=s2.NOTE_EVENT_DTYPE)
    ev["kind"] = 1; ev["note"] = 36 + np.arange(voices) % 61; ev["velocity"] = 1.0
    a.note_events(ev); b.note_events(ev)
    W = 2048
    ora = s2o.OracleSynth(W)
    ora.config = oracle_cfg_from_patch(patch)
    for i in range(W):
        ora.note_on(int(ev["note"][i]))
    for k in range(4):
        if k == 2:
            # re-trigger the 2 048 oldest voices (the allocation policy takes them in index order: exactly the window)
            re = np.zeros(W, dtype=s2.NOTE_EVENT_DTYPE)
            re["kind"] = 1; re["note"] = 40 + np.arange(W) % 50; re["velocity"] = 1.0
            a.note_events(re); b.note_events(re)
            for i in range(W):
                ora.note_on(int(re["note"][i]))
        mix = a.sample(np.empty(1024, dtype=np.float32))
        pv = b.render_voices(1024)
        assert_bits_equal(pv[:W], ora.render_voices(1024, threads=_threads()), "SVF, first %d voices, buffer %d" % (W, k))
        assert_bits_equal(mix, s2o.mix_tree(pv, a.block_voices, 1), "SVF mix vs tree over the GPU's rows, buffer %d" % k)


def _poke_note_on(ora, j, note, program=0):
    """*voice = Voice { .. } (synth.rs:63-69) on oracle voice j, without the allocation policy"""
    v = ora.p.contents.voices[j]
    C.memset(C.byref(v), 0, C.sizeof(v))
    v.note = note; v.velocity = 1.0; v.has_current = 1; v.current_frame_offset = 0; v.has_release = 0; v.program = program


def _poke_note_off(ora, j):
    """synth.rs:74-75 on oracle voice j"""
    v = ora.p.contents.voices[j]
    if v.has_current and not v.has_release:
        v.has_release = 1; v.release_frame_offset = v.current_frame_offset


def test_config3_1048576_voices_as_8_shards_on_one_device():
    """BASELINE config [3] at its full pool: 1 048 576 voices dealt out to 8 shards in runs of 64 (what eight GPUs hold),
    all eight on this one device.  (a) windows of every shard's per-voice rows bit for bit against the oracle — the
    events that hit a window are found with the host-only allocation policy (s2r_voice_pool, itself checked against the
    oracle's O(V) policy on the CPU) and applied to the window's oracle voices directly; (b) every shard's partial row
    equals the oracle's tree over that shard's rows; (c) the rank-ordered sum of the eight partial rows is what ONE
    handle with a device list of eight (s2r_config.devices = {0 x 8}) returns."""
    import torch
    V, N, G, F = 1048576, 8, 64, 1024
    per = V // N
    shards = [s2.Synth(V, max_frames=F, shard_interleave=G, shard_index=k, shard_count=N) for k in range(N)]
    twins = [s2.Synth(V, max_frames=F, shard_interleave=G, shard_index=k, shard_count=N) for k in range(N)]
    one = s2.Synth(V, max_frames=F, shard_interleave=G, devices=[0] * N)
    assert one.device_count == N and one.shard_voices == V
    policy = s2.VoicePool(V)
    # windows (pool indices): the first voices, a stretch in the middle, the last voices (where note_off's
    # "last active voice holding the note" lands)
    win = np.concatenate([np.arange(0, 1024), np.arange(V // 2 - 512, V // 2 + 512), np.arange(V - 1024, V)])
    slot_of = {int(p): i for i, p in enumerate(win)}
    ora = s2o.OracleSynth(win.size)
    rows = torch.zeros((N, F), dtype=torch.float32, device="cuda")
    out = torch.zeros(F, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.RandomState(11)
    for b in range(3):
        if b == 0:
            ev = np.zeros(V, dtype=s2.NOTE_EVENT_DTYPE)
            ev["kind"] = 1; ev["note"] = 36 + np.arange(V) % 61; ev["velocity"] = 1.0
        else:
            n = 4096
            ev = np.zeros(n, dtype=s2.NOTE_EVENT_DTYPE)
            ev["kind"] = rng.randint(0, 2, n); ev["note"] = 36 + rng.randint(0, 61, n); ev["velocity"] = 1.0
            ev["frame"] = np.sort(rng.randint(0, F // 16, n)) * 16 if b == 2 else 0
        for s in shards + [one]:
            s.note_events(ev)
        # Per-voice rows come from the twins.  s2r_render_voices takes no timed events, so the twins (and the window's
        # oracle voices) are driven the way the reference's caller would: the events of a boundary, then the frames up
        # to the next one.  The window's truth: replay the policy, apply what lands in the window, render.
        want = np.zeros((win.size, F), dtype=np.float32)
        pvs = [np.zeros((per, F), dtype=np.float32) for _ in range(N)]
        bounds = sorted(set([0] + [int(f) for f in ev["frame"]] + [F]))
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            batch = ev[(ev["frame"] == lo)]
            for e in batch:
                if e["kind"] == 1:
                    idx = policy.note_on(int(e["note"]), 1.0)
                    if idx in slot_of:
                        _poke_note_on(ora, slot_of[idx], int(e["note"]))
                else:
                    idx = policy.note_off(int(e["note"]))
                    if idx in slot_of:
                        _poke_note_off(ora, slot_of[idx])
            ub = batch.copy(); ub["frame"] = 0
            for t in twins:
                t.note_events(ub)
            n = hi - lo
            want[:, lo:hi] = ora.render_voices(n, SR, threads=_threads())
            policy.advance(n)
            for k, t in enumerate(twins):
                pvs[k][:, lo:hi] = t.render_voices(n, SR)
        for k, s in enumerate(shards):
            s.fill_device(rows[k].data_ptr(), F, SR, stream)
        s2.sum_partials_device(rows.data_ptr(), N, F, out.data_ptr(), stream)
        torch.cuda.synchronize()
        rows_h = rows.cpu().numpy()
        acc = np.zeros(F, dtype=np.float32)
        for k in range(N):
            pool_idx = s2.synth.shard_pool_indices(V, k, N, G)
            sel = np.nonzero(np.isin(pool_idx, win))[0]
            assert sel.size == win.size // N
            got = pvs[k][sel]
            assert_bits_equal(got, want[[slot_of[int(p)] for p in pool_idx[sel]]], "shard %d, window rows, buffer %d" % (k, b))
            part = s2o.mix_tree_partial(pvs[k], shards[k].block_voices)
            assert_bits_equal(rows_h[k], part, "shard %d partial row, buffer %d" % (k, b))
            acc = acc + part
        assert_bits_equal(out.cpu().numpy(), acc, "rank-ordered sum of the 8 partial rows, buffer %d" % b)
        assert_bits_equal(one.sample(np.empty(F, dtype=np.float32)), acc, "one handle over a device list of 8, buffer %d" % b)


def test_two_buffers_in_flight_where_two_workgroups_share_a_compute_unit():
    """131 072 voices are two render workgroups per compute unit — every register taken: the fills of s2r_fill_begin then
    stay on ONE stream (on two, the render kernel's waiting workgroups would keep out the chain-heads workgroups they wait
    for: DESIGN.md 4.2b).  The buffers of s2r_fill_begin / s2r_fill_end with timed events equal those of the same events
    through s2r_fill, and the smaller pool next to it (one workgroup per unit: two streams) says the same."""
    import bench
    for voices in (131072, 65536):
        cyc = bench.make_c3_events(voices, bench.PERIOD)
        a = s2.Synth(voices, max_frames=bench.FRAMES)
        b = s2.Synth(voices, max_frames=bench.FRAMES)
        pending = []
        for k in range(10):
            ev = cyc[k % bench.PERIOD]
            a.note_events(ev); a.sample_begin(bench.FRAMES, SR)
            b.note_events(ev)
            pending.append((k, b.sample(np.empty(bench.FRAMES, dtype=np.float32), SR).copy()))
            if len(pending) == 2:
                kk, want = pending.pop(0)
                assert_bits_equal(a.sample_end(np.empty(bench.FRAMES, dtype=np.float32)), want, "%d voices, buffer %d" % (voices, kk))
        kk, want = pending.pop(0)
        assert_bits_equal(a.sample_end(np.empty(bench.FRAMES, dtype=np.float32)), want, "%d voices, buffer %d" % (voices, kk))


def test_config4_262144_voices_as_8_shards_4x_oversampled_dpw_svf():
    """BASELINE config [4] at its full pool: 262 144 voices, alias-suppressed (DPW) saw + SVF, rendered at 4 x 48 kHz and
    decimated, dealt out to 8 shards in runs of 64 — all eight on this one device.  (a) windows of every shard's per-voice
    rows at the 4x rate bit for bit against the oracle; (b) every shard's partial row equals the oracle's tree over that
    shard's rows; (c) the rank-ordered sum of the eight partial rows is what ONE handle over a device list of eight returns
    at the 4x rate; (d) that handle's s2r_fill_oversampled output is the oracle's decimator applied to the same mix."""
    import torch
    from helpers import oracle_cfg_from_patch
    V, N, G, frames_out = 262144, 8, 64, 64
    F4 = 4 * frames_out
    per = V // N
    patch = make_patch(osc_kind=s2.OSC_DPW_SAW, lpf_kind=s2.FILT_SVF_LP, lpf_freq=4000.0, lpf_q=1.2, mod_env_to_lpf_freq=2.0)
    def mk(**kw):
        s = s2.Synth(V, max_frames=F4, shard_interleave=G, **kw)
        s.set_patch(patch)
        return s
    shards = [mk(shard_index=k, shard_count=N) for k in range(N)]
    twins = [mk(shard_index=k, shard_count=N) for k in range(N)]
    one = mk(devices=[0] * N)
    one_os = mk(devices=[0] * N)
    ev = np.zeros(V, dtype=s2.NOTE_EVENT_DTYPE)
    ev["kind"] = 1; ev["note"] = (np.arange(V) * 13) % 90 + 24; ev["velocity"] = 1.0
    for s in shards + twins + [one, one_os]:
        s.note_events(ev)                                    # (an empty pool hands out its voices in index order: voice i plays ev[i])
    win = np.concatenate([np.arange(0, 512), np.arange(V - 512, V)])
    slot_of = {int(p): i for i, p in enumerate(win)}
    ora = s2o.OracleSynth(win.size)
    ora.config = oracle_cfg_from_patch(patch)
    for p in win:
        ora.note_on(int(ev["note"][p]))
    rows = torch.zeros((N, F4), dtype=torch.float32, device="cuda")
    out = torch.zeros(F4, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    hist = np.zeros(62, dtype=np.float32)
    for b in range(2):
        want = ora.render_voices(F4, 4 * SR, threads=_threads())
        for k, s in enumerate(shards):
            s.fill_device(rows[k].data_ptr(), F4, 4 * SR, stream)
        s2.sum_partials_device(rows.data_ptr(), N, F4, out.data_ptr(), stream)
        torch.cuda.synchronize()
        rows_h = rows.cpu().numpy()
        acc = np.zeros(F4, dtype=np.float32)
        for k in range(N):
            pv = twins[k].render_voices(F4, 4 * SR)
            pool_idx = s2.synth.shard_pool_indices(V, k, N, G)
            sel = np.nonzero(np.isin(pool_idx, win))[0]
            assert sel.size == win.size // N
            assert_bits_equal(pv[sel], want[[slot_of[int(p)] for p in pool_idx[sel]]], "shard %d, window rows at 192 kHz, buffer %d" % (k, b))
            part = s2o.mix_tree_partial(pv, shards[k].block_voices)
            assert_bits_equal(rows_h[k], part, "shard %d partial row, buffer %d" % (k, b))
            acc = acc + part
            del pv
        assert_bits_equal(out.cpu().numpy(), acc, "rank-ordered sum of the 8 partial rows, buffer %d" % b)
        assert_bits_equal(one.sample(np.empty(F4, dtype=np.float32), 4 * SR), acc, "one handle over a device list of 8 at the 4x rate, buffer %d" % b)
        x = np.concatenate([hist, acc])
        assert_bits_equal(one_os.sample_oversampled(frames_out, SR), s2o.decimate4(x, frames_out), "decimated output of the device-list handle, buffer %d" % b)
        hist = x[-62:]
