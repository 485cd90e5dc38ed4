"""GPU parity of the ONE-LAUNCH fill (the render kernel builds its own chain heads from the fill's records grouped by
workgroup; the last workgroups to finish add the rows up: S2rMixTail) and of the POOL-RESIDENT kernel
(s2r_set_resident, DESIGN.md 4.2d: the shard's whole grid stays on the device, a fill is a posted command), bit for bit
against the CPU oracle driven as the reference's caller drives Synth — MIDI applied between 16-frame sample() calls
(s2_bin/src/main.rs:138-147, synth.rs:154-203) — and against the launches they replace."""
import os
import time

import numpy as np
import pytest

from helpers import assert_bits_equal, make_patch, oracle_cfg_from_patch
from oracle import s2o
import synth2_amd as s2

pytestmark = pytest.mark.gpu
SR = 48000


def _threads():
    return max(1, min(32, len(os.sched_getaffinity(0))))


def _random_batch(rng, n, frames, timed=True, notes=(36, 97)):
    ev = np.zeros(n, dtype=s2.NOTE_EVENT_DTYPE)
    ev["kind"] = rng.randint(0, 2, n)
    ev["note"] = rng.randint(notes[0], notes[1], n)
    ev["velocity"] = 1.0
    if timed:
        ev["frame"] = np.sort(rng.randint(0, max(1, frames // 16), n)) * 16
    return ev


def _drive(gpu, ora, batches, frames_of, what, groups=1, ring=True, pause=None):
    """the batches through s2r_note_events + s2r_fill_begin / s2r_fill_end with two buffers in flight (ring) or s2r_fill, every
    buffer against the oracle's rows through the documented tree"""
    queue = []
    for k, ev in enumerate(batches):
        frames = frames_of(k)
        gpu.note_events(ev)
        if ring:
            gpu.sample_begin(frames, SR)
        else:
            got_now = gpu.sample(np.empty(frames, dtype=np.float32), SR).copy()
        pv = ora.render_events(ev, frames, SR, threads=_threads())
        want = s2o.mix_tree(pv, gpu.block_voices, groups)
        del pv
        if not ring:
            assert_bits_equal(got_now, want, "%s, buffer %d" % (what, k))
            continue
        queue.append((k, frames, want))
        if pause and k in pause:
            time.sleep(pause[k])
        if len(queue) == 2:
            kk, ff, w = queue.pop(0)
            assert_bits_equal(gpu.sample_end(np.empty(ff, dtype=np.float32)), w, "%s, buffer %d" % (what, kk))
    while queue:
        kk, ff, w = queue.pop(0)
        assert_bits_equal(gpu.sample_end(np.empty(ff, dtype=np.float32)), w, "%s, buffer %d" % (what, kk))


@pytest.mark.parametrize("voices,ring", [(2048, False), (2048, True), (131072, True)])
def test_one_launch_fill_against_the_oracle(voices, ring):
    """the one-launch form without a resident kernel: synchronous fills (every pool of more than one workgroup takes it), and
    the fills of s2r_fill_begin where the two streams do not apply (131 072 voices: two workgroups per compute unit) — timed
    and untimed events, fills with a scalar tail, a fill without events, more events than ride in the kernel arguments"""
    rng = np.random.RandomState(voices % 977 + (7 if ring else 0))
    gpu = s2.Synth(voices, max_frames=1024)
    ora = s2o.OracleSynth(voices)
    n_big = 3000 if voices <= 2048 else 20000
    batches = [_random_batch(rng, n_big, 1024, timed=False), _random_batch(rng, 700, 1024), np.zeros(0, dtype=s2.NOTE_EVENT_DTYPE),
               _random_batch(rng, 40, 1000, timed=False), _random_batch(rng, 900, 1000), _random_batch(rng, 300, 16), _random_batch(rng, 2500, 1024)]
    lens = [1024, 1024, 1024, 1000, 1000, 16, 1024]
    _drive(gpu, ora, batches, lambda k: lens[k], "one launch per fill, %d voices, %s" % (voices, "ring" if ring else "s2r_fill"), ring=ring)
    assert not gpu.resident_active


@pytest.mark.parametrize("osc", [s2.OSC_SAW, s2.OSC_SINE])
def test_pool_resident_kernel_against_the_oracle(osc):
    """s2r_set_resident on a pool of eight workgroups: ring fills and synchronous fills through the pool-resident kernel —
    timed and untimed events, restarts inside fills, fills of every shape — then a knob that stops it, and on again"""
    rng = np.random.RandomState(31 + osc)
    patch = make_patch(osc_kind=osc, mod_env_to_osc_freq=0.0 if osc == s2.OSC_SAW else 0.6, noise=0.0 if osc == s2.OSC_SAW else 0.2)
    patch.amp_env.attack_ms = 5.0; patch.amp_env.decay_ms = 20.0; patch.amp_env.release_ms = 15.0; patch.mod_env.decay_ms = 30.0
    voices = 2048
    gpu = s2.Synth(voices, max_frames=1024)
    gpu.set_patch(patch)
    gpu.set_resident(True)
    ora = s2o.OracleSynth(voices)
    ora.config = oracle_cfg_from_patch(patch)
    batches = [_random_batch(rng, 2500, 1024, timed=False)] + [_random_batch(rng, int(rng.randint(0, 900)), 1024) for _ in range(10)]
    _drive(gpu, ora, batches, lambda k: 1024, "pool-resident, osc %d, ring" % osc)
    assert gpu.resident_active
    lens = [1024, 1000, 16, 1, 512, 777]
    batches = [_random_batch(rng, int(rng.randint(0, 600)), lens[k]) for k in range(6)]
    _drive(gpu, ora, batches, lambda k: lens[k], "pool-resident, osc %d, s2r_fill" % osc, ring=False)
    assert gpu.resident_active
    gpu.set_flat_shortcut(True)                                  # (any knob stops the kernel)
    assert not gpu.resident_active
    batches = [_random_batch(rng, 500, 1024) for _ in range(4)]
    _drive(gpu, ora, batches, lambda k: 1024, "pool-resident again, osc %d" % osc)
    assert gpu.resident_active
    gpu.set_resident(False)
    assert not gpu.resident_active
    _drive(gpu, ora, [_random_batch(rng, 300, 1024)], lambda k: 1024, "switched off, osc %d" % osc)
    assert not gpu.resident_active


@pytest.mark.parametrize("case", ["svf", "lp2_fm", "bank"])
def test_pool_resident_kernel_of_the_other_render_kernel(case):
    """the pool-resident form of s2r_render_general_kernel: a patch with the SVF, one with a dsp_filters.rs filter and mod-to-pitch,
    and a patch bank with program changes (per-lane patches, a DPW voice among them) — ring fills (two-stream form) and
    synchronous ones (chain heads and mix inside the kernel) against the oracle"""
    rng = np.random.RandomState({"svf": 3, "lp2_fm": 4, "bank": 5}[case])
    voices = 2048
    gpu = s2.Synth(voices, max_frames=1024)
    ora = s2o.OracleSynth(voices)
    if case == "bank":
        bank = [make_patch(osc_kind=s2.OSC_SAW), make_patch(osc_kind=s2.OSC_SINE, lpf_kind=s2.FILT_LP2, lpf_freq=900.0, noise=0.2),
                make_patch(osc_kind=s2.OSC_DPW_SAW, lpf_kind=s2.FILT_SVF_LP, lpf_q=1.3)]
        gpu.set_patch_bank(bank)
        ora.set_bank([oracle_cfg_from_patch(q) for q in bank])
    else:
        patch = (make_patch(lpf_kind=s2.FILT_SVF_LP, lpf_freq=900.0, lpf_q=1.4) if case == "svf" else
                 make_patch(osc_kind=s2.OSC_TRIANGLE, lpf_kind=s2.FILT_LP2, lpf_freq=1500.0, lpf_damping=0.8, mod_env_to_osc_freq=0.5, noise=0.1))
        gpu.set_patch(patch)
        ora.config = oracle_cfg_from_patch(patch)
    gpu.set_resident(True)

    def batch(n, frames, timed=True):
        ev = _random_batch(rng, n, frames, timed)
        if case == "bank":                                       # program changes among the events (kind 2: note = the program)
            pc = rng.rand(ev.size) < 0.1
            ev["kind"][pc] = 2; ev["note"][pc] = rng.randint(0, 3, int(pc.sum()))
        return ev

    batches = [batch(2500, 1024, timed=False)] + [batch(int(rng.randint(0, 700)), 1024) for _ in range(7)]
    _drive(gpu, ora, batches, lambda k: 1024, "pool-resident general kernel (%s), ring" % case)
    assert gpu.resident_active
    lens = [1024, 1000, 16, 512]
    batches = [batch(int(rng.randint(0, 500)), lens[k]) for k in range(4)]
    _drive(gpu, ora, batches, lambda k: lens[k], "pool-resident general kernel (%s), s2r_fill" % case, ring=False)
    assert gpu.resident_active


def test_pool_resident_kernel_leaves_when_idle_and_is_started_again():
    """the kernel's patience is 2 ms: pauses shorter and longer than that between fills, with one fill in flight across the
    pause or none — every buffer still the oracle's"""
    rng = np.random.RandomState(5)
    voices = 1024
    gpu = s2.Synth(voices, max_frames=256)
    gpu.set_resident(True)
    ora = s2o.OracleSynth(voices)
    batches = [_random_batch(rng, 1500, 256, timed=False)] + [_random_batch(rng, 200, 256) for _ in range(12)]
    _drive(gpu, ora, batches, lambda k: 256, "idle exits, ring", pause={2: 0.0005, 4: 0.0021, 6: 0.004, 8: 0.02})
    for k in range(6):
        ev = _random_batch(rng, 100, 256)
        gpu.note_events(ev)
        got = gpu.sample(np.empty(256, dtype=np.float32), SR).copy()
        want = s2o.mix_tree(ora.render_events(ev, 256, SR, threads=_threads()), gpu.block_voices, 1)
        assert_bits_equal(got, want, "idle exits, s2r_fill %d" % k)
        time.sleep([0.0, 0.0019, 0.0022, 0.003, 0.01, 0.0][k])


def test_bench_c3_path_through_the_pool_resident_kernel():
    """what bench.py times when the resident kernel is on: bench.make_c3_events(65536) through s2r_note_events + s2r_fill_begin /
    s2r_fill_end, the pool filled first, then 20 buffers of the schedule — every buffer against the oracle"""
    import bench
    V = 65536
    cyc = bench.make_c3_events(V, bench.PERIOD)
    gpu = s2.Synth(V, max_frames=bench.FRAMES)
    gpu.load_patch("synth mySynth {\n\n}\n")
    gpu.set_resident(True)
    ora = s2o.OracleSynth(V)
    on = np.zeros(V, dtype=s2.NOTE_EVENT_DTYPE)
    on["kind"] = 1; on["note"] = 36 + np.arange(V) % 61; on["velocity"] = 1.0
    _drive(gpu, ora, [on] + [cyc[k % bench.PERIOD] for k in range(20)], lambda k: bench.FRAMES, "C3 through the pool-resident kernel")
    assert gpu.resident_active


@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("n,interleave", [(2, 0), (4, 64)])
def test_device_list_exchange_in_the_kernels(n, interleave, resident):
    """a device list whose shards each take ONE launch per fill (or none: resident) and exchange their rows through a counter in
    the parent's memory, the first shard adding them up: equal to the rank-ordered sum of the shards' trees over the oracle's
    rows, ring fills and synchronous ones"""
    rng = np.random.RandomState(77 + n)
    V = 4096
    multi = s2.Synth(V, max_frames=1024, devices=[0] * n, shard_interleave=interleave)
    if resident:
        multi.set_resident(True)
    ora = s2o.OracleSynth(V)

    def want_of(pv):
        acc = np.zeros(pv.shape[1], dtype=np.float32)
        for k in range(n):
            idx = s2.synth.shard_pool_indices(V, k, n, interleave) if interleave else np.arange(k * V // n, (k + 1) * V // n)
            acc = acc + s2o.mix_tree_partial(pv[idx], multi.block_voices)
        return acc

    queue = []
    batches = [_random_batch(rng, 5000, 1024, timed=False)] + [_random_batch(rng, int(rng.randint(0, 1200)), 1024) for _ in range(7)]
    for k, ev in enumerate(batches):
        multi.note_events(ev)
        multi.sample_begin(1024, SR)
        queue.append((k, want_of(ora.render_events(ev, 1024, SR, threads=_threads()))))
        if len(queue) == 2:
            kk, w = queue.pop(0)
            assert_bits_equal(multi.sample_end(np.empty(1024, dtype=np.float32)), w, "device list of %d, ring buffer %d" % (n, kk))
    kk, w = queue.pop(0)
    assert_bits_equal(multi.sample_end(np.empty(1024, dtype=np.float32)), w, "device list of %d, ring buffer %d" % (n, kk))
    for k in range(3):
        ev = _random_batch(rng, 400, 1000)
        multi.note_events(ev)
        got = multi.sample(np.empty(1000, dtype=np.float32), SR).copy()
        assert_bits_equal(got, want_of(ora.render_events(ev, 1000, SR, threads=_threads())), "device list of %d, s2r_fill %d" % (n, k))
    assert multi.resident_active == resident


def test_a_voice_restarted_among_other_programs_in_the_pool_resident_bank_kernel():
    """Round 4's stray result, kept as found (delta-debugged from the test above): one note held, then 26 note-ons inside the next
    buffer — programs 0, 2 (DPW + SVF: the wave leaves the branch-free chunk) and, last, 1 (sine + LP2 + noise, started at frame
    976).  A build with an unrelated block added to fused_tail rendered that last voice 7e-4 off in the pool-resident kernel only
    (synth2_amd/build.py, -ftrivial-auto-var-init); its 48 frames go through the frame loop with fresh LP2 coefficients per frame."""
    voices = 2048
    bank = [make_patch(osc_kind=s2.OSC_SAW), make_patch(osc_kind=s2.OSC_SINE, lpf_kind=s2.FILT_LP2, lpf_freq=900.0, noise=0.2),
            make_patch(osc_kind=s2.OSC_DPW_SAW, lpf_kind=s2.FILT_SVF_LP, lpf_q=1.3)]

    def evs(lst):
        ev = np.zeros(len(lst), dtype=s2.NOTE_EVENT_DTYPE)
        for j, (kind, note, frame) in enumerate(lst):
            ev["kind"][j] = kind; ev["note"][j] = note; ev["frame"][j] = frame; ev["velocity"][j] = 1.0
        return ev

    b0 = evs([(1, 89, 0)])
    b1 = evs([(1, 36, 816), (1, 45, 816), (1, 87, 816), (1, 81, 832), (1, 60, 832), (1, 47, 832), (1, 50, 848), (1, 76, 848), (1, 78, 848),
              (1, 88, 848), (1, 74, 848), (1, 56, 864), (1, 95, 864), (1, 84, 864), (1, 70, 880), (1, 44, 880), (1, 64, 880), (1, 53, 880),
              (2, 2, 912), (1, 52, 912), (1, 42, 944), (1, 96, 944), (1, 71, 960), (1, 51, 960), (1, 70, 960), (1, 87, 960), (2, 1, 960), (1, 46, 976)])
    for resident in (False, True):
        gpu = s2.Synth(voices, max_frames=1024)
        ora = s2o.OracleSynth(voices)
        gpu.set_patch_bank(bank)
        ora.set_bank([oracle_cfg_from_patch(q) for q in bank])
        gpu.set_resident(resident)
        for k, ev in enumerate((b0, b1)):
            gpu.note_events(ev)
            got = gpu.sample(1024, SR).copy()
            pv = ora.render_events(ev, 1024, SR, threads=_threads())
            assert_bits_equal(got, s2o.mix_tree(pv, gpu.block_voices, 1), "resident %d, buffer %d" % (resident, k))
        st = gpu.export_state()
        for i in range(27):
            v = ora.voice(i)
            for name, a, b in (("phase", st["phase_accum"][i], v.state.phase_accum), ("x1", st["filt_x1"][i], v.state.x1), ("x2", st["filt_x2"][i], v.state.x2),
                               ("y1", st["filt_y1"][i], v.state.y1), ("y2", st["filt_y2"][i], v.state.y2), ("lpf_last", st["lpf_last"][i], v.state.lpf_last)):
                assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32), "resident %d, voice %d: %s" % (resident, i, name)
        gpu.close()
