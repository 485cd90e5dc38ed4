"""GPU parity of the resident render kernel (s2r_set_low_latency): the reference's own call pattern —
Synth::sample per 16 frames with MIDI applied in between (s2_bin/src/main.rs:138-147, synth.rs:154-203) — rendered by a
kernel that stays on the device between calls, bit for bit against the CPU oracle; and every way in and out of it."""
import time

import numpy as np
import pytest

from helpers import Pair, assert_bits_equal, make_patch
import synth2_amd as s2

pytestmark = pytest.mark.gpu


def _lowlat_pair(voices=8, **kw):
    pr = Pair(voices, **kw)
    pr.gpu.set_low_latency(True)
    return pr


def test_reference_call_pattern_through_the_resident_kernel():
    """test_reference_call_pattern_16_frame_fills with the resident kernel: 400 fills of 16 frames, note-ons and note-offs in
    between; the kernel is started by the first fill and takes every one after it"""
    pr = _lowlat_pair(8)
    rng = np.random.RandomState(7)
    held = []
    assert not pr.gpu.low_latency_active
    for k in range(400):
        r = rng.randint(0, 20)
        if r == 0:
            n = int(rng.randint(40, 90))
            pr.note_on(n)
            held.append(n)
        elif r == 1 and held:
            pr.note_off(held.pop(rng.randint(len(held))))
        g, o, _ = pr.sample(16)
        assert_bits_equal(g, o, "16-frame fill %d" % k)
        assert pr.gpu.low_latency_active
    # the same handle with the mode switched off again: launches per fill, same bits
    pr.gpu.set_low_latency(False)
    assert not pr.gpu.low_latency_active
    for k in range(8):
        g, o, _ = pr.sample(16)
        assert_bits_equal(g, o, "after switching off, fill %d" % k)
    assert not pr.gpu.low_latency_active


@pytest.mark.parametrize("osc", [s2.OSC_SQUARE, s2.OSC_SAW, s2.OSC_TRIANGLE, s2.OSC_SINE])
def test_resident_kernel_every_oscillator_and_fill_length(osc):
    """all four oscillators (one resident kernel each), with and without mod-to-pitch and noise; fills of 16 frames, of a
    length with a scalar tail, of the handle's longest, of one frame; 250 voices in four waves"""
    for fm in (0.0, 0.7):
        patch = make_patch(osc_kind=osc, mod_env_to_osc_freq=fm, noise=0.25 if fm else 0.0, lpf_freq=1500.0)
        patch.amp_env.attack_ms = 3.0; patch.amp_env.decay_ms = 11.0; patch.mod_env.decay_ms = 17.0; patch.amp_env.release_ms = 9.0
        pr = _lowlat_pair(250, patch=patch, max_frames=1024)
        for i in range(250):
            pr.note_on(30 + i % 70)
            if i % 9 == 8:                                       # at most nine events ride in one command
                g, o, _ = pr.sample(16)
                assert_bits_equal(g, o, "osc %d fm %g, while the pool fills, voice %d" % (osc, fm, i))
        for k, frames in enumerate([16, 100, 1024, 1, 16, 333, 16]):
            if k == 3:
                pr.note_off(30 + 5); pr.note_off(30 + 6)
            g, o, _ = pr.sample(frames)
            assert_bits_equal(g, o, "osc %d fm %g, %d frames" % (osc, fm, frames))
            assert pr.gpu.low_latency_active


def test_resident_kernel_gives_way_and_comes_back():
    """every other entry point stops the resident kernel (it holds the stream and arguments built from the patch) and the next
    eligible fill starts it again; fills it cannot take go the ordinary way — all with the oracle's bits"""
    pr = _lowlat_pair(8)
    for n in (60, 64, 67):
        pr.note_on(n)
    g, o, _ = pr.sample(16); assert_bits_equal(g, o, "first")
    assert pr.gpu.low_latency_active
    # checkpoint: stops it
    st = pr.gpu.export_state()
    assert not pr.gpu.low_latency_active and st["started"].sum() == 3
    g, o, _ = pr.sample(16); assert_bits_equal(g, o, "after export_state")
    assert pr.gpu.low_latency_active
    # a new patch: stops it, the next fill runs with the new patch's tables
    patch = make_patch(lpf_freq=700.0, mod_env_to_lpf_freq=3.0)
    pr.gpu.set_patch(patch)
    from helpers import oracle_cfg_from_patch
    pr.cpu.config = oracle_cfg_from_patch(patch)
    assert not pr.gpu.low_latency_active
    for k in range(3):
        g, o, _ = pr.sample(16); assert_bits_equal(g, o, "new patch, fill %d" % k)
    assert pr.gpu.low_latency_active
    # more note-ons than voices: the events fold to one record per voice (eight) and still ride in one command
    for n in range(40, 52):
        pr.note_on(n)
    g, o, _ = pr.sample(16); assert_bits_equal(g, o, "12 note-ons into 8 voices")
    assert pr.gpu.low_latency_active
    # more records than a command holds (nine), in a bigger pool: that fill is a launch, the next one resident again
    wide = _lowlat_pair(32)
    wide.note_on(50)
    g, o, _ = wide.sample(16); assert_bits_equal(g, o, "32 voices, first")
    assert wide.gpu.low_latency_active
    for n in range(60, 72):
        wide.note_on(n)
    g, o, _ = wide.sample(16); assert_bits_equal(g, o, "32 voices, 12 events")
    assert not wide.gpu.low_latency_active
    g, o, _ = wide.sample(16); assert_bits_equal(g, o, "32 voices, after 12 events")
    assert wide.gpu.low_latency_active
    # per-voice rows (mix disabled) in between
    gv, ov = pr.render_voices(16)
    assert_bits_equal(gv, ov, "render_voices")
    assert not pr.gpu.low_latency_active
    # a timed event: the ordinary way (the oracle applies it at its 16-frame boundary)
    ev = np.zeros(1, dtype=s2.NOTE_EVENT_DTYPE)
    ev["kind"] = 1; ev["note"] = 72; ev["velocity"] = 1.0; ev["frame"] = 16
    pr.gpu.note_events(ev)
    want = np.empty(32, dtype=np.float32)
    want[:16] = pr_cpu_sample(pr, 16)[1]
    pr.cpu.note_on(72)
    want[16:] = pr_cpu_sample(pr, 16)[1]
    g = pr.gpu.sample(np.empty(32, dtype=np.float32))
    assert_bits_equal(g, want, "timed event")
    assert not pr.gpu.low_latency_active
    # the two-buffer interface and the resident kernel alternate
    pr.gpu.sample_begin(16)
    _, o3 = pr_cpu_sample(pr, 16)
    assert_bits_equal(pr.gpu.sample_end(np.empty(16, dtype=np.float32)), o3, "fill_begin / fill_end")
    g, o, _ = pr.sample(16); assert_bits_equal(g, o, "after fill_begin / fill_end")
    assert pr.gpu.low_latency_active
    # stereo: its own resident kernel (the output layout is a kernel argument)
    gs = pr.gpu.sample_stereo(16)
    _, o4 = pr_cpu_sample(pr, 16)
    assert_bits_equal(gs[:, 0], o4, "stereo, left"); assert_bits_equal(gs[:, 1], o4, "stereo, right")
    assert pr.gpu.low_latency_active
    g, o, _ = pr.sample(16); assert_bits_equal(g, o, "mono after stereo")
    # another sample rate: restarted with that rate's tables
    g = pr.gpu.sample(np.empty(16, dtype=np.float32), 44100)
    pv = pr.cpu.render_voices(16, 44100)
    from oracle import s2o
    assert_bits_equal(g, s2o.mix_tree(pv, pr.block_voices, 1), "44.1 kHz")
    assert pr.gpu.low_latency_active


def pr_cpu_sample(pr, frames, sr=48000):
    from oracle import s2o
    pv = pr.cpu.render_voices(frames, sr)
    return pv, s2o.mix_tree(pv, pr.block_voices, 1)


def test_resident_kernel_leaves_when_idle_and_is_started_again():
    """no fill for longer than its idle limit (1 ms): the kernel leaves by itself; the next fill finds it gone and starts
    another — also when the command was posted just as it left"""
    pr = _lowlat_pair(8)
    pr.note_on(57); pr.note_on(64)
    for pause in (0.0, 0.004, 0.0, 0.0009, 0.00101, 0.0011, 0.00095, 0.02, 0.0):
        if pause:
            t = time.perf_counter()
            while time.perf_counter() - t < pause:
                pass
        g, o, _ = pr.sample(16)
        assert_bits_equal(g, o, "after %.2f ms without a fill" % (pause * 1e3))
    # a sweep of pauses around the limit: whichever side of the race a command lands on, the fill is rendered once
    for k in range(60):
        pause = 0.0009 + 0.000005 * k
        t = time.perf_counter()
        while time.perf_counter() - t < pause:
            pass
        g, o, _ = pr.sample(16)
        assert_bits_equal(g, o, "pause %.4f ms" % (pause * 1e3))


def test_resident_kernel_refused_or_ignored_where_it_does_not_apply():
    """a device list refuses the mode; a pool of more than one workgroup, a patch bank, another filter kind accept it and
    render the ordinary way"""
    with pytest.raises(s2.S2rError):
        s2.Synth(512, devices=[0, 0]).set_low_latency(True)
    big = _lowlat_pair(512)
    big.note_on(60)
    g, o, _ = big.sample(16); assert_bits_equal(g, o, "two workgroups")
    assert not big.gpu.low_latency_active
    svf = _lowlat_pair(8, patch=make_patch(lpf_kind=s2.FILT_SVF_LP, lpf_q=1.2))
    svf.note_on(60)
    g, o, _ = svf.sample(16); assert_bits_equal(g, o, "state-variable filter")
    assert not svf.gpu.low_latency_active
    bank = _lowlat_pair(8)
    bank.set_bank([make_patch(), make_patch(osc_kind=s2.OSC_SINE)])
    bank.program_change(1); bank.note_on(60)
    g, o, _ = bank.sample(16); assert_bits_equal(g, o, "patch bank")
    assert not bank.gpu.low_latency_active
    seeded = _lowlat_pair(8, patch=make_patch(noise=0.5), seeds=[3, 1, 4, 1, 5, 9, 2, 6])
    seeded.note_on(60)
    g, o, _ = seeded.sample(16); assert_bits_equal(g, o, "seed overrides ride with the note-on: a launch")
    g, o, _ = seeded.sample(16); assert_bits_equal(g, o, "no events: resident")
    assert seeded.gpu.low_latency_active


@pytest.mark.parametrize("block", [0, 64, 128])
def test_resident_kernel_granules_and_completion_word_side_by_side(block):
    """fills of up to 64 frames come back as tagged 8-byte granules, longer ones through the completion word: lengths on both
    sides of the limit in turn, mono and stereo, in workgroups of 64, 128 and 256 voices"""
    pr = _lowlat_pair(60, block_voices=block, max_frames=256)
    for n in (50, 57, 62, 69, 74):
        pr.note_on(n)
    for k, frames in enumerate([64, 65, 63, 1, 256, 16, 64, 128, 2, 64, 65, 16]):
        if k == 5:
            pr.note_off(57); pr.note_on(81)
        if k % 3 == 2:
            gs = pr.gpu.sample_stereo(frames)
            _, o = pr_cpu_sample(pr, frames)
            assert_bits_equal(gs[:, 0], o, "block %d, %d frames, stereo left" % (block, frames))
            assert_bits_equal(gs[:, 1], o, "block %d, %d frames, stereo right" % (block, frames))
        else:
            g, o, _ = pr.sample(frames)
            assert_bits_equal(g, o, "block %d, %d frames" % (block, frames))
        assert pr.gpu.low_latency_active
