"""GPU parity: the HIP path, called through the C ABI (synth2_amd.Synth -> libs2r.so), against
the CPU oracle on the same seeded note streams.  The bar is BIT-EXACT float32 (tolerance 0
ULP) per voice, and bit-exact for the mix when the oracle sums through the documented tree;
the deviation from the reference's sequential voice order is reported, not asserted to be 0.
north_star tolerance: within 1 ULP fp32 of the CPU s2_lib path.
"""
import numpy as np
import pytest

from helpers import Pair, assert_bits_equal, lcg, make_patch, ulp_diff
from oracle import s2o
import synth2_amd as s2

pytestmark = pytest.mark.gpu

SR = 48000


def test_c1_one_voice_default_patch():
    """BASELINE config 0: example.synth2 (empty body == default patch), 1 voice, 48 kHz,
    1024-frame buffers; note 69 on at frame 0, off at frame 24000."""
    pr = Pair(8)
    pr.gpu.load_patch("synth mySynth {\n\n}\n")
    pr.note_on(69)
    frames_done = 0
    released = False
    for _ in range(40):
        if not released and frames_done + 1024 > 24000:
            # fills of 1024 cannot land exactly on 24000: split the buffer like s2_bin would
            n = 24000 - frames_done
            g, o, pv = pr.sample(n)
            assert_bits_equal(g, o, "pre-release partial buffer")
            pr.note_off(69)
            released = True
            g, o, pv = pr.sample(1024 - n)
            assert_bits_equal(g, o, "post-release partial buffer")
            frames_done += 1024
            continue
        g, o, pv = pr.sample(1024)
        assert_bits_equal(g, o, "buffer at frame %d" % frames_done)
        # with one voice the tree and the reference's sequential order coincide
        assert_bits_equal(o, s2o.mix_sequential(pv), "tree vs sequential, 1 voice")
        frames_done += 1024
    assert np.any(g == 0.0)      # envelope has ended


@pytest.mark.parametrize("osc", [s2.OSC_SQUARE, s2.OSC_SAW, s2.OSC_TRIANGLE, s2.OSC_SINE])
@pytest.mark.parametrize("fm", [0.0, 3.5])
def test_per_voice_all_oscillators(osc, fm):
    """every oscillator kind, with and without oscillator FM (mod_env_to_osc_freq), mix off"""
    patch = make_patch(osc_kind=osc, mod_env_to_osc_freq=fm, noise=0.25, osc_gain=0.75)
    patch.mod_env.attack_ms = 5.0
    patch.mod_env.sustain = 0.3
    patch.mod_env.release_ms = 40.0
    pr = Pair(64, patch)
    for v in range(40):
        pr.note_on(30 + (v * 7) % 70)
    for k in range(6):
        g, o = pr.render_voices(512)
        assert_bits_equal(g, o, "osc %d fm %g block %d" % (osc, fm, k))
        if k == 2:
            for v in range(0, 40, 3):
                pr.note_off(30 + (v * 7) % 70)


def test_c2_1024_voices_one_workgroup():
    """BASELINE config 1: 1024 voices, default patch, one 1024-thread workgroup"""
    _run_c2(block_voices=1024)


def test_c2_1024_voices_blocks_of_256():
    _run_c2(block_voices=256)


@pytest.mark.parametrize("stream", [3, 4, 1, 0])
def test_coefficient_stream_is_bit_neutral(stream):
    """64-voice groups with a moving mod envelope get their LPF coefficients from the ahead-of-time
    pass; with it off they are computed in-lane.  Mixed population: some groups fully flat, some
    partly moving, some restarted mid-run, ragged fills (tail frames bypass the stream)."""
    V = 1024
    pr = Pair(V, max_frames=1024)
    pr.gpu.set_coeff_stream(stream)      # 3: events in the classification launch where possible, 4: separate kernels, 1: default policy, 0: in-lane
    for v in range(V):
        pr.note_on(36 + v % 61)          # 1024 events: more than one launch carries -> separate kernels
    for b, n in enumerate([1024, 1024, 512, 1000, 1024, 16, 1024, 1024, 1024, 1024, 1024, 1024]):
        if b in (3, 5, 8, 10):
            for k in range(40):              # restart the 40 oldest voices: their groups start moving again
                pr.note_on(40 + (k + b) % 50)
        if b == 6:
            for note in range(36, 97, 3):
                pr.note_off(note)
        g, o, _pv = pr.sample(n)
        assert_bits_equal(g, o, "stream=%s buffer %d" % (stream, b))


@pytest.mark.parametrize("mode", [3, 4])
def test_coefficient_stream_every_group_moving(mode):
    """every group moving at once (the stream has a slot per group), events arriving in batches on
    both sides of what one preparation launch carries (288), releases folded onto restarts"""
    V = 512
    pr = Pair(V)
    pr.gpu.set_coeff_stream(mode)
    rng = np.random.RandomState(mode)
    for b in range(6):
        n_on = [288, 100, 289, 0, 17, 300][b]
        for k in range(n_on):
            pr.note_on(36 + int(rng.randint(0, 61)))
        for k in range(n_on // 5):
            pr.note_off(36 + int(rng.randint(0, 61)))     # some land on voices restarted in the same batch
        g, o, _pv = pr.sample(1024 if b != 3 else 1000)
        assert_bits_equal(g, o, "mode %d buffer %d" % (mode, b))
    st = pr.gpu.export_state()
    for v in range(V):
        cv = pr.cpu.voice(v)
        assert bool(st["started"][v]) == bool(cv.has_current) and bool(st["released"][v]) == bool(cv.has_release)
        if cv.has_current:
            assert st["current_frame_offset"][v] == cv.current_frame_offset


@pytest.mark.parametrize("fm", [0.0, -2.0])
@pytest.mark.parametrize("flat", [True, False])
def test_flat_envelope_shortcut_is_bit_neutral(flat, fm):
    """the LPF-coefficient reuse while the mod envelope is flat (sustain / end stage) must not
    change a bit; exercised with voices entering and leaving flat stages at different times,
    a non-zero sustain, and FM so the cached oscillator constants are used too"""
    patch = make_patch(mod_env_to_osc_freq=fm)
    patch.mod_env.attack_ms = 2.0
    patch.mod_env.decay_ms = 10.0
    patch.mod_env.sustain = 0.25
    patch.mod_env.release_ms = 15.0
    pr = Pair(256, patch, flat_shortcut=flat)
    for b in range(12):
        for v in range(16):
            pr.note_on(36 + (b * 16 + v) % 61)
        if b >= 3:
            for n in range(36, 97, 5):
                pr.note_off(n + b % 5)
        g, o, _pv = pr.sample(512)
        assert_bits_equal(g, o, "flat=%s fm=%g buffer %d" % (flat, fm, b))


def test_sample_rate_without_fast_division():
    """a rate outside the verified whitelist takes the true-division kernel variant"""
    pr = Pair(64)
    for v in range(20):
        pr.note_on(40 + v)
    for _ in range(3):
        g, o, _pv = pr.sample(512, 37123)
        assert_bits_equal(g, o, "sr 37123")
        g, o, _pv = pr.sample(512, 44100)
        assert_bits_equal(g, o, "sr 44100")


def _run_c2(block_voices):
    V = 1024
    pr = Pair(V, block_voices=block_voices)
    for v in range(V):
        pr.note_on(36 + (v % 61))
    # per-note release countdown derived from the LCG so every ADSR stage is live
    worst = 0
    for b in range(24):
        if b >= 2:
            for note in range(36, 97):
                if lcg(note * 131 + b) % 5 == 0:
                    pr.note_off(note)
        g, o, pv = pr.sample(1024)
        assert_bits_equal(g, o, "C2 buffer %d (block_voices %d)" % (b, block_voices))
        worst = max(worst, ulp_diff(o, s2o.mix_sequential(pv)))
    print("C2 tree-vs-sequential mix deviation: %d ULP max" % worst)


@pytest.mark.parametrize("frames", [1, 7, 15, 17, 100, 1000, 1023])
def test_tail_frames_use_scalar_path(frames):
    """frames % 16 != 0: the tail goes through the scalar path with its different semantics
    (multiplicative gains, release from the current level, libm powf) — process.rs:39-48"""
    patch = make_patch(noise=0.5, osc_gain=0.5, mod_env_to_osc_freq=1.25)
    pr = Pair(16, patch)
    for v in range(10):
        pr.note_on(40 + 3 * v)
    for k in range(5):
        g, o = pr.render_voices(frames)
        assert_bits_equal(g, o, "tail %d call %d" % (frames, k))
        if k == 1:
            pr.note_off(43)
            pr.note_off(61)


def test_reference_call_pattern_16_frame_fills():
    """s2_bin applies MIDI between 16-frame sample() calls (main.rs:138-143)"""
    pr = Pair(8)
    rng = np.random.RandomState(7)
    held = []
    for k in range(400):
        r = rng.randint(0, 20)
        if r == 0:
            n = int(rng.randint(40, 90))
            pr.note_on(n)
            held.append(n)
        elif r == 1 and held:
            pr.note_off(held.pop(rng.randint(len(held))))
        g, o, pv = pr.sample(16)
        assert_bits_equal(g, o, "16-frame fill %d" % k)


def test_voice_stealing_and_retrigger():
    pr = Pair(8)
    for n in range(60, 72):          # 12 note-ons into 8 voices: steals the oldest
        pr.note_on(n)
        g, o, _ = pr.sample(64)
        assert_bits_equal(g, o)
    pr.note_on(60); pr.note_on(60)   # same note twice, then off releases the LAST active match
    pr.note_off(60)
    g, o, _ = pr.sample(256)
    assert_bits_equal(g, o)
    pr.note_off(60)
    pr.note_off(60)                  # nothing active any more: no-op
    g, o, _ = pr.sample(256)
    assert_bits_equal(g, o)


def test_note_on_and_off_between_same_fill():
    pr = Pair(8)
    pr.note_on(64); pr.note_off(64)  # release offset 0
    pr.note_on(70)
    for _ in range(4):
        g, o, _ = pr.sample(512)
        assert_bits_equal(g, o)


def test_seeded_noise_variant():
    """NoiseState.seed = v so voices are decorrelated (SURVEY §8d variant)"""
    V = 256
    seeds = ((np.arange(V, dtype=np.uint64) * 2654435761) % (1 << 32)).astype(np.uint32)
    pr = Pair(V, make_patch(noise=1.0), seeds=seeds)
    for v in range(V):
        pr.note_on(36 + v % 61)
    for _ in range(3):
        g, o = pr.render_voices(256)
        assert_bits_equal(g, o, "seeded noise")


def test_mix_groups_reproduce_multi_gpu_order():
    """1 GPU with mix_groups=2 == the order a 2-GPU run produces; and two shard handles,
    partial mixes combined in rank order, give the same bits."""
    V = 1024
    pr = Pair(V, mix_groups=2)
    sh = [s2.Synth(V, shard_begin=r * 512, shard_voices=512) for r in range(2)]
    for v in range(V):
        pr.note_on(36 + (v % 61))
        for s in sh:
            s.note_on(36 + (v % 61))
    import torch
    for b in range(3):
        g, o, pv = pr.sample(1024)
        assert_bits_equal(g, o, "groups=2 vs oracle tree")
        rows = torch.empty((2, 1024), dtype=torch.float32, device="cuda")
        out = torch.empty(1024, dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for r, s in enumerate(sh):
            s.fill_device(rows[r].data_ptr(), 1024, SR, st)
        s2.sum_partials_device(rows.data_ptr(), 2, 1024, out.data_ptr(), st)
        torch.cuda.synchronize()
        assert_bits_equal(out.cpu().numpy(), g, "2 shards combined vs 1 GPU with mix_groups=2")
        for r in range(2):
            assert_bits_equal(rows[r].cpu().numpy(), s2o.mix_tree_partial(pv[r * 512:(r + 1) * 512], 256), "shard %d partial" % r)


def test_stereo_is_mono_on_both_channels():
    pr = Pair(8)
    pr.note_on(57)
    lr = pr.gpu.sample_stereo(512)
    o = s2o.mix_tree(pr.cpu.render_voices(512), pr.block_voices, 1)
    assert_bits_equal(lr[:, 0], o)
    assert_bits_equal(lr[:, 1], o)


def test_export_import_roundtrip_and_large_offsets():
    """checkpoint/resume; offsets past 2^24 where u32->f32 rounds (SURVEY §7 hard parts)"""
    pr = Pair(8)
    for n in (50, 62, 74):
        pr.note_on(n)
    g, o, _ = pr.sample(1024)
    st = pr.gpu.export_state()
    # push voices far out: the oracle gets the same edit
    for v in range(3):
        st["current_frame_offset"][v] = (1 << 24) + 12345 * (v + 1)
        pr.cpu.voice(v).current_frame_offset = int(st["current_frame_offset"][v])
    st["released"][1] = 1
    st["release_frame_offset"][1] = (1 << 24) + 20000
    pr.cpu.voice(1).has_release = 1
    pr.cpu.voice(1).release_frame_offset = (1 << 24) + 20000
    pr.gpu.import_state(st)
    for _ in range(3):
        g, o, _ = pr.sample(1024)
        assert_bits_equal(g, o, "after import at 2^24+")
    st2 = pr.gpu.export_state()
    for v in range(3):
        assert st2["current_frame_offset"][v] == pr.cpu.voice(v).current_frame_offset
        assert np.float32(st2["phase_accum"][v]).view(np.uint32) == np.float32(pr.cpu.voice(v).state.phase_accum).view(np.uint32)
        assert np.float32(st2["lpf_last"][v]).view(np.uint32) == np.float32(pr.cpu.voice(v).state.lpf_last).view(np.uint32)


def test_noise_division_all_u16_values():
    """hash_noise's two-operation exact quotient v/65535 on the device for all 65536 values: 65536 consecutive
    offsets hit every low-16 pattern of h (the multiplier is odd); compare a long noise-only render, through the
    general arithmetic (noise level 1) and through the small-offset chunk (noise level 0, which still ADDS the
    noise value, process.rs:353-356)."""
    for level in (1.0, 0.0):
        patch = make_patch(noise=level, osc_gain=0.0)
        patch.amp_env.attack_ms = 0.0
        pr = Pair(64, patch)
        pr.note_on(69)
        for _ in range(33):
            g, o = pr.render_voices(2048)
            assert_bits_equal(g[0], o[0], "noise quotient, level %g" % level)


def test_offsets_across_2_pow_24():
    """The branch-free chunk takes exact f32 offsets and a 16-bit noise hash while every offset of a wave's run is
    below 2^24, and the general arithmetic (hashnoise.rs:37: the offset goes through f32, which rounds there)
    above: waves below, across and above 2^24 in one pool, with the reference's seed 0 and with per-voice seeds."""
    V = 192
    seeds = ((np.arange(V, dtype=np.uint64) * 2654435761 + 12345) & 0xffffffff).astype(np.uint32)
    for noise, sd in ((0.0, None), (0.0, seeds), (0.25, seeds)):
        pr = Pair(V, make_patch(noise=noise), seeds=sd)
        for v in range(V):
            pr.note_on(36 + v % 61)
        g, o, _ = pr.sample(1024)
        assert_bits_equal(g, o, "before the edit")
        st = pr.gpu.export_state()
        for v in range(V):
            if v < 64:
                off = 20000 + 977 * v                                   # a wave that stays small
            elif v < 128:
                off = (1 << 24) - 3000 + 61 * (v - 64) if v % 3 else 50000 + v   # lanes cross 2^24 inside the fills
            else:
                off = (1 << 24) + 100 + 4001 * (v - 128)                # a wave beyond it
            st["current_frame_offset"][v] = off
            pr.cpu.voice(v).current_frame_offset = off
        pr.gpu.import_state(st)
        for frames in (1024, 2048, 1000):
            g, o, _ = pr.sample(frames)
            assert_bits_equal(g, o, "noise %g, %d frames across 2^24" % (noise, frames))
        gv, ov = pr.render_voices(512)
        assert_bits_equal(gv, ov, "per voice, noise %g" % noise)


def test_error_statuses():
    s = s2.Synth(8, max_frames=256)
    with pytest.raises(s2.S2rError) as e:
        s.sample(np.empty(257, dtype=np.float32))
    assert e.value.status == -6
    st = s.export_state()
    st["started"][0] = 1
    st["current_frame_offset"][0] = 0xFFFFFFF0
    s.import_state(st)
    with pytest.raises(s2.S2rError) as e:
        s.sample(np.empty(32, dtype=np.float32))
    assert e.value.status == -7          # the reference panics here (process.rs:36)
    with pytest.raises(s2.S2rError) as e:
        s.load_patch("synth x { osc.gain = 3 }")
    assert e.value.status == -5
    with pytest.raises(s2.S2rError) as e:
        s.load_patch("synth x { bogus = 3 }")
    assert e.value.status == -4
    assert s.sample(np.empty(0, dtype=np.float32)).size == 0      # empty buffer is fine


def _full_size(voices, buffers, churn):
    """BASELINE-size pools against the (multi-threaded) oracle, bit for bit, plus the
    size-independent properties: determinism across handles and shard/whole equality."""
    import os
    pr = Pair(voices, max_frames=1024)
    pr.threads = min(32, len(os.sched_getaffinity(0)))
    ev = np.zeros(voices, dtype=s2.NOTE_EVENT_DTYPE)
    ev["kind"] = 1
    ev["note"] = 36 + np.arange(voices) % 61
    ev["velocity"] = 1.0
    pr.note_events(ev)
    rng = np.random.RandomState(voices % 9973)
    for b in range(buffers):
        if b:
            ch = np.zeros(2 * churn, dtype=s2.NOTE_EVENT_DTYPE)
            ch["kind"][churn:] = 1
            ch["note"] = 36 + rng.randint(0, 61, 2 * churn)
            ch["velocity"] = 1.0
            pr.note_events(ch)
        g, o, pv = pr.sample(1024)
        assert_bits_equal(g, o, "%d voices, buffer %d" % (voices, b))
    seq = s2o.mix_sequential(pv)
    exact = pv.astype(np.float64).sum(axis=0).astype(np.float32)
    print("%d voices: tree vs the reference's sequential order %d ULP; vs the exactly rounded sum: tree %d ULP, sequential %d ULP"
          % (voices, ulp_diff(o, seq), ulp_diff(o, exact), ulp_diff(seq, exact)))
    _record_mix_deviation(voices, o, seq, exact)


def _record_mix_deviation(voices, tree, seq, exact):
    """north_star asks for "within 1 ULP" of the CPU path; a 64 k-term fp32 sum cannot be within 1 ULP of ANY other
    association of itself, so the achieved distance of the GPU's tree (== the oracle's tree, asserted bit for bit above)
    from the reference's sequential order is put on file: gpurun_out/mix_deviation.json, copied to profiles/r02/."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "mix_deviation.json")
    try:
        data = json.load(open(path))
    except Exception:
        data = {"what": "GPU mix (documented tree, bit-equal to the oracle's tree) vs the reference's sequential voice order "
                        "(synth.rs:177-195) and vs the correctly rounded sum of the voices; last buffer of the test, 1024 frames",
                "rows": {}}
    peak = float(np.max(np.abs(exact)))
    rel = np.abs(tree.astype(np.float64) - seq.astype(np.float64)) / max(peak, 1e-30)
    data["rows"][str(voices)] = {
        "voices": voices,
        "tree_vs_sequential_max_ulp": ulp_diff(tree, seq),
        "tree_vs_exact_max_ulp": ulp_diff(tree, exact),
        "sequential_vs_exact_max_ulp": ulp_diff(seq, exact),
        "tree_vs_sequential_max_abs_over_peak": float(rel.max()),
        "mix_peak": peak,
        "note": "ULP distances are per sample and large where the mix passes near zero; the last column relates the "
                "difference to the buffer's peak",
    }
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(data, open(path, "w"), indent=1)


def test_mix_deviation_1024_voices():
    """the same record for BASELINE config 1's pool size"""
    _full_size(1024, 2, 8)


def test_c3_full_size_65536_voices():
    """BASELINE config 2: 65 536 voices on one MI355X (the bench workload), 3 buffers with churn"""
    _full_size(65536, 3, 128)


def test_c4_shard_size_131072_voices():
    """BASELINE config 3's per-GPU share: 1 048 576 voices / 8 GPUs"""
    _full_size(131072, 2, 256)


@pytest.mark.parametrize("voices", [8, 300])
def test_timed_events_reproduce_the_16_frame_call_pattern(voices):
    """note events stamped with a frame offset take effect INSIDE one 1024-frame launch exactly as
    if the caller had called sample() 16 frames at a time with MIDI applied in between, which is
    what s2_bin does (main.rs:138-143): the oracle is driven that way, chunk by chunk."""
    pr = Pair(voices, max_frames=1024)
    rng = np.random.RandomState(voices * 7 + 1)
    held = []
    for b in range(8):
        frames = 1024 if b != 5 else 1000                 # one ragged buffer: tail frames + an event at the tail boundary
        n_ev = int(rng.randint(0, 40))
        times = np.sort(rng.randint(0, (frames + 15) // 16, n_ev)) * 16
        if b == 5 and n_ev:
            times[-1] = 992                               # the scalar tail starts here
        ev = np.zeros(n_ev, dtype=s2.NOTE_EVENT_DTYPE)
        for i, t in enumerate(times):
            on = (not held) or rng.randint(0, 3) > 0
            if on:
                note = int(rng.randint(40, 90)); held.append(note)
            else:
                note = held.pop(int(rng.randint(len(held))))
            ev[i] = (1 if on else 0, note, int(t), 1.0)
        pr.gpu.note_events(ev)
        g = pr.gpu.sample(np.empty(frames, dtype=np.float32))
        pv = np.zeros((voices, frames), dtype=np.float32)
        k = 0
        for c in range(0, frames, 16):
            while k < n_ev and ev["frame"][k] == c:
                if ev["kind"][k]:
                    pr.cpu.note_on(int(ev["note"][k]))
                else:
                    pr.cpu.note_off(int(ev["note"][k]))
                k += 1
            n = min(16, frames - c)
            pv[:, c:c + n] = pr.cpu.render_voices(n)
        assert k == n_ev
        o = s2o.mix_tree(pv, pr.block_voices, 1)
        assert_bits_equal(g, o, "timed events, %d voices, buffer %d" % (voices, b))
    st = pr.gpu.export_state()
    for v in range(voices):
        cv = pr.cpu.voice(v)
        assert bool(st["started"][v]) == bool(cv.has_current)
        if cv.has_current:
            assert st["current_frame_offset"][v] == cv.current_frame_offset
            assert bool(st["released"][v]) == bool(cv.has_release)
            if cv.has_release:
                assert st["release_frame_offset"][v] == cv.release_frame_offset


def test_timed_event_errors():
    s = s2.Synth(8, max_frames=256)
    ev = np.zeros(1, dtype=s2.NOTE_EVENT_DTYPE)
    ev[0] = (1, 60, 24, 1.0)                 # not a multiple of 16
    with pytest.raises(s2.S2rError):
        s.note_events(ev)
    ev[0] = (1, 60, 256, 1.0)                # no fill of this handle is long enough to contain frame 256
    with pytest.raises(s2.S2rError):
        s.note_events(ev)
    s.sample(np.empty(256, dtype=np.float32))        # ... and the handle is not stuck
    ev[0] = (1, 60, 128, 1.0)
    s.note_events(ev)
    with pytest.raises(s2.S2rError):         # a fill that ends before the queued event
        s.sample(np.empty(64, dtype=np.float32))
    with pytest.raises(s2.S2rError):         # untimed events cannot follow timed ones
        s.note_on(61)
    s.sample(np.empty(256, dtype=np.float32))        # the fill that contains it still works


def test_rejected_event_batch_leaves_the_handle_untouched():
    """a batch is validated as a whole before the first event is applied — an error in its last event
    must not leave the earlier ones half-applied (host pool and device state would diverge)."""
    pair = Pair(8, max_frames=256)
    pair.note_on(60); pair.note_on(64)
    g, want, _ = pair.sample(256)
    assert_bits_equal(g, want, "before")
    for bad_tail in [(7, 60, 0, 1.0),        # unknown kind
                     (1, 62, 40, 1.0),       # frame % 16
                     (1, 62, 16, 1.0),       # out of order (after the event at 32)
                     (0, 62, 4096, 0.0),     # not inside any fill
                     (2, 5, 0, 0.0)]:        # program past the bank
        ev = np.zeros(4, dtype=s2.NOTE_EVENT_DTYPE)
        ev[0] = (1, 67, 0, 1.0); ev[1] = (0, 60, 0, 0.0); ev[2] = (1, 72, 32, 1.0); ev[3] = bad_tail
        with pytest.raises(s2.S2rError):
            pair.gpu.note_events(ev)
        # nothing of the batch happened: the oracle, which never saw it, still agrees — allocation included
        assert pair.gpu.L.s2r_double_release_count(pair.gpu.h) == 0
        pair.note_on(65); pair.note_off(64)
        g, want, _ = pair.sample(256)
        assert_bits_equal(g, want, "after a rejected batch %r" % (bad_tail,))
        pair.note_on(64)
    pair.gpu.note_off(99)                    # nobody holds note 99: ignored (synth.rs:72-80), counted
    pair.cpu.note_off(99)
    assert pair.gpu.L.s2r_double_release_count(pair.gpu.h) == 1
    g, want, _ = pair.sample(256)
    assert_bits_equal(g, want, "after an unmatched note_off")


def test_more_timed_events_than_the_initial_buffers_hold():
    """the timed-event buffers start at max(4096, shard voices) records and grow on demand"""
    voices = 64
    pair = Pair(voices, max_frames=2048)
    rng = np.random.default_rng(3)
    n = 6000
    ev = np.zeros(n, dtype=s2.NOTE_EVENT_DTYPE)
    frames = np.sort(rng.integers(1, 128, n)) * 16
    ev["kind"] = rng.integers(0, 2, n); ev["note"] = rng.integers(50, 60, n); ev["frame"] = frames; ev["velocity"] = 1.0
    pair.gpu.note_events(ev)
    got = pair.gpu.sample(np.empty(2048, dtype=np.float32))
    pv = np.zeros((voices, 2048), dtype=np.float32)
    pos = 0
    k = 0
    while pos < 2048:                        # the oracle in the reference's 16-frame call pattern
        while k < n and ev["frame"][k] == pos:
            (pair.cpu.note_on if ev["kind"][k] == 1 else pair.cpu.note_off)(int(ev["note"][k]))
            k += 1
        pv[:, pos:pos + 16] = pair.cpu.render_voices(16, SR)
        pos += 16
    assert_bits_equal(got, s2o.mix_tree(pv, pair.block_voices, 1), "6000 timed events in one fill")


# ---------------------------------------------------------------------------------------------
# dsp_filters.rs filters as the layer's filter (lpf.kind != onepole)
# ---------------------------------------------------------------------------------------------
DSP_KINDS = [s2.FILT_LP1, s2.FILT_HP1, s2.FILT_LP2, s2.FILT_HP2, s2.FILT_BP2,
             s2.FILT_SVF_LP, s2.FILT_SVF_BP, s2.FILT_SVF_HP]


@pytest.mark.parametrize("kind", DSP_KINDS)
@pytest.mark.parametrize("osc,fm", [(s2.OSC_SAW, 0.0), (s2.OSC_SQUARE, 2.5), (s2.OSC_TRIANGLE, 0.0), (s2.OSC_SINE, -1.5)])
def test_dsp_filters_per_voice(kind, osc, fm):
    """every first/second-order filter of dsp_filters.rs:25-180 at the modulated cutoff, mix off"""
    patch = make_patch(osc_kind=osc, mod_env_to_osc_freq=fm, noise=0.25, osc_gain=0.75, lpf_kind=kind,
                       lpf_freq=900.0, mod_env_to_lpf_freq=3.0, lpf_damping=0.6, lpf_q=1.7)
    patch.mod_env.attack_ms = 5.0
    patch.mod_env.sustain = 0.3
    patch.mod_env.release_ms = 40.0
    pr = Pair(96, patch)
    for v in range(70):
        pr.note_on(30 + (v * 7) % 70)
    for k in range(6):
        g, o = pr.render_voices(512)
        assert_bits_equal(g, o, "filter %d osc %d fm %g block %d" % (kind, osc, fm, k))
        if k == 2:
            for v in range(0, 70, 3):
                pr.note_off(30 + (v * 7) % 70)


@pytest.mark.parametrize("kind", DSP_KINDS)
@pytest.mark.parametrize("frames", [1, 15, 17, 1000])
def test_dsp_filters_tail_frames_and_mix(kind, frames):
    """ragged fills (scalar tail runs the same filter at the libm-modulated cutoff) and the mix tree"""
    patch = make_patch(lpf_kind=kind, noise=0.5, osc_gain=0.5, mod_env_to_osc_freq=1.25, lpf_damping=1.41421354)
    pr = Pair(300, patch, block_voices=128)
    for v in range(200):
        pr.note_on(40 + (3 * v) % 60)
    for k in range(4):
        g, o, pv = pr.sample(frames)
        assert_bits_equal(g, o, "filter %d, %d frames, call %d" % (kind, frames, k))
        if k == 1:
            pr.note_off(43)
            pr.note_off(61)


def test_dsp_filter_cutoff_range_and_damping_extremes():
    """theta = 2 pi f / sr far past pi (cutoff above Nyquist: the reference does not clamp; theta
    reaches ~2700 here, the large-argument reduction of sinf/cosf) and the ends of Unipolar<10>
    damping; the host refuses only a cutoff whose 2 pi f overflows f32"""
    for damping in (0.0, 0.2, 10.0):
        patch = make_patch(lpf_kind=s2.FILT_LP2, lpf_freq=20000.0, mod_env_to_lpf_freq=10.0, lpf_damping=damping)
        patch.mod_env.decay_ms = 20.0
        pr = Pair(16, patch)
        for v in range(12):
            pr.note_on(36 + 5 * v)
        for k in range(3):
            with np.errstate(all="ignore"):
                g, o = pr.render_voices(256)
            assert_bits_equal(g, o, "damping %g block %d" % (damping, k))
    for q in (0.2, 3.0, 10.0):                       # band-pass: tan(theta / (2 q)) up to ~6800 rad
        patch = make_patch(lpf_kind=s2.FILT_BP2, lpf_freq=20000.0, mod_env_to_lpf_freq=10.0, lpf_q=q)
        patch.mod_env.decay_ms = 20.0
        pr = Pair(16, patch)
        for v in range(12):
            pr.note_on(36 + 5 * v)
        for k in range(3):
            with np.errstate(all="ignore"):
                g, o = pr.render_voices(256)
            assert_bits_equal(g, o, "q %g block %d" % (q, k))
    for kind in (s2.FILT_SVF_LP, s2.FILT_SVF_HP):   # SVF: tan(pi f / sr) far past Nyquist, extreme resonance
        for q in (0.05, 10.0):
            patch = make_patch(lpf_kind=kind, lpf_freq=20000.0, mod_env_to_lpf_freq=10.0, lpf_q=q)
            patch.mod_env.decay_ms = 20.0
            pr = Pair(16, patch)
            for v in range(12):
                pr.note_on(36 + 5 * v)
            for k in range(3):
                with np.errstate(all="ignore"):
                    g, o = pr.render_voices(256)
                assert_bits_equal(g, o, "svf %d q %g block %d" % (kind, q, k))
    s = s2.Synth(8, max_frames=256)
    s.set_patch(make_patch(lpf_kind=s2.FILT_SVF_BP, lpf_q=0.0))
    with pytest.raises(s2.S2rError) as e:
        s.sample(np.empty(256, dtype=np.float32), 48000)
    assert e.value.status == -5
    s.set_patch(make_patch(lpf_kind=s2.FILT_BP2, lpf_q=0.0))
    with pytest.raises(s2.S2rError) as e:
        s.sample(np.empty(256, dtype=np.float32), 48000)
    assert e.value.status == -5
    s.set_patch(make_patch(lpf_kind=s2.FILT_HP2, lpf_freq=3.0e37, mod_env_to_lpf_freq=10.0))
    with pytest.raises(s2.S2rError) as e:
        s.sample(np.empty(256, dtype=np.float32), 48000)
    assert e.value.status == -5


def test_dsp_filter_state_survives_patch_switches_and_checkpoints():
    """x1,x2,y1,y2 belong to the voice (st::Layer): they are zeroed by note_on only, are carried by
    export/import, and are left alone while the patch runs the one-pole"""
    lp2 = make_patch(lpf_kind=s2.FILT_LP2, lpf_freq=1500.0, lpf_damping=0.4)
    one = make_patch()
    pr = Pair(8, lp2)
    for n in (50, 57, 64):
        pr.note_on(n)
    g, o = pr.render_voices(256)
    assert_bits_equal(g, o, "lp2 first")
    st = pr.gpu.export_state()
    for v in range(3):
        cs = pr.cpu.voice(v).state
        for a, b in (("filt_x1", cs.x1), ("filt_x2", cs.x2), ("filt_y1", cs.y1), ("filt_y2", cs.y2)):
            assert np.float32(st[a][v]).view(np.uint32) == np.float32(b).view(np.uint32)
    pr.gpu.set_patch(one); pr.cpu.config = __import__("helpers").oracle_cfg_from_patch(one)
    g, o = pr.render_voices(256)
    assert_bits_equal(g, o, "one-pole in between")
    pr.note_on(70)                                   # restarted under the one-pole: its filter state is zero
    pr.gpu.import_state(pr.gpu.export_state())       # round trip through the checkpoint format
    pr.gpu.set_patch(lp2); pr.cpu.config = __import__("helpers").oracle_cfg_from_patch(lp2)
    for k in range(2):
        g, o = pr.render_voices(256)
        assert_bits_equal(g, o, "lp2 resumed %d" % k)


@pytest.mark.parametrize("kind", [s2.FILT_LP2, s2.FILT_HP1, s2.FILT_BP2, s2.FILT_SVF_LP])
def test_dsp_filters_with_timed_events(kind):
    """timed events (16-frame boundaries inside one launch) under the dsp filters"""
    voices = 100
    patch = make_patch(lpf_kind=kind, lpf_freq=700.0, lpf_damping=0.9)
    pr = Pair(voices, patch, max_frames=1024)
    rng = np.random.RandomState(5 + kind)
    held = []
    for b in range(5):
        frames = 1024 if b != 3 else 1000
        n_ev = int(rng.randint(1, 40))
        times = np.sort(rng.randint(0, (frames + 15) // 16, n_ev)) * 16
        ev = np.zeros(n_ev, dtype=s2.NOTE_EVENT_DTYPE)
        for i, t in enumerate(times):
            on = (not held) or rng.randint(0, 3) > 0
            if on:
                note = int(rng.randint(40, 90)); held.append(note)
            else:
                note = held.pop(int(rng.randint(len(held))))
            ev[i] = (1 if on else 0, note, int(t), 1.0)
        pr.gpu.note_events(ev)
        g = pr.gpu.sample(np.empty(frames, dtype=np.float32))
        pv = np.zeros((voices, frames), dtype=np.float32)
        k = 0
        for c in range(0, frames, 16):
            while k < n_ev and ev["frame"][k] == c:
                if ev["kind"][k]:
                    pr.cpu.note_on(int(ev["note"][k]))
                else:
                    pr.cpu.note_off(int(ev["note"][k]))
                k += 1
            n = min(16, frames - c)
            pv[:, c:c + n] = pr.cpu.render_voices(n)
        o = s2o.mix_tree(pv, pr.block_voices, 1)
        assert_bits_equal(g, o, "timed events under filter %d, buffer %d" % (kind, b))


def test_dsp_filters_full_size_linearity_free_properties():
    """65536 voices under LP2, the quick form (whole pools against the oracle: tests/test_gpu_full_size.py): (a) a 2048-voice
    window of per-voice rows bit-exactly and (b) that the mix equals the documented tree over the
    GPU's own per-voice rows (the summation order is size-independent)."""
    voices = 65536
    patch = make_patch(lpf_kind=s2.FILT_LP2, lpf_freq=1200.0, lpf_damping=0.7, noise=0.1)
    a = s2.Synth(voices, max_frames=256)
    b = s2.Synth(voices, max_frames=256)
    a.set_patch(patch); b.set_patch(patch)
    ev = np.zeros(voices, dtype=s2.NOTE_EVENT_DTYPE)
    ev["kind"] = 1
    ev["note"] = (np.arange(voices) * 13) % 100 + 20
    ev["velocity"] = 1.0
    a.note_events(ev); b.note_events(ev)
    ora = s2o.OracleSynth(2048)
    ora.config = __import__("helpers").oracle_cfg_from_patch(patch)
    for i in range(2048):
        ora.note_on(int(ev["note"][i]))
    for k in range(2):
        mix = a.sample(np.empty(256, dtype=np.float32))
        pv = b.render_voices(256)
        assert_bits_equal(pv[:2048], ora.render_voices(256, threads=8), "first 2048 voices, buffer %d" % k)
        assert_bits_equal(mix, s2o.mix_tree(pv, a.block_voices, 1), "mix vs tree over GPU rows, buffer %d" % k)


# ---------------------------------------------------------------------------------------------
# patch bank: per-voice patches selected by program change
# ---------------------------------------------------------------------------------------------
def _bank():
    b = []
    for i, (osc, filt, fm) in enumerate([(s2.OSC_SAW, s2.FILT_ONEPOLE, 0.0), (s2.OSC_SQUARE, s2.FILT_LP2, 1.5),
                                         (s2.OSC_TRIANGLE, s2.FILT_HP1, 0.0), (s2.OSC_SINE, s2.FILT_BP2, -2.0),
                                         (s2.OSC_SINE, s2.FILT_ONEPOLE, 3.0), (s2.OSC_SAW, s2.FILT_HP2, 0.0),
                                         (s2.OSC_SQUARE, s2.FILT_LP1, 0.5)]):
        p = make_patch(osc_kind=osc, lpf_kind=filt, mod_env_to_osc_freq=fm, noise=0.05 * i, osc_gain=1.0 - 0.1 * i,
                       lpf_freq=300.0 + 250.0 * i, mod_env_to_lpf_freq=4.0 - i, lpf_damping=0.5 + 0.2 * i, lpf_q=1.0 + 0.5 * i)
        p.amp_env.attack_ms = 3.0 + 10.0 * i
        p.amp_env.release_ms = 20.0 + 30.0 * i
        p.mod_env.attack_ms = 2.0 * i
        p.mod_env.decay_ms = 30.0 + 20.0 * i
        p.mod_env.sustain = 0.1 * i
        p.mod_env.release_ms = 15.0 * i
        b.append(p)
    return b


def test_sharded_exchange_through_rccl_with_one_rank():
    """the N-rank exchange (all-gather + rank-ordered combine, and the reduce variant) executed through RCCL by one rank
    on this box's GPU — the collective's calls, shapes, work handles and stream ordering, short of a second GPU"""
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rccl_one_rank_worker.py")],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "RCCL_ONE_RANK_OK" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


def test_fill_begin_mixed_with_the_other_fill_calls():
    """a fill begun and not yet ended leaves its mix to whoever comes next (s2r_host.cpp DeferredMix): another fill_begin
    (with or without timed events), a synchronous s2r_fill, s2r_render_voices, or its own fill_end — the buffers come
    out as if every fill had been a plain s2r_fill"""
    a = s2.Synth(300, max_frames=256)
    b = s2.Synth(300, max_frames=256)
    rng = np.random.RandomState(9)

    def events(timed):
        ev = np.zeros(12, dtype=s2.NOTE_EVENT_DTYPE)
        ev["kind"] = rng.randint(0, 2, 12); ev["note"] = rng.randint(40, 90, 12); ev["velocity"] = 1.0
        ev["frame"] = np.sort(rng.randint(0, 16, 12)) * 16 if timed else 0
        return ev
    init = np.zeros(200, dtype=s2.NOTE_EVENT_DTYPE); init["kind"] = 1; init["note"] = 40 + np.arange(200) % 50; init["velocity"] = 1.0
    a.note_events(init); b.note_events(init)
    want, got = [], []
    plan = ["begin", "begin_timed", "end", "sync", "end", "begin", "voices", "end", "begin", "begin", "end", "end"]
    for step in plan:
        if step == "end":
            got.append(b.sample_end(np.empty(256, dtype=np.float32)).copy())
            continue
        if step == "voices":
            pa, pb = a.render_voices(256), b.render_voices(256)
            assert_bits_equal(pb, pa, "render_voices behind a fill in flight")
            continue
        ev = events(step == "begin_timed")
        a.note_events(ev); b.note_events(ev)
        w = a.sample(np.empty(256, dtype=np.float32)).copy()
        if step == "sync":
            g = b.sample(np.empty(256, dtype=np.float32)).copy()
            assert_bits_equal(g, w, "s2r_fill behind a fill in flight")
        else:
            want.append(w)
            b.sample_begin(256)
    assert len(want) == len(got)
    for k, (w, g) in enumerate(zip(want, got)):
        assert_bits_equal(g, w, "begun fill %d" % k)


@pytest.mark.parametrize("frames", [1024, 1040, 2048])
def test_super_chunk_layouts(frames):
    """the whole fill as one super-chunk (frames <= 1024 on a grid no larger than the device: group sums in one LDS
    buffer) and 256-frame super-chunks in two buffers (longer fills) give the same mix, with note events inside"""
    pr = Pair(700, max_frames=2048)
    rng = np.random.RandomState(frames)
    for v in range(600):
        pr.note_on(30 + (v * 11) % 70)
    for b in range(3):
        n_ev = 40
        times = np.sort(rng.randint(0, frames // 16, n_ev)) * 16
        ev = np.zeros(n_ev, dtype=s2.NOTE_EVENT_DTYPE)
        ev["kind"] = rng.randint(0, 2, n_ev); ev["note"] = 30 + rng.randint(0, 70, n_ev); ev["frame"] = times; ev["velocity"] = 1.0
        pr.gpu.note_events(ev)
        g = pr.gpu.sample(np.empty(frames, dtype=np.float32))
        pv = np.zeros((700, frames), dtype=np.float32)
        k = 0
        for c in range(0, frames, 16):
            while k < n_ev and ev["frame"][k] == c:
                (pr.cpu.note_on if ev["kind"][k] == 1 else pr.cpu.note_off)(int(ev["note"][k]))
                k += 1
            pv[:, c:c + 16] = pr.cpu.render_voices(16)
        assert_bits_equal(g, s2o.mix_tree(pv, pr.block_voices, 1), "%d frames, buffer %d" % (frames, b))


@pytest.mark.parametrize("fams", list(range(1, 16)))
def test_patch_bank_chunk_every_family_set(fams):
    """chunk_bank is instantiated per set of filter families (one-pole; LP1/HP1; LP2/HP2/BP2; SVF) x oscillator set
    (sine / the others / both) x FM or not: every instantiation once, one wave with lanes without a voice, envelopes
    settled (the branch-free chunk), a release in between, per-voice rows against the oracle"""
    kinds = {0: [s2.FILT_ONEPOLE], 1: [s2.FILT_LP1, s2.FILT_HP1], 2: [s2.FILT_LP2, s2.FILT_HP2, s2.FILT_BP2],
             3: [s2.FILT_SVF_LP, s2.FILT_SVF_BP, s2.FILT_SVF_HP]}
    filters = [k for f in range(4) if fams & (1 << f) for k in kinds[f]]
    for oscs in ([s2.OSC_SINE], [s2.OSC_SAW, s2.OSC_SQUARE, s2.OSC_TRIANGLE], [s2.OSC_SAW, s2.OSC_SINE, s2.OSC_TRIANGLE, s2.OSC_SQUARE]):
        for fm in (0.0, 1.25):
            bank = []
            for i in range(max(len(filters), len(oscs)) + 1):
                p = make_patch(osc_kind=oscs[i % len(oscs)], lpf_kind=filters[i % len(filters)], mod_env_to_osc_freq=fm if i % 2 == 0 else 0.0,
                               noise=0.03 * i, osc_gain=1.0 - 0.07 * i, lpf_freq=250.0 + 300.0 * i, mod_env_to_lpf_freq=3.0 - 0.5 * i,
                               lpf_damping=0.6 + 0.2 * i, lpf_q=0.8 + 0.4 * i)
                p.amp_env.attack_ms = 1.0; p.amp_env.decay_ms = 2.0
                p.mod_env.attack_ms = 1.0; p.mod_env.decay_ms = 2.0 + i; p.mod_env.sustain = 0.15 * (i % 5)
                bank.append(p)
            pr = Pair(64, max_frames=512)
            pr.set_bank(bank)
            for v in range(52):                                  # 12 lanes stay without a voice
                pr.program_change(v % len(bank))
                pr.note_on(33 + (v * 7) % 60)
            what = "families %x, oscillators %s, fm %g" % (fams, oscs, fm)
            g, o = pr.render_voices(512)                         # attack and decay, then settled chunks
            assert_bits_equal(g, o, what + ", first buffer")
            for n in range(33, 60, 4):
                pr.note_off(n)
            g, o = pr.render_voices(304)
            assert_bits_equal(g, o, what + ", after releases")


@pytest.mark.parametrize("frames", [512, 1000])
def test_patch_bank_per_voice(frames):
    """every voice renders with the patch its note_on's program selected: all oscillator kinds and
    all six filters side by side in one wavefront, with stealing, releases and ragged fills"""
    bank = _bank()
    pr = Pair(200, max_frames=1024)
    pr.set_bank(bank)
    rng = np.random.RandomState(11)
    held = []
    for k in range(6):
        for _ in range(60):
            pr.program_change(int(rng.randint(len(bank))))
            n = int(rng.randint(30, 100)); pr.note_on(n); held.append(n)
        for _ in range(15):
            pr.note_off(held.pop(int(rng.randint(len(held)))))
        g, o = pr.render_voices(frames)
        assert_bits_equal(g, o, "bank, %d frames, block %d" % (frames, k))
    st = pr.gpu.export_state()
    for v in range(200):
        if pr.cpu.voice(v).has_current:
            assert st["program"][v] == pr.cpu.voice(v).program


def test_patch_bank_mix_checkpoint_and_shrinking_bank():
    bank = _bank()
    pr = Pair(300, block_voices=128, max_frames=512)
    pr.set_bank(bank)
    for v in range(260):
        pr.program_change(v % len(bank))
        pr.note_on(30 + (v * 5) % 70)
    g, o, pv = pr.sample(512)
    assert_bits_equal(g, o, "bank mix")
    pr.gpu.import_state(pr.gpu.export_state())               # the program index travels with the checkpoint
    g, o, pv = pr.sample(500)
    assert_bits_equal(g, o, "bank mix after checkpoint round trip")
    pr.set_bank(bank[:3])                                    # voices started with programs 3..6 fall back to patch 0
    g, o, pv = pr.sample(512)
    assert_bits_equal(g, o, "bank shrunk to 3")
    pr.set_bank(bank[:1])                                    # bank of one: back on the tuned single-patch kernel
    g, o, pv = pr.sample(512)
    assert_bits_equal(g, o, "bank of one")
    with pytest.raises(s2.S2rError):
        pr.gpu.program_change(1)


def test_patch_bank_timed_events_with_program_changes():
    """program changes inside a timed event batch: each timed note_on carries the program current at
    its place in the stream"""
    bank = _bank()
    voices = 64
    pr = Pair(voices, max_frames=1024)
    pr.set_bank(bank)
    rng = np.random.RandomState(3)
    held = []
    for b in range(5):
        frames = 1024 if b != 2 else 1000
        n_ev = int(rng.randint(5, 50))
        times = np.sort(rng.randint(0, (frames + 15) // 16, n_ev)) * 16
        rows = []
        for t in times:
            if rng.randint(0, 2):
                rows.append((2, int(rng.randint(len(bank))), int(t), 0.0))
            on = (not held) or rng.randint(0, 3) > 0
            if on:
                note = int(rng.randint(40, 90)); held.append(note)
            else:
                note = held.pop(int(rng.randint(len(held))))
            rows.append((1 if on else 0, note, int(t), 1.0))
        ev = np.array(rows, dtype=s2.NOTE_EVENT_DTYPE)
        pr.gpu.note_events(ev)
        g = pr.gpu.sample(np.empty(frames, dtype=np.float32))
        pv = np.zeros((voices, frames), dtype=np.float32)
        k = 0
        for c in range(0, frames, 16):
            while k < len(ev) and ev["frame"][k] == c:
                if ev["kind"][k] == 2:
                    pr.cpu.program_change(int(ev["note"][k]))
                elif ev["kind"][k] == 1:
                    pr.cpu.note_on(int(ev["note"][k]))
                else:
                    pr.cpu.note_off(int(ev["note"][k]))
                k += 1
            n = min(16, frames - c)
            pv[:, c:c + n] = pr.cpu.render_voices(n)
        assert k == len(ev)
        assert_bits_equal(g, s2o.mix_tree(pv, pr.block_voices, 1), "bank + timed events, buffer %d" % b)


@pytest.mark.parametrize("kind", [s2.FILT_SVF_LP, s2.FILT_LP2, s2.FILT_ONEPOLE])
def test_config4_shape_192khz(kind):
    """BASELINE config [4]'s shape (4x oversampled rate, resonant filter) as a parity case: the path
    at sample_rate = 192 000 — every rate-dependent constant (Ms::as_samples, periods, filter
    coefficients) changes.  Decimation back to 48 kHz is the caller's (out of scope)."""
    patch = make_patch(osc_kind=s2.OSC_TRIANGLE, lpf_kind=kind, lpf_freq=2500.0, mod_env_to_lpf_freq=3.0, lpf_q=2.0,
                       lpf_damping=0.5, noise=0.02)
    pr = Pair(512, patch, max_frames=1024)
    for v in range(400):
        pr.note_on(24 + (v * 7) % 96)
    for k in range(4):
        g, o, pv = pr.sample(1024, sr=192000)
        assert_bits_equal(g, o, "192 kHz, filter %d, buffer %d" % (kind, k))
        if k == 1:
            for n in range(24, 120, 5):
                pr.note_off(n)


@pytest.mark.parametrize("osc", [s2.OSC_SQUARE, s2.OSC_SAW, s2.OSC_TRIANGLE, s2.OSC_SINE])
def test_fm_patch_after_the_mod_envelope_settles(osc):
    """oscillator FM with the mod envelope in a flat stage (sustain at 0.4, later the end stage): the
    period constants are the ones cached at stage entry and the wave takes the branch-free chunks;
    short envelope times so that attack, decay, sustain, release and end all fall inside the run"""
    patch = make_patch(osc_kind=osc, mod_env_to_osc_freq=2.75, mod_env_to_lpf_freq=5.0, noise=0.1)
    patch.mod_env.attack_ms = 3.0
    patch.mod_env.decay_ms = 12.0
    patch.mod_env.sustain = 0.4
    patch.mod_env.release_ms = 20.0
    pr = Pair(192, patch, max_frames=1024)
    for v in range(150):
        pr.note_on(28 + (v * 5) % 80)
    for k in range(8):
        g, o = pr.render_voices(1024 if k != 5 else 1000)
        assert_bits_equal(g, o, "fm settled, osc %d, buffer %d" % (osc, k))
        if k == 2:
            for n in range(28, 108, 3):
                pr.note_off(n)
        if k == 4:
            for v in range(20):
                pr.note_on(40 + v)          # steals the oldest voices: their groups move again


@pytest.mark.parametrize("base", [(1 << 24) - 700, (1 << 25) + 3, (1 << 28) + 99, (1 << 31) - 1500, (1 << 32) - 9000])
def test_very_old_voices_on_the_branch_free_path(base):
    """voices that have been sounding for minutes to a day: offsets where u32 -> f32 rounds (the
    envelope time, the noise's `offset.cast::<u32>()` of the ROUNDED float, the "last frame of the run"
    test of the branch-free runs), sustained and released, with noise on"""
    patch = make_patch(noise=0.3)
    pr = Pair(70, patch, max_frames=1024)
    for v in range(70):
        pr.note_on(30 + v)
    pr.sample(256)
    st = pr.gpu.export_state()
    for v in range(70):
        off = base + 37 * v
        st["current_frame_offset"][v] = off
        pr.cpu.voice(v).current_frame_offset = off
        if v % 3 == 0:                                   # released long ago or just now
            rel = off - (5 if v % 2 else 4000 + v)
            st["released"][v] = 1; st["release_frame_offset"][v] = rel
            pr.cpu.voice(v).has_release = 1; pr.cpu.voice(v).release_frame_offset = rel
    pr.gpu.import_state(st)
    for k in range(4):
        g, o = pr.render_voices(1024 if k != 2 else 1000)
        assert_bits_equal(g, o, "offsets from %d, buffer %d" % (base, k))


@pytest.mark.parametrize("kind", [s2.FILT_ONEPOLE, s2.FILT_SVF_LP])
def test_oversampled_4x_fill(kind):
    """BASELINE config [4]'s shape end to end (build-defined, self-oracle): the path rendered at 192 kHz
    and decimated to 48 kHz by the 63-tap filter, history carried from call to call; ragged sizes"""
    patch = make_patch(osc_kind=s2.OSC_SAW, lpf_kind=kind, lpf_freq=3000.0, mod_env_to_lpf_freq=2.0, lpf_q=2.0)
    pr = Pair(300, patch, max_frames=4096)
    for v in range(200):
        pr.note_on(28 + (v * 7) % 90)
    hist = np.zeros(62, dtype=np.float32)
    for b, frames in enumerate([1024, 1024, 500, 1, 37, 1024]):
        g = pr.gpu.sample_oversampled(frames, SR)
        pv = pr.cpu.render_voices(4 * frames, 4 * SR)
        x = np.concatenate([hist, s2o.mix_tree(pv, pr.block_voices, 1)])
        o = s2o.decimate4(x, frames)
        hist = x[-62:]
        assert_bits_equal(g, o, "oversampled, filter %d, buffer %d (%d frames)" % (kind, b, frames))
        if b == 2:
            for n in range(28, 118, 4):
                pr.note_off(n)
    with pytest.raises(s2.S2rError):
        pr.gpu.sample_oversampled(1025, SR)              # 4 x 1025 > max_frames


@pytest.mark.parametrize("osc", [s2.OSC_SQUARE, s2.OSC_SAW, s2.OSC_TRIANGLE, s2.OSC_SINE])
@pytest.mark.parametrize("mode", [3, 4, 0])
def test_fm_patch_through_the_coefficient_stream(osc, mode):
    """oscillator FM while the mod envelope moves: the stream carries the LPF coefficient, the period
    and its reciprocal per frame and the branch-free chunk reads them (mode 3/4: forced on for this small
    pool, events in the classification launch / separate kernels; 0: everything in-lane)"""
    patch = make_patch(osc_kind=osc, mod_env_to_osc_freq=3.25, mod_env_to_lpf_freq=6.0, noise=0.15, osc_gain=0.8)
    patch.mod_env.attack_ms = 40.0
    patch.mod_env.decay_ms = 60.0
    patch.mod_env.sustain = 0.35
    patch.mod_env.release_ms = 50.0
    pr = Pair(320, patch, max_frames=1024)
    pr.gpu.set_coeff_stream(mode)
    for v in range(250):
        pr.note_on(26 + (v * 5) % 85)
    for k in range(9):
        g, o, _pv = pr.sample(1024 if k != 4 else 1000)
        assert_bits_equal(g, o, "fm through the stream, osc %d, mode %d, buffer %d" % (osc, mode, k))
        if k == 2:
            for n in range(26, 111, 3):
                pr.note_off(n)
        if k in (3, 6):
            for v in range(30):
                pr.note_on(35 + v)


@pytest.mark.parametrize("kind", DSP_KINDS)
@pytest.mark.parametrize("fm,mode", [(0.0, 3), (1.75, 4)])
def test_dsp_filters_through_the_coefficient_stream(kind, fm, mode):
    """the first/second-order filters and the SVF with a moving mod envelope: alpha / beta / gamma per frame
    come from the stream (plus period and 1/period under oscillator FM), forced on for this small pool"""
    patch = make_patch(osc_kind=s2.OSC_SAW if kind % 2 else s2.OSC_SINE, lpf_kind=kind, lpf_freq=700.0, mod_env_to_lpf_freq=4.0,
                       mod_env_to_osc_freq=fm, lpf_damping=0.7, lpf_q=2.5, noise=0.1)
    patch.mod_env.attack_ms = 30.0
    patch.mod_env.decay_ms = 45.0
    patch.mod_env.sustain = 0.3
    patch.mod_env.release_ms = 35.0
    pr = Pair(320, patch, max_frames=1024)
    pr.gpu.set_coeff_stream(mode)
    for v in range(250):
        pr.note_on(26 + (v * 5) % 85)
    for k in range(8):
        g, o, _pv = pr.sample(1024 if k != 3 else 1000)
        assert_bits_equal(g, o, "filter %d through the stream, fm %g, mode %d, buffer %d" % (kind, fm, mode, k))
        if k == 2:
            for n in range(26, 111, 3):
                pr.note_off(n)
        if k == 5:
            for v in range(30):
                pr.note_on(35 + v)


@pytest.mark.parametrize("patch_kw", [dict(), dict(mod_env_to_osc_freq=2.0, osc_kind=2), dict(lpf_kind=3, lpf_freq=900.0, mod_env_to_lpf_freq=3.0)])
def test_timed_events_with_the_coefficient_stream(patch_kw):
    """note events inside a fill on groups that stream their coefficients: the coefficient pass follows each
    voice's event chain (restart -> new offsets and pitch from that frame on, release -> the release stage), and
    the render kernel runs branch-free up to the wave's next event"""
    voices = 1088
    patch = make_patch(**patch_kw)
    patch.mod_env.decay_ms = 60.0
    patch.mod_env.sustain = 0.25
    patch.mod_env.release_ms = 30.0
    pr = Pair(voices, patch, max_frames=1024)
    pr.threads = 8
    pr.gpu.set_coeff_stream(3)
    rng = np.random.RandomState(17)
    held = []
    for b in range(4):
        frames = 1024 if b != 2 else 1000
        n_ev = int(rng.randint(20, 200)) if b else voices
        times = np.sort(rng.randint(0, (frames + 15) // 16, n_ev)) * 16 if b else np.zeros(n_ev, dtype=np.int64)
        ev = np.zeros(n_ev, dtype=s2.NOTE_EVENT_DTYPE)
        for i, t in enumerate(times):
            on = (not held) or b == 0 or rng.randint(0, 3) > 0
            if on:
                note = int(rng.randint(30, 100)); held.append(note)
            else:
                note = held.pop(int(rng.randint(len(held))))
            ev[i] = (1 if on else 0, note, int(t), 1.0)
        pr.gpu.note_events(ev)
        g = pr.gpu.sample(np.empty(frames, dtype=np.float32))
        pv = np.zeros((voices, frames), dtype=np.float32)
        k = 0
        for c in range(0, frames, 16):
            while k < n_ev and ev["frame"][k] == c:
                (pr.cpu.note_on if ev["kind"][k] else pr.cpu.note_off)(int(ev["note"][k]))
                k += 1
            n = min(16, frames - c)
            pv[:, c:c + n] = pr.cpu.render_voices(n, threads=8)
        assert k == n_ev
        assert_bits_equal(g, s2o.mix_tree(pv, pr.block_voices, 1), "timed events + stream, %s, buffer %d" % (patch_kw, b))


# ---------------------------------------------------------------------------------------------
# alias-suppressed (DPW) oscillators — build-defined, self-oracle (BASELINE config [4])
# ---------------------------------------------------------------------------------------------
DPW_KINDS = [s2.OSC_DPW_SAW, s2.OSC_DPW_SQUARE, s2.OSC_DPW_TRIANGLE]


@pytest.mark.parametrize("osc", DPW_KINDS)
@pytest.mark.parametrize("filt,fm", [(s2.FILT_ONEPOLE, 0.0), (s2.FILT_SVF_LP, 0.0), (s2.FILT_LP2, 2.5), (s2.FILT_ONEPOLE, -1.5)])
def test_dpw_oscillators_per_voice(osc, filt, fm):
    patch = make_patch(osc_kind=osc, lpf_kind=filt, lpf_freq=3000.0, lpf_q=2.0, mod_env_to_osc_freq=fm, noise=0.0, osc_gain=0.0)
    pr = Pair(96, patch, max_frames=1024)
    for v in range(80):
        pr.note_on(30 + (v * 5) % 90)
    for k, frames in enumerate((1024, 1000, 16, 1024)):          # a ragged fill: the scalar tail carries the state too
        g, o = pr.render_voices(frames)
        assert_bits_equal(g, o, "dpw osc %d filt %d fm %g, fill %d" % (osc, filt, fm, k))
        if k == 1:
            for n in range(30, 120, 7):
                pr.note_off(n)
            pr.note_on(64); pr.note_on(65)                       # restarts: the differentiator's memory starts over
    g, want, _ = pr.sample(512)
    assert_bits_equal(g, want, "mix")


def test_dpw_with_timed_events_and_checkpoint():
    patch = make_patch(osc_kind=s2.OSC_DPW_SAW, lpf_kind=s2.FILT_SVF_LP, lpf_freq=5000.0, lpf_q=1.5)
    pr = Pair(40, patch, max_frames=1024)
    rng = np.random.RandomState(11)
    for b in range(4):
        n_ev = 12
        times = np.sort(rng.randint(0, 64, n_ev)) * 16
        ev = np.zeros(n_ev, dtype=s2.NOTE_EVENT_DTYPE)
        ev["kind"] = rng.randint(0, 2, n_ev) | (b == 0); ev["note"] = rng.randint(50, 70, n_ev); ev["frame"] = times; ev["velocity"] = 1.0
        pr.gpu.note_events(ev)
        g = pr.gpu.sample(np.empty(1024, dtype=np.float32))
        pv = np.zeros((40, 1024), dtype=np.float32)
        k = 0
        for c in range(0, 1024, 16):
            while k < n_ev and ev["frame"][k] == c:
                (pr.cpu.note_on if ev["kind"][k] else pr.cpu.note_off)(int(ev["note"][k])); k += 1
            pv[:, c:c + 16] = pr.cpu.render_voices(16, SR)
        assert_bits_equal(g, s2o.mix_tree(pv, pr.block_voices, 1), "buffer %d" % b)
        if b == 1:                                               # checkpoint round trip keeps the memory (osc_z)
            st = pr.gpu.export_state()
            pr.gpu.import_state(st)


def test_config4_share_32768_voices_4x_oversampled_dpw_svf():
    """BASELINE config [4]'s per-GPU share at full size: 262 144 voices / 8 GPUs = 32 768 voices, alias-suppressed
    oscillator + SVF, 4x oversampled (rendered at 192 kHz, decimated to 48 kHz).  The quick form (the whole pool against the oracle,
    timed events and all: test_bench_config4_leg_at_32768_voices_against_the_oracle):
    (a) a 1 024-voice window of per-voice rows at the 4x rate bit for bit against the oracle, (b) the 4x-rate mix equals
    the documented tree over the GPU's own rows, (c) the decimated output equals the oracle's decimator applied to that
    mix (the summation order and the decimator are size-independent)."""
    voices, frames_out = 32768, 256
    patch = make_patch(osc_kind=s2.OSC_DPW_SAW, lpf_kind=s2.FILT_SVF_LP, lpf_freq=4000.0, lpf_q=1.2, mod_env_to_lpf_freq=2.0)
    a = s2.Synth(voices, max_frames=4 * frames_out)
    b = s2.Synth(voices, max_frames=4 * frames_out)
    c = s2.Synth(voices, max_frames=4 * frames_out)
    ev = np.zeros(voices, dtype=s2.NOTE_EVENT_DTYPE)
    ev["kind"] = 1; ev["note"] = (np.arange(voices) * 13) % 90 + 24; ev["velocity"] = 1.0
    for s in (a, b, c):
        s.set_patch(patch); s.note_events(ev)
    ora = s2o.OracleSynth(1024)
    ora.config = __import__("helpers").oracle_cfg_from_patch(patch)
    for i in range(1024):
        ora.note_on(int(ev["note"][i]))
    hist = np.zeros(62, dtype=np.float32)
    for k in range(2):
        mix4 = a.sample(np.empty(4 * frames_out, dtype=np.float32), 4 * SR)
        pv = b.render_voices(4 * frames_out, 4 * SR)
        assert_bits_equal(pv[:1024], ora.render_voices(4 * frames_out, 4 * SR, threads=8), "first 1024 voices at 192 kHz, buffer %d" % k)
        assert_bits_equal(mix4, s2o.mix_tree(pv, a.block_voices, 1), "4x-rate mix vs tree over GPU rows, buffer %d" % k)
        out = c.sample_oversampled(frames_out, SR)
        x = np.concatenate([hist, mix4])
        assert_bits_equal(out, s2o.decimate4(x, frames_out), "decimated output, buffer %d" % k)
        hist = x[-62:]


def test_fill_begin_end_two_buffers_in_flight():
    """s2r_fill_begin / s2r_fill_end (two buffers in flight, s2_bin's arrangement) return what s2r_fill returns"""
    a = s2.Synth(300, max_frames=512)
    b = s2.Synth(300, max_frames=512)
    rng = np.random.RandomState(5)
    want, got = [], []
    buf = np.empty(512, dtype=np.float32)
    for k in range(7):
        ev = np.zeros(20, dtype=s2.NOTE_EVENT_DTYPE)
        ev["kind"] = rng.randint(0, 2, 20) | (k == 0); ev["note"] = rng.randint(40, 90, 20); ev["velocity"] = 1.0
        ev["frame"] = np.sort(rng.randint(0, 32, 20)) * 16 if k % 2 else 0
        frames = 512 if k != 4 else 100
        a.note_events(ev); b.note_events(ev)
        want.append(a.sample(np.empty(frames, dtype=np.float32)).copy())
        b.sample_begin(frames)
        if k:
            n = want[k - 1].size
            got.append(b.sample_end(np.empty(n, dtype=np.float32)).copy())
    got.append(b.sample_end(np.empty(want[-1].size, dtype=np.float32)).copy())
    for k, (w, g) in enumerate(zip(want, got)):
        assert_bits_equal(g, w, "buffer %d" % k)
    with pytest.raises(s2.S2rError):
        b.sample_end(buf)                                        # nothing in flight
    b.sample_begin(16); b.sample_begin(16)
    with pytest.raises(s2.S2rError):
        b.sample_begin(16)                                       # a third one
    b.sample_end(np.empty(16, dtype=np.float32)); b.sample_end(np.empty(16, dtype=np.float32))
