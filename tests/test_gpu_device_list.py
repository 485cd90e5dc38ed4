"""One Synth over a LIST of devices (s2r_config.devices): the pool cut into one shard per device, the
allocation policy of synth.rs:61-120 run once per event by the handle, every event routed to the shard that holds the
chosen voice, the shards' partial mixes written into the first device's rows and added there in shard order.  A one-GPU
box names its only device several times — the code path is the one N GPUs take except that the rows do not cross xGMI."""
import numpy as np
import pytest

from helpers import Pair, assert_bits_equal, make_patch
from oracle import s2o
import synth2_amd as s2

pytestmark = pytest.mark.gpu

SR = 48000


def _rank_ordered(pv, n, interleave, block):
    """the association an N-shard run produces: every shard's tree, the partial rows added in shard order from +0.0"""
    V = pv.shape[0]
    acc = np.zeros(pv.shape[1], dtype=np.float32)
    for k in range(n):
        idx = s2.synth.shard_pool_indices(V, k, n, interleave) if interleave else np.arange(k * V // n, (k + 1) * V // n)
        acc = acc + s2o.mix_tree_partial(pv[idx], block)
    return acc


@pytest.fixture(params=["", "S2R_FORCE_PEER", "S2R_FORCE_STAGE"])
def multi_device_branch(request):
    """the two multi-device branches of s2r_create / enqueue_multi, reachable on a box with ONE device (read
    at s2r_create): S2R_FORCE_PEER — a shard on the parent's own device goes through the peer set-up (hipDeviceCanAccessPeer,
    hipDeviceEnablePeerAccess) and writes its row into the parent's rows with system-scope stores, counting in on the
    parent's word; S2R_FORCE_STAGE — every shard renders into a row of its own that hipMemcpyPeerAsync moves, events order
    the parent's stream, s2r_sum_rows_kernel adds the rows"""
    import os
    knob = request.param
    if knob:
        os.environ[knob] = "1"
    yield knob
    if knob:
        del os.environ[knob]


@pytest.mark.parametrize("n,interleave", [(2, 0), (2, 64), (4, 64), (3, 0)])
def test_device_list_matches_the_oracle_and_mix_groups(n, interleave, multi_device_branch):
    """contiguous shards: bit-equal to ONE device with mix_groups = N (and to the oracle's tree with that many groups);
    dealt-out shards: bit-equal to the rank-ordered sum of the oracle's per-shard trees.  Untimed and timed events,
    restarts, a ragged fill, the voice index every note_on reports.  Under each of the multi-device branches."""
    V = 256 * n * 2
    pr = Pair(V, max_frames=1024, devices=[0] * n, shard_interleave=interleave)
    ref = s2.Synth(V, max_frames=1024, mix_groups=n) if not interleave else None
    assert pr.gpu.device_count == n and pr.gpu.shard_voices == V
    rng = np.random.RandomState(n * 7 + interleave)
    for v in range(V - 100):
        pr.note_on(36 + v % 61)                       # (asserts the chosen pool index against the oracle's policy)
        if ref:
            ref.note_on(36 + v % 61)
    for b, frames in enumerate([1024, 1000, 1024, 512, 1024]):
        if b in (1, 3):
            for k in range(150):                      # past the idle voices: the oldest ones are stolen, shard after shard
                note = 40 + int(rng.randint(0, 50))
                pr.note_on(note)
                if ref:
                    ref.note_on(note)
            for note in range(36, 97, 4):
                pr.note_off(note)
                if ref:
                    ref.note_off(note)
        timed = None
        if b == 2:
            m = 60
            timed = np.zeros(m, dtype=s2.NOTE_EVENT_DTYPE)
            timed["kind"] = rng.randint(0, 2, m); timed["note"] = 36 + rng.randint(0, 61, m); timed["velocity"] = 1.0
            timed["frame"] = np.sort(rng.randint(0, frames // 16, m)) * 16
            pr.gpu.note_events(timed)
            if ref:
                ref.note_events(timed)
            pv = pr.cpu.render_events(timed, frames, SR, threads=4)
            g = pr.gpu.sample(np.empty(frames, dtype=np.float32), SR)
        else:
            g = pr.gpu.sample(np.empty(frames, dtype=np.float32), SR)
            pv = pr.cpu.render_voices(frames, SR, threads=4)
        assert_bits_equal(g, _rank_ordered(pv, n, interleave, pr.block_voices), "device list of %d, interleave %d, buffer %d" % (n, interleave, b))
        if ref:
            assert_bits_equal(g, s2o.mix_tree(pv, pr.block_voices, n), "vs the oracle's tree with %d groups, buffer %d" % (n, b))
            assert_bits_equal(g, ref.sample(np.empty(frames, dtype=np.float32), SR), "vs one device with mix_groups=%d, buffer %d" % (n, b))


def test_device_list_every_entry_point(multi_device_branch):
    """render_voices in pool order, fill_begin / fill_end with two buffers in flight, the stereo copy, the 4x-oversampled
    fill, a patch bank with program changes, noise seeds, export -> import into a single-device handle and back"""
    V, n = 1024, 2
    bank = [make_patch(osc_kind=s2.OSC_SAW), make_patch(osc_kind=s2.OSC_SINE, lpf_kind=s2.FILT_LP2, lpf_freq=900.0, noise=0.2),
            make_patch(osc_kind=s2.OSC_DPW_SAW, lpf_kind=s2.FILT_SVF_LP, lpf_q=1.3)]
    multi = s2.Synth(V, max_frames=2048, devices=[0] * n, shard_interleave=64)
    single = s2.Synth(V, max_frames=2048)
    pr_cpu = s2o.OracleSynth(V)
    from helpers import oracle_cfg_from_patch
    for s in (multi, single):
        s.set_patch_bank(bank)
    pr_cpu.set_bank([oracle_cfg_from_patch(p) for p in bank])
    for s in (multi, single):
        s.set_noise_seed(5, 12345); s.set_noise_seed(700, 99)
    for v in range(600):
        prog = v % 3
        for s in (multi, single):
            s.program_change(prog)
        pr_cpu.program_change(prog)
        assert multi.note_on(30 + v % 70) == single.note_on(30 + v % 70)
        pr_cpu.note_on(30 + v % 70)
    pr_cpu.set_seed(5, 12345)
    # (the seed of voice 700 is not live: the voice is idle, and a note_on resets the reference's seed — the override is the build's variant)
    # per-voice rows, pool order
    a, b = multi.render_voices(512), single.render_voices(512)
    assert_bits_equal(a, b, "render_voices: device list vs single device")
    assert_bits_equal(a, pr_cpu.render_voices(512, threads=4), "render_voices vs the oracle")
    # two buffers in flight
    for k in range(4):
        frames = 1024 if k != 2 else 300
        multi.sample_begin(frames); single.sample_begin(frames)
        if k:
            n_prev = 1024 if k - 1 != 2 else 300
            assert multi.pending_frames == n_prev
            g1 = multi.sample_end(np.empty(n_prev, dtype=np.float32)); g2 = single.sample_end(np.empty(n_prev, dtype=np.float32))
            pv = pr_cpu.render_voices(n_prev, threads=4)
            assert_bits_equal(g1, _rank_ordered(pv, n, 64, multi.block_voices), "fill_begin/end on a device list, buffer %d" % (k - 1))
            assert_bits_equal(g2, s2o.mix_tree(pv, single.block_voices, 1), "fill_begin/end single, buffer %d" % (k - 1))
    with pytest.raises(s2.S2rError):
        multi.sample_end(np.empty(100, dtype=np.float32))        # too small for the 1024-frame fill in flight: refused, nothing consumed
    g1 = multi.sample_end(np.empty(1024, dtype=np.float32)); single.sample_end(np.empty(1024, dtype=np.float32))
    pv = pr_cpu.render_voices(1024, threads=4)
    assert_bits_equal(g1, _rank_ordered(pv, n, 64, multi.block_voices), "last buffer in flight")
    # stereo
    lr = multi.sample_stereo(256)
    mono = _rank_ordered(pr_cpu.render_voices(256, threads=4), n, 64, multi.block_voices)
    single.sample(np.empty(256, dtype=np.float32))
    assert_bits_equal(lr[:, 0], mono, "stereo L"); assert_bits_equal(lr[:, 1], mono, "stereo R")
    # checkpoint: the device list's state (pool order) equals the single-device handle's, and goes into a fresh device list of 4
    st = multi.export_state()
    assert st.size == V
    st1 = single.export_state()
    for f in ("started", "released", "current_frame_offset", "release_frame_offset", "program", "note"):
        assert np.array_equal(st[f], st1[f]), f
    assert_bits_equal(st["phase_accum"], st1["phase_accum"]); assert_bits_equal(st["lpf_last"], st1["lpf_last"])
    other = s2.Synth(V, max_frames=2048, devices=[0] * 4, shard_interleave=16, block_voices=64)
    other.set_patch_bank(bank)
    other.import_state(st)
    for s in (multi, single, other):
        s.note_on(64); s.note_off(40)
    pr_cpu.note_on(64); pr_cpu.note_off(40)
    pv = pr_cpu.render_voices(512, threads=4)
    assert_bits_equal(multi.sample(np.empty(512, dtype=np.float32)), _rank_ordered(pv, n, 64, 256), "after the checkpoint, device list of 2")
    assert_bits_equal(other.sample(np.empty(512, dtype=np.float32)), _rank_ordered(pv, 4, 16, 64), "imported into a device list of 4")


def test_device_list_rejects_what_it_cannot_be():
    with pytest.raises(s2.S2rError):
        s2.Synth(1000, devices=[0, 0])                          # not a multiple of devices x block_voices
    with pytest.raises(s2.S2rError):
        s2.Synth(1024, devices=[0, 99])                         # no such device
    with pytest.raises(s2.S2rError):
        s2.Synth(1024, devices=[0, 0], shard_begin=512)         # a device list shards by itself
    s = s2.Synth(1024, devices=[0, 0])
    import torch
    row = torch.zeros(1024, device="cuda")
    with pytest.raises(s2.S2rError):
        s.fill_device(row.data_ptr(), 1024)                     # the per-shard building block
