"""`s2r_process_layers` — the C-ABI twin of the reference's lower-level public entry, process::process_layer_buf_simd(&sc::Layer,
&mut st::Layer, Hz, SampleRateKhz, offset, Option<u32>, &mut [f32]) (process.rs:14-49; pub through try3/mod.rs), for callers that keep
their own st::Layer — against the oracle's restatement of that function, layer by layer, bit for bit; and `s2r_set_voice_log`, the
reference's `log::debug!("using new voice index {} for note {}")` (synth.rs:118) as a callback."""
import ctypes as C
import numpy as np
import pytest

import synth2_amd as s2
from oracle import s2o
from helpers import assert_bits_equal, oracle_cfg_from_patch

pytestmark = pytest.mark.gpu
SR = 48000


def _oracle_layers(patch, layers, frames):
    """the oracle's process_layer_buf_simd on copies of `layers`: rows, and the states afterwards"""
    L = s2o.lib()
    cfg = oracle_cfg_from_patch(patch)
    rows = np.zeros((layers.size, frames), dtype=np.float32)
    after = layers.copy()
    for i in range(layers.size):
        c = layers[i]
        st = s2o.LayerState()
        st.has_phase = 1; st.phase_accum = float(c["phase_accum"]); st.seed = int(c["noise_seed"]); st.lpf_last = float(c["lpf_last"])
        st.x1, st.x2, st.y1, st.y2 = (float(c[k]) for k in ("filt_x1", "filt_x2", "filt_y1", "filt_y2"))
        z = float(c["osc_z"])
        st.has_z = 0 if np.isnan(z) else 1
        st.dpw_z = 0.0 if np.isnan(z) else z
        rc = L.s2o_process_layer_buf_simd(C.byref(cfg), C.byref(st), float(c["pitch_hz"]), SR, int(c["offset"]), int(c["has_release"]),
                                          int(c["release_offset"]), s2o._fp(rows[i]), frames)
        assert rc == 0
        after[i]["phase_accum"] = st.phase_accum; after[i]["lpf_last"] = st.lpf_last
        for k, v in (("filt_x1", st.x1), ("filt_x2", st.x2), ("filt_y1", st.y1), ("filt_y2", st.y2)):
            after[i][k] = v
    return rows, after


def _random_layers(rng, n):
    a = np.zeros(n, dtype=s2.LAYER_CALL_DTYPE)
    a["pitch_hz"] = np.exp(rng.uniform(np.log(20.0), np.log(9000.0), n)).astype(np.float32)     # any Hz, not only MIDI notes
    a["offset"] = rng.choice([0, 16, 4800, 9584, 9600, 20000, 123456], n) + 16 * rng.randint(0, 40, n)
    rel = rng.rand(n) < 0.4
    a["has_release"] = rel
    a["release_offset"] = np.where(rel, (a["offset"] * rng.rand(n)).astype(np.uint32), 0)
    a["phase_accum"] = rng.rand(n).astype(np.float32) * np.float32(0.999)
    a["lpf_last"] = (rng.rand(n).astype(np.float32) - np.float32(0.5))
    a["osc_z"] = np.float32(np.nan)
    return a


@pytest.mark.parametrize("frames", [16, 100, 7, 1024])
@pytest.mark.parametrize("patch_text", ["synth d { }",
                                        "synth n { osc.kind = sine  noise = 0.25  mod_env_to_osc_freq = 0.5 }",
                                        "synth f { lpf.kind = svf_lp  lpf.q = 2.5  osc.kind = triangle }"])
def test_process_layers_equals_process_layer_buf_simd(frames, patch_text):
    rng = np.random.RandomState(frames + len(patch_text))
    n = 300
    patch = s2.parse_patch(patch_text)
    ws = s2.Synth(512, max_frames=1024)                   # the workspace: more voices than layers
    ws.set_patch(patch)
    layers = _random_layers(rng, n)
    if "noise" in patch_text:
        layers["noise_seed"] = rng.randint(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    for call in range(2):                                  # the second call continues the first's states, `frames` later
        want_rows, want_after = _oracle_layers(patch, layers, frames)
        before = layers.copy()
        got_rows = ws.process_layers(layers, frames, SR)
        assert_bits_equal(got_rows, want_rows, "call %d" % call)
        for k in ("phase_accum", "lpf_last", "filt_x1", "filt_x2", "filt_y1", "filt_y2"):
            assert_bits_equal(layers[k], want_after[k], "call %d: %s" % (call, k))
        for k in ("pitch_hz", "offset", "release_offset", "has_release", "noise_seed"):      # by value / untouched
            assert np.array_equal(layers[k], before[k]), k
        layers["offset"] += frames                         # (the caller's step: synth.rs:197)
    ws.close()


def test_process_layers_with_a_patch_bank_and_dpw_oscillators():
    """`program` picks the layer's sc::Layer from the handle's bank; a DPW oscillator's differentiator memory (osc_z: NaN = none yet)
    travels with the layer like the rest of its state"""
    rng = np.random.RandomState(77)
    bank = [s2.parse_patch("synth a { osc.kind = dpw_saw  lpf.kind = lp2  lpf.damping = 0.9 }"),
            s2.parse_patch("synth b { osc.kind = square  noise = 0.1 }"),
            s2.parse_patch("synth c { osc.kind = dpw_triangle  lpf.kind = svf_bp  lpf.q = 1.5  mod_env_to_lpf_freq = 4 }")]
    ws = s2.Synth(256, max_frames=256)
    ws.set_patch_bank(bank)
    n, frames = 200, 112
    layers = _random_layers(rng, n)
    layers["program"] = rng.randint(0, 3, n)
    layers["phase_accum"] = 0.0                            # (a DPW voice without history starts at phase 0 like a fresh note_on)
    L = s2o.lib()
    for call in range(3):
        want = np.zeros((n, frames), dtype=np.float32)
        after = layers.copy()
        for i in range(n):
            c = layers[i]
            cfg = oracle_cfg_from_patch(bank[int(c["program"])])
            st = s2o.LayerState()
            st.has_phase = 1; st.phase_accum = float(c["phase_accum"]); st.seed = int(c["noise_seed"]); st.lpf_last = float(c["lpf_last"])
            st.x1, st.x2, st.y1, st.y2 = (float(c[k]) for k in ("filt_x1", "filt_x2", "filt_y1", "filt_y2"))
            z = float(c["osc_z"])
            st.has_z = 0 if np.isnan(z) else 1
            st.dpw_z = 0.0 if np.isnan(z) else z
            assert L.s2o_process_layer_buf_simd(C.byref(cfg), C.byref(st), float(c["pitch_hz"]), SR, int(c["offset"]), int(c["has_release"]),
                                                int(c["release_offset"]), s2o._fp(want[i]), frames) == 0
            after[i]["phase_accum"] = st.phase_accum; after[i]["lpf_last"] = st.lpf_last
            after[i]["filt_x1"], after[i]["filt_x2"], after[i]["filt_y1"], after[i]["filt_y2"] = st.x1, st.x2, st.y1, st.y2
            after[i]["osc_z"] = st.dpw_z if st.has_z else np.float32(np.nan)
        got = ws.process_layers(layers, frames, SR)
        assert_bits_equal(got, want, "call %d" % call)
        for k in ("phase_accum", "lpf_last", "filt_x1", "filt_x2", "filt_y1", "filt_y2", "osc_z"):
            assert_bits_equal(layers[k], after[k], "call %d: %s" % (call, k))
        layers["offset"] += frames
    ws.close()


def test_process_layers_errors():
    ws = s2.Synth(256, max_frames=64)
    a = np.zeros(257, dtype=s2.LAYER_CALL_DTYPE)
    with pytest.raises(s2.S2rError):
        ws.process_layers(a, 16, SR)                       # more layers than voices
    a = np.zeros(1, dtype=s2.LAYER_CALL_DTYPE)
    a["pitch_hz"] = 440.0
    a["offset"] = 0xFFFFFFF8
    with pytest.raises(s2.S2rError) as e:
        ws.process_layers(a, 16, SR)                       # process.rs:36 `checked_add(16).expect("overflow")`
    assert e.value.status == -7, e.value.status
    a["offset"] = 0xFFFFFFE0
    ws.process_layers(a, 16, SR)                           # the last chunk that fits
    ws.close()


def test_voice_log_reports_the_policy_s_choice_per_note_on():
    """synth.rs:118: one line per note_on with the index next_voice chose — here a callback, for single events and batches"""
    gpu = s2.Synth(256, max_frames=64)
    ora = s2o.OracleSynth(256)
    seen = []
    gpu.set_voice_log(lambda i, note: seen.append((i, note)))
    want = []
    for note in (60, 61, 60):
        want.append((ora.next_voice_index(), note)); ora.note_on(note)
        gpu.note_on(note)
    ev = np.zeros(300, dtype=s2.NOTE_EVENT_DTYPE)          # a batch that wraps the pool: steals included
    ev["kind"] = 1; ev["note"] = 40 + np.arange(300) % 50; ev["velocity"] = 1.0
    ev["kind"][100:110] = 0
    for e in ev:
        if e["kind"] == 1:
            want.append((ora.next_voice_index(), int(e["note"]))); ora.note_on(int(e["note"]))
        else:
            ora.note_off(int(e["note"]))
    gpu.note_events(ev)
    assert seen == want
    gpu.set_voice_log(None)
    gpu.note_on(70); ora.note_on(70)
    assert len(seen) == len(want)
    assert_bits_equal(gpu.render_voices(64, SR), ora.render_voices(64, SR), "after the logged events")
    gpu.close()
