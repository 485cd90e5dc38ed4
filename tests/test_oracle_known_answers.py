"""Pins the CPU oracle (oracle/, test infrastructure) against every known answer the reference
holds for this path: its four unit tests (hashnoise.rs:70-98, lookup.rs:250-310), its table data
(tables.rs via CRC and, when /root/reference is mounted, literal by literal), the notebook's
recorded ADSR / modulate_freq outputs (Untitled.ipynb cell 4), and the values derived op by op
from the source text in SURVEY.md §8c."""
import os
import re
import zlib

import numpy as np
import pytest

from oracle import s2o

L = s2o.lib()
f32 = np.float32


def bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


# ---------------------------------------------------------------- reference unit tests

def test_ref_test_hash_word():
    """hashnoise.rs:70-83: scalar hash_word == lane 0 of hash_word_x16; value re-derived in SURVEY §4"""
    start, word = 0xFF00FF00, 0x11111111
    h = L.s2o_hash_word(start, word)
    s16 = np.full(16, start, dtype=np.uint32)
    w16 = np.full(16, word, dtype=np.uint32)
    out = np.zeros(16, dtype=np.uint32)
    L.s2o_hash_word_x16(s2o._up(s16), s2o._up(w16), s2o._up(out))
    assert h == out[0] == 0xb1bdd11e
    assert np.all(out == h)


def test_ref_test_hash_word_dist():
    """hashnoise.rs:85-98, run as the reference runs it: every one of the 200 004 words through the ORACLE's hash_word
    (the x16 form, 16 words per call, and the scalar form on the remainder), sum of popcounts == 200004 * 16"""
    count = 200004
    ones = 0
    start = np.zeros(16, dtype=np.uint32)
    out = np.zeros(16, dtype=np.uint32)
    for base in range(0, count - count % 16, 16):
        w = np.arange(base, base + 16, dtype=np.uint32)
        L.s2o_hash_word_x16(s2o._up(start), s2o._up(w), s2o._up(out))
        ones += int(np.unpackbits(out.view(np.uint8)).sum())
    for k in range(count - count % 16, count):
        ones += bin(L.s2o_hash_word(0, k)).count("1")
    assert ones == count * 16 == 3200064
    # the closed form (rotl(0, 5) ^ i) * 0x9e3779b9 mod 2^32 agrees, and so does the scalar function on a sample
    i = np.arange(count, dtype=np.uint64)
    h = (i * 0x9e3779b9) & 0xffffffff
    assert int(np.unpackbits(h.astype(">u4").view(np.uint8)).sum()) == ones
    for k in (0, 1, 2, 3, 12345, 200003):
        assert L.s2o_hash_word(0, k) == int(h[k])


def _lookup_x16(fn, table, value, rng):
    t = np.asarray(table, dtype=np.float32)
    v = np.full(16, value, dtype=np.float32)
    r = np.full(16, rng, dtype=np.float32)
    out = np.zeros(16, dtype=np.float32)
    fn(s2o._fp(t), t.size, s2o._fp(v), s2o._fp(r), s2o._fp(out))
    return out


def test_ref_test_table_lookup_x16_versions():
    """lookup.rs:250-279 (the function named test_table_lookup exercises the x16 variants)"""
    ex, inc = L.s2o_table_lookup_exclusive_x16, L.s2o_table_lookup_inclusive_x16
    assert np.all(_lookup_x16(ex, [0, 1, 2, 3], 0.0, 4.0) == 0.0)
    assert np.all(_lookup_x16(ex, [0, 1, 2, 3], 0.5, 4.0) == 0.5)
    assert np.all(_lookup_x16(ex, [0, 1, 2, 3], 3.5, 4.0) == 1.5)      # wraps to index 0
    assert np.all(_lookup_x16(inc, [0, 1, 2, 3, 4], 0.0, 4.0) == 0.0)
    assert np.all(_lookup_x16(inc, [0, 1, 2, 3, 4], 0.5, 4.0) == 0.5)
    assert np.all(_lookup_x16(inc, [0, 1, 2, 3, 4], 3.0, 4.0) == 3.0)
    assert np.all(_lookup_x16(inc, [0, 1, 2, 3, 4], 4.0, 4.0) == 4.0)


def test_ref_test_table_lookup_scalar_versions():
    """lookup.rs:281-310"""
    def ex(t, v, r):
        a = np.asarray(t, dtype=np.float32)
        return L.s2o_table_lookup_exclusive(s2o._fp(a), a.size, v, r, None)

    def inc(t, v, r):
        a = np.asarray(t, dtype=np.float32)
        return L.s2o_table_lookup_inclusive(s2o._fp(a), a.size, v, r, None)
    assert ex([0, 1, 2, 3], 0.0, 4.0) == 0.0
    assert ex([0, 1, 2, 3], 0.5, 4.0) == 0.5
    assert ex([0, 1, 2, 3], 3.5, 4.0) == 1.5
    assert inc([0, 1, 2, 3, 4], 0.0, 4.0) == 0.0
    assert inc([0, 1, 2, 3, 4], 0.5, 4.0) == 0.5
    assert inc([0, 1, 2, 3, 4], 3.0, 4.0) == 3.0
    assert inc([0, 1, 2, 3, 4], 4.0, 4.0) == 4.0


# ---------------------------------------------------------------- SIN_TABLE

SIN_TABLE_CRC32 = 0x55293b66     # CRC-32 of the 1024 reference literals as little-endian f32


def test_sin_table_crc():
    assert zlib.crc32(s2o.sin_table().tobytes()) == SIN_TABLE_CRC32


def test_sin_table_spot_values():
    """tables.rs:3,4,258,514,1025"""
    t = s2o.sin_table()
    assert t[0] == 0.0
    assert t[1] == f32(0.006135884672403335571289062500)
    assert t[2] == f32(0.012271538376808166503906250000)
    assert t[256] == 1.0
    assert t[1023] == f32(-0.006135727278888225555419921875)


@pytest.mark.skipif(not os.path.exists("/root/reference/components/s2_lib/src/try3/tables.rs"),
                    reason="reference tree not mounted (GPU box)")
def test_sin_table_against_reference_literals():
    txt = open("/root/reference/components/s2_lib/src/try3/tables.rs").read()
    vals = [float(x) for x in re.findall(r"^\s*(-?\d+\.\d+),\s*$", txt, re.M)]
    assert len(vals) == 1024
    ref = np.array(vals, dtype=np.float64).astype(np.float32)
    assert all(float(a) == b for a, b in zip(ref, vals))         # the literals are exact f32 values
    assert np.array_equal(bits(ref), bits(s2o.sin_table()))
    assert zlib.crc32(ref.tobytes()) == SIN_TABLE_CRC32


# ---------------------------------------------------------------- SURVEY §8c derived values

AMP = (4800.0, 4800.0, 0.5, 4800.0)       # default amp ADSR at 48 kHz (A, D, S, R in samples)
MOD = (0.0, 9600.0, 0.0, 0.0)


def test_ms_as_samples_default_patch():
    assert L.s2o_ms_as_samples(100.0, 48000) == 4800.0
    assert L.s2o_ms_as_samples(200.0, 48000) == 9600.0
    assert L.s2o_ms_as_samples(0.0, 48000) == 0.0


@pytest.mark.parametrize("t,release,want", [
    (0, None, 0.0), (1, None, 0.00020833334), (2, None, 0.00041666668), (3, None, 0.000625),
    (4799, None, 0.9997917), (4800, None, 1.0), (4801, None, 0.9998958), (7200, None, 0.75),
    (9599, None, 0.5001042), (9600, None, 0.5),
    (12000, 12000, 0.5), (12001, 12000, 0.49989584), (14400, 12000, 0.25), (16799, 12000, 0.00010415912),
    (16800, 12000, 0.0),
    (5000, 3000, 0.9791667), (9601, 3000, 0.49989584), (14399, 3000, 0.00010415912), (14400, 3000, 0.0),
])
def test_amp_adsr_x16_known_values(t, release, want):
    """simdtest.rs:270-331 incl. the release clamp max(release, attack+decay)"""
    assert s2o.adsr_x16_at(*AMP, t, release) == f32(want)


@pytest.mark.parametrize("t,want", [(0, 1.0), (1, 0.9998958), (2, 0.9997917), (4800, 0.5), (9599, 0.000104129314), (9600, 0.0)])
def test_mod_adsr_x16_known_values(t, want):
    assert s2o.adsr_x16_at(*MOD, t) == f32(want)


def test_adsr_x16_whole_chunk_matches_lanewise():
    off = np.arange(4790, 4806, dtype=np.uint32)
    chunk = s2o.adsr_x16(*AMP, off)
    for i, o in enumerate(off):
        assert chunk[i] == s2o.adsr_x16_at(*AMP, int(o))


def test_notebook_cell4_adsr_and_modulate_freq():
    """Untitled.ipynb cell 4 (f64 prototype): ADSR(attack 0, decay 10, sustain 0, release 10),
    release at 30, offsets 0..39 -> 1.0, 0.9, ... then 0; modulate_freq(100, env, 1.0) ->
    200, 186.6066, 174.1101, ... 107.1773, 100.  Same stage logic as AdsrX16; f32 vs f64, so
    compared to 1e-6 relative."""
    env = np.concatenate([s2o.adsr_x16(0.0, 10.0, 0.0, 10.0, np.arange(k, k + 16, dtype=np.uint32), 30)
                          for k in (0, 16, 32)])[:40]
    want = [1.0 - 0.1 * k for k in range(10)] + [0.0] * 30
    assert np.allclose(env, want, rtol=0, atol=1e-6)
    freq = np.array([L.s2o_modulate_freq_unipolar(100.0, float(e), 1.0) for e in env])
    assert np.allclose(freq[:3], [200.0, 186.60659830736148, 174.11011265922482], rtol=1e-6)
    assert np.allclose(freq[9], 107.17734625362931, rtol=1e-6)
    assert np.all(freq[10:] == 100.0)


def test_note_to_pitch():
    assert L.s2o_note_to_pitch(69) == 440.0
    assert L.s2o_note_to_pitch(81) == 880.0
    assert L.s2o_note_to_pitch(57) == 220.0


def test_saw_note69_first_samples():
    """pitch 440, period 109.09091, phases 0, 0.009166666, ...; saw 1.0, 0.9816667, ..."""
    cfg = s2o.lib().s2o_default_config()
    cfg.noise = 0.0
    # isolate the oscillator: render the x16 path and undo the +gain/+noise/LPF by reading state
    st = s2o.LayerState()
    period = f32(48000.0) / f32(440.0)
    assert period == f32(109.09091)
    inv = f32(1.0) / period
    assert inv == f32(0.009166666)
    ph = f32(0.0)
    phases, saws = [], []
    for _ in range(6):
        phases.append(ph)
        off = np.float32(np.float64(period) * np.float64(ph))          # fma(period, ph, 0)
        x = np.fmod(off, period)
        saws.append(np.float32(np.float64(f32(-2.0) / period) * np.float64(x) + 1.0))
        ph = np.fmod(f32(ph + inv), f32(1.0))
    assert [float(p) for p in phases] == [float(f32(v)) for v in (0.0, 0.009166666, 0.018333333, 0.0275, 0.036666665, 0.04583333)]
    assert [float(s) for s in saws] == [float(f32(v)) for v in (1.0, 0.9816667, 0.9633333, 0.945, 0.9266667, 0.90833336)]


@pytest.mark.parametrize("off,h,v,want", [(0, 0x0, 0, -1.0), (1, 0x9e3779b9, 31161, -0.049027264),
                                          (2, 0x3c6ef372, 62322, 0.9019455), (3, 0xdaa66d2b, 27947, -0.14711225)])
def test_noise_seed0_known_values(off, h, v, want):
    assert L.s2o_hash_word(0, off) == h
    assert h & 0xffff == v
    assert L.s2o_hash_noise(0, float(off)) == f32(want)
    o = np.zeros(16, dtype=np.float32)
    L.s2o_hash_noise_x16(0, s2o._fp(np.arange(off, off + 16, dtype=np.float32)), s2o._fp(o))
    assert o[0] == f32(want)


def test_noise_offset_collapses_past_2_24():
    """offsets are converted u32 -> f32 -> u32 (process.rs:348, hashnoise.rs:37): 2^24+1 == 2^24"""
    a = L.s2o_hash_noise(0, float(f32(16777217)))
    b = L.s2o_hash_noise(0, float(f32(16777216)))
    assert a == b


# ---------------------------------------------------------------- composed path / semantics

def test_x16_and_sisd_paths_disagree_like_the_reference():
    """process.rs:342-345,353-356 ADD the gains on the x16 path; :287,292 MULTIPLY on the scalar
    path.  With the default patch (noise = 0.0) the x16 path still mixes in full-scale noise."""
    cfg = L.s2o_default_config()
    st16, st1 = s2o.LayerState(), s2o.LayerState()
    b16 = np.zeros(16, dtype=np.float32)
    b15 = np.zeros(15, dtype=np.float32)
    L.s2o_process_layer_buf_simd(cfg, st16, 440.0, 48000, 100, 0, 0, s2o._fp(b16), 16)
    L.s2o_process_layer_buf_simd(cfg, st1, 440.0, 48000, 100, 0, 0, s2o._fp(b15), 15)
    assert not np.array_equal(b16[:15], b15)
    # frame 0 of a fresh voice: saw = 1.0, x16: (1 + 1) + (noise(100) + 0) through the LPF, scalar: 1*1 + noise*0
    assert abs(b15[0]) < abs(b16[0]) or b16[0] != b15[0]


def test_synth_sample_equals_per_voice_sequential_mix():
    s = s2o.OracleSynth(8)
    t = s2o.OracleSynth(8)
    for n in (60, 64, 67, 72):
        s.note_on(n)
        t.note_on(n)
    a = s.sample(1000)
    pv = t.render_voices(1000)
    assert np.array_equal(bits(a), bits(s2o.mix_sequential(pv)))
    # a tree over <= 64 voices of which most are silent stays within a couple of ULP of the sequential order
    tree = s2o.mix_tree(pv, 256, 1)
    assert np.allclose(tree, a, rtol=1e-6, atol=1e-7)


def test_voice_allocation_and_release_policy():
    """synth.rs:61-120: idle voices first by index, then the oldest; note_off hits the LAST active match"""
    s = s2o.OracleSynth(8)
    picks = []
    for i in range(8):
        picks.append(s.next_voice_index())
        s.note_on(60 + i)
    assert picks == list(range(8))
    s.sample(32)
    assert s.next_voice_index() == 0           # all equally old: first index (strict >)
    s.note_on(80)                              # steals voice 0
    s.sample(32)
    assert s.next_voice_index() == 1
    s.note_on(61)                              # voice 1 now also holds note 61 ... it replaced the old 61
    s.note_on(61)                              # voice 2
    s.note_off(61)                             # last active match = voice 2
    assert s.voice(2).has_release == 1 and s.voice(1).has_release == 0
    s.note_off(61)
    assert s.voice(1).has_release == 1
    s.note_off(61)                             # nothing active: no-op
    assert s.p.contents.double_release == 0


def test_offset_saturates_and_panic_flag():
    s = s2o.OracleSynth(8)
    s.note_on(60)
    s.voice(0).current_frame_offset = 0xFFFFFFF8
    s.sample(16)
    assert s.panicked                          # process.rs:36 "overflow"
    assert s.voice(0).current_frame_offset == 0xFFFFFFFF      # synth.rs:197 saturating_add


def test_dsp_filters_restatement_properties():
    """dsp_filters.rs:25-180 has no tests or golden vectors in the reference ("parity unpinned" for
    this row beyond the restatement itself): check the textbook properties the coefficient
    formulas imply — unit DC gain for the low-passes, zero DC gain for the high-passes, and the
    exact first output alpha * x (x2 for the second-order forms) from a zero state."""
    import ctypes as C
    L = s2o.lib()
    L.s2o_dsp_filter_process.restype = C.c_float
    L.s2o_dsp_filter_process.argtypes = [C.c_int] + [C.POINTER(C.c_float)] * 4 + [C.c_uint32, C.c_float, C.c_float, C.c_float]
    for kind, dc in ((1, 1.0), (2, 0.0), (3, 1.0), (4, 0.0), (5, 0.0), (6, 1.0), (7, 0.0), (8, 0.0)):
        st = [C.c_float(0.0) for _ in range(4)]
        y = 0.0
        for i in range(4000):
            y = L.s2o_dsp_filter_process(kind, st[0], st[1], st[2], st[3], 48000, 1000.0, 1.41421354, 1.0)
        assert abs(y - dc) < 1e-4, (kind, y)
    # first output from rest, LP1: alpha = (1 - cos/(1+sin))/2 in f32 steps
    f = np.float32
    theta = f(f(f(2.0) * f(np.pi)) * f(1000.0)) / f(48000.0)
    gamma = f(np.cos(theta, dtype=np.float32)) / f(f(1.0) + f(np.sin(theta, dtype=np.float32)))
    alpha = f(f(1.0) - gamma) / f(2.0)
    st = [C.c_float(0.0) for _ in range(4)]
    y = L.s2o_dsp_filter_process(1, st[0], st[1], st[2], st[3], 48000, 1000.0, 1.0, 0.5)
    assert abs(y - float(alpha * f(0.5))) <= 1e-9
    assert st[0].value == 0.5 and st[2].value == y


def test_decimator_taps_are_a_sane_lowpass():
    """the build-defined 4x decimator: unit DC gain, linear phase, >= 70 dB down from 0.16 cycles per input
    sample (what lies above folds to below 17 kHz at 48 kHz out), within 1 dB up to 0.08 (15 kHz)"""
    h = s2o.decim4_taps().astype(np.float64)
    assert h.size == 63 and abs(h.sum() - 1.0) < 1e-6
    assert np.allclose(h, h[::-1], atol=1e-9)
    f = np.linspace(0.16, 0.5, 400)
    resp = np.abs(np.exp(-2j * np.pi * np.outer(f, np.arange(63))) @ h)
    assert 20 * np.log10(resp.max()) < -70.0
    passband = np.abs(np.exp(-2j * np.pi * np.outer(np.linspace(0, 0.08, 50), np.arange(63))) @ h)
    assert np.all(np.abs(20 * np.log10(passband)) < 1.0)


def test_phased_offset_stays_below_the_period():
    """oscillators.rs:217-239 computes offset = fma(period, phase, 0) and the basic oscillators then take
    offset % period (:66,105,154; lookup.rs:195).  For 0 <= phase <= 1 - 2^-24 (every value fmodf(.., 1.0) can return)
    the product rounds to a float strictly below period — period * 2^-24 is at least half an ulp of period — so the
    remainder is the offset itself: the GPU's branch-free chunk relies on it (s2r_kern_common.h chunk_fast).
    Exhaustive over every float period of ten binades at the two largest phases, plus random pairs."""
    big = np.float32(1.0) - np.float32(2.0 ** -24)
    assert np.nextafter(big, np.float32(2.0)) == np.float32(1.0)
    for e in (-20, -3, 0, 1, 5, 6, 7, 10, 20, 60):
        bits = np.float32(2.0 ** e).view(np.uint32) + np.arange(0, 1 << 23, dtype=np.uint32)
        period = bits.view(np.float32)
        for ph in (big, np.nextafter(big, np.float32(0.0))):
            off = period * ph                       # one correctly rounded float32 product == fma(period, ph, 0)
            assert off.dtype == np.float32
            assert not np.any(off >= period), e
    rng = np.random.default_rng(7)
    period = rng.uniform(0.5, 6000.0, 5_000_000).astype(np.float32)
    ph = np.minimum(rng.uniform(0.0, 1.0, 5_000_000).astype(np.float32), big)
    off = period * ph
    assert not np.any(off >= period)
    assert np.array_equal(np.fmod(off, period), off)


@pytest.mark.parametrize("voices,block,groups,frames", [(8, 256, 1, 100), (300, 64, 1, 33), (1000, 256, 3, 64), (4096 + 17, 128, 2, 16), (70, 1024, 1, 7)])
def test_mix_tree_row_form_equals_the_scalar_statement(voices, block, groups, frames):
    """the tree a row at a time (what the full-size checks use) performs the additions of the frame-at-a-time statement
    in the same order: bit-equal on rows with signed zeros, ragged pools and every grouping"""
    rng = np.random.RandomState(voices + frames)
    pv = (rng.randn(voices, frames) * rng.choice([1e-3, 1.0, 1e3], (voices, 1))).astype(np.float32)
    pv[rng.rand(voices, frames) < 0.1] = -0.0
    pv[rng.rand(voices) < 0.2] = 0.0
    assert np.array_equal(bits(s2o.mix_tree(pv, block, groups)), bits(s2o.mix_tree(pv, block, groups, scalar=True)))
    assert np.array_equal(bits(s2o.mix_tree_partial(pv, block)), bits(s2o.mix_tree_partial(pv, block, scalar=True)))


def test_render_events_is_the_16_frame_call_pattern():
    """s2o_render_events_mt (events applied at their 16-frame boundary, the stretches in between rendered by the thread
    pool) against the reference caller's loop spelled out: apply MIDI, sample(16 frames), repeat (main.rs:138-147)"""
    V = 96
    rng = np.random.RandomState(3)
    a, b = s2o.OracleSynth(V), s2o.OracleSynth(V)
    ev_dtype = np.dtype([("kind", np.uint8), ("note", np.uint8), ("frame", np.uint16), ("velocity", np.float32)])
    for k in range(6):
        frames = 512 if k != 3 else 500
        n = int(rng.randint(0, 60))
        ev = np.zeros(n, dtype=ev_dtype)
        ev["kind"] = rng.randint(0, 2, n) | (k == 0); ev["note"] = rng.randint(50, 60, n); ev["velocity"] = 1.0
        ev["frame"] = np.sort(rng.randint(0, (frames + 15) // 16, n)) * 16
        pv, mix = a.render_events(ev, frames, threads=3, per_voice=True, mix=False), None
        want = np.zeros((V, frames), dtype=np.float32)
        i = 0
        for c in range(0, frames, 16):
            while i < n and ev["frame"][i] == c:
                (b.note_on if ev["kind"][i] else b.note_off)(int(ev["note"][i])); i += 1
            m = min(16, frames - c)
            want[:, c:c + m] = b.render_voices(m)
        assert np.array_equal(bits(pv), bits(want)), k
    # the mix form (thread partials added in thread order): one thread is the reference's sequential order
    c, d = s2o.OracleSynth(V), s2o.OracleSynth(V)
    on = np.zeros(40, dtype=ev_dtype); on["kind"] = 1; on["note"] = 40 + np.arange(40); on["velocity"] = 1.0
    m1 = c.render_events(on, 256, threads=1, per_voice=False, mix=True)
    for e in on:
        d.note_on(int(e["note"]))
    assert np.array_equal(bits(m1), bits(d.sample(256)))
