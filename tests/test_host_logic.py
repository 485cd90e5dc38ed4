"""CPU-only tests of the product's host logic through the C ABI (no device needed): the
voice-allocation policy against the oracle's restatement of synth.rs:61-120, the .synth2
loader, and that libs2r.so loads and exports every symbol include/s2r.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import s2o
import synth2_amd as s2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    L = s2.load_library()
    assert L.s2r_abi_version() == 4
    hdr = open(os.path.join(ROOT, "include", "s2r.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(s2r_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 30
    raw = C.CDLL(s2.synth.lib_path())
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, "declared in include/s2r.h but not exported: %s" % missing


def test_no_cpu_fallback_without_a_device(has_gpu):
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(s2.S2rError) as e:
        s2.Synth(8)
    assert e.value.status == -2      # S2R_ERR_NO_DEVICE
    with pytest.raises(s2.S2rError) as e:
        s2.Synth(512, devices=[0, 0])       # a device list needs its devices just the same
    assert e.value.status == -2


def test_status_strings():
    L = s2.load_library()
    assert L.s2r_status_string(0) == b"ok"
    assert b"overflow" in L.s2r_status_string(-7)


# ---------------------------------------------------------------- .synth2 loader

def test_example_synth2_is_the_default_patch():
    """example.synth2:1-3 (`synth mySynth {` / blank / `}`) == Synth::default_config (synth.rs:125-152)"""
    p = s2.parse_patch("synth mySynth {\n\n}\n")
    d = s2.default_patch()
    assert bytes(p) == bytes(d)
    assert (p.osc_kind, p.osc_gain, p.noise, p.lpf_freq) == (s2.OSC_SAW, 1.0, 0.0, 200.0)
    assert (p.amp_env.attack_ms, p.amp_env.decay_ms, p.amp_env.sustain, p.amp_env.release_ms) == (100.0, 100.0, 0.5, 100.0)
    assert (p.mod_env.attack_ms, p.mod_env.decay_ms, p.mod_env.sustain, p.mod_env.release_ms) == (0.0, 200.0, 0.0, 0.0)
    assert (p.mod_env_to_osc_freq, p.mod_env_to_lpf_freq) == (0.0, 10.0)
    # and equals the oracle's restatement of default_config
    o = s2o.lib().s2o_default_config()
    assert bytes(o) == bytes(p)


def test_patch_full_grammar():
    p = s2.parse_patch("""
        # a comment
        synth lead {
            osc.kind = sine;  osc.gain = 0.25
            noise = 0.125,
            lpf.freq = 1234.5      // Hz
            amp_env.attack = 1; amp_env.decay = 2; amp_env.sustain = 0.75; amp_env.release = 3
            mod_env.attack = 4; mod_env.decay = 5; mod_env.sustain = 0.5; mod_env.release = 6
            mod_env_to_osc_freq = -2.5
            mod_env_to_lpf_freq = 7
        }""")
    assert p.osc_kind == s2.OSC_SINE and p.osc_gain == 0.25 and p.noise == 0.125 and p.lpf_freq == 1234.5
    assert (p.amp_env.attack_ms, p.amp_env.decay_ms, p.amp_env.sustain, p.amp_env.release_ms) == (1, 2, 0.75, 3)
    assert (p.mod_env.attack_ms, p.mod_env.decay_ms, p.mod_env.sustain, p.mod_env.release_ms) == (4, 5, 0.5, 6)
    assert (p.mod_env_to_osc_freq, p.mod_env_to_lpf_freq) == (-2.5, 7.0)
    for kind, code in (("square", 0), ("saw", 1), ("triangle", 2), ("sine", 3)):
        assert s2.parse_patch("synth x { osc.kind = %s }" % kind).osc_kind == code
    # the filter selector: filters.rs one-pole by default, dsp_filters.rs:25-180 by name
    d = s2.parse_patch("synth x { }")
    assert d.lpf_kind == s2.FILT_ONEPOLE and d.lpf_damping == np.float32(2.0 ** 0.5)
    assert d.lpf_q == 3.0
    for kind, code in (("onepole", 0), ("lp1", 1), ("hp1", 2), ("lp2", 3), ("hp2", 4), ("bp2", 5),
                       ("svf_lp", 6), ("svf_bp", 7), ("svf_hp", 8)):
        q = s2.parse_patch("synth x { lpf.kind = %s; lpf.damping = 0.25; lpf.q = 1.5 }" % kind)
        assert q.lpf_kind == code and q.lpf_damping == 0.25 and q.lpf_q == 1.5


@pytest.mark.parametrize("text,status", [
    ("", -4), ("synth", -4), ("synth x", -4), ("synth x {", -4), ("synth x { } trailing", -4),
    ("synth x { bogus = 1 }", -4), ("synth x { osc.kind = sawtooth }", -4), ("synth x { osc.gain 1 }", -4),
    ("synth x { osc.gain = }", -4),
    ("synth x { osc.gain = 1.5 }", -5),               # Unipolar<1>, units.rs:55-65
    ("synth x { noise = -0.1 }", -5),
    ("synth x { amp_env.sustain = 2 }", -5),
    ("synth x { mod_env_to_lpf_freq = 10.5 }", -5),   # Bipolar<10>
    ("synth x { mod_env_to_osc_freq = -11 }", -5),
    ("synth x { amp_env.attack = -1 }", -5),
    ("synth x { lpf.freq = nan }", -5),
    ("synth x { osc.kind = 7 }", -5),
    ("synth x { lpf.kind = bandpass }", -4),
    ("synth x { lpf.kind = 9 }", -5),
    ("synth x { lpf.q = 11 }", -5),                   # Unipolar<10>, dsp_filters.rs:195
    ("synth x { lpf.damping = 10.5 }", -5),           # Unipolar<10>, dsp_filters.rs:96
])
def test_patch_errors(text, status):
    with pytest.raises(s2.S2rError) as e:
        s2.parse_patch(text)
    assert e.value.status == status


# ---------------------------------------------------------------- voice pool policy

@pytest.mark.parametrize("voices,seed", [(1, 0), (2, 1), (8, 2), (8, 3), (64, 4), (300, 5)])
def test_voice_pool_matches_reference_policy(voices, seed):
    """random note_on/note_off/advance streams: the O(log V) pool picks exactly the voices the
    reference's O(V) scans pick (synth.rs:82-120)."""
    rng = np.random.RandomState(seed)
    pool = s2.VoicePool(voices)
    ora = s2o.OracleSynth(voices)
    for step in range(3000):
        r = rng.randint(0, 10)
        note = int(rng.randint(50, 50 + max(2, voices // 2)))
        if r < 5:
            want = ora.next_voice_index()
            assert pool.next_voice() == want
            ora.note_on(note)
            assert pool.note_on(note) == want
        elif r < 8:
            before = [(ora.voice(i).has_current, ora.voice(i).has_release) for i in range(voices)]
            ora.note_off(note)
            after = [(ora.voice(i).has_current, ora.voice(i).has_release) for i in range(voices)]
            changed = [i for i in range(voices) if before[i] != after[i]]
            got = pool.note_off(note)
            assert got == (changed[0] if changed else -1)
        else:
            n = int(rng.choice([16, 16, 16, 64, 1024, 1]))
            pool.advance(n)
            for i in range(voices):      # what Synth::sample does to every started voice (synth.rs:197)
                v = ora.voice(i)
                if v.has_current:
                    v.current_frame_offset = min(0xFFFFFFFF, v.current_frame_offset + n)
    for i in range(voices):
        q = pool.query(i)
        v = ora.voice(i)
        assert bool(q.started) == bool(v.has_current)
        if v.has_current:
            assert q.note == v.note and q.current_frame_offset == v.current_frame_offset
            assert bool(q.released) == bool(v.has_release)
            if v.has_release:
                assert q.release_frame_offset == v.release_frame_offset


def test_voice_pool_many_steals_of_a_held_note_stays_bounded():
    """a note that is re-triggered forever without note_off must not grow the lazy heaps without bound"""
    pool = s2.VoicePool(4)
    for _ in range(20000):
        pool.note_on(60)
        pool.advance(16)
    assert pool.note_off(60) in (0, 1, 2, 3)


def _serde_f32(v):
    """serde_json's rendering of one f32, restated independently of the C++ (digits from numpy's
    shortest-round-trip printer, layout by ryu's format32 rules)"""
    v = np.float32(v)
    if not np.isfinite(v):
        return "null"
    sign = "-" if np.signbit(v) else ""
    v = abs(v)
    if v == 0:
        return sign + "0.0"
    sci = np.format_float_scientific(v, unique=True, trim="-", exp_digits=1)     # d.ddde-7 / de+12
    mant, exp = sci.split("e")
    digits = mant.replace(".", "")
    n, k = len(digits), int(exp) - (len(digits) - 1)
    kk = n + k
    if 0 <= k and kk <= 13:
        return sign + digits + "0" * k + ".0"
    if 0 < kk <= 13:
        return sign + digits[:kk] + "." + digits[kk:]
    if -6 < kk <= 0:
        return sign + "0." + "0" * (-kk) + digits
    return sign + digits[0] + ("." + digits[1:] if n > 1 else "") + "e" + str(kk - 1)


def test_stream_frame_json_is_serde_json_for_vec_f32():
    """threads.rs:303-305 `serde_json::to_string(&buffer32)`: layout rules, null for non-finite values,
    and — the property the consumer (www/streamer.js JSON.parse) needs — every number reads back as
    the same f32"""
    import json
    edge = np.array([0.0, -0.0, 1.0, -1.5, 0.1, 1e-7, 1.5e-7, 123456.78, 1e13, 1e12, 9999999e6, 16777216.0, 3.4028235e38,
                     1e-45, 1.17549435e-38, 0.001234, 1e-5, 1e-6, 0.30000001192092896, np.nan, np.inf, -np.inf], dtype=np.float32)
    assert s2.stream_frame_json(edge[:8]) == "[0.0,-0.0,1.0,-1.5,0.1,1e-7,1.5e-7,123456.78]"
    rng = np.random.RandomState(5)
    bits = rng.randint(0, 2 ** 32, 60000, dtype=np.uint64).astype(np.uint32)
    audio = (rng.standard_normal(4096) * 0.3).astype(np.float32)           # what a frame really holds
    for x in (edge, bits.view(np.float32), audio):
        text = s2.stream_frame_json(x)
        assert text == "[" + ",".join(_serde_f32(v) for v in x) + "]"
        back = np.array([np.nan if v is None else v for v in json.loads(text)], dtype=np.float32)
        fin = np.isfinite(x)
        assert np.array_equal(back[fin].view(np.uint32), x[fin].view(np.uint32))
        assert np.all(np.isnan(back[~fin]))
    assert s2.stream_frame_json(np.zeros(0, dtype=np.float32)) == "[]"


def test_u16_over_65535_two_operation_quotient():
    """s2r_div_u16_by_65535 (s2r_math.h): fma(v, 0x1.0001p-32, v * 2^-16) is RN(v / 65535) for every integer
    0 <= v <= 65535 (hashnoise.rs:46, value / u16_max).  v * (2^-16 + 2^-32 + 2^-48) has at most 49 significant
    bits, so the f64 evaluation below is exact and its one rounding to f32 is the fma's."""
    v = np.arange(65536, dtype=np.float64)
    c = float.fromhex("0x1.0001p-32")
    got = (v * c + v * 2.0 ** -16).astype(np.float32)
    want = v.astype(np.float32) / np.float32(65535.0)
    assert want.dtype == np.float32
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # and the noise value built from it is never a zero, which is what lets a 0.0 noise level skip its add
    n = (got.astype(np.float64) * 2.0 - 1.0).astype(np.float32)
    assert not np.any(n == 0.0)


def test_rust_shim_and_integration_excerpt_follow_the_header():
    """the (uncompilable here) Rust shim declares only entry points the header has, with the header's ABI version, and the
    excerpt INTEGRATION.md prints is made of lines of that file"""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "s2r.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(s2r_[a-z0-9_]+)\s*\(", hdr))
    rs = open(os.path.join(ROOT, "rust", "s2_lib_gpu", "src", "lib.rs")).read()
    used = set(re.findall(r"pub fn (s2r_[a-z0-9_]+)\(", rs))
    assert used and used <= declared, sorted(used - declared)
    assert {"s2r_fill_begin", "s2r_fill_end", "s2r_set_patch_bank", "s2r_fill_stereo", "s2r_fill_oversampled", "s2r_abi_version"} <= used
    ver = int(re.search(r"#define S2R_ABI_VERSION (\d+)", hdr).group(1))
    assert "pub const S2R_ABI_VERSION: u32 = %d;" % ver in rs
    # struct layouts: the field lists of s2r_config, in order
    cfg_c = hdr[:hdr.index("} s2r_config;")]
    cfg_c = cfg_c[cfg_c.rindex("typedef struct {"):]
    fields_c = re.findall(r"\b(?:uint32_t|int32_t)\s+([a-z0-9_]+)(?:\[[A-Z_0-9]+\])?;", cfg_c)
    cfg_rs = re.search(r"pub struct S2rConfig \{(.*?)\}", rs, re.S).group(1)
    fields_rs = re.findall(r"pub ([a-z0-9_]+):", cfg_rs)
    assert fields_c == fields_rs, (fields_c, fields_rs)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    excerpt = re.search(r"```rust\n(.*?)```", doc, re.S).group(1)
    rs_lines = {l.strip() for l in rs.splitlines()}
    for line in excerpt.splitlines():
        line = line.strip()
        if line.startswith("pub fn "):
            assert line in rs_lines, "INTEGRATION.md prints a line that is not in lib.rs: " + line


def test_the_library_is_bound_to_its_sources_by_content(tmp_path):
    """libs2r.so carries the hash of the sources it was built from (s2r_build_id, put there by
    synth2_amd/build.py) and the loader compares THAT with the sources on disk — not file times.  A source edited without
    its time moving (or with a time older than the library's) makes the library stale; the untouched tree does not."""
    import shutil
    from synth2_amd import build as b
    L = s2.load_library()
    have = L.s2r_build_id().decode()
    assert have == b.build_id() == b.embedded_build_id()
    assert have.split("-")[0] == b.source_hash()
    import bench
    assert bench.kernel_source_hash() == b.source_hash()      # the hash the committed profiles are keyed by
    assert not b.needs_build()
    # the same sources somewhere else, one kernel source edited, every file OLDER than the library
    csrc = tmp_path / "csrc"
    shutil.copytree(b.CSRC, csrc)
    assert not b.needs_build(str(csrc))
    f = csrc / "s2r_aux.hip"
    f.write_bytes(f.read_bytes() + b"\n// edited\n")
    old = os.path.getmtime(b.LIB) - 3600.0
    for g in csrc.iterdir():
        os.utime(g, (old, old))
    assert b.source_hash(str(csrc)) != b.source_hash()
    assert b.needs_build(str(csrc)), "a library built from other sources must be refused or rebuilt, whatever the times say"


@pytest.mark.parametrize("threads", [0, 3])
def test_voice_pool_batch_form_equals_event_by_event(threads):
    """S2rVoicePool::resolve_batch — what s2r_note_events runs; with `threads` > 0 the queue on the caller's thread and the
    notes' sets shared among worker threads — makes the choices of the event-by-event policy (itself held against the oracle's
    O(V) restatement of synth.rs:61-120 above) on random streams: stolen voices still held, note_offs that find nothing,
    several boundaries per batch, batches below and above the threshold, a pool smaller than a batch."""
    rng = np.random.RandomState(100 + threads)
    for voices, n_batches, n_ev in ((64, 30, 300), (5000, 12, 3000), (70000, 4, 9000)):
        a = s2.VoicePool(voices)
        b = s2.VoicePool(voices)
        b.set_threads(threads, 256)
        ora = s2o.OracleSynth(voices) if voices <= 5000 else None
        for k in range(n_batches):
            n = int(rng.randint(1, n_ev))
            ev = np.zeros(n, dtype=s2.NOTE_EVENT_DTYPE)
            ev["kind"] = (rng.rand(n) < 0.55).astype(np.uint8)
            ev["note"] = rng.randint(30, 30 + (12 if k % 3 else 90), n)
            ev["velocity"] = rng.rand(n).astype(np.float32)
            ev["frame"] = np.sort(rng.randint(0, 64, n)) * 16 if k % 2 else 0
            want = np.empty(n, dtype=np.int64)
            t = 0
            for j in range(n):
                f = int(ev["frame"][j])
                if f > t:
                    a.advance(f - t)
                    if ora is not None:
                        for v in range(voices):              # (the oracle's clock is its voices' offsets: synth.rs:196-199)
                            vo = ora.p.contents.voices[v]
                            if vo.has_current:
                                vo.current_frame_offset += f - t
                    t = f
                if ev["kind"][j] == 1:
                    if ora is not None:
                        assert a.next_voice() == ora.next_voice_index()
                        ora.note_on(int(ev["note"][j]), float(ev["velocity"][j]))
                    want[j] = a.note_on(int(ev["note"][j]), float(ev["velocity"][j]))
                else:
                    want[j] = a.note_off(int(ev["note"][j]))
                    if ora is not None:
                        ora.note_off(int(ev["note"][j]))
            got, t_end = b.resolve(ev)
            assert t_end == t
            assert np.array_equal(got, want), "batch %d of %d voices: first difference at event %d" % (k, voices, int(np.nonzero(got != want)[0][0]))
            a.advance(1024 - t); b.advance(1024 - t)
            if ora is not None:
                for v in range(voices):
                    vo = ora.p.contents.voices[v]
                    if vo.has_current:
                        vo.current_frame_offset += 1024 - t
            for v in rng.randint(0, voices, 40):
                qa, qb = a.query(int(v)), b.query(int(v))
                assert (qa.note, qa.started, qa.released, qa.current_frame_offset) == (qb.note, qb.started, qb.released, qb.current_frame_offset)
                if qa.released:
                    assert qa.release_frame_offset == qb.release_frame_offset
