"""ctypes binding of the CPU oracle (oracle/libs2oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; nothing under synth2_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libs2oracle.so")

OSC_SQUARE, OSC_SAW, OSC_TRIANGLE, OSC_SINE = 0, 1, 2, 3


class AdsrCfg(C.Structure):
    _fields_ = [("attack_ms", C.c_float), ("decay_ms", C.c_float), ("sustain", C.c_float), ("release_ms", C.c_float)]


class LayerCfg(C.Structure):
    _fields_ = [("osc_kind", C.c_int32), ("osc_gain", C.c_float), ("noise", C.c_float), ("lpf_freq", C.c_float),
                ("amp_env", AdsrCfg), ("mod_env", AdsrCfg),
                ("mod_env_to_osc_freq", C.c_float), ("mod_env_to_lpf_freq", C.c_float),
                ("lpf_kind", C.c_int32), ("lpf_damping", C.c_float), ("lpf_q", C.c_float)]


class LayerState(C.Structure):
    _fields_ = [("has_phase", C.c_int32), ("phase_accum", C.c_float), ("seed", C.c_uint32), ("lpf_last", C.c_float),
                ("x1", C.c_float), ("x2", C.c_float), ("y1", C.c_float), ("y2", C.c_float),
                ("has_z", C.c_int32), ("dpw_z", C.c_float)]


class Voice(C.Structure):
    _fields_ = [("note", C.c_uint8), ("velocity", C.c_float), ("has_current", C.c_int32),
                ("current_frame_offset", C.c_uint32), ("has_release", C.c_int32),
                ("release_frame_offset", C.c_uint32), ("state", LayerState), ("program", C.c_uint32)]


class SynthS(C.Structure):
    _fields_ = [("config", LayerCfg), ("num_voices", C.c_uint32), ("voices", C.POINTER(Voice)),
                ("panicked", C.c_int32), ("double_release", C.c_uint64),
                ("bank", C.POINTER(LayerCfg)), ("bank_size", C.c_uint32), ("current_program", C.c_uint32)]


class Tree(C.Structure):
    _fields_ = [("block_voices", C.c_uint32), ("groups", C.c_uint32)]


_NATIVE = os.path.join(_HERE, "_build", "libs2oracle_native.so")
_use_native = False


def use_native_build():
    """bench.py's cpu_baseline legs: the same sources compiled on THIS host with -march=native, into
    oracle/_build/.  Must be called before the library is first loaded.  Same bits: -ffp-contract=off, no fast-math."""
    global _use_native
    assert _lib is None, "the oracle library is already loaded"
    # -B: always rebuilt — oracle/_build travels with the repo snapshot, and a -march=native object made on another
    # host (the build container) may hold instructions this one lacks
    subprocess.check_call(["make", "-B", "-C", _HERE, "_build/libs2oracle_native.so"], stdout=subprocess.DEVNULL)
    _use_native = True


def build(force=False):
    if os.environ.get("S2O_LIB"):                # an instrumented build of the same sources (tests/test_sanitizers.py)
        return os.environ["S2O_LIB"]
    if _use_native:
        return _NATIVE
    if force or not os.path.exists(_LIB):
        subprocess.check_call(["make", "-C", _HERE, "libs2oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None
_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    P = C.POINTER(SynthS)
    L.s2o_default_config.restype = LayerCfg
    L.s2o_synth_new.restype = P
    L.s2o_synth_new.argtypes = [C.c_uint32]
    L.s2o_synth_free.argtypes = [P]
    L.s2o_note_on.argtypes = [P, C.c_uint8, C.c_float]
    L.s2o_set_bank.argtypes = [P, C.POINTER(LayerCfg), C.c_uint32]
    L.s2o_program_change.argtypes = [P, C.c_uint32]
    L.s2o_note_off.argtypes = [P, C.c_uint8]
    L.s2o_next_voice_index.restype = C.c_uint32
    L.s2o_next_voice_index.argtypes = [P]
    L.s2o_sample.argtypes = [P, _f32p, C.c_size_t, C.c_uint32]
    L.s2o_render_voices.argtypes = [P, _f32p, C.c_size_t, C.c_uint32]
    L.s2o_render_voices_mt.argtypes = [P, _f32p, C.c_size_t, C.c_uint32, C.c_int]
    L.s2o_sample_mt.argtypes = [P, _f32p, C.c_size_t, C.c_uint32, C.c_int]
    L.s2o_render_events_mt.argtypes = [P, C.c_void_p, C.c_size_t, _f32p, _f32p, C.c_size_t, C.c_uint32, C.c_int]
    L.s2o_events_seconds.argtypes = [C.POINTER(C.c_double), C.c_int]
    L.s2o_pool_pin.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.s2o_mix_sequential.argtypes = [_f32p, C.c_uint32, C.c_size_t, _f32p]
    L.s2o_mix_tree.argtypes = [_f32p, C.c_uint32, C.c_size_t, Tree, _f32p]
    L.s2o_mix_tree_partial.argtypes = [_f32p, C.c_uint32, C.c_size_t, C.c_uint32, _f32p]
    L.s2o_mix_tree_scalar.argtypes = [_f32p, C.c_uint32, C.c_size_t, Tree, _f32p]
    L.s2o_mix_tree_partial_scalar.argtypes = [_f32p, C.c_uint32, C.c_size_t, C.c_uint32, _f32p]
    L.s2o_process_layer_buf_simd.restype = C.c_int
    L.s2o_process_layer_buf_simd.argtypes = [C.POINTER(LayerCfg), C.POINTER(LayerState), C.c_float, C.c_uint32,
                                             C.c_uint32, C.c_int, C.c_uint32, _f32p, C.c_size_t]
    L.s2o_note_to_pitch.restype = C.c_float
    L.s2o_note_to_pitch.argtypes = [C.c_uint8]
    L.s2o_ms_as_samples.restype = C.c_float
    L.s2o_ms_as_samples.argtypes = [C.c_float, C.c_uint32]
    L.s2o_adsr_x16.argtypes = [C.c_float] * 4 + [_u32p, C.c_int, C.c_uint32, _f32p]
    L.s2o_adsr_scalar.restype = C.c_float
    L.s2o_adsr_scalar.argtypes = [C.c_float] * 4 + [C.c_uint32, C.c_int, C.c_uint32]
    L.s2o_hash_word.restype = C.c_uint32
    L.s2o_hash_word.argtypes = [C.c_uint32, C.c_uint32]
    L.s2o_hash_word_x16.argtypes = [_u32p, _u32p, _u32p]
    L.s2o_hash_noise_x16.argtypes = [C.c_uint32, _f32p, _f32p]
    L.s2o_hash_noise.restype = C.c_float
    L.s2o_hash_noise.argtypes = [C.c_uint32, C.c_float]
    for n in ("exclusive", "inclusive"):
        f = getattr(L, "s2o_table_lookup_" + n)
        f.restype = C.c_float
        f.argtypes = [_f32p, C.c_uint32, C.c_float, C.c_float, C.POINTER(C.c_int)]
    for n in ("exclusive", "inclusive", "periodic"):
        getattr(L, "s2o_table_lookup_%s_x16" % n).argtypes = [_f32p, C.c_uint32, _f32p, _f32p, _f32p]
    L.s2o_sin_table.restype = _f32p
    L.s2o_lpf_process.restype = C.c_float
    L.s2o_lpf_process.argtypes = [_f32p, C.c_uint32, C.c_float, C.c_float]
    L.s2o_modulate_freq_unipolar_x16.argtypes = [C.c_float, _f32p, C.c_float, _f32p]
    L.s2o_modulate_freq_unipolar.restype = C.c_float
    L.s2o_modulate_freq_unipolar.argtypes = [C.c_float, C.c_float, C.c_float]
    L.s2o_sleef_powf.restype = C.c_float
    L.s2o_sleef_powf.argtypes = [C.c_float, C.c_float]
    L.s2o_decim4_taps.argtypes = [_f32p]
    L.s2o_decimate4.argtypes = [_f32p, C.c_size_t, _f32p, _f32p]
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _up(a):
    return a.ctypes.data_as(_u32p)


class OracleSynth:
    """s2_lib::try3::synth::Synth restated on the CPU (reference synth.rs:9-203)."""

    def __init__(self, num_voices=8, config=None):
        self.L = lib()
        self.p = self.L.s2o_synth_new(num_voices)
        self.num_voices = num_voices
        if config is not None:
            self.p.contents.config = config

    def __del__(self):
        if getattr(self, "p", None):
            self.L.s2o_synth_free(self.p)
            self.p = None

    @property
    def config(self):
        return self.p.contents.config

    @config.setter
    def config(self, c):
        self.p.contents.config = c

    def set_bank(self, cfgs):
        arr = (LayerCfg * len(cfgs))(*cfgs)
        self.L.s2o_set_bank(self.p, arr, len(cfgs))

    def program_change(self, program):
        self.L.s2o_program_change(self.p, program)

    def note_on(self, note, velocity=1.0):
        self.L.s2o_note_on(self.p, note, velocity)

    def note_off(self, note):
        self.L.s2o_note_off(self.p, note)

    def next_voice_index(self):
        return self.L.s2o_next_voice_index(self.p)

    def voice(self, i):
        return self.p.contents.voices[i]

    def set_seed(self, i, seed):
        self.p.contents.voices[i].state.seed = seed

    def sample(self, frames, sample_rate=48000):
        out = np.zeros(frames, dtype=np.float32)
        self.L.s2o_sample(self.p, _fp(out), frames, sample_rate)
        return out

    def sample_mt(self, frames, sample_rate=48000, threads=1):
        out = np.zeros(frames, dtype=np.float32)
        self.L.s2o_sample_mt(self.p, _fp(out), frames, sample_rate, threads)
        return out

    def render_voices(self, frames, sample_rate=48000, threads=1):
        out = np.zeros((self.num_voices, frames), dtype=np.float32)
        if threads > 1:
            self.L.s2o_render_voices_mt(self.p, _fp(out), frames, sample_rate, threads)
        else:
            self.L.s2o_render_voices(self.p, _fp(out), frames, sample_rate)
        return out

    def render_events(self, events, frames, sample_rate=48000, threads=1, per_voice=True, mix=False):
        """one buffer in s2_bin's pattern (main.rs:138-147): the events (structured array kind/note/frame/velocity, the
        layout of libs2r's s2r_note_event, sorted by frame) applied between 16-frame sample() calls"""
        ev = np.ascontiguousarray(events)
        assert ev.dtype.itemsize == 8
        pv = np.zeros((self.num_voices, frames), dtype=np.float32) if per_voice else None
        mx = np.zeros(frames, dtype=np.float32) if mix else None
        self.L.s2o_render_events_mt(self.p, ev.ctypes.data, ev.size, _fp(pv) if per_voice else None, _fp(mx) if mix else None,
                                    frames, sample_rate, threads)
        return pv if not mix else (mx if not per_voice else (pv, mx))

    @property
    def panicked(self):
        return bool(self.p.contents.panicked)


def pool_pin(cpus):
    """timing legs: thread t of the pool on CPU cpus[t] (thread 0 is the caller: os.sched_setaffinity); [] unpins"""
    arr = (C.c_int * max(1, len(cpus)))(*cpus)
    lib().s2o_pool_pin(arr, len(cpus))


def events_seconds(reset=True):
    """(seconds inside note_on / note_off, seconds rendering) accumulated by render_events"""
    out = (C.c_double * 2)()
    lib().s2o_events_seconds(out, 1 if reset else 0)
    return float(out[0]), float(out[1])


def decim4_taps():
    L = lib()
    h = np.zeros(63, dtype=np.float32)
    L.s2o_decim4_taps(_fp(h))
    return h


def decimate4(x_with_history, n_out):
    """x_with_history: 62 samples of history followed by 4*n_out new samples"""
    L = lib()
    x = np.ascontiguousarray(x_with_history, dtype=np.float32)
    assert x.size == 62 + 4 * n_out
    h = decim4_taps()
    out = np.zeros(n_out, dtype=np.float32)
    L.s2o_decimate4(_fp(x), n_out, _fp(h), _fp(out))
    return out


def mix_sequential(per_voice):
    L = lib()
    pv = np.ascontiguousarray(per_voice, dtype=np.float32)
    out = np.zeros(pv.shape[1], dtype=np.float32)
    L.s2o_mix_sequential(_fp(pv), pv.shape[0], pv.shape[1], _fp(out))
    return out


def mix_tree(per_voice, block_voices=256, groups=1, scalar=False):
    """the GPU's summation tree; scalar=True: the frame-at-a-time statement of it"""
    L = lib()
    pv = np.ascontiguousarray(per_voice, dtype=np.float32)
    out = np.zeros(pv.shape[1], dtype=np.float32)
    (L.s2o_mix_tree_scalar if scalar else L.s2o_mix_tree)(_fp(pv), pv.shape[0], pv.shape[1], Tree(block_voices, groups), _fp(out))
    return out


def mix_tree_partial(per_voice, block_voices=256, scalar=False):
    L = lib()
    pv = np.ascontiguousarray(per_voice, dtype=np.float32)
    out = np.zeros(pv.shape[1], dtype=np.float32)
    (L.s2o_mix_tree_partial_scalar if scalar else L.s2o_mix_tree_partial)(_fp(pv), pv.shape[0], pv.shape[1], block_voices, _fp(out))
    return out


def adsr_x16(attack, decay, sustain, release, offsets, release_offset=None):
    L = lib()
    off = np.asarray(offsets, dtype=np.uint32)
    assert off.size == 16
    out = np.zeros(16, dtype=np.float32)
    L.s2o_adsr_x16(attack, decay, sustain, release, _up(off), release_offset is not None,
                   release_offset or 0, _fp(out))
    return out


def adsr_x16_at(attack, decay, sustain, release, t, release_offset=None):
    """value at absolute frame t (lane 0 of a chunk starting at t)"""
    return adsr_x16(attack, decay, sustain, release, np.arange(t, t + 16, dtype=np.uint64).astype(np.uint32),
                    release_offset)[0]


def sin_table():
    return np.ctypeslib.as_array(lib().s2o_sin_table(), shape=(1024,)).copy()
