"""Host-side mirror of ``s2_lib::try3::synth`` over the C ABI of libs2r (include/s2r.h).

Python is only the test / bench driver here (the reference's host language, Rust, is not in
this image; rust/s2_lib_gpu holds the uncompiled Rust shim and include/s2_synth.hpp the C++
one).  Names and argument meaning follow the reference
(/root/reference/components/s2_lib/src/try3/synth.rs:9-21,53-80,154-156;
units.rs:11-14):

    synth = Synth()                      # Synth::new()
    synth.note_on(Note(69), Velocity(1.0))
    synth.sample(buffer, SampleRateKhz(48000))
    synth.note_off(Note(69))

There is no CPU fallback: importing works anywhere, constructing a Synth needs libs2r.so and
a gfx950 device and raises otherwise.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

OSC_SQUARE, OSC_SAW, OSC_TRIANGLE, OSC_SINE = 0, 1, 2, 3
OSC_DPW_SAW, OSC_DPW_SQUARE, OSC_DPW_TRIANGLE = 4, 5, 6      # build-defined alias-suppressed shapes

S2R_OK = 0
S2R_ERR_INVALID = -1
S2R_ERR_NO_DEVICE = -2
S2R_ERR_HIP = -3
S2R_ERR_PATCH_SYNTAX = -4
S2R_ERR_PATCH_RANGE = -5
S2R_ERR_TOO_MANY_FRAMES = -6
S2R_ERR_OFFSET_OVERFLOW = -7
S2R_ERR_OUT_OF_MEMORY = -8


class S2rError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("libs2r status %d: %s" % (status, message))
        self.status = status


class Adsr(C.Structure):
    """static_config.rs:38-44"""
    _fields_ = [("attack_ms", C.c_float), ("decay_ms", C.c_float), ("sustain", C.c_float), ("release_ms", C.c_float)]


class Patch(C.Structure):
    """static_config.rs:4-24 (sc::Layer)"""
    _fields_ = [("osc_kind", C.c_int32), ("osc_gain", C.c_float), ("noise", C.c_float), ("lpf_freq", C.c_float),
                ("amp_env", Adsr), ("mod_env", Adsr),
                ("mod_env_to_osc_freq", C.c_float), ("mod_env_to_lpf_freq", C.c_float),
                ("lpf_kind", C.c_int32), ("lpf_damping", C.c_float), ("lpf_q", C.c_float)]


# s2r_filter_kind: filters.rs one-pole (the reference's live path) and dsp_filters.rs:25-180
FILT_ONEPOLE, FILT_LP1, FILT_HP1, FILT_LP2, FILT_HP2, FILT_BP2 = 0, 1, 2, 3, 4, 5
FILT_SVF_LP, FILT_SVF_BP, FILT_SVF_HP = 6, 7, 8      # build-defined state-variable filter


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("total_voices", C.c_uint32), ("shard_begin", C.c_uint32),
                ("shard_voices", C.c_uint32), ("max_frames", C.c_uint32), ("device", C.c_int32),
                ("block_voices", C.c_uint32), ("mix_groups", C.c_uint32), ("reserved0", C.c_uint32),
                ("shard_interleave", C.c_uint32), ("shard_index", C.c_uint32), ("shard_count", C.c_uint32),
                ("n_devices", C.c_uint32), ("devices", C.c_int32 * 16)]


class VoiceState(C.Structure):
    _fields_ = [("note", C.c_uint8), ("started", C.c_uint8), ("released", C.c_uint8), ("program", C.c_uint8),
                ("current_frame_offset", C.c_uint32), ("release_frame_offset", C.c_uint32),
                ("pitch_hz", C.c_float), ("phase_accum", C.c_float), ("lpf_last", C.c_float),
                ("noise_seed", C.c_uint32), ("velocity", C.c_float),
                ("filt_x1", C.c_float), ("filt_x2", C.c_float), ("filt_y1", C.c_float), ("filt_y2", C.c_float),
                ("osc_z", C.c_float)]


VOICE_STATE_DTYPE = np.dtype([("note", np.uint8), ("started", np.uint8), ("released", np.uint8), ("program", np.uint8),
                              ("current_frame_offset", np.uint32), ("release_frame_offset", np.uint32),
                              ("pitch_hz", np.float32), ("phase_accum", np.float32), ("lpf_last", np.float32),
                              ("noise_seed", np.uint32), ("velocity", np.float32),
                              ("filt_x1", np.float32), ("filt_x2", np.float32), ("filt_y1", np.float32), ("filt_y2", np.float32),
                              ("osc_z", np.float32)])
assert VOICE_STATE_DTYPE.itemsize == C.sizeof(VoiceState)
LAYER_CALL_DTYPE = np.dtype([("pitch_hz", np.float32), ("offset", np.uint32), ("release_offset", np.uint32),
                             ("has_release", np.uint8), ("program", np.uint8), ("_pad", np.uint8, (2,)),
                             ("phase_accum", np.float32), ("lpf_last", np.float32), ("noise_seed", np.uint32),
                             ("filt_x1", np.float32), ("filt_x2", np.float32), ("filt_y1", np.float32), ("filt_y2", np.float32),
                             ("osc_z", np.float32)])
assert LAYER_CALL_DTYPE.itemsize == 48
VOICE_LOG_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint8)
NOTE_EVENT_DTYPE = np.dtype([("kind", np.uint8), ("note", np.uint8), ("frame", np.uint16), ("velocity", np.float32)])
assert NOTE_EVENT_DTYPE.itemsize == 8


class Note(int):
    """synth.rs:16 ``pub struct Note(pub u8)``"""
    def __new__(cls, v):
        if not 0 <= int(v) <= 255:
            raise ValueError("Note is a u8")
        return super().__new__(cls, int(v))


class Velocity(float):
    """synth.rs:18 ``pub struct Velocity(pub Unipolar<1>)`` (stored, never used in rendering)"""


class SampleRateKhz(int):
    """units.rs:14 — named Khz in the reference but holds Hz (units.rs:21)"""


_lib = None
_f32p = C.POINTER(C.c_float)


def lib_path():
    return _build.LIB


def load_library():
    """dlopen libs2r.so; builds it first when it was not built from the sources on disk (s2r_build_id: content, not times)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if _build.needs_build():
        # no library, or one built from other sources: rebuild, and fail loudly if that is impossible — a stale
        # libs2r.so would let tests and the bench pass against old code
        try:
            path = _build.build()
        except Exception as e:
            raise RuntimeError("libs2r.so is missing or was not built from these sources (build id %s, sources %s) and could not be rebuilt: %s" % (
                _build.embedded_build_id(), _build.build_id(), e)) from e
    L = C.CDLL(path)
    if not _build.AB_LIB:
        L.s2r_build_id.restype = C.c_char_p
        have = L.s2r_build_id().decode()
        if have != _build.build_id():
            raise RuntimeError("libs2r.so says it was built from %s; the sources on disk are %s" % (have, _build.build_id()))
    H = C.c_void_p
    sig = {
        "s2r_abi_version": (C.c_uint32, []),
        "s2r_build_id": (C.c_char_p, []),
        "s2r_status_string": (C.c_char_p, [C.c_int]),
        "s2r_create": (C.c_int, [C.POINTER(Config), C.POINTER(H)]),
        "s2r_destroy": (None, [H]),
        "s2r_load_patch": (C.c_int, [H, C.c_char_p, C.c_size_t]),
        "s2r_set_patch": (C.c_int, [H, C.POINTER(Patch)]),
        "s2r_get_patch": (C.c_int, [H, C.POINTER(Patch)]),
        "s2r_stream_frame_json": (C.c_size_t, [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]),
        "s2r_set_patch_bank": (C.c_int, [H, C.POINTER(Patch), C.c_uint32]),
        "s2r_patch_bank_size": (C.c_uint32, [H]),
        "s2r_program_change": (C.c_int, [H, C.c_uint32]),
        "s2r_default_patch": (None, [C.POINTER(Patch)]),
        "s2r_note_on": (C.c_int, [H, C.c_uint8, C.c_float]),
        "s2r_note_on_ex": (C.c_int, [H, C.c_uint8, C.c_float, C.POINTER(C.c_uint32)]),
        "s2r_note_off": (C.c_int, [H, C.c_uint8]),
        "s2r_note_events": (C.c_int, [H, C.c_void_p, C.c_size_t]),
        "s2r_fill": (C.c_int, [H, _f32p, C.c_size_t, C.c_uint32]),
        "s2r_fill_begin": (C.c_int, [H, C.c_size_t, C.c_uint32]),
        "s2r_fill_end": (C.c_int, [H, _f32p, C.c_size_t]),
        "s2r_fill_pending_frames": (C.c_size_t, [H]),
        "s2r_fills_in_flight": (C.c_uint32, [H]),
        "s2r_fill_stereo": (C.c_int, [H, _f32p, C.c_size_t, C.c_uint32]),
        "s2r_fill_oversampled": (C.c_int, [H, _f32p, C.c_size_t, C.c_uint32]),
        "s2r_fill_device": (C.c_int, [H, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]),
        "s2r_fill_device_root": (C.c_int, [H, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]),
        "s2r_sum_partials_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_size_t, C.c_void_p, C.c_void_p]),
        "s2r_render_voices": (C.c_int, [H, _f32p, C.c_size_t, C.c_uint32]),
        "s2r_process_layers": (C.c_int, [H, C.c_void_p, C.c_uint32, _f32p, C.c_size_t, C.c_uint32]),
        "s2r_set_voice_log": (C.c_int, [H, VOICE_LOG_FN, C.c_void_p]),
        "s2r_export_state": (C.c_int, [H, C.c_void_p]),
        "s2r_import_state": (C.c_int, [H, C.c_void_p]),
        "s2r_set_noise_seed": (C.c_int, [H, C.c_uint32, C.c_uint32]),
        "s2r_shard_voices": (C.c_uint32, [H]),
        "s2r_block_voices": (C.c_uint32, [H]),
        "s2r_device_count": (C.c_uint32, [H]),
        "s2r_double_release_count": (C.c_uint64, [H]),
        "s2r_set_timing": (C.c_int, [H, C.c_int]),
        "s2r_set_low_latency": (C.c_int, [H, C.c_int]),
        "s2r_low_latency_active": (C.c_int, [H]),
        "s2r_set_resident": (C.c_int, [H, C.c_int]),
        "s2r_resident_active": (C.c_int, [H]),
        "s2r_quiesce": (C.c_int, [H]),
        "s2r_exchange_create": (C.c_int, [H, C.c_uint32, C.c_void_p, C.c_size_t]),
        "s2r_exchange_attach": (C.c_int, [H, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]),
        "s2r_set_flat_shortcut": (C.c_int, [H, C.c_int]),
        "s2r_set_coeff_stream": (C.c_int, [H, C.c_int]),
        "s2r_last_render_ms": (C.c_float, [H]),
        "s2r_last_error": (C.c_char_p, [H]),
        "s2r_parse_patch_text": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(Patch), C.c_char_p, C.c_size_t]),
        "s2r_voice_pool_create": (H, [C.c_uint32]),
        "s2r_voice_pool_destroy": (None, [H]),
        "s2r_voice_pool_note_on": (C.c_uint32, [H, C.c_uint8, C.c_float]),
        "s2r_voice_pool_note_off": (C.c_int64, [H, C.c_uint8]),
        "s2r_voice_pool_advance": (None, [H, C.c_uint64]),
        "s2r_voice_pool_next_voice": (C.c_uint32, [H]),
        "s2r_voice_pool_set_threads": (None, [H, C.c_uint32, C.c_size_t]),
        "s2r_voice_pool_resolve": (C.c_uint32, [H, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]),
        "s2r_voice_pool_query": (C.c_int, [H, C.c_uint32, C.POINTER(VoiceState)]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    # s2r_fill once more, taking the buffer's address as an integer (Synth.sample)
    L._fill_raw = C.CFUNCTYPE(C.c_int, H, C.c_void_p, C.c_size_t, C.c_uint32)(("s2r_fill", L))
    _lib = L
    return L


def default_patch():
    """Synth::default_config() (synth.rs:125-152)"""
    p = Patch()
    load_library().s2r_default_patch(C.byref(p))
    return p


def parse_patch(text):
    """.synth2 text -> Patch (host only, no device needed)"""
    L = load_library()
    p = Patch()
    raw = text.encode() if isinstance(text, str) else bytes(text)
    err = C.create_string_buffer(256)
    rc = L.s2r_parse_patch_text(raw, len(raw), C.byref(p), err, len(err))
    if rc != S2R_OK:
        raise S2rError(rc, err.value.decode())
    return p


def stream_frame_json(samples):
    """one buffer as the text frame of the reference's websocket audio server (threads.rs:303-305):
    serde_json's rendering of a Vec<f32> (host only, no device needed)"""
    L = load_library()
    a = np.ascontiguousarray(samples, dtype=np.float32)
    cap = L.s2r_stream_frame_json(None, a.size, None, 0)      # the worst case for n samples (3 + 18 n)
    buf = C.create_string_buffer(cap)
    n = L.s2r_stream_frame_json(a.ctypes.data_as(C.c_void_p), a.size, buf, cap)
    return buf.raw[:n].decode("ascii")


def shard_pool_indices(total_voices, shard_index, shard_count, interleave):
    """pool indices of a round-robin shard's local voices, in local order (s2r_config.shard_interleave)"""
    local = np.arange(total_voices // shard_count)
    return ((local // interleave) * shard_count + shard_index) * interleave + local % interleave


class VoicePool:
    """The allocation policy of Synth (synth.rs:61-120) without a device."""

    def __init__(self, total_voices):
        self.L = load_library()
        self.p = self.L.s2r_voice_pool_create(total_voices)
        if not self.p:
            raise MemoryError("s2r_voice_pool_create")

    def __del__(self):
        if getattr(self, "p", None):
            self.L.s2r_voice_pool_destroy(self.p)
            self.p = None

    def note_on(self, note, velocity=1.0):
        return self.L.s2r_voice_pool_note_on(self.p, note, velocity)

    def note_off(self, note):
        return self.L.s2r_voice_pool_note_off(self.p, note)

    def advance(self, frames):
        self.L.s2r_voice_pool_advance(self.p, frames)

    def set_threads(self, worker_threads, batch_threshold=4096):
        self.L.s2r_voice_pool_set_threads(self.p, worker_threads, batch_threshold)

    def resolve(self, events, frames_moved=0):
        """the batch form: (voice per event, frame of the last event)"""
        ev = np.ascontiguousarray(events, dtype=NOTE_EVENT_DTYPE)
        out = np.empty(ev.size, dtype=np.int64)
        t = self.L.s2r_voice_pool_resolve(self.p, ev.ctypes.data, ev.size, frames_moved, out.ctypes.data)
        return out, int(t)

    def next_voice(self):
        return self.L.s2r_voice_pool_next_voice(self.p)

    def query(self, i):
        st = VoiceState()
        rc = self.L.s2r_voice_pool_query(self.p, i, C.byref(st))
        if rc != S2R_OK:
            raise S2rError(rc, "voice pool query")
        return st


class Synth:
    """``s2_lib::try3::synth::Synth`` on an MI355X.

    ``num_voices`` replaces the reference's ``NUM_VOICES = 8`` (synth.rs:7).  ``devices=[d0, d1, ...]``: ONE Synth over
    several GPUs (s2r_config.devices: the pool cut into one shard per device, the allocation policy run once, the
    shards' partial mixes added in shard order on d0).  With one process per GPU instead (torch.distributed,
    synth2_amd/sharded.py) every rank builds a Synth over the same pool with its own ``shard_begin`` /
    ``shard_voices`` (or ``shard_interleave`` / ``shard_index`` / ``shard_count``) and feeds it the same note events.
    """

    def __init__(self, num_voices=8, max_frames=2048, device=-1, shard_begin=0, shard_voices=0,
                 block_voices=0, mix_groups=0, shard_interleave=0, shard_index=0, shard_count=1, devices=None):
        self.L = load_library()
        self.h = C.c_void_p()
        devs = list(devices) if devices is not None else []
        if len(devs) > 16:
            raise ValueError("a device list holds at most 16 devices")
        cfg = Config(C.sizeof(Config), num_voices, shard_begin, shard_voices, max_frames, device, block_voices, mix_groups,
                     0, shard_interleave, shard_index, shard_count, len(devs), (C.c_int32 * 16)(*devs))
        rc = self.L.s2r_create(C.byref(cfg), C.byref(self.h))
        if rc != S2R_OK:
            self.h = None
            raise S2rError(rc, self.L.s2r_status_string(rc).decode())
        self.num_voices = num_voices
        self.max_frames = max_frames
        self.shard_voices = self.L.s2r_shard_voices(self.h)
        self.block_voices = self.L.s2r_block_voices(self.h)
        self.device_count = self.L.s2r_device_count(self.h)
        self._fill_raw = self.L._fill_raw

    # Synth::new() (synth.rs:54-59)
    @classmethod
    def new(cls):
        return cls()

    def close(self):
        if getattr(self, "h", None):
            self.L.s2r_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != S2R_OK:
            raise S2rError(rc, self.L.s2r_last_error(self.h).decode() or self.L.s2r_status_string(rc).decode())

    # --- patch ---
    def load_patch(self, text):
        raw = text.encode() if isinstance(text, str) else bytes(text)
        self._check(self.L.s2r_load_patch(self.h, raw, len(raw)))

    def set_patch(self, patch):
        self._check(self.L.s2r_set_patch(self.h, C.byref(patch)))

    def set_patch_bank(self, patches):
        """1..256 patches; the current program picks the one a note_on gives its voice"""
        arr = (Patch * len(patches))(*patches)
        self._check(self.L.s2r_set_patch_bank(self.h, arr, len(patches)))

    @property
    def patch_bank_size(self):
        return self.L.s2r_patch_bank_size(self.h)

    def program_change(self, program):
        self._check(self.L.s2r_program_change(self.h, int(program)))

    def get_patch(self):
        p = Patch()
        self._check(self.L.s2r_get_patch(self.h, C.byref(p)))
        return p

    # --- events (synth.rs:61-80) ---
    def note_on(self, note, velocity=Velocity(1.0)):
        idx = C.c_uint32()
        self._check(self.L.s2r_note_on_ex(self.h, int(note), float(velocity), C.byref(idx)))
        return idx.value

    def note_off(self, note):
        self._check(self.L.s2r_note_off(self.h, int(note)))

    def note_events(self, events):
        """A batch of events (structured array NOTE_EVENT_DTYPE: kind 1=on 0=off, note, velocity),
        applied in order in one call — s2_bin's apply_all_midi_messages (main.rs:170-187)."""
        ev = np.ascontiguousarray(events, dtype=NOTE_EVENT_DTYPE)
        self._check(self.L.s2r_note_events(self.h, ev.ctypes.data, ev.size))

    # --- Synth::sample(&mut [f32], SampleRateKhz) (synth.rs:154-169) ---
    def sample(self, buffer, sample_rate=SampleRateKhz(48000)):
        """Overwrites ``buffer`` (1-D float32, C-contiguous) and returns it."""
        if isinstance(buffer, int):
            buffer = np.empty(buffer, dtype=np.float32)
        assert buffer.dtype == np.float32 and buffer.flags["C_CONTIGUOUS"] and buffer.ndim == 1
        # (the array's address without building a ctypes view of it: a microsecond per call that a 16-frame fill notices)
        rc = self._fill_raw(self.h, buffer.__array_interface__["data"][0], buffer.size, int(sample_rate))
        if rc:
            self._check(rc)
        return buffer

    def sample_begin(self, frames, sample_rate=SampleRateKhz(48000)):
        """first half of sample(): queue the fill (at most two in flight, like s2_bin's two buffers)"""
        self._check(self.L.s2r_fill_begin(self.h, int(frames), int(sample_rate)))

    def sample_end(self, buffer):
        """second half: wait for the oldest fill in flight and copy it into ``buffer``"""
        assert buffer.dtype == np.float32 and buffer.flags["C_CONTIGUOUS"] and buffer.ndim == 1
        self._check(self.L.s2r_fill_end(self.h, buffer.ctypes.data_as(_f32p), buffer.size))
        return buffer

    @property
    def pending_frames(self):
        """frames of the oldest fill begun and not yet ended (0: none in flight)"""
        return int(self.L.s2r_fill_pending_frames(self.h))

    def sample_oversampled(self, frames, sample_rate=SampleRateKhz(48000)):
        """build-defined 4x oversampling: rendered at 4 * sample_rate, decimated to `frames` samples"""
        out = np.empty(frames, dtype=np.float32)
        self._check(self.L.s2r_fill_oversampled(self.h, out.ctypes.data_as(_f32p), frames, int(sample_rate)))
        return out

    def sample_stereo(self, frames, sample_rate=SampleRateKhz(48000)):
        out = np.empty(2 * frames, dtype=np.float32)
        self._check(self.L.s2r_fill_stereo(self.h, out.ctypes.data_as(_f32p), frames, int(sample_rate)))
        return out.reshape(frames, 2)

    def render_voices(self, frames, sample_rate=SampleRateKhz(48000)):
        """Mix disabled: (shard_voices, frames) float32."""
        out = np.empty((self.shard_voices, frames), dtype=np.float32)
        self._check(self.L.s2r_render_voices(self.h, out.ctypes.data_as(_f32p), frames, int(sample_rate)))
        return out

    def fill_device(self, dev_ptr, frames, sample_rate=SampleRateKhz(48000), stream=None):
        """Partial mix of this shard into device memory at ``dev_ptr`` on ``stream`` (async)."""
        self._check(self.L.s2r_fill_device(self.h, C.c_void_p(dev_ptr), frames, int(sample_rate),
                                           C.c_void_p(stream) if stream else None))

    def fill_device_root(self, dev_ptr, frames, sample_rate=SampleRateKhz(48000), stream=None):
        """Final (root-added) mix of a single-shard synth into device memory (async)."""
        self._check(self.L.s2r_fill_device_root(self.h, C.c_void_p(dev_ptr), frames, int(sample_rate),
                                                C.c_void_p(stream) if stream else None))

    def process_layers(self, layers, frames, sample_rate=SampleRateKhz(48000)):
        """process::process_layer_buf_simd (process.rs:14-49) for layers the caller keeps: `layers` (LAYER_CALL_DTYPE) is
        updated in place (the st::Layer fields); returns (n_layers, frames) float32.  The handle is the workspace."""
        assert layers.dtype == LAYER_CALL_DTYPE and layers.flags["C_CONTIGUOUS"]
        out = np.empty((layers.size, frames), dtype=np.float32)
        self._check(self.L.s2r_process_layers(self.h, layers.ctypes.data, layers.size, out.ctypes.data_as(_f32p), frames, int(sample_rate)))
        return out

    def set_voice_log(self, fn):
        """synth.rs:118's log::debug! as a callback fn(voice_index, note) per note_on; None switches it off"""
        self._voice_log = VOICE_LOG_FN((lambda user, i, note: fn(int(i), int(note)))) if fn else VOICE_LOG_FN()
        self._check(self.L.s2r_set_voice_log(self.h, self._voice_log, None))

    # --- state ---
    def export_state(self):
        arr = np.zeros(self.shard_voices, dtype=VOICE_STATE_DTYPE)
        self._check(self.L.s2r_export_state(self.h, arr.ctypes.data))
        return arr

    def import_state(self, arr):
        arr = np.ascontiguousarray(arr, dtype=VOICE_STATE_DTYPE)
        assert arr.size == self.shard_voices
        self._check(self.L.s2r_import_state(self.h, arr.ctypes.data))

    def set_noise_seed(self, voice_index, seed):
        self._check(self.L.s2r_set_noise_seed(self.h, voice_index, seed))

    def set_flat_shortcut(self, enabled=True):
        self._check(self.L.s2r_set_flat_shortcut(self.h, 1 if enabled else 0))

    def set_coeff_stream(self, enabled=True):
        """False/0: filter coefficients computed in-lane; True/1: read from the patch's coefficient tables, few untimed
        events ride in the render kernel's arguments; 2: tables, events always through their own launch; 3 / 4: synonyms
        of 1 / 2"""
        self._check(self.L.s2r_set_coeff_stream(self.h, int(enabled)))

    def set_low_latency(self, enabled=True):
        """a resident render kernel between sample() calls (small pools, s2r.h: s2r_set_low_latency): the reference's own
        16-frames-per-call pattern without a launch per call"""
        self._check(self.L.s2r_set_low_latency(self.h, 1 if enabled else 0))

    @property
    def low_latency_active(self):
        return bool(self.L.s2r_low_latency_active(self.h))

    def set_resident(self, enabled=True):
        """keep the shard's whole render grid on the device between fills (s2r.h: s2r_set_resident): a fill is a posted
        command, not a launch; with two fills in flight the workgroups run ahead of each other"""
        self._check(self.L.s2r_set_resident(self.h, 1 if enabled else 0))

    @property
    def resident_active(self):
        return bool(self.L.s2r_resident_active(self.h))

    def quiesce(self):
        """stop any resident kernel of the handle and wait for it (before a device-wide synchronize)"""
        self._check(self.L.s2r_quiesce(self.h))

    def exchange_create(self, n_ranks):
        """rank 0 of a group of processes (one per GPU): the rows' block; returns the 64 handle bytes for the other ranks"""
        buf = C.create_string_buffer(64)
        self._check(self.L.s2r_exchange_create(self.h, n_ranks, buf, 64))
        return buf.raw

    def exchange_attach(self, rank, n_ranks, handle):
        buf = C.create_string_buffer(bytes(handle), 64)
        self._check(self.L.s2r_exchange_attach(self.h, rank, n_ranks, buf, 64))

    def set_timing(self, enabled=True):
        self._check(self.L.s2r_set_timing(self.h, 1 if enabled else 0))

    def last_render_ms(self):
        return float(self.L.s2r_last_render_ms(self.h))


def sum_partials_device(rows_ptr, n_rows, frames, out_ptr, stream=None):
    """out = (+0.0 + row0) + row1 + ... on device, rank order (multi-GPU root combine)."""
    rc = load_library().s2r_sum_partials_device(C.c_void_p(rows_ptr), n_rows, frames, C.c_void_p(out_ptr),
                                                C.c_void_p(stream) if stream else None)
    if rc != S2R_OK:
        raise S2rError(rc, "s2r_sum_partials_device")
