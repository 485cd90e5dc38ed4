"""Shared drivers for the parity tests: the same note-event stream is fed to the CPU oracle
(oracle/s2o.py, test infrastructure) and to the HIP path (synth2_amd.Synth over the C ABI)."""
import ctypes as C

import numpy as np

from oracle import s2o
import synth2_amd as s2


def oracle_cfg_from_patch(p):
    """synth2_amd.Patch -> oracle LayerCfg (same field order, separate struct types)."""
    c = s2o.LayerCfg()
    c.osc_kind = p.osc_kind
    c.osc_gain = p.osc_gain
    c.noise = p.noise
    c.lpf_freq = p.lpf_freq
    for name in ("amp_env", "mod_env"):
        src, dst = getattr(p, name), getattr(c, name)
        dst.attack_ms, dst.decay_ms, dst.sustain, dst.release_ms = src.attack_ms, src.decay_ms, src.sustain, src.release_ms
    c.mod_env_to_osc_freq = p.mod_env_to_osc_freq
    c.mod_env_to_lpf_freq = p.mod_env_to_lpf_freq
    c.lpf_kind = p.lpf_kind
    c.lpf_damping = p.lpf_damping
    c.lpf_q = p.lpf_q
    return c


def make_patch(**kw):
    p = s2.default_patch()
    for k, v in kw.items():
        if "." in k:
            a, b = k.split(".")
            setattr(getattr(p, a), b, v)
        else:
            setattr(p, k, v)
    return p


def lcg(seed):
    """the C2/C3 generator of SURVEY.md §8d"""
    return (1103515245 * seed + 12345) % (1 << 31)


class Pair:
    """An oracle synth and a GPU synth driven in lockstep."""

    def __init__(self, num_voices, patch=None, max_frames=2048, block_voices=0, mix_groups=0, seeds=None, flat_shortcut=True, devices=None,
                 shard_interleave=0):
        self.gpu = s2.Synth(num_voices, max_frames=max_frames, block_voices=block_voices, mix_groups=mix_groups, devices=devices,
                            shard_interleave=shard_interleave)
        self.cpu = s2o.OracleSynth(num_voices)
        if not flat_shortcut:
            self.gpu.set_flat_shortcut(False)
        self.block_voices = self.gpu.block_voices
        self.groups = mix_groups or 1
        self.seeds = seeds
        if patch is not None:
            self.gpu.set_patch(patch)
            self.cpu.config = oracle_cfg_from_patch(patch)
        if seeds is not None:
            for v, sd in enumerate(seeds):
                self.gpu.set_noise_seed(v, int(sd))

    def note_on(self, note, velocity=1.0):
        want = self.cpu.next_voice_index()
        self.cpu.note_on(note, velocity)
        got = self.gpu.note_on(note, velocity)
        assert got == want, "voice allocation differs: gpu %d oracle %d" % (got, want)
        if self.seeds is not None:          # reference resets seed to 0 on note_on; the variant keeps seeds
            self.cpu.set_seed(got, int(self.seeds[got]))
        return got

    def note_off(self, note):
        self.cpu.note_off(note)
        self.gpu.note_off(note)

    def set_bank(self, patches):
        self.gpu.set_patch_bank(patches)
        self.cpu.set_bank([oracle_cfg_from_patch(p) for p in patches])

    def program_change(self, program):
        self.gpu.program_change(program)
        self.cpu.program_change(program)

    def render_voices(self, frames, sr=48000):
        return self.gpu.render_voices(frames, sr), self.cpu.render_voices(frames, sr, threads=self.threads)

    threads = 1

    def note_events(self, events):
        """batch form; voice choices are cross-checked by the single-event tests"""
        self.gpu.note_events(events)
        for e in events:
            if e["kind"] == 1:
                self.cpu.note_on(int(e["note"]), float(e["velocity"]))
            else:
                self.cpu.note_off(int(e["note"]))

    def sample(self, frames, sr=48000):
        """GPU mono mix and the oracle's mix through the same tree"""
        g = self.gpu.sample(np.empty(frames, dtype=np.float32), sr)
        pv = self.cpu.render_voices(frames, sr, threads=self.threads)
        return g, s2o.mix_tree(pv, self.block_voices, self.groups), pv


def assert_bits_equal(a, b, what=""):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    au, bu = a.view(np.uint32), b.view(np.uint32)
    # a NaN produced by an invalid operation (inf - inf in a filter driven unstable) has an
    # implementation-defined sign/payload (x86 sets the sign bit, the GPU does not): NaN == NaN here
    bad = np.nonzero((au != bu) & ~(np.isnan(a) & np.isnan(b)))
    if bad[0].size:
        idx = tuple(x[0] for x in bad)
        raise AssertionError("%s: %d of %d values differ bitwise; first at %s: %r vs %r" % (
            what, bad[0].size, a.size, idx, a[idx], b[idx]))


def ulp_diff(a, b):
    """max distance in units of float32 ULP (monotone integer mapping)"""
    def key(x):
        u = np.ascontiguousarray(x, dtype=np.float32).view(np.int32).astype(np.int64)
        return np.where(u < 0, -(u & 0x7fffffff), u)
    return int(np.max(np.abs(key(a) - key(b)))) if np.size(a) else 0
