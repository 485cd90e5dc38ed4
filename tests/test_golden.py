"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU: the oracle must still produce them (guards the restatement and the host libm it calls).
GPU: the HIP path must produce them too — this check has no oracle in the loop at run time."""
import os
import sys
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg          # noqa: E402

NAMES = sorted(mg.cases().keys())


def _crc(a):
    return zlib.crc32(np.ascontiguousarray(a, dtype=np.float32).tobytes()) & 0xffffffff


def _check(name, pv, mix):
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    assert pv.shape[1] == int(g["frames"])
    assert np.array_equal(pv[:, :256].view(np.uint32), g["head"].view(np.uint32)), "first 256 frames per voice"
    assert np.array_equal(mix[:256].view(np.uint32), g["mix_head"].view(np.uint32))
    assert np.array_equal(mix[-256:].view(np.uint32), g["mix_tail"].view(np.uint32))
    assert [_crc(r) for r in pv] == [int(c) for c in g["per_voice_crc"]], "CRC-32 of every voice's whole output"
    assert _crc(mix) == int(g["mix_crc"])


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name):
    from oracle import s2o
    cfg_kw, voices, script, fills = mg.cases()[name]
    pv, sr = mg.render(name, cfg_kw, voices, script, fills)
    _check(name, pv, s2o.mix_sequential(pv))


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_reproduces_golden(name):
    import synth2_amd as s2
    cfg_kw, voices, script, fills = mg.cases()[name]
    sr = 192000 if name.endswith("192k") else 48000
    patch = s2.default_patch()
    for k, v in cfg_kw.items():
        setattr(patch, k, v)
    a = s2.Synth(voices, max_frames=1024)        # per-voice rows
    b = s2.Synth(voices, max_frames=1024)        # the mix (8 voices: tree == the reference's sequential order)
    a.set_patch(patch); b.set_patch(patch)
    pos, pending, rows, mixes = 0, sorted(script), [], []
    for n in fills:
        while pending and pending[0][0] <= pos:
            _, kind, note = pending.pop(0)
            for s in (a, b):
                (s.note_on(s2.Note(note)) if kind == "on" else s.note_off(s2.Note(note)))
        rows.append(a.render_voices(n, sr))
        mixes.append(b.sample(np.empty(n, dtype=np.float32), sr).copy())
        pos += n
    _check(name, np.concatenate(rows, axis=1), np.concatenate(mixes))
