#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/, test infrastructure).

The reference itself cannot be run here (Rust nightly, no toolchain in the image — DESIGN.md 3), so
these vectors are NOT reference outputs: they freeze what the oracle — pinned to the reference's own
unit tests, known answers and the host libm the reference would call — produced on the build host
(glibc 2.35, x86-64).  They guard against drift: a different host libm, a compiler that contracts
an expression, an accidental edit of the restatement.  tests/test_golden.py checks the oracle
against them on the CPU and the HIP path against them on the GPU (no oracle in that loop).

    python tests/golden/make_golden.py        # rewrites the .npz files
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import s2o          # noqa: E402

SR = 48000


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a, dtype=np.float32).tobytes()) & 0xffffffff


def cases():
    """name -> (config kwargs, note script).  A script is a list of (frame, 'on'|'off', note); buffers of 1024
    frames except where a 'split' entry forces a shorter fill (tail path)."""
    out = {}
    # BASELINE config [0] / SURVEY 8d C1: note 69 on at frame 0, off at frame 24000, 40 buffers of 1024
    out["c1_default_patch_note69"] = (dict(), 8, [(0, "on", 69), (24000, "off", 69)],
                                      [1024] * 23 + [448, 576] + [1024] * 16)
    out["saw_fm_noise"] = (dict(osc_kind=1, mod_env_to_osc_freq=2.5, noise=0.25, osc_gain=0.75), 8,
                           [(0, "on", 45), (0, "on", 52), (0, "on", 57), (2048, "off", 52)], [1024] * 8)
    out["square_lp2"] = (dict(osc_kind=0, lpf_kind=3, lpf_freq=900.0, mod_env_to_lpf_freq=3.0, lpf_damping=0.6), 8,
                         [(0, "on", 40), (0, "on", 64), (1024, "off", 40)], [1024] * 6)
    out["triangle_bp2"] = (dict(osc_kind=2, lpf_kind=5, lpf_freq=1200.0, mod_env_to_lpf_freq=2.0, lpf_q=1.5), 8,
                           [(0, "on", 33), (0, "on", 70)], [1024] * 4 + [777])
    out["sine_hp1_tail"] = (dict(osc_kind=3, lpf_kind=2, lpf_freq=500.0, noise=0.1), 8,
                            [(0, "on", 60), (0, "on", 61)], [1000, 1000, 17, 1])
    out["saw_svf_lp_192k"] = (dict(osc_kind=1, lpf_kind=6, lpf_freq=2500.0, mod_env_to_lpf_freq=3.0, lpf_q=2.0), 8,
                              [(0, "on", 48), (0, "on", 55)], [1024] * 4)
    return out


def render(name, cfg_kw, voices, script, fills):
    sr = 192000 if name.endswith("192k") else SR
    syn = s2o.OracleSynth(voices)
    cfg = s2o.lib().s2o_default_config()
    for k, v in cfg_kw.items():
        setattr(cfg, k, v)
    syn.config = cfg
    pos = 0
    rows = []
    pending = sorted(script)
    for n in fills:
        while pending and pending[0][0] <= pos:
            _, kind, note = pending.pop(0)
            (syn.note_on if kind == "on" else syn.note_off)(note)
        rows.append(syn.render_voices(n, sr))
        pos += n
    return np.concatenate(rows, axis=1), sr


def main():
    for name, (cfg_kw, voices, script, fills) in cases().items():
        pv, sr = render(name, cfg_kw, voices, script, fills)
        mix = s2o.mix_sequential(pv)            # <= 16 voices: the reference's own order == the GPU tree
        np.savez_compressed(os.path.join(HERE, name + ".npz"),
                            head=pv[:, :256], mix_head=mix[:256], mix_tail=mix[-256:],
                            per_voice_crc=np.array([crc(r) for r in pv], dtype=np.uint32),
                            mix_crc=np.uint32(crc(mix)), frames=np.uint32(pv.shape[1]), sample_rate=np.uint32(sr))
        print("%-28s %d voices x %d frames  mix crc %08x" % (name, voices, pv.shape[1], crc(mix)))


if __name__ == "__main__":
    main()
