#!/usr/bin/env python3
"""How far the LPF-cutoff chain — the only stage of the path whose arithmetic IEEE 754 does not pin — is from the TRUTH.

The reference computes, per frame (default patch, synth.rs:125-152):
    f_lpf = sleef_pow(2, mod * 10) * 200                      process.rs:148-152,231-250   (sleef 0.3.2, source not in the tree)
    x     = expf(((-2 * PI_f32) * f_lpf) / sr)                filters.rs:20-21             (the host libm)
    y     = fma(1 - x, in, x * y_prev)                        filters.rs:23-33
The oracle (and, bit for bit, the GPU — tests/test_gpu_parity.py) evaluates these with its restatement of SLEEF's xpowf
and glibc's expf.  Neither is the Rust crate the reference links, so "within 1 ULP of the reference" is argued through the
truth: this script evaluates the same formulas on the same f32 inputs in binary64 with libm's double pow / exp (error
< 1e-16 relative: exact for the purpose) and reports the distance of the oracle's f32 results from it in f32 ULPs.  Any
implementation whose pow and exp are each within e ULP of the truth is within (this + e) ULP of the oracle, stage by stage.

    python tools/truth_report.py [out.json]        (tests/test_truth_model.py asserts the bounds this prints)
"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import s2o          # noqa: E402   (test infrastructure)

SR = 48000
PI32 = np.float32(3.14159274101257324)


def ulps(approx32, truth64):
    """|approx - truth| in units of the f32 ulp at the truth (fractional)"""
    t = np.asarray(truth64, dtype=np.float64)
    a = np.asarray(approx32, dtype=np.float32).astype(np.float64)
    with np.errstate(divide="ignore"):
        e = np.floor(np.log2(np.maximum(np.abs(t), 2.0 ** -126)))
    ulp = 2.0 ** (e - 23)
    return np.abs(a - t) / ulp


def stats(u):
    u = np.asarray(u, dtype=np.float64).ravel()
    return {"max": float(u.max()), "p99": float(np.percentile(u, 99)), "p50": float(np.percentile(u, 50)), "mean": float(u.mean())}


def report(notes=range(36, 97)):
    L = s2o.lib()
    libm = ctypes.CDLL("libm.so.6")
    libm.expf.restype = ctypes.c_float
    libm.expf.argtypes = [ctypes.c_float]
    syn0 = s2o.OracleSynth(1)                       # (kept alive: `config` is a view into it)
    cfg = s2o.LayerCfg.from_buffer_copy(syn0.config)
    A = L.s2o_ms_as_samples(cfg.mod_env.attack_ms, SR)
    D = L.s2o_ms_as_samples(cfg.mod_env.decay_ms, SR)
    n = int(np.ceil(A + D)) + 16                      # the whole decay of the mod envelope, a little of the sustain
    n = (n + 15) // 16 * 16
    t = np.arange(n, dtype=np.uint32)

    def env(e):
        a, d, rl = (L.s2o_ms_as_samples(ms, SR) for ms in (e.attack_ms, e.decay_ms, e.release_ms))
        return np.concatenate([s2o.adsr_x16(a, d, e.sustain, rl, t[c:c + 16]) for c in range(0, n, 16)]).astype(np.float32)

    mod = env(cfg.mod_env)
    amount, freq = np.float32(cfg.mod_env_to_lpf_freq), np.float32(cfg.lpf_freq)
    # ---- stage 1: f_lpf ----
    f_or = np.empty(n, dtype=np.float32)
    for c in range(0, n, 16):
        out = np.empty(16, dtype=np.float32)
        L.s2o_modulate_freq_unipolar_x16(float(freq), mod[c:c + 16].ctypes.data_as(ctypes.POINTER(ctypes.c_float)), float(amount),
                                         out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        f_or[c:c + 16] = out
    m32 = (mod * amount).astype(np.float32)           # the reference rounds mod * amount to f32 first (process.rs:243)
    f_tr = np.exp2(m32.astype(np.float64)) * float(freq)
    u_f = ulps(f_or, f_tr)
    # ---- stage 2: x = expf(arg), arg from the ORACLE's f_lpf (the stage alone) and from the truth (the chain) ----
    arg32 = ((np.float32(-2.0) * PI32) * f_or / np.float32(SR)).astype(np.float32)
    x_or = np.array([libm.expf(float(a)) for a in arg32], dtype=np.float32)      # the call the oracle makes
    x_tr_stage = np.exp(arg32.astype(np.float64))
    x_tr_chain = np.exp((-2.0 * float(PI32)) * f_tr / SR)
    u_x_stage, u_x_chain = ulps(x_or, x_tr_stage), ulps(x_or, x_tr_chain)
    # ---- stage 3: the filter's output over the decay, per note: oracle voice vs an f64 recurrence on the same input ----
    u_y, worst_abs = [], 0.0
    for note in notes:
        syn = s2o.OracleSynth(1)
        syn.note_on(note)
        y_or = syn.render_voices(n, SR)[0]                       # includes the amp envelope: out = y * amp
        amp = env(cfg.amp_env)
        # the filter input, reconstructed from the oracle's own stages: saw + gain + noise (all pure IEEE, bit-exact)
        st = s2o.LayerState()
        c0 = s2o.LayerCfg.from_buffer_copy(cfg)
        c0.lpf_freq = 1e9                                        # x -> 0: the "filter" passes its input through (a0 = 1)
        c0.mod_env_to_lpf_freq = 0.0
        buf = np.empty(n, dtype=np.float32)
        # amplitude 1: sustain 1, no attack / decay
        c0.amp_env.attack_ms = 0.0; c0.amp_env.decay_ms = 0.0; c0.amp_env.sustain = 1.0
        rc = L.s2o_process_layer_buf_simd(ctypes.byref(c0), ctypes.byref(st), L.s2o_note_to_pitch(note), SR, 0, 0, 0,
                                          buf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n)
        assert rc == 0
        xin = buf.astype(np.float64)
        y = 0.0
        y_tr = np.empty(n)
        for i in range(n):
            y = (1.0 - x_tr_chain[i]) * xin[i] + x_tr_chain[i] * y
            y_tr[i] = y
        out_tr = y_tr * amp.astype(np.float64)
        # relative to the ulp of the signal's running peak (the filter output passes through zero)
        peak = np.maximum.accumulate(np.maximum(np.abs(out_tr), 1e-30))
        e = np.abs(y_or.astype(np.float64) - out_tr) / (2.0 ** (np.floor(np.log2(peak)) - 23))
        u_y.append(e[1:])                                        # (frame 0 is exactly 0 on both sides: amp = 0)
        worst_abs = max(worst_abs, float(np.max(np.abs(y_or.astype(np.float64) - out_tr))))
    u_y = np.concatenate(u_y)
    return {
        "what": "distance of the oracle's (== the GPU's) f32 results from a binary64 evaluation of the same formulas on the same f32 inputs",
        "patch": "default (synth.rs:125-152): lpf 200 Hz, mod_env_to_lpf_freq 10, mod ADSR 0 / 200 ms / 0 / 0, 48 kHz",
        "frames": int(n), "notes": [int(min(notes)), int(max(notes))],
        "f_lpf_ulp": stats(u_f),
        "x_stage_ulp": stats(u_x_stage),
        "x_chain_ulp": stats(u_x_chain),
        "lpf_output_ulp_of_running_peak": stats(u_y),
        "lpf_output_max_abs_error": worst_abs,
    }


if __name__ == "__main__":
    r = report()
    text = json.dumps(r, indent=1)
    print(text)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text + "\n")
