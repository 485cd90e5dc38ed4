"""Render-kernel time under the bench's note churn (128 note-ons + 128 note-offs per buffer per 64k voices)
for a few patch shapes: where a moving mod envelope on a handful of waves sets the launch time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2
from bench import make_events
V = 65536
cases = [("default patch", {}), ("oscillator FM 2.5", dict(mod_env_to_osc_freq=2.5)), ("sine", dict(osc_kind=s2.OSC_SINE)),
         ("lp2 filter", dict(lpf_kind=s2.FILT_LP2)), ("svf_lp filter", dict(lpf_kind=s2.FILT_SVF_LP)), ("bp2 + FM", dict(lpf_kind=s2.FILT_BP2, mod_env_to_osc_freq=1.5)),
         ("no LPF modulation", dict(mod_env_to_lpf_freq=0.0))]
for name, kw in cases:
    s = s2.Synth(V, max_frames=1024)
    p = s2.default_patch()
    for k, v in kw.items(): setattr(p, k, v)
    s.set_patch(p)
    ev = np.zeros(V, dtype=s2.NOTE_EVENT_DTYPE); ev["kind"] = 1; ev["note"] = 36 + np.arange(V) % 61; ev["velocity"] = 1.0
    s.note_events(ev)
    s.set_timing(True)
    buf = np.empty(1024, dtype=np.float32)
    ts = []
    for k in range(50):
        s.note_events(make_events(V, 128, k))
        s.sample(buf); ts.append(s.last_render_ms())
    print("%-20s render kernel %.3f ms per buffer in steady churn (first buffer %.3f)" % (name, np.mean(ts[25:]), ts[0]))
