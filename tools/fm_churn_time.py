"""Render-kernel time under the bench's note churn for a patch WITH oscillator FM against the default patch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2
from bench import make_events
V = 65536
for fm in (0.0, 2.5):
    s = s2.Synth(V, max_frames=1024)
    p = s2.default_patch(); p.mod_env_to_osc_freq = fm
    s.set_patch(p)
    ev = np.zeros(V, dtype=s2.NOTE_EVENT_DTYPE); ev["kind"] = 1; ev["note"] = 36 + np.arange(V) % 61; ev["velocity"] = 1.0
    s.note_events(ev)
    s.set_timing(True)
    buf = np.empty(1024, dtype=np.float32)
    ts = []
    for k in range(60):
        s.note_events(make_events(V, 128, k))
        s.sample(buf); ts.append(s.last_render_ms())
    print("mod_env_to_osc_freq %.1f: render kernel %.3f ms per buffer with 128 note-ons + 128 note-offs per buffer (steady state)" % (fm, np.mean(ts[30:])))
