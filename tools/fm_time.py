"""Times the render kernel for an oscillator-FM patch (mod_env_to_osc_freq != 0): while the mod envelope
moves (first buffers) and once it is flat."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2
V = 65536
for fm in (0.0, 2.5):
    s = s2.Synth(V, max_frames=1024)
    p = s2.default_patch(); p.mod_env_to_osc_freq = fm
    s.set_patch(p)
    ev = np.zeros(V, dtype=s2.NOTE_EVENT_DTYPE); ev["kind"] = 1; ev["note"] = (np.arange(V) * 13) % 100 + 20; ev["velocity"] = 1.0
    s.note_events(ev)
    s.set_timing(True)
    buf = np.empty(1024, dtype=np.float32)
    ts = []
    for k in range(24):
        s.sample(buf); ts.append(s.last_render_ms())
    print("mod_env_to_osc_freq %.1f: first %.3f ms  settled %.3f ms" % (fm, ts[0], np.mean(ts[12:])))
