"""Times the settled one-pole render kernel for each oscillator kind (65536 voices x 1024 frames)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2
V = 65536
for osc, name in ((s2.OSC_SAW, "saw"), (s2.OSC_SQUARE, "square"), (s2.OSC_TRIANGLE, "triangle"), (s2.OSC_SINE, "sine")):
    s = s2.Synth(V, max_frames=1024)
    p = s2.default_patch(); p.osc_kind = osc
    s.set_patch(p)
    ev = np.zeros(V, dtype=s2.NOTE_EVENT_DTYPE); ev["kind"] = 1; ev["note"] = (np.arange(V) * 13) % 100 + 20; ev["velocity"] = 1.0
    s.note_events(ev)
    s.set_timing(True)
    buf = np.empty(1024, dtype=np.float32)
    ts = []
    for k in range(24):
        s.sample(buf); ts.append(s.last_render_ms())
    print("%-9s first %.3f ms  settled %.3f ms  (%.3g voice-samples/s)" % (name, ts[0], np.mean(ts[12:]), V * 1024 / (np.mean(ts[12:]) * 1e-3)))
