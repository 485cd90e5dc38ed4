"""Times the render kernel under each filter kind (65536 voices, 1024 frames, settled state)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2

V = int(os.environ.get("V", 65536))
for kind in [int(k) for k in os.environ.get("KINDS", "0,1,2,3,4,5,6,8").split(",")]:
    s = s2.Synth(V, max_frames=1024)
    p = s2.default_patch(); p.lpf_kind = kind
    s.set_patch(p)
    ev = np.zeros(V, dtype=s2.NOTE_EVENT_DTYPE)
    ev["kind"] = 1; ev["note"] = (np.arange(V) * 13) % 100 + 20; ev["velocity"] = 1.0
    s.note_events(ev)
    s.set_timing(True)
    buf = np.empty(1024, dtype=np.float32)
    ts = []
    for k in range(24):
        s.sample(buf)
        ts.append(s.last_render_ms())
    print("kind %d: first %.3f ms  settled %.3f ms  (%.3g voice-samples/s)" % (kind, ts[0], np.mean(ts[12:]), V * 1024 / (np.mean(ts[12:]) * 1e-3)))
