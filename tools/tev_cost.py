"""development aid: what timed events cost the render kernel.  C3's periodic schedule (bench.py) with its note-offs
(a) on their own 16-frame boundaries, (b) all at frame 16, (c) all at the buffer's start (untimed), (d) only every
k-th one timed.  Prints the render kernel's mean time per 1024-frame launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2
from bench import make_c3_events, FRAMES, SR, PERIOD

V = int(os.environ.get("V", 65536))
base = make_c3_events(V, PERIOD)


def variant(kind):
    out = []
    for ev in base:
        e = ev.copy()
        timed = e["frame"] > 0
        if kind == "same16":
            e["frame"][timed] = 16
        elif kind == "start":
            e["frame"][:] = 0
        elif kind.startswith("every"):
            k = int(kind[5:])
            idx = np.nonzero(timed)[0]
            drop = idx[np.arange(idx.size) % k != 0]
            e["frame"][drop] = 0
            e = e[np.argsort(e["frame"], kind="stable")]
        out.append(e)
    return out


def run(kind):
    cyc = variant(kind)
    s = s2.Synth(V, max_frames=FRAMES)
    buf = np.empty(FRAMES, dtype=np.float32)
    for k in range(PERIOD + 4):
        s.note_events(cyc[k % PERIOD]); s.sample(buf, SR)
    s.set_timing(True)
    ts = []
    for k in range(PERIOD + 4, 2 * PERIOD + 4):
        s.note_events(cyc[k % PERIOD]); s.sample(buf, SR); ts.append(s.last_render_ms())
    ts = np.array(ts)
    print("%-8s kernel ms: mean %.4f  min %.4f  max %.4f   by phase of the period (8 buffers each): %s" % (
        kind, ts.mean(), ts.min(), ts.max(), " ".join("%.3f" % ts[i:i + 8].mean() for i in range(0, PERIOD, 8))))


for kind in (sys.argv[1:] or ["c3", "same16", "start", "every4", "every16", "every64"]):
    run(kind)
