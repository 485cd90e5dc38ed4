"""A/B of two builds of libs2r.so on the bench workload: S2R_AB_LIB=<path> selects the library (default: the
in-tree build).  Prints the render kernel's time settled (no events) and under the bench's note churn, and the
wall time per buffer of the synchronous API under churn."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd.build as _b
if os.environ.get("S2R_AB_LIB"):
    _b.LIB = os.environ["S2R_AB_LIB"]
    _b.needs_build = lambda: False
import synth2_amd as s2
from bench import make_events
V = int(os.environ.get("V", 65536))
s = s2.Synth(V, max_frames=1024)
ev = np.zeros(V, dtype=s2.NOTE_EVENT_DTYPE); ev["kind"] = 1; ev["note"] = 36 + np.arange(V) % 61; ev["velocity"] = 1.0
s.note_events(ev)
buf = np.empty(1024, dtype=np.float32)
s.set_timing(True)
ts = []
for k in range(24):
    s.sample(buf); ts.append(s.last_render_ms())
print("settled: first %.4f  mean[12:] %.4f ms" % (ts[0], np.mean(ts[12:])))
ts = []
for k in range(80):
    s.note_events(make_events(V, 128, k))
    s.sample(buf); ts.append(s.last_render_ms())
print("churn:   mean[40:] %.4f  min %.4f  max %.4f ms" % (np.mean(ts[40:]), np.min(ts[40:]), np.max(ts[40:])))
s.set_timing(False)
t0 = time.perf_counter()
for k in range(80, 208):
    s.note_events(make_events(V, 128, k))
    s.sample(buf)
print("churn wall per buffer (sync API): %.4f ms" % ((time.perf_counter() - t0) / 128 * 1e3))
