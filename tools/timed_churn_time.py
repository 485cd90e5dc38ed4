"""The bench's note churn delivered as TIMED events (each stamped with a random 16-frame boundary inside the
buffer, the way a caller batching s2_bin's MIDI would) against the same churn applied at buffer starts."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth2_amd as s2
from bench import make_events
V = 65536
for timed in (False, True):
    s = s2.Synth(V, max_frames=1024)
    ev = np.zeros(V, dtype=s2.NOTE_EVENT_DTYPE); ev["kind"] = 1; ev["note"] = 36 + np.arange(V) % 61; ev["velocity"] = 1.0
    s.note_events(ev)
    s.set_timing(True)
    rng = np.random.RandomState(1)
    out = torch.zeros(1024, device="cuda"); st = torch.cuda.current_stream().cuda_stream
    ts = []; t0 = None
    for k in range(80):
        e = make_events(V, 128, k)
        if timed:
            e["frame"] = np.sort(rng.randint(0, 64, len(e))) * 16
        if k == 30:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        s.note_events(e)
        s.fill_device_root(out.data_ptr(), 1024, 48000, st)
        if k < 30:
            torch.cuda.synchronize(); ts.append(s.last_render_ms())
    torch.cuda.synchronize()
    print("%s: %.1f us per buffer end to end (render kernel %.3f ms in the warm-up)" % ("timed events  " if timed else "events at start", (time.perf_counter() - t0) / 50 * 1e6, np.mean(ts[15:])))
