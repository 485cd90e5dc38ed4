"""Wall time per s2r_fill for the reference's own call pattern: 8 voices, 16 frames per call
(s2_bin, main.rs:138-143) — a launch per call, and through the resident kernel (s2r_set_low_latency)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2
for voices in (8, 256, 1024):
    for lowlat in (False, True):
        s = s2.Synth(voices, max_frames=2048)
        s.set_low_latency(lowlat)
        for n in (57, 64, 69): s.note_on(n)
        buf = np.empty(16, dtype=np.float32)
        for _ in range(200): s.sample(buf)
        ts = []
        for rep in range(5):
            t = time.perf_counter()
            for _ in range(2000): s.sample(buf)
            ts.append((time.perf_counter() - t) / 2000)
        # ... and with a note event before every fourth call
        t = time.perf_counter()
        for k in range(2000):
            if k % 4 == 0: (s.note_on if k % 8 == 0 else s.note_off)(60 + (k // 8) % 12)
            s.sample(buf)
        dte = (time.perf_counter() - t) / 2000
        print("%5d voices, 16-frame fills, %-16s %5.1f us per call (best of 5 x 2000: %5.1f; with note events %5.1f); resident kernel on the device: %s  (real time needs < %.0f us)" % (
            voices, "resident kernel:" if lowlat else "launch per call:", np.median(ts) * 1e6, min(ts) * 1e6, dte * 1e6, s.low_latency_active, 16 / 48000 * 1e6))
        del s
