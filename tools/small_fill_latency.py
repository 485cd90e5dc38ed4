"""Wall time per s2r_fill for the reference's own call pattern: 8 voices, 16 frames per call
(s2_bin, main.rs:138-143), with and without the coefficient stream."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2
for voices in (8, 1024):
    for mode in (1, 0):
        s = s2.Synth(voices, max_frames=2048)
        s.set_coeff_stream(mode)
        for n in (57, 64, 69): s.note_on(n)
        buf = np.empty(16, dtype=np.float32)
        for _ in range(200): s.sample(buf)
        t = time.perf_counter()
        for _ in range(2000): s.sample(buf)
        dt = (time.perf_counter() - t) / 2000
        print("%5d voices, 16-frame fills, coefficient stream %d: %.1f us per call (real time needs < %.0f us)" % (voices, mode, dt * 1e6, 16 / 48000 * 1e6))
