"""development aid: the GPU's own timeline of the C3 bench steps, without a profiler in the way of the host calls.
    S2R_STAMPS=1 python tools/gpu_timeline.py
The diagnostic library's render and mix kernels record their first entry and last exit on the GPU's 100 MHz clock
(s_memrealtime); this prints, per step, each launch's duration and the idle gap in front of it, next to the host's time in
the three calls."""
import ctypes as C
import os
import sys
import time
os.environ["S2R_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2
from bench import make_c3_events, FRAMES, SR, PERIOD

V = int(os.environ.get("V", 65536))
N = int(os.environ.get("N", 40))
cyc = make_c3_events(V, PERIOD)
s = s2.Synth(V, max_frames=FRAMES)
L = s.L
L.s2r_debug_timeline.restype = C.c_uint32
L.s2r_debug_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
buf = [np.empty(FRAMES, dtype=np.float32) for _ in range(2)]
for k in range(2 * PERIOD):
    s.note_events(cyc[k % PERIOD]); s.sample(buf[0], SR)
assert L.s2r_debug_timeline(s.h, None, 4 * N + 16), "this is not the diagnostic build"
k0 = 2 * PERIOD
s.note_events(cyc[k0 % PERIOD]); s.sample_begin(FRAMES, SR)
host = []
t0 = time.perf_counter()
for k in range(k0 + 1, k0 + 1 + N):
    a = time.perf_counter(); s.note_events(cyc[k % PERIOD])
    b = time.perf_counter(); s.sample_begin(FRAMES, SR)
    c = time.perf_counter(); s.sample_end(buf[k & 1])
    d = time.perf_counter(); host.append((b - a, c - b, d - c))
s.sample_end(buf[0])
wall = time.perf_counter() - t0
tl = np.zeros((4 * N + 16, 2), dtype=np.uint64)
n = L.s2r_debug_timeline(s.h, tl.ctypes.data, tl.shape[0])
tl = tl[:n].astype(np.int64)
order = np.argsort(tl[:, 0])
tl = tl[order]
dur = (tl[:, 1] - tl[:, 0]) / 100.0                     # us
gap = np.concatenate([[0.0], (tl[1:, 0] - tl[:-1, 1]) / 100.0])
print("%d steps in %.1f us each (host wall); host calls: note_events %.1f, fill_begin %.1f, fill_end %.1f us" % (
    N, wall / N * 1e6, *(1e6 * np.mean([h[i] for h in host]) for i in range(3))))
print("launches recorded: %d; GPU span per step %.1f us" % (n, (tl[-1, 1] - tl[0, 0]) / 100.0 / max(1, n // 2)))
for i in range(min(n, 24)):
    print("  launch %2d (slot %2d): gap %6.1f us, runs %6.1f us  %s" % (i, order[i], gap[i], dur[i], "render" if dur[i] > 20 else "mix+heads"))
big = dur > 20
print("median: render %.1f us, gap before render %.1f us; mix+heads %.1f us, gap before it %.1f us" % (
    np.median(dur[big]), np.median(gap[big][1:]), np.median(dur[~big]), np.median(gap[~big][1:])))
