"""development aid: where a 16-frame fill of a small pool spends its time inside the render kernel's fill loop — launched
per call, or run by the resident kernel (s2r_set_low_latency).  Needs the diagnostic library:
    S2R_STAMPS=1 python tools/stamps_small.py [voices]"""
import ctypes as C
import os
import sys
os.environ["S2R_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2

V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for lowlat in (False, True):
    s = s2.Synth(V, max_frames=2048)
    L = s.L
    L.s2r_debug_read_stamps.restype = C.c_uint32
    L.s2r_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    waves = L.s2r_debug_read_stamps(s.h, None, 0)          # arm (before the resident kernel starts: it is an argument)
    assert waves, "this is not the diagnostic build"
    s.set_low_latency(lowlat)
    for n in (57, 64, 69): s.note_on(n)
    buf = np.empty(16, dtype=np.float32)
    st = np.zeros((waves, 16), dtype=np.uint64)
    rows = []
    for rep in range(6):
        for _ in range(50): s.sample(buf)
        L.s2r_debug_read_stamps(s.h, st.ctypes.data, waves)     # (stops the resident kernel; the next fill starts it again)
        t = st.astype(np.int64)[0]
        rows.append((t[1] - t[0], t[2] - t[1], t[3] - t[2], t[15] - t[3], t[15] - t[0]))
    r = np.median(np.array(rows), axis=0)
    print("%d voices, 16 frames, %s: cycles inside the fill: prologue %d, chunk work %d, barrier + combine %d, write-back + completion word %d; total %d (%.2f us at 2.4 GHz)" % (
        V, "resident kernel" if lowlat else "launch per call", r[0], r[1], r[2], r[3], r[4], r[4] / 2400.0))
