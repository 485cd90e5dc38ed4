"""development aid: why does the SAME settled chunk cost 941 cycles in one fill and 1 113-1 156 in the next two
(profiles/r03/stamps_c3.txt)?  The fills tools/stamps.py looks at differ in what the HOST did before them: the first follows 72
fills back to back, the others follow a read-back and a page of printing.  This tool renders the same population after pauses of
several lengths and prints, per pause, the median wave's cycles (s_memtime) and the kernel's wall time (HIP events).
    S2R_STAMPS=1 python tools/stamps_swing.py"""
import ctypes as C
import os
import sys
import time
os.environ["S2R_STAMPS"] = "1"
os.environ["S2R_FUSED"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2
from bench import make_c3_events, FRAMES, SR, PERIOD

V = 65536
cyc = make_c3_events(V, PERIOD)
s = s2.Synth(V, max_frames=FRAMES)
L = s.L
L.s2r_debug_read_stamps.restype = C.c_uint32
L.s2r_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
buf = np.empty(FRAMES, dtype=np.float32)
k = 0
for _ in range(PERIOD + 8):
    s.note_events(cyc[k % PERIOD]); s.sample(buf, SR); k += 1
waves = L.s2r_debug_read_stamps(s.h, None, 0)
assert waves
st = np.zeros((waves, 16), dtype=np.uint64)
s.set_timing(True)
quiet = np.zeros(0, dtype=s2.NOTE_EVENT_DTYPE)
print("pause before the fill | median wave cycles (s_memtime) | cycles per settled chunk (median wave's run) | kernel ms (HIP events)")
for pause_ms in (0.0, 0.0, 0.1, 0.5, 2.0, 10.0, 50.0, 0.0, 0.0):
    for _ in range(6):                                   # six fills back to back, no events: the same settled population every time
        s.note_events(quiet); s.sample(buf, SR)
    if pause_ms:
        time.sleep(pause_ms * 1e-3)
    s.note_events(quiet); s.sample(buf, SR)
    ms = s.last_render_ms()
    L.s2r_debug_read_stamps(s.h, st.ctypes.data, waves)
    t = st.astype(np.int64)
    total = t[:, 15] - t[:, 0]
    x = st[:, 10]
    runchunks = (x >> np.uint64(48)).astype(np.int64); t_run = (x & np.uint64(0xffffff)).astype(np.int64)
    per = t_run / np.maximum(1, runchunks)
    print("%8.1f ms            | %9d                      | %9.0f                                    | %.4f" % (pause_ms, np.median(total), np.median(per), ms))
