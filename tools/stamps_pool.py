"""development aid: where a ONE-LAUNCH fill's time goes (in-kernel chain heads, render, ticket mix), wave by wave, a launch per
fill or through the pool-resident kernel.  Needs the diagnostic library:
    S2R_STAMPS=1 python tools/stamps_pool.py [fused|pool|two]
(slots: 0 entry, 7 heads built, 1 prologue done, 2 work done, 3 combine done, 4 ticket taken, 5 mixers: every row in, 6 mixed, 15 exit;
 9 the workgroup's place among the arrivals)"""
import ctypes as C
import os
import sys
os.environ["S2R_STAMPS"] = "1"
mode = sys.argv[1] if len(sys.argv) > 1 else "pool"
os.environ["S2R_FUSED"] = "1" if mode == "two" else "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2
from bench import make_c3_events, FRAMES, SR, PERIOD

V = int(os.environ.get("V", 65536))
cyc = make_c3_events(V, PERIOD)
s = s2.Synth(V, max_frames=FRAMES)
L = s.L
L.s2r_debug_read_stamps.restype = C.c_uint32
L.s2r_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
waves = L.s2r_debug_read_stamps(s.h, None, 0)          # arm (before the resident kernel is started: its arguments carry the pointer)
assert waves, "this is not the diagnostic build"
if mode == "pool":
    s.set_resident(True)
buf = np.empty(FRAMES, dtype=np.float32)
n_in = 0
def step(k):
    global n_in
    s.note_events(cyc[k % PERIOD]); s.sample_begin(FRAMES, SR); n_in += 1
    if n_in == 2:
        s.sample_end(buf); n_in -= 1
for k in range(2 * PERIOD + 8):
    step(k)
st = np.zeros((waves, 16), dtype=np.uint64)
for rep in range(3):
    for j in range(5):
        step(2 * PERIOD + 8 + 5 * rep + j)
    while n_in:
        s.sample_end(buf); n_in -= 1
    L.s2r_debug_read_stamps(s.h, st.ctypes.data, waves)       # (stops a resident kernel: the last fill's stamps)
    t = st.astype(np.int64)
    w0 = t[::4]                                               # wave 0 of every workgroup
    def col(a, b, x=t):
        return x[:, a] - x[:, b]
    print("%s, after fill %d: per wave, cycles of the shader clock" % (mode, 2 * PERIOD + 8 + 5 * rep + 4))
    for name, a, b in (("heads built", 7, 0), ("prologue (incl. heads)", 1, 0), ("work", 2, 1), ("combine", 3, 2), ("state out + ticket", 4, 3), ("total to ticket", 4, 0)):
        c = col(a, b)
        c = c[(t[:, a] > 0) & (t[:, b] > 0)]
        if c.size:
            print("   %-24s median %7d  p90 %7d  max %7d   (%d waves)" % (name, np.median(c), np.percentile(c, 90), c.max(), c.size))
    mix = t[(t[:, 5] > 0) & (t[:, 4] > 0) & (t[:, 5] > t[:, 4])]
    if mix.size:
        print("   mixers (%d waves): wait for the last row  median %7d max %7d | mix median %7d max %7d" % (
            mix.shape[0], np.median(mix[:, 5] - mix[:, 4]), (mix[:, 5] - mix[:, 4]).max(), np.median(mix[:, 6] - mix[:, 5]), (mix[:, 6] - mix[:, 5]).max()))
    rt = t[:, 8][t[:, 8] > 0]
    if rt.size:
        print("   rows done (100 MHz clock): first workgroup -> last  %.1f us" % ((rt.max() - rt.min()) / 100.0))
