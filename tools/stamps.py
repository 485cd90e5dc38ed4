"""development aid: where the one-pole render kernel's time goes, wave by wave.  Needs the diagnostic library:
    S2R_STAMPS=1 python tools/stamps.py [workload]      (workload: c3 | start | every64 | churn ...; see tools/tev_cost.py)
builds libs2r_stamps.so (s_memtime stamps at the kernel's phase boundaries) and prints, for one fill in steady state,
the per-phase cycle counts of the median wave, of the slowest wave and of the waves that had timed events."""
import ctypes as C
import os
import sys
os.environ["S2R_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2
from bench import make_c3_events, FRAMES, SR, PERIOD

V = int(os.environ.get("V", 65536))
kind = sys.argv[1] if len(sys.argv) > 1 else "c3"
base = make_c3_events(V, PERIOD)
cyc = []
for ev in base:
    e = ev.copy()
    timed = e["frame"] > 0
    if kind == "start":
        e["frame"][:] = 0
    elif kind.startswith("every"):
        k = int(kind[5:])
        idx = np.nonzero(timed)[0]
        e["frame"][idx[np.arange(idx.size) % k != 0]] = 0
        e = e[np.argsort(e["frame"], kind="stable")]
    cyc.append(e)
s = s2.Synth(V, max_frames=FRAMES)
L = s.L
L.s2r_debug_read_stamps.restype = C.c_uint32
L.s2r_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
buf = np.empty(FRAMES, dtype=np.float32)
for k in range(PERIOD + 8):
    s.note_events(cyc[k % PERIOD]); s.sample(buf, SR)
waves = L.s2r_debug_read_stamps(s.h, None, 0)          # arm
assert waves, "this is not the diagnostic build"
st = np.zeros((waves, 16), dtype=np.uint64)
for rep in range(3):
    k = PERIOD + 8 + rep
    ev = cyc[k % PERIOD]
    s.note_events(ev); s.sample(buf, SR)
    L.s2r_debug_read_stamps(s.h, st.ctypes.data, waves)
    t = st.astype(np.int64)
    t0 = t[:, 0].min()
    total = t[:, 15] - t[:, 0]
    # (a fill of up to 1024 frames on a grid no larger than the device is ONE super-chunk: slots 2, 3 and then 15)
    n_sc = int(np.count_nonzero(t[0, 2:10])) // 2
    names = ["prologue"] + ["sc%d work" % i if j == 0 else "sc%d combine" % i for i in range(n_sc) for j in range(2)] + ["epilogue"]
    cols = [t[:, 1] - t[:, 0]] + [t[:, 2 + i] - t[:, 1 + i] for i in range(2 * n_sc)] + [t[:, 15] - t[:, 1 + 2 * n_sc]]
    t[:, 15] = np.where(t[:, 15] == 0, t[:, 0], t[:, 15])
    slow = int(np.argmax(t[:, 15]))
    print("fill %d (%s): kernel span %.1f us at 100 MHz-free ticks=cycles: first entry -> last exit %d cycles; wave total median %d, max %d (wave %d)" % (
        k, kind, 0.0, int(t[:, 15].max() - t0), int(np.median(total)), int(total.max()), int(np.argmax(total))))
    print("   entry skew (last wave's entry - first's): %d cycles" % int(t[:, 0].max() - t0))
    ev_w = np.nonzero(t[:, 11] > 0)[0]
    work = sum(cols[1 + 2 * i] for i in range(n_sc))
    print("   waves with timed events: %d; events per such wave: median %d max %d" % (ev_w.size, int(np.median(t[ev_w, 11])) if ev_w.size else 0, int(t[:, 11].max())))
    order = np.argsort(-total)[:12]
    x = st[:, 10]
    runchunks = (x >> np.uint64(48)).astype(np.int64); t_dense = ((x >> np.uint64(24)) & np.uint64(0xffffff)).astype(np.int64); t_run = (x & np.uint64(0xffffff)).astype(np.int64)
    y = st[:, 13]
    t_top = (y >> np.uint64(32)).astype(np.int64); t_sel = (y & np.uint64(0xffffffff)).astype(np.int64)
    z = st[:, 12]
    n_dense = (z & np.uint64(0xfffff)).astype(np.int64); t_pick1 = ((z >> np.uint64(20)) & np.uint64(0x3fffff)).astype(np.int64)
    t_casc = ((z >> np.uint64(42)) & np.uint64(0x3fffff)).astype(np.int64)
    med = np.argsort(total)[len(total) // 2 - 2: len(total) // 2 + 2]
    print("   %6s %7s %7s %7s %7s %5s | %4s %6s %7s %6s | %7s = %6s + %6s + %6s | %6s | %5s %7s" % (
        "wave", "total", "prolog", "work", "combine", "epil", "runs", "chunks", "in runs", "/chunk", "choose", "look", "stage", "setup", "top", "dense", "cyc"))
    for tag, ws in (("slowest", order[:8]), ("median", med)):
        for w in ws:
            print("   %6d %7d %7d %7d %7d %5d | %4d %6d %7d %6d | %7d = %6d + %6d + %6d | %6d | %5d %7d  %s" % (
                w, total[w], cols[0][w], work[w], sum(cols[2 + 2 * i][w] for i in range(n_sc)), cols[-1][w], t[w, 14], runchunks[w], t_run[w],
                t_run[w] // max(1, runchunks[w]), t_sel[w], t_pick1[w], t_casc[w], t_sel[w] - t_pick1[w] - t_casc[w], t_top[w], n_dense[w], t_dense[w], tag))
    for n, c in zip(names, cols):
        print("   %-12s median %7d   p99 %7d   max %7d   | last-exiting wave %d: %7d" % (n, int(np.median(c)), int(np.percentile(c, 99)), int(c.max()), slow, int(c[slow])))
