"""development aid: what one handle over a device list costs the host per C3 step, by the number of shards
(the allocation policy runs ONCE per event whatever the number of devices; the shards' launches run
side by side on per-device host threads).  The pool and the event stream stay the same — 65 536 voices, 2 048 events per
step — and are cut into N shards; on a one-GPU box every shard lives on device 0.
    python tools/devlist_host_cost.py [N ...]"""
import sys, os, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # (N shards on ONE device, each with a resident kernel on a stream of its own: s2r.h, s2r_set_resident)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2
from bench import make_c3_events, FRAMES, SR, PERIOD

V = int(os.environ.get("V", 65536))
cyc = make_c3_events(V, PERIOD)
modes = os.environ.get("MODES", "launch,resident").split(",")
for mode, n in [(m, int(a)) for m in modes for a in (sys.argv[1:] or [1, 2, 4, 8])]:
    s = s2.Synth(V, max_frames=FRAMES, devices=[0] * n, shard_interleave=64) if n > 1 else s2.Synth(V, max_frames=FRAMES)
    if mode == "resident":
        s.set_resident(True)
    bufs = [np.empty(FRAMES, dtype=np.float32) for _ in range(2)]
    for k in range(PERIOD):
        s.note_events(cyc[k]); s.sample(bufs[0], SR)
    t_ev = t_b = t_e = 0.0
    reps = 128
    s.note_events(cyc[0]); s.sample_begin(FRAMES, SR)
    t0 = time.perf_counter()
    for k in range(1, reps + 1):
        a = time.perf_counter(); s.note_events(cyc[k % PERIOD])
        b = time.perf_counter(); s.sample_begin(FRAMES, SR)
        c = time.perf_counter(); s.sample_end(bufs[k & 1])
        d = time.perf_counter()
        t_ev += b - a; t_b += c - b; t_e += d - c
    s.sample_end(bufs[0])
    tot = time.perf_counter() - t0
    print("%-8s device list of %d (%5d voices per shard): step %6.1f us = note_events %5.1f + fill_begin %5.1f + fill_end (wait + copy) %5.1f" % (
        mode, n, V // n, tot / reps * 1e6, t_ev / reps * 1e6, t_b / reps * 1e6, t_e / reps * 1e6))
    s.close()
