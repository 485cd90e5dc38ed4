"""development aid: host cost of s2r_note_events per event for big pools (what every rank of an
N-GPU run pays for the whole pool's events)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
from bench import make_events
for world in (1, 8):
    total = 65536 * world
    s = s2.Synth(total, max_frames=1024, shard_begin=0, shard_voices=65536)
    init = np.zeros(total, dtype=s2.NOTE_EVENT_DTYPE); init["kind"] = 1; init["note"] = 36 + np.arange(total) % 61
    t = time.perf_counter(); s.note_events(init); dt = time.perf_counter() - t
    print("world %d: initial %d note-ons: %.1f ms (%.0f ns each)" % (world, total, dt * 1e3, dt * 1e9 / total))
    evs = [make_events(total, 128, k) for k in range(50)]
    buf = np.empty(1024, dtype=np.float32)
    t = time.perf_counter()
    for e in evs:
        s.note_events(e)
    dt = time.perf_counter() - t
    n = sum(len(e) for e in evs)
    print("world %d: %d churn events per step: %.1f us per step (%.0f ns per event)" % (world, len(evs[0]), dt * 1e6 / 50, dt * 1e9 / n))
    s.close()
