"""One patch-bank population rendered 20 times (for rocprofv3 runs: tools/bank_pmc.sh).  argv[1]: 2 = two default patches, 8 = the mixed bank of tools/bank_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2

V = 65536
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bank = []
for i in range(n):
    p = s2.default_patch()
    if n == 8:
        p.osc_kind = i % 4
        p.lpf_kind = [0, 3, 0, 5, 1, 0, 4, 2][i]
    bank.append(p)
s = s2.Synth(V, max_frames=1024)
s.set_patch_bank(bank)
ev = np.zeros(2 * V, dtype=s2.NOTE_EVENT_DTYPE)
ev["kind"][0::2] = 2
ev["note"][0::2] = np.arange(V) % len(bank)
ev["kind"][1::2] = 1
ev["note"][1::2] = (np.arange(V) * 13) % 100 + 20
ev["velocity"] = 1.0
s.note_events(ev)
buf = np.empty(1024, dtype=np.float32)
for k in range(20):
    s.sample(buf)
print("done")
