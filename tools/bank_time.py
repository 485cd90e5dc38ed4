"""Times one fill of 65536 voices rendering through a patch bank (general kernel) against the
single-patch kernels on the same note population."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth2_amd as s2

V = int(os.environ.get("V", 65536))
FR = int(os.environ.get("FR", 1024))      # frames per fill


def run(bank, label):
    s = s2.Synth(V, max_frames=1024)
    s.set_patch_bank(bank)
    rows = []
    for i in range(V):
        if len(bank) > 1 and i % 64 == 0:
            pass
    ev = np.zeros(2 * V, dtype=s2.NOTE_EVENT_DTYPE)
    ev["kind"][0::2] = 2
    ev["note"][0::2] = np.arange(V) % len(bank)
    ev["kind"][1::2] = 1
    ev["note"][1::2] = (np.arange(V) * 13) % 100 + 20
    ev["velocity"] = 1.0
    s.note_events(ev)
    s.set_timing(True)
    buf = np.empty(FR, dtype=np.float32)
    ts = []
    for k in range(24):
        s.sample(buf)
        ts.append(s.last_render_ms())
    print("%-40s first %.3f ms  settled %.3f ms  (%.3g voice-samples/s)" % (label, ts[0], np.mean(ts[12:]), V * FR / (np.mean(ts[12:]) * 1e-3)))


d = s2.default_patch()
run([d], "bank of 1 (tuned one-pole kernel)")
run([d, d], "bank of 2 x default patch")
mixed = []
for i in range(8):
    p = s2.default_patch()
    p.osc_kind = i % 4
    p.lpf_kind = [0, 3, 0, 5, 1, 0, 4, 2][i]
    mixed.append(p)
run(mixed, "bank of 8 (4 osc kinds, 6 filters)")
