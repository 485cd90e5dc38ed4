"""development aid: render-kernel time vs frames per fill and vs voices (fixed-cost vs per-frame cost)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth2_amd as s2

def run(voices, frames, flat=True, lanes=0, reps=12):
    s = s2.Synth(voices, max_frames=4096)
    ev = np.zeros(voices, dtype=s2.NOTE_EVENT_DTYPE); ev["kind"] = 1; ev["note"] = 36 + np.arange(voices) % 61
    s.note_events(ev)
    out = torch.zeros(4096, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(4): s.fill_device(out.data_ptr(), 4096, 48000, st)   # get past the mod decay (9600 frames)
    s.set_flat_shortcut(flat)
    s.set_timing(True)
    ms = []
    for _ in range(reps):
        s.fill_device(out.data_ptr(), frames, 48000, st)
        ms.append(s.last_render_ms())
    return float(np.median(ms)), 1

import itertools
cases = [(65536, 0), (131072, 0), (16384, 0)] if len(sys.argv) < 2 else [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1:]]
for voices, lanes in cases:
    for flat in (True, False):
        row = []
        for frames in (64, 256, 1024, 4096):
            t, L = run(voices, frames, flat, lanes)
            row.append("%d:%.4f" % (frames, t))
        print("voices %6d L=%d flat=%-5s  ms by frames  %s" % (voices, L, flat, "  ".join(row)))
