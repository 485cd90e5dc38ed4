"""development aid: where the host's time goes in one C3 bench step (s2r_note_events / s2r_fill_begin / s2r_fill_end), and
the same step with 16-frame fills (the GPU never the bottleneck): if the host's own time per step approaches the render
kernel's, the GPU waits for the host."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("WITH_TORCH") == "1":          # (does the bench's own environment — torch loaded, its CUDA context up — cost the calls anything?)
    import torch
    torch.cuda.init(); torch.zeros(1, device="cuda")
import synth2_amd as s2
from bench import make_c3_events, FRAMES, SR, PERIOD

V = int(os.environ.get("V", 65536))
cyc = make_c3_events(V, PERIOD)
for frames in (FRAMES, 16):
    s = s2.Synth(V, max_frames=FRAMES)
    bufs = [np.empty(frames, dtype=np.float32) for _ in range(2)]
    cyc_f = cyc
    if frames != FRAMES:
        cyc_f = []
        for e in cyc:
            e = e.copy(); e["frame"] = 0; cyc_f.append(e)
    for k in range(PERIOD):
        s.note_events(cyc_f[k]); s.sample(bufs[0][:frames], SR)
    t_ev = t_b = t_e = 0.0
    n = 128
    s.note_events(cyc_f[0]); s.sample_begin(frames, SR)
    t0 = time.perf_counter()
    for k in range(1, n + 1):
        a = time.perf_counter()
        s.note_events(cyc_f[k % PERIOD])
        b = time.perf_counter()
        s.sample_begin(frames, SR)
        c = time.perf_counter()
        s.sample_end(bufs[k & 1])
        d = time.perf_counter()
        t_ev += b - a; t_b += c - b; t_e += d - c
    s.sample_end(bufs[0])
    tot = time.perf_counter() - t0
    print("frames %4d: step %.1f us = note_events %.1f + fill_begin %.1f + fill_end (wait + copy) %.1f" % (
        frames, tot / n * 1e6, t_ev / n * 1e6, t_b / n * 1e6, t_e / n * 1e6))
    s.close()
