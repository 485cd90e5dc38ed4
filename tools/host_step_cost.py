"""development aid: HOST time per bench step as rank 0 of an N-GPU run sees it (events of the whole
pool + the launches of one fill), measured with 16-frame fills so that the GPU is never the
bottleneck.  If this exceeds the GPU's time per 1024-frame buffer, the host limits the scaling."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth2_amd as s2
from bench import make_events

for world in (1, 2, 4, 8):
    total = 65536 * world
    s = s2.Synth(total, max_frames=1024, shard_begin=0, shard_voices=65536)
    init = np.zeros(total, dtype=s2.NOTE_EVENT_DTYPE); init["kind"] = 1; init["note"] = 36 + np.arange(total) % 61
    s.note_events(init)
    out = torch.zeros(1024, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    evs = [make_events(total, 128, k) for k in range(200)]
    for e in evs[:20]:
        s.note_events(e); s.fill_device(out.data_ptr(), 16, 48000, st)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for e in evs[20:]:
        s.note_events(e)
        s.fill_device(out.data_ptr(), 16, 48000, st)
    dt_host = time.perf_counter() - t
    torch.cuda.synchronize()
    t = time.perf_counter()
    for e in evs[20:]:
        s.note_events(e)
    dt_ev = time.perf_counter() - t
    print("world %d: host %.1f us per step (events alone %.1f us, %d events per step)" % (world, dt_host * 1e6 / 180, dt_ev * 1e6 / 180, len(evs[0])))
    s.close()
