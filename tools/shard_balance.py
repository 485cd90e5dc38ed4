"""What one rank of an 8-GPU run costs per buffer under the bench's note traffic, with contiguous shards
(the allocation policy's sweep lands all of a step's 1024 restarts on one rank) and with the pool dealt
out in runs of 64 voices.  One GPU plays rank 0 (the busy one for the first 64 steps) and rank 5."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth2_amd as s2
from bench import make_events, FRAMES, SR

world, vpr = 8, 65536
total = world * vpr
events = [make_events(total, 128, k) for k in range(72)]
init = np.zeros(total, dtype=s2.NOTE_EVENT_DTYPE); init["kind"] = 1; init["note"] = 36 + (np.arange(total) % 61); init["velocity"] = 1.0
out = torch.zeros(FRAMES, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for layout in ("contiguous", "dealt out in runs of 64"):
    for rank in (0, 5):
        kw = dict(shard_begin=rank * vpr, shard_voices=vpr) if layout == "contiguous" else dict(shard_interleave=64, shard_index=rank, shard_count=world)
        s = s2.Synth(total, max_frames=FRAMES, **kw)
        s.load_patch("synth mySynth {\n\n}\n")
        s.note_events(init)
        for k in range(8):
            s.note_events(events[k]); s.fill_device(out.data_ptr(), FRAMES, SR, st)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for k in range(8, 72):
            s.note_events(events[k]); s.fill_device(out.data_ptr(), FRAMES, SR, st)
        torch.cuda.synchronize()
        print("%-24s rank %d: %.1f us per buffer" % (layout, rank, (time.perf_counter() - t) / 64 * 1e6))
        s.close()
