"""development aid: N launches of the render kernel in a settled state (all voices flat or not)
for PMC profiling.  usage: flat_probe.py <voices> <frames> <flat 0|1> [lanes]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synth2_amd as s2
voices, frames, flat = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 0
s = s2.Synth(voices, max_frames=4096)
ev = np.zeros(voices, dtype=s2.NOTE_EVENT_DTYPE); ev["kind"] = 1; ev["note"] = 36 + np.arange(voices) % 61
s.note_events(ev)
out = torch.zeros(4096, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(4): s.fill_device(out.data_ptr(), 4096, 48000, st)
s.set_flat_shortcut(bool(flat))
torch.cuda.synchronize()
for _ in range(10): s.fill_device(out.data_ptr(), frames, 48000, st)
torch.cuda.synchronize()
