"""Worker for test_sharded_exchange_through_rccl_with_one_rank: ONE rank joins an RCCL ("nccl") process group on the GPU
and drives ShardedSynth through the N-rank exchange code (all-gather of partial rows + rank-ordered combine, then the
reduce variant): the calls, tensor shapes, work handles and stream ordering of the multi-GPU path, on the only kind of
box this repository's tests see.  The mixes must equal the plain single-handle path bit for bit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synth2_amd as s2                       # noqa: E402
from synth2_amd.sharded import ShardedSynth   # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    V, F = 1024, 512
    ref = s2.Synth(V, max_frames=F)
    for mode in ("all_gather", "all_gather_no_overlap", "reduce"):
        sh = ShardedSynth(V, max_frames=F, rank=0, world=1, device=dev, force_exchange=True,
                          overlap=(mode != "all_gather_no_overlap"), reduce_to_root=(mode == "reduce"))
        plain = s2.Synth(V, max_frames=F)
        rng = np.random.RandomState(3)
        for k in range(6):
            ev = np.zeros(40, dtype=s2.NOTE_EVENT_DTYPE)
            ev["kind"] = rng.randint(0, 2, 40) | (k == 0); ev["note"] = 36 + rng.randint(0, 61, 40); ev["velocity"] = 1.0
            frames = F if k != 3 else 100
            sh.note_events(ev); plain.note_events(ev)
            sh.fill(frames, 48000)
            sh.flush()
            torch.cuda.synchronize()
            got = sh.mix[:frames].cpu().numpy()
            want = plain.sample(np.empty(frames, dtype=np.float32))
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "%s: buffer %d differs" % (mode, k)
    del ref
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_ONE_RANK_OK")


if __name__ == "__main__":
    main()
