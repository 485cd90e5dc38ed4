"""Worker for tests/test_sharded_gloo.py: one of WORLD_SIZE CPU processes joined by gloo.

It drives the PRODUCT's sharding layer (synth2_amd.sharded.ShardedSynth: double-buffered
partials, all-gather, rank-ordered root combine) with an injected renderer that produces this
rank's partial mix from the CPU oracle (test infrastructure) — there is no GPU here — and checks
on rank 0 that the combined buffers equal the oracle's mix tree with groups == world size."""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import s2o                      # noqa: E402
import synth2_amd as s2                     # noqa: E402
from synth2_amd.sharded import ShardedSynth  # noqa: E402

BLOCK = 256


class OracleShardRenderer:
    """stands in for synth2_amd.Synth on a machine without a GPU"""

    def __init__(self, total, indices):
        self.ora = s2o.OracleSynth(total)
        self.pool = s2.VoicePool(total)          # the product's allocation policy, run alongside
        self.indices = indices                   # pool indices of this rank's voices, in local order
        self.rows = None

    def note_events(self, events):
        for e in events:
            if e["kind"] == 1:
                want = self.ora.next_voice_index()
                self.ora.note_on(int(e["note"]), float(e["velocity"]))
                assert self.pool.note_on(int(e["note"]), float(e["velocity"])) == want
            else:
                self.ora.note_off(int(e["note"]))
                self.pool.note_off(int(e["note"]))

    def fill_device(self, ptr, frames, sample_rate, stream):
        self.rows = self.ora.render_voices(frames, sample_rate)
        self.pool.advance(frames)
        part = s2o.mix_tree_partial(self.rows[self.indices], BLOCK)
        ctypes.memmove(ptr, part.ctypes.data, frames * 4)


def combine_numpy(rows, n_rows, frames, out):
    r = rows.numpy()
    acc = np.zeros(frames, dtype=np.float32)          # accum = splat(0.0), synth.rs:176
    for k in range(n_rows):
        acc = acc + r[k, :frames]
    out[:frames] = torch.from_numpy(acc)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vpr, frames = 512, 256
    total = vpr * world
    overlap = os.environ.get("S2R_OVERLAP", "1") == "1"
    inter = int(os.environ.get("S2R_INTERLEAVE", "0"))     # 0: contiguous ranges, else runs of `inter` voices dealt out
    shard = (lambda r: s2.shard_pool_indices(total, r, world, inter)) if inter else (lambda r: np.arange(r * vpr, (r + 1) * vpr))
    reduce = os.environ.get("S2R_REDUCE", "0") == "1"     # one reduce(SUM) to rank 0 instead of all-gather + ordered sum
    sh = ShardedSynth(vpr, max_frames=frames, rank=rank, world=world, device=torch.device("cpu"),
                      renderer=OracleShardRenderer(total, shard(rank)), combine=combine_numpy, overlap=overlap,
                      reduce_to_root=reduce)

    def same(got, want, what):
        if reduce and world > 2:
            # the collective picks the association: equal to rounding, not to the bit (why all-gather is the default)
            assert np.allclose(got, want, rtol=2e-6, atol=2e-6), what
        else:
            # two ranks: a + b == b + a, so even the reduce is the rank-ordered sum bit for bit
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), what
    ev = np.zeros(total, dtype=s2.NOTE_EVENT_DTYPE)
    ev["kind"] = 1
    ev["note"] = 36 + np.arange(total) % 61
    ev["velocity"] = 1.0
    sh.note_events(ev)
    rng = np.random.RandomState(5)
    expected = []
    for k in range(6):
        churn = np.zeros(40, dtype=s2.NOTE_EVENT_DTYPE)
        churn["kind"] = rng.randint(0, 2, 40)
        churn["note"] = 36 + rng.randint(0, 61, 40)
        churn["velocity"] = 1.0
        sh.note_events(churn)
        n = frames if k != 3 else 100                  # one short (ragged, non multiple of 16) buffer
        sh.fill(n, 48000)
        want = np.zeros(n, dtype=np.float32)               # accum = splat(0.0), then the ranks' partial rows in rank order
        for r in range(world):
            want = want + s2o.mix_tree_partial(sh.renderer.rows[shard(r)], BLOCK)
        if not inter:                                      # contiguous ranges: the association of one GPU with mix_groups
            assert np.array_equal(want.view(np.uint32), s2o.mix_tree(sh.renderer.rows, BLOCK, world).view(np.uint32))
        expected.append((n, want))
        if not overlap or k == 5:
            sh.flush()
        if rank == 0 and (not overlap):
            got = sh.mix.numpy()[:n]
            same(got, expected[-1][1], "buffer %d differs" % k)
    sh.flush()
    if rank == 0:
        n, want = expected[-1]
        got = sh.mix.numpy()[:n]
        same(got, want, "last buffer differs")
        # the 1-rank order (groups = 1) generally differs in the last bits: that is why mix_groups exists
        print("SHARDED_OK world=%d overlap=%d reduce=%d" % (world, overlap, reduce))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
