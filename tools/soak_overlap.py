"""One-off soak (not part of the test suite) of the two-stream mode (s2r_fill_begin / s2r_fill_end): the bench's
C3 schedule at 65 536 voices, every buffer against the oracle, with random pauses between the host's calls so that the
kernels of the two streams meet in every order.      python tools/soak_overlap.py [buffers] [mode]
mode: two (default: the two streams) | resident (s2r_set_resident: the pool-resident kernel, with pauses past its 2 ms patience
among the draws) | fused (S2R_FUSED=2: one launch per fill) | devlist (one handle over {0} x 4, resident)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import assert_bits_equal
from oracle import s2o
import synth2_amd as s2
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
MODE = sys.argv[2] if len(sys.argv) > 2 else "two"
if MODE == "fused":
    os.environ["S2R_FUSED"] = "2"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
V = 65536
cyc = bench.make_c3_events(V, bench.PERIOD)
gpu = s2.Synth(V, max_frames=bench.FRAMES, devices=[0] * 4, shard_interleave=64) if MODE == "devlist" else s2.Synth(V, max_frames=bench.FRAMES)
if MODE in ("resident", "devlist"):
    gpu.set_resident(True)
ora = s2o.OracleSynth(V)
threads = max(1, min(32, len(os.sched_getaffinity(0))))
rng = np.random.RandomState(5)
queue = []
t0 = time.time()


def pause():
    r = rng.rand()
    if r < 0.5:
        return
    t = time.perf_counter()
    d = float(rng.choice([5e-6, 2e-5, 5e-5, 2e-4, 1e-3] + ([3e-3] if MODE in ("resident", "devlist") else [])))
    while time.perf_counter() - t < d:
        pass


def end_one():
    k, want = queue.pop(0)
    pause()
    assert_bits_equal(gpu.sample_end(np.empty(bench.FRAMES, dtype=np.float32)), want, "buffer %d" % k)


for k in range(N):
    ev = cyc[k % bench.PERIOD]
    pause(); gpu.note_events(ev)
    pause(); gpu.sample_begin(bench.FRAMES, bench.SR)
    pv = ora.render_events(ev, bench.FRAMES, bench.SR, threads=threads)
    if MODE == "devlist":                                    # the rank-ordered sum of the four shards' trees
        want = np.zeros(bench.FRAMES, dtype=np.float32)
        for sk in range(4):
            want = want + s2o.mix_tree_partial(pv[s2.synth.shard_pool_indices(V, sk, 4, 64)], gpu.block_voices)
        queue.append((k, want))
    else:
        queue.append((k, s2o.mix_tree(pv, gpu.block_voices, 1)))
    del pv
    while len(queue) > (1 if rng.rand() < 0.85 else 0):
        end_one()
    if k % 50 == 0:
        print("buffer %d ok, %.0f s" % (k, time.time() - t0), flush=True)
while queue:
    end_one()
print("soak ok (%s): %d buffers of %d voices through s2r_fill_begin / s2r_fill_end, every one equal to the oracle's" % (MODE, N, V))
