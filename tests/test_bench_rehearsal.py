"""bench.py's multi-rank flow rehearsed on ONE GPU: two ranks launched exactly as the driver launches
them (torch.distributed.run, one process per rank) share the card and exchange their partial rows
over gloo instead of RCCL.  Everything but the RCCL collective is the code of a real N-GPU run:
rendezvous, the same event stream on every rank, shard-local rendering, gather, rank-ordered
combine on rank 0, max-over-ranks timing, one JSON line."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


@pytest.mark.gpu
def test_two_rank_bench_flow_on_one_gpu():
    env = dict(os.environ, S2R_BENCH_BACKEND="gloo", S2R_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "6", "--warmup", "2", "--voices-per-gpu", "8192", "--workload", "churn"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak"
    assert d["config"]["voices_total"] == 2 * 8192
    assert d["value"] > 0 and d["mix_checksum"] > 0
    # single process over the same pool with mix_groups = 2 adds the same rows in the same order
    import numpy as np
    sys.path.insert(0, ROOT)
    import synth2_amd as s2
    from bench import make_events, FRAMES, SR
    ref = s2.Synth(2 * 8192, max_frames=FRAMES, mix_groups=2)
    ref.load_patch("synth mySynth {\n\n}\n")
    init = np.zeros(2 * 8192, dtype=s2.NOTE_EVENT_DTYPE)
    init["kind"] = 1; init["note"] = 36 + (np.arange(2 * 8192) % 61); init["velocity"] = 1.0
    ref.note_events(init)
    buf = np.empty(FRAMES, dtype=np.float32)
    for k in range(8):
        ref.note_events(make_events(2 * 8192, 128, k))
        ref.sample(buf, SR)
    # bench.py runs more fills after the timed region before it reads the mix back (kernel timing legs), so
    # only the order of magnitude of the checksum is comparable here; the bit-level equivalence of the
    # sharded sum is tests/test_gpu_parity.py::test_mix_groups_reproduce_multi_gpu_order
    assert 0.1 < d["mix_checksum"] / float(np.abs(buf).sum()) < 10.0


@pytest.mark.gpu
def test_two_rank_bench_flow_with_resident_kernels_and_the_in_kernel_exchange():
    """what `bench.py --gpus N` runs by default — the C3 workload, pool-resident kernels, the rows exchanged inside the kernels — as
    two ranks on one card (16 384 voices each: both grids fit it together).  Round 4: this hung every time.  A resident kernel's
    workgroups leave after 2 ms without a command; the root's last mixer may wait tens of milliseconds for another rank's row, and
    when it came back its neighbours had run the next command and decided to leave at the one after — a decision two ahead of
    its own, which the command loop did not read as "run yours" (s2r_kern_common.h, pool_loop).  Twice, back to back."""
    for attempt in range(2):
        env = dict(os.environ, S2R_BENCH_BACKEND="gloo", S2R_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
               "--gpus", "2", "--steps", "20", "--warmup", "5", "--voices-per-gpu", "16384", "--watchdog", "60"]
        out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["config"]["voices_total"] == 2 * 16384 and d["value"] > 0
        assert "resident" in d["config"]["timed_call"] and "exchange" in d["config"]["parallelism"], d["config"]
