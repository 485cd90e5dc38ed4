"""The N > 1 path on CPU: world_size-2 (and 3) gloo runs of the product's sharding layer."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,overlap,interleave,reduce", [(2, 1, 0, 0), (2, 0, 64, 0), (3, 1, 0, 0), (2, 1, 64, 0), (4, 1, 16, 0),
                                                             (2, 1, 64, 1), (3, 0, 64, 1)])
def test_sharded_partials_allgather_and_ordered_combine(world, overlap, interleave, reduce):
    """interleave 0: contiguous ranges per rank; else the pool dealt out in runs of `interleave` voices.
    reduce 1: SURVEY 8(e)'s fallback, one reduce(SUM) to rank 0 — bit-equal to the rank-ordered sum for two ranks,
    equal to rounding for three (the worker asserts exactly that)."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   S2R_OVERLAP=str(overlap), S2R_INTERLEAVE=str(interleave), S2R_REDUCE=str(reduce), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_sharded_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-2000:])
    assert "SHARDED_OK world=%d overlap=%d reduce=%d" % (world, overlap, reduce) in outs[0]
