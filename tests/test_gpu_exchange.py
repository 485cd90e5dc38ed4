"""One process per GPU without a collective in the step (s2r_exchange_create / _attach, s2r.h): two processes sharing this box's
card, each rendering its contiguous half of the pool, the rows exchanged through the root's IPC-mapped block and added by the
root's last workgroup — bit for bit what ONE device returns with mix_groups = 2."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import assert_bits_equal
import synth2_amd as s2

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("resident,pace,K,n", [(0, "lockstep", 12, 2), (1, "lockstep", 12, 2), (0, "free", 60, 2), (1, "free", 60, 2),
                                               (0, "free", 40, 4), (1, "free", 40, 4)])
def test_two_processes_exchange_rows_through_the_roots_block(tmp_path, resident, pace, K, n):
    """pace "free": nothing outside the library keeps the ranks together — the root pauses at random (up to 4 ms, past a resident
    kernel's patience), the other rank runs ahead as far as its own two fills in flight allow and must wait, inside its kernels,
    for the root to have added up a rows slot's previous fill before it writes the next one's row there"""
    V = 16384                                                    # (n = 4: four processes on the card, 16 workgroups each)
    K = int(os.environ.get("S2R_EXCHANGE_K", K))                 # (a soak: S2R_EXCHANGE_K=2000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_exchange_worker.py"), str(r), str(n), str(tmp_path), str(V), str(K), str(resident)] + (["free"] if pace == "free" else []),
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(n)]
    outs = [p.communicate(timeout=300) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d: %s\n%s" % (r, so[-1000:], se[-3000:])
    got = np.load(os.path.join(str(tmp_path), "out0.npy"))
    other = np.load(os.path.join(str(tmp_path), "out1.npy"))
    assert not other.any(), "a rank other than the root returns silence"
    for r in range(2, n):
        assert not np.load(os.path.join(str(tmp_path), "out%d.npy" % r)).any()
    # the same events on ONE device with mix_groups = 2
    one = s2.Synth(V, max_frames=1024, mix_groups=n)
    rng = np.random.RandomState(2024)
    for k in range(K):
        m = 3000 if k == 0 else int(rng.randint(0, 700))
        ev = np.zeros(m, dtype=s2.NOTE_EVENT_DTYPE)
        ev["kind"] = rng.randint(0, 2, m); ev["note"] = rng.randint(36, 97, m); ev["velocity"] = 1.0
        if k % 3:
            ev["frame"] = np.sort(rng.randint(0, 64, m)) * 16
        one.note_events(ev)
        assert_bits_equal(got[k], one.sample(np.empty(1024, dtype=np.float32), 48000), "two processes vs one device with mix_groups = 2, buffer %d" % k)
