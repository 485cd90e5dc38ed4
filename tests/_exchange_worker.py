"""one rank of a group of processes sharing the card (tests/test_gpu_exchange.py): its shard of the pool, the exchange through the
root's block (s2r_exchange_create / _attach), the same events on every rank; rank 0 saves the buffers it gets
usage: _exchange_worker.py RANK N DIR VOICES_TOTAL BUFFERS RESIDENT [free]
free: no barrier between the ranks' steps — the root dawdles (pauses of up to 4 ms: past a resident kernel's patience), the others run
as fast as their own fills allow, and what keeps a rank from writing fill k + 2's row over fill k's unread one is the library's."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth2_amd as s2  # noqa: E402

rank, n, d, V, K, resident = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
free = len(sys.argv) > 7 and sys.argv[7] == "free"
pause_rng = np.random.RandomState(7)


def wait_for(name, timeout=120.0):
    t0 = time.time()
    while not os.path.exists(os.path.join(d, name)):
        if time.time() - t0 > timeout:
            sys.exit("rank %d: timed out waiting for %s" % (rank, name))
        time.sleep(0.001)


def touch(name, data=b"1"):
    tmp = os.path.join(d, name + ".tmp")
    with open(tmp, "wb") as f:
        f.write(data)
    os.rename(tmp, os.path.join(d, name))


s = s2.Synth(V, max_frames=1024, shard_begin=rank * (V // n), shard_voices=V // n)
if rank == 0:
    touch("handle", s.exchange_create(n))
else:
    wait_for("handle")
    s.exchange_attach(rank, n, open(os.path.join(d, "handle"), "rb").read())
if resident:
    s.set_resident(True)
touch("ready%d" % rank)
for r in range(n):
    wait_for("ready%d" % r)
rng = np.random.RandomState(2024)
out = np.zeros((K, 1024), dtype=np.float32)
in_flight = []
for k in range(K):
    m = 3000 if k == 0 else int(rng.randint(0, 700))
    ev = np.zeros(m, dtype=s2.NOTE_EVENT_DTYPE)
    ev["kind"] = rng.randint(0, 2, m); ev["note"] = rng.randint(36, 97, m); ev["velocity"] = 1.0
    if k % 3:
        ev["frame"] = np.sort(rng.randint(0, 64, m)) * 16
    s.note_events(ev)
    if k % 5 == 4:                                   # a synchronous fill among the ring fills
        while in_flight:
            s.sample_end(out[in_flight.pop(0)])
        s.sample(out[k], 48000)
    else:
        s.sample_begin(1024, 48000)
        in_flight.append(k)
        if len(in_flight) == 2:
            s.sample_end(out[in_flight.pop(0)])
    if free:
        if rank == 0 and pause_rng.rand() < 0.5:
            t_end = time.perf_counter() + float(pause_rng.choice([2e-4, 1e-3, 4e-3]))
            while time.perf_counter() < t_end:
                pass
        continue
    # (the ranks stay within a few fills of each other: the root's last workgroup waits 50 ms at most for a rank's row)
    touch("step%d_%d" % (rank, k))
    for r in range(n):
        wait_for("step%d_%d" % (r, max(0, k - 1)))
while in_flight:
    s.sample_end(out[in_flight.pop(0)])
np.save(os.path.join(d, "out%d.npy" % rank), out)
print("rank %d done" % rank)
