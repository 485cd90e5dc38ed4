"""One-off soak (not part of the test suite): a few thousand buffers with continuous note churn, GPU
against the oracle on every buffer, offsets growing past 2^24 on the way (import of old voices)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from helpers import Pair, make_patch, assert_bits_equal
V = int(os.environ.get("V", 192)); N = int(os.environ.get("N", 2500))
patch = make_patch(noise=0.2, mod_env_to_lpf_freq=6.0)
patch.amp_env.release_ms = 300.0
pr = Pair(V, patch, max_frames=1024)
pr.threads = 16
rng = np.random.RandomState(2)
held = []
t0 = time.time()
for b in range(N):
    for _ in range(int(rng.randint(0, 4))):
        if (not held) or rng.rand() < 0.55:
            n = int(rng.randint(24, 100)); held.append(n); pr.note_on(n)
        else:
            pr.note_off(held.pop(int(rng.randint(len(held)))))
    if b == N // 3:                      # make a third of the voices 2^24 - 300000 frames old
        st = pr.gpu.export_state()
        for v in range(0, V, 3):
            if st["started"][v]:
                off = (1 << 24) - 300000 + 17 * v
                st["current_frame_offset"][v] = off; pr.cpu.voice(v).current_frame_offset = off
                if st["released"][v]:
                    st["release_frame_offset"][v] = off - 100; pr.cpu.voice(v).release_frame_offset = off - 100
        pr.gpu.import_state(st)
    frames = 1024 if b % 7 else 1000
    g, o, _ = pr.sample(frames)
    assert_bits_equal(g, o, "soak buffer %d" % b)
    if b % 500 == 0:
        print("buffer %d ok, %.0f s" % (b, time.time() - t0), flush=True)
print("soak ok: %d buffers, %d voices" % (N, V))
