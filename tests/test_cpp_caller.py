"""The C++ caller examples/s2_render.cpp — s2_bin's synth loop (main.rs:132-147) over the C++
mirror of `Synth` (include/s2_synth.hpp) and libs2r's C ABI, no Python in the product path —
against the oracle driven exactly like s2_bin: MIDI applied between 16-frame sample() calls."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SCRIPT = """# frame kind note [velocity]
0 on 57
0 on 64 90
37 on 69            # not on a chunk boundary: applied before the chunk that starts at 48
2000 off 64
2048 on 72
4100 off 57
4100 on 45 64
6000 off 69
9000 on 50
9017 off 50
"""


def _exe():
    import __graft_entry__ as ge
    return ge.build_example()


def test_example_builds_against_the_public_headers():
    """compiles and links on the CPU box (headers + exported symbols); running it needs the GPU"""
    assert os.path.exists(_exe())


def _oracle(script, total, buffer_frames, patch_cfg=None):
    from oracle import s2o
    syn = s2o.OracleSynth(8)
    if patch_cfg is not None:
        syn.config = patch_cfg
    msgs = []
    for line in script.splitlines():
        line = line.split("#")[0].split()
        if len(line) >= 3:
            msgs.append((int(line[0]), line[1] == "on", int(line[2])))
    msgs.sort(key=lambda m: m[0])
    out = np.zeros(total, dtype=np.float32)
    k = 0
    for pos in range(0, total, buffer_frames):
        n = min(buffer_frames, total - pos)
        for c in range(0, n, 16):
            while k < len(msgs) and msgs[k][0] <= pos + c:
                (syn.note_on if msgs[k][1] else syn.note_off)(msgs[k][2])
                k += 1
            m = min(16, n - c)
            out[pos + c:pos + c + m] = syn.sample(m)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["reference_loop", "reference_loop_launch_per_call", "batched"])
@pytest.mark.parametrize("buffer_frames", [2048, 1000])
def test_cpp_caller_matches_the_oracle(tmp_path, mode, buffer_frames):
    exe = _exe()
    script = tmp_path / "notes.txt"
    script.write_text(SCRIPT)
    out = tmp_path / "out.f32"
    total = 12000
    cmd = [exe, str(script), str(out), "--buffer", str(buffer_frames), "--frames", str(total)]
    if mode == "batched":
        cmd.append("--batched")
    elif mode == "reference_loop_launch_per_call":       # (the default loop keeps the resident kernel between its 16-frame calls)
        cmd.append("--launch-per-call")
    subprocess.check_call(cmd)
    got = np.fromfile(out, dtype=np.float32)
    want = _oracle(SCRIPT, total, buffer_frames)
    assert got.shape == want.shape
    # 8 voices: the GPU's summation tree is the reference's own sequential order
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "C++ caller (%s) differs from the oracle" % mode


@pytest.mark.gpu
def test_cpp_caller_with_a_patch_file(tmp_path):
    from oracle import s2o
    exe = _exe()
    (tmp_path / "p.synth2").write_text("synth lead { osc.kind = square; lpf.kind = lp2; lpf.freq = 800; lpf.damping = 0.5; noise = 0.1 }\n")
    (tmp_path / "n.txt").write_text(SCRIPT)
    subprocess.check_call([exe, str(tmp_path / "n.txt"), str(tmp_path / "o.f32"), "--patch", str(tmp_path / "p.synth2"),
                           "--frames", "8192", "--buffer", "1024", "--batched"])
    cfg = s2o.lib().s2o_default_config()
    cfg.osc_kind = 0; cfg.lpf_kind = 3; cfg.lpf_freq = 800.0; cfg.lpf_damping = 0.5; cfg.noise = 0.1
    want = _oracle(SCRIPT, 8192, 1024, cfg)
    got = np.fromfile(tmp_path / "o.f32", dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
