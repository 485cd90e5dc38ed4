"""ASan + UBSan over the host-side code that runs without a GPU (the GPU side cannot be sanitised on
this pool): the voice pool under random traffic and checkpoint restores, the .synth2 parser under a
mutation fuzz, the streamer text frame with the buffer sized exactly as advertised."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_voice_pool_and_parser_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "san_host")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "synth2_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "san_host.cpp"),
                           os.path.join(ROOT, "synth2_amd", "csrc", "s2r_patch.cpp"),
                           os.path.join(ROOT, "synth2_amd", "csrc", "s2r_stream.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "pool ok" in out.stdout
    # s2r_stream_frame_json with the buffer sized exactly as advertised (16 chars per sample overflowed)
    assert "stream ok" in out.stdout and "longest element 16 chars" in out.stdout, out.stdout
