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
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "synth2_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "san_host.cpp"),
                           os.path.join(ROOT, "synth2_amd", "csrc", "s2r_patch.cpp"),
                           os.path.join(ROOT, "synth2_amd", "csrc", "s2r_stream.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "pool ok" in out.stdout
    # s2r_stream_frame_json with the buffer sized exactly as advertised (16 chars per sample overflowed)
    assert "stream ok" in out.stdout and "longest element 16 chars" in out.stdout, out.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_policy_threads_under_tsan(tmp_path):
    """the allocation policy's batch form on four threads (S2rVoicePool::resolve_batch: the queue on the caller's thread, the
    notes' sets shared among three workers) under ThreadSanitizer, its choices and the voices' clocks equal to the
    event-by-event policy's on random batches (tests/native/tsan_policy.cpp)"""
    exe = str(tmp_path / "tsan_policy")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-I", os.path.join(ROOT, "synth2_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "tsan_policy.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    if out.returncode != 0 and "unexpected memory mapping" in out.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow here")
    assert out.returncode == 0 and "policy threads ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_oracle_known_answers_under_asan_ubsan():
    """the C oracle itself once under -fsanitize=address,undefined — the known-answer suite (every vector the
    reference holds for the path) and a multi-voice render with note traffic, in a child interpreter that preloads the
    sanitizer runtime and loads the instrumented build of oracle/*.c."""
    import sys
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_build/libs2oracle_san.so"], stdout=subprocess.DEVNULL)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               S2O_LIB=os.path.join(ROOT, "oracle", "_build", "libs2oracle_san.so"))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_known_answers.py"), "-x", "-q",
                          "-p", "no:cacheprovider", "-k", "not sin_table_against_reference and not phased_offset"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    code = ("import numpy as np\nfrom oracle import s2o\n"
            "s = s2o.OracleSynth(70)\n"
            "rng = np.random.RandomState(3)\n"
            "for b in range(6):\n"
            "    for k in range(25):\n"
            "        (s.note_on if rng.rand() < 0.6 else s.note_off)(int(rng.randint(40, 80)))\n"
            "    pv = s.render_voices(1000 if b == 3 else 1024, 48000, threads=3)\n"
            "    s2o.mix_tree(pv, 64, 2); s2o.mix_sequential(pv)\n"
            "print('render ok')\n")
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "render ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])
