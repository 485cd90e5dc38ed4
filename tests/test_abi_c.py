"""The C ABI from C: include/s2r.h compiled as C99 with -pedantic -Werror, linked against libs2r.so,
exercising the entry points that need no GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_header_is_c99_and_cpu_entry_points_link(tmp_path):
    import synth2_amd
    synth2_amd.load_library()                       # builds libs2r.so if needed
    lib_dir = os.path.join(ROOT, "synth2_amd")
    exe = str(tmp_path / "abi_check")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "abi_check.c"), "-o", exe,
                           "-L", lib_dir, "-ls2r", "-Wl,-rpath," + lib_dir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert out.stdout.startswith("abi ok")
