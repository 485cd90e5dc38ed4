"""s2r_math.h on the device against the same header on the host.  libm_xcheck pins the HOST build of
these routines to the host libm over all 2^32 inputs; this closes the remaining hop — the gfx950
build of the same source (different compiler back end, different fma/contract decisions if the flags
were ever wrong) — over every 257th bit pattern of every routine."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_device_math_matches_host_math(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "dev_math_check")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           "-Wno-unused-value", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "synth2_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "dev_math_check.hip"), "-o", exe])
    out = subprocess.run([exe, "257"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    for name in ("sinf", "cosf", "tanf", "expf", "pow2_libm", "pow2_sleef"):
        assert any(ln.startswith(name) and " 0 device/host mismatches" in ln for ln in out.stdout.splitlines()), out.stdout
