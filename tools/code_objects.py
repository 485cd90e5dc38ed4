#!/usr/bin/env python3
"""Pulls the gfx950 code objects out of libs2r.so (clang offload bundles in .hip_fatbin) and disassembles them.
Used by the CPU tests that check properties of the generated ISA and by hand:  python tools/code_objects.py [lib] > all.s"""
import os
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def extract(lib_path):
    """-> list of ELF images (bytes) for amdgcn gfx950"""
    data = open(lib_path, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            break
        pos = i + len(MAGIC)
        (n,) = struct.unpack_from("<Q", data, pos)
        if n == 0 or n > 64:
            continue
        p = pos + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode("ascii", "replace")
            p += 24 + tlen
            if "gfx950" in triple and size:
                out.append(data[i + off:i + off + size])
    return out


def disassemble(lib_path):
    texts = []
    for k, elf in enumerate(extract(lib_path)):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(elf)
            name = f.name
        try:
            texts.append(subprocess.run([OBJDUMP, "-d", name], capture_output=True, text=True, check=True).stdout)
        finally:
            os.unlink(name)
    return texts


READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def kernel_metadata(lib_path):
    """-> {kernel name: {".private_segment_fixed_size": n, ".sgpr_spill_count": n, ".vgpr_spill_count": n, ".vgpr_count": n,
    ".sgpr_count": n, ".group_segment_fixed_size": n}} from the code objects' amdhsa notes"""
    import re
    out = {}
    for elf in extract(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(elf)
            name = f.name
        try:
            notes = subprocess.run([READELF, "--notes", name], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(name)
        # one "- .agpr_count:" item per kernel; scalar keys are "    .key:   value" lines at the item's indent
        for item in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
            kv = dict(re.findall(r"\n\s{4}(\.[a-z_]+):\s+(\S+)", item))
            if ".name" in kv:
                out[kv[".name"]] = {k: int(v) for k, v in kv.items() if v.isdigit()}
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "synth2_amd", "libs2r.so")
    for t in disassemble(lib):
        sys.stdout.write(t)
