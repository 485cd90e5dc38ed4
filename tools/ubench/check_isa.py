#!/usr/bin/env python3
"""Disassembles tools/ubench/_build/issue_rates3 (gfx950 code object) and checks that each kernel's loop holds the
instruction it claims, the right number of times, and nothing the compiler derived from it (no v_pk_* where a scalar
form is measured, no folded constants).  Prints one line per kernel; exit code 1 on a mismatch.
usage: python tools/ubench/check_isa.py [path-to-.out-or-.o]"""
import collections
import re
import subprocess
import sys
import os

HERE = os.path.dirname(os.path.abspath(__file__))
obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "_build", "issue_rates3-hip-amdgcn-amd-amdhsa-gfx950.out")
MNEMONIC = {"fma": "v_fma_f32", "add": "v_add_f32", "mul": "v_mul_f32", "pkfma": "v_pk_fma_f32", "pkmul": "v_pk_mul_f32",
            "pkadd": "v_pk_add_f32", "fract": "v_fract_f32", "cvtu": "v_cvt_f32_u32", "xor": "v_xor_b32", "addu": "v_add_u32",
            "ashr": "v_ashrrev_i32", "mullo": "v_mul_lo_u32", "pkmullo16": "v_pk_mul_lo_u16", "pkaddu16": "v_pk_add_u16",
            "cndmask": "v_cndmask_b32", "cmp": "v_cmp_lt_f32", "rcp": "v_rcp_f32", "fma64": "v_fma_f64", "readlane": "v_readlane_b32", "max3": "v_max3_f32", "salu": "s_add_u32", "fma_salu": "s_add_u32",
            "cmp_sor": "v_cmp_eq_f32", "cnd_sgpr": "v_cndmask_b32", "ldsw": "ds_write_b32", "fma_ldsw": "ds_write_b32",
            "fma3_ldsw": "ds_write_b32", "fma_pkfma": "v_pk_fma_f32", "ldsw2": "ds_write2_b32", "ldsw64": "ds_write_b64",
            "ldsw128": "ds_write_b128", "ldswtid": "ds_write_addtid_b32", "fma3_ldswtid": "ds_write_addtid_b32",
            "ldsr": "ds_read_b32", "ldsr2": "ds_read2_b32", "ldsr64": "ds_read_b64", "ldsr128": "ds_read_b128",
            "fma3_ldsr128": "ds_read_b128"}
UNROLL = 16
dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", obj], capture_output=True, text=True, check=True).stdout
kern = None
counts = collections.defaultdict(collections.Counter)
for line in dis.splitlines():
    m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
    if m:
        kern = m.group(1)
        continue
    m = re.match(r"^\s+(\S+)", line)
    if kern and m:
        counts[kern][m.group(1)] += 1
bad = 0
for k in sorted(counts):
    m = re.search(r"ub_(\w+)_ilp(\d+)", k)
    if not m:
        continue
    name, ilp = m.group(1), int(m.group(2))
    want = UNROLL * ilp
    # encodings carry a suffix (_e32, _e64, _sdwa, _dpp); the compiler may unroll the outer loop (x2, x4); the epilogue
    # (sum of the chains, cvt of y.x) may add a few of the same mnemonic
    got = sum(c for mn, c in counts[k].items() if mn == MNEMONIC[name] or mn.startswith(MNEMONIC[name] + "_e"))
    if name.startswith("lds") or "_lds" in name:
        got = counts[k][MNEMONIC[name]]
    ok = any(want * f <= got <= want * f + 2 * ilp + 2 for f in (1, 2, 4))
    others = sum(c for mn, c in counts[k].items() if mn.startswith("v_pk_") and not MNEMONIC[name].startswith("v_pk_"))
    ok = ok and others <= 2 * ilp + 2    # nothing packed behind the measured scalar form (the epilogue sum may pack a few)
    print("%-28s %-18s in loop: want %3d, disassembly has %3d  %s" % (k[:28], MNEMONIC[name], want, got, "ok" if ok else "MISMATCH"))
    bad += not ok
sys.exit(1 if bad else 0)
