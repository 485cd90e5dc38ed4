#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh output directory: per-kernel average duration from the
kernel trace and per-dispatch averages of every PMC counter, as one small text/JSON file
suitable for committing under profiles/.

usage: tools/summarize_prof.py gpurun_out/prof_<tag> [kernel-substring] > profiles/rNN/<tag>.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench      # noqa: E402
SETUP = int(os.environ.get("S2R_PROF_SETUP", str(bench.C3_SETUP)))      # bench.py's untimed set-up buffers (workload c3)


def newest(pattern):
    """gpurun merges a run's files INTO the local directory: an earlier run of the same tag leaves its files beside the new
    ones.  Only the newest file of a kind is the run being summarised."""
    files = glob.glob(pattern, recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []


def main():
    root = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "s2r_render_kernel"
    out = {"source": root, "kernel_filter": want}
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        out["kernel_source_hash"] = bench.kernel_source_hash()     # bench.py quotes these counters only for this build
    except Exception:
        pass
    for f in newest(os.path.join(root, "trace", "**", "*kernel_stats.csv")):
        rows = list(csv.DictReader(open(f)))
        out["kernel_stats"] = [{"name": r["Name"][:120], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                "pct": float(r["Percentage"])} for r in rows[:8]]
    for f in newest(os.path.join(root, "trace", "**", "*kernel_trace.csv")):
        rows = [r for r in csv.DictReader(open(f)) if want in r["Kernel_Name"]]
        if rows:
            rows.sort(key=lambda r: int(r["Start_Timestamp"]))
            # the resources of the kernel the TIMED steps run (bench.py's launch order below), not of whichever variant
            # of the template happens to be first or last in the trace
            w0 = SETUP + int(os.environ.get("S2R_PROF_WARMUP", "4"))
            r = rows[min(w0, len(rows) - 1)]
            out["dispatch"] = {k: r[k] for k in ("Kernel_Name", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count",
                                                  "SGPR_Count", "Workgroup_Size_X", "Grid_Size_X") if k in r}
            out["dispatch_per_kernel_name"] = {}
            for q in rows:
                out["dispatch_per_kernel_name"].setdefault(q["Kernel_Name"][:100], {k: q[k] for k in ("LDS_Block_Size", "Scratch_Size", "VGPR_Count", "SGPR_Count") if k in q})
            d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
            out["dispatch"]["avg_ns"] = sum(d) / len(d)
            out["dispatch"]["n"] = len(d)
            # bench.py launches the render kernel in a fixed order: the untimed set-up (one period + 2 buffers for workload
            # c3), warm-up, the K timed steps, min(K, 64) steps through the synchronous host API, 16 launches timed with HIP
            # events (-> roofline.kernel_ms), 16 more with every shortcut off (-> roofline_valu.kernel_ms); the same slices
            # of the trace, for comparison
            w = SETUP + int(os.environ.get("S2R_PROF_WARMUP", "4")); k = int(os.environ.get("S2R_PROF_STEPS", "16"))
            m = min(max(k, 4), 16); sy = min(k, 64)
            def avg(a, b):
                seg = d[a:b]
                return sum(seg) / len(seg) if seg else None
            out["dispatch"]["phases_avg_ns"] = {"setup_and_warmup": avg(0, w), "timed_steps": avg(w, w + k),
                                                "host_api_sync_steps": avg(w + k, w + k + sy),
                                                # (each of the two HIP-event loops primes one launch: two fills in flight)
                                                "kernel_ms_launches": avg(w + k + sy + 1, w + k + sy + m + 1),
                                                "kernel_ms_all_in_lane_launches": avg(w + k + sy + m + 2, w + k + sy + 2 * m + 2)}
            out["dispatch"]["avg_ns"] = avg(w, w + k)            # the timed steps
    # PMC passes: one row per dispatch and counter; the same launch order as the trace, so the same
    # slices.  `pmc_avg_per_dispatch` is the TIMED-STEPS phase (steady state); the warm-up launches,
    # where every group streams coefficients, are reported separately.
    w = SETUP + int(os.environ.get("S2R_PROF_WARMUP", "4")); k = int(os.environ.get("S2R_PROF_STEPS", "16"))
    counters = defaultdict(list)
    for sub in ("pmc1", "pmc2", "pmc3", "pmc4", "pmc_fetch", "pmc_write"):
        for f in newest(os.path.join(root, sub, "**", "*counter_collection.csv")):
            rows = [r for r in csv.DictReader(open(f)) if want in r.get("Kernel_Name", "")]
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            per = defaultdict(list)
            for r in rows:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for name, vals in per.items():
                counters[name] = vals
    def mean(v):
        return sum(v) / len(v) if v else None
    out["pmc_avg_per_dispatch"] = {n: mean(v[w:w + k]) for n, v in sorted(counters.items())}
    out["pmc_avg_first_launches"] = {n: mean(v[:8]) for n, v in sorted(counters.items())}
    out["pmc_dispatches"] = {n: len(v) for n, v in sorted(counters.items())}
    pm = out["pmc_avg_per_dispatch"]
    if pm.get("SQ_INSTS_VALU_FLOPS_FP32") is not None:
        # SQ_INSTS_VALU_FLOPS_FP32 counts flops per LANE per wave-instruction (add / mul 1, fma 2, a packed instruction its two
        # halves): checked against the per-type counters of the same launches — ADD + MUL + 2 FMA is its value without the
        # packed instructions' second halves.  x 64 lanes = flops; integer and conversion instructions count 1 per lane.
        lanes = 64.0
        fp = (pm["SQ_INSTS_VALU_FLOPS_FP32"] + (pm.get("SQ_INSTS_VALU_FLOPS_FP32_TRANS") or 0.0) + (pm.get("SQ_INSTS_VALU_FLOPS_FP64") or 0.0)) * lanes
        other = ((pm.get("SQ_INSTS_VALU_IOPS") or 0.0) + (pm.get("SQ_INSTS_VALU_CVT") or 0.0)) * lanes
        unpacked = (pm.get("SQ_INSTS_VALU_ADD_F32") or 0.0) + (pm.get("SQ_INSTS_VALU_MUL_F32") or 0.0) + 2.0 * (pm.get("SQ_INSTS_VALU_FMA_F32") or 0.0)
        out["executed_flop_eq"] = {
            "flop_eq_per_launch": fp + other, "fp32_flops_per_launch": fp, "int_and_cvt_ops_per_launch": other,
            "packed_second_halves_per_lane": pm["SQ_INSTS_VALU_FLOPS_FP32"] - unpacked,
            "model": "64 x (SQ_INSTS_VALU_FLOPS_FP32 [+ _TRANS + FP64] + SQ_INSTS_VALU_IOPS + SQ_INSTS_VALU_CVT) per launch, timed steps"}
    out["pmc_phase"] = "dispatches %d..%d of the render kernel (bench.py's timed steps)" % (w, w + k - 1)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
