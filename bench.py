#!/usr/bin/env python3
"""bench.py — voice-render throughput on MI355X (BASELINE.json metric; SURVEY.md §8(d) is the definition).

    python bench.py --gpus N --steps K --warmup W

A *step* is one 1024-frame buffer of the whole voice pool: the buffer's note events are handed over
(`s2r_note_events`), the fill renders every voice's 1024 frames (oscillator -> envelopes -> LPF) and mixes them down, and
the 4 KiB mix is copied to (pinned) host memory — every buffer, inside the timed region, like the events' H2D.  The steps
run through the host-buffer API with TWO buffers in flight (`s2r_fill_begin` / `s2r_fill_end`: buffer k is queued, then
buffer k - 1 is waited for and copied into the caller's memory — the arrangement s2_bin itself uses between its synth and
audio threads, audio_player.rs:56-60; the library then runs the previous buffer's mix and the next buffer's event preparation on a
second stream beside the render kernels), one fence before and one after the K steps as the bench contract prescribes; the
strictly one-at-a-time form (`s2r_fill`: the caller waits for every buffer before it hands over the next events) is timed
on the same workload and printed as `value_host_api_sync`.

Workload `c3` (default; config.workload names it): SURVEY §8(d)'s C3 — 65 536 voices per GPU, the reference's own patch
(`example.synth2`, empty body == Synth::default_config: saw + amp/mod ADSR + the one-pole LPF; the reference has no
SVF), `note[v] = 36 + (v mod 61)`, note-off `16 * (512 + lcg(v) mod 2048)` frames after the note-on (LCG seeded by v) —
made PERIODIC so that the envelope-stage mix is stationary and the result does not depend on --steps / --warmup: a voice
lives one C3 life (on -> LCG note-off -> release -> silent) every 64 buffers, the lives start 1/64 of the pool per
buffer, in index order (the allocation policy — oldest voice, lowest index — then restarts exactly the voices whose
turn it is).  Note-ons land on buffer starts, note-offs on their 16-frame boundary INSIDE the buffer (timed events, the
granularity s2_bin's loop has, main.rs:138-143).  Before the warm-up the population is aged by six whole periods
(untimed set-up).  Workload `churn` is round 1's: everything on at frame 0, then 128 note-offs + 128 note-ons per
buffer per 64 k voices.

N GPUs: the pool is N x as large (weak scaling; `--voices-total` fixes the pool instead: strong scaling), dealt out to
the ranks in runs of 64 voices; every rank sees the same event stream; per buffer one all-gather of the 4 KiB partial
mixes over RCCL and a rank-ordered sum on rank 0, whose result is copied to host memory every buffer.

`value` = voice-samples/s = voices x frames x steps / wall time of the K timed steps (max over ranks).
Extra objects on the JSON line: `roofline` (the render kernel against the HBM roof, as BASELINE's north_star asks — this
path is NOT HBM-bound), `roofline_valu` (the bound that applies), `cpu_baseline` (+ `cpu_baseline_legs`:
the CPU oracle, a C restatement of s2_lib, timed on this host on bounded samples; rank 0, N = 1 only).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

# (a device list rehearsed on ONE device keeps a resident kernel per shard, each on a stream of its own: a hardware queue for
# every stream; read by the HIP runtime at its first call.  N devices need nothing: s2r.h, s2r_set_resident)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR = 48000
FRAMES = 1024
C3_SETUP = 6 * 64 + 2      # untimed set-up buffers of workload c3 (tools/summarize_prof.py maps launches to phases with it)
PERIOD = 64                     # buffers between two lives of a voice (workload c3)
# algorithmic HBM bytes per started voice per fill: read pitch, offset, release, flags, phase,
# lpf_last, seed (7 x 4 B) and write offset, phase, lpf_last (3 x 4 B)        
BYTES_PER_VOICE_FILL = 28 + 12
# fp32-equivalent flops per voice-sample of the x16 path with the default patch
FLOPS_PER_VOICE_SAMPLE = 250.0
PROFILE_ROUND = "r04"                 # profiles/<round>/c3_summary.json: the rocprofv3 summary bench.py quotes counters from
HBM_PEAK_GBS = 8000.0
VALU_PEAK_TFLOPS = 157.3
# one wave-instruction per SIMD per 2 cycles (SIMD-32, MI355X_MICROARCH.md): 1024 SIMDs x 2.4 GHz / 2
VALU_ISSUE_PEAK_PER_S = 1024 * 2.4e9 / 2.0


def lcg(x):
    return (1103515245 * x + 12345) % (1 << 31)


def make_events(total_voices, churn_per_64k, step, rng_seed=1):
    """workload `churn`: `churn` note-offs then `churn` note-ons over the whole pool (voice stealing picks the oldest
    voice), notes 36..96, all at the buffer's start."""
    import synth2_amd as s2
    n = max(1, total_voices * churn_per_64k // 65536)
    ev = np.zeros(2 * n, dtype=s2.NOTE_EVENT_DTYPE)
    x = lcg(rng_seed * 7919 + step)
    for i in range(n):
        x = lcg(x)
        ev["kind"][i] = 0
        ev["note"][i] = 36 + x % 61
    for i in range(n):
        x = lcg(x)
        ev["kind"][n + i] = 1
        ev["note"][n + i] = 36 + x % 61
        ev["velocity"][n + i] = 1.0
    return ev


def make_c3_events(total_voices, period=PERIOD, frames=FRAMES):
    """workload `c3`: the event batch of every buffer of one period (the schedule repeats every `period` buffers).
    Buffer b starts the lives of voices [b * V / period, (b + 1) * V / period) at its frame 0 and carries the note-offs
    that fall inside it, each at its own 16-frame boundary."""
    import synth2_amd as s2
    v = np.arange(total_voices, dtype=np.int64)
    note = (36 + v % 61).astype(np.uint8)
    delay = 16 * (512 + ((1103515245 * v + 12345) % (1 << 31)) % 2048)          # frames from note-on to note-off
    per = max(1, total_voices // period)
    start_buf = np.minimum(v // per, period - 1)
    off_buf = (start_buf + delay // frames) % period
    off_frame = delay % frames
    batches = []
    for b in range(period):
        on = np.nonzero(start_buf == b)[0]
        off = np.nonzero(off_buf == b)[0]
        off = off[np.argsort(off_frame[off], kind="stable")]
        ev = np.zeros(on.size + off.size, dtype=s2.NOTE_EVENT_DTYPE)
        ev["kind"][:on.size] = 1
        ev["note"][:on.size] = note[on]
        ev["velocity"][:on.size] = 1.0
        ev["kind"][on.size:] = 0
        ev["note"][on.size:] = note[off]
        ev["frame"][on.size:] = off_frame[off]
        if os.environ.get("S2R_BENCH_QUANTIZE") == "1":       # measurement aid: the note-offs moved to the buffer's start
            ev["frame"][:] = 0
        batches.append(ev)      # ordered by frame: the note-ons (frame 0), then the note-offs by their frame
    return batches


def host_cpus():
    """(threads this process may run at once, physical cores among them): the affinity mask cut down to the cgroup's CPU
    quota, and the distinct (package, core) pairs of the CPUs in the mask (SMT siblings count once)."""
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(float(q) / float(per)))
    except Exception:
        pass
    cores = set()
    try:
        cur = {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif not line.strip():
                if cur.get("processor") is not None and int(cur["processor"]) in cpus:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur["processor"])))
                cur = {}
    except Exception:
        pass
    physical = len(cores) if cores else len(cpus)
    usable = min(len(cpus), quota) if quota else len(cpus)
    return usable, min(physical, usable)


def one_cpu_per_core(n):
    """n CPU numbers of the affinity mask on n different physical cores (for pinning the CPU baseline's threads: two
    of them on the SMT siblings of one core would share it), or [] when the topology cannot be read"""
    cpus = set(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else set()
    seen, picks, cur = set(), [], {}
    try:
        for line in list(open("/proc/cpuinfo")) + [""]:
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif not line.strip():
                if cur.get("processor") is not None and int(cur["processor"]) in cpus:
                    core = (cur.get("physical id", "0"), cur.get("core id", cur["processor"]))
                    if core not in seen:
                        seen.add(core); picks.append(int(cur["processor"]))
                cur = {}
    except Exception:
        return []
    return picks[:n] if len(picks) >= n else []


def cpu_oracle_rate(voices, buffers, threads, c3=False):
    """The CPU oracle (test infrastructure) timed as the reported CPU baseline: voices x 1024 frames x buffers.
    c3=True: the bench's own event schedule (make_c3_events), driven as the reference's caller drives Synth — events applied
    between 16-frame sample() calls (main.rs:138-147) — after one untimed period; returns the rate over the RENDERING time
    (the reference's O(voices) scans per note event, an artefact of an 8-voice design at this pool size, are timed apart)."""
    from oracle import s2o
    s = s2o.OracleSynth(voices)
    if c3:
        period = PERIOD if voices >= PERIOD else 1
        cyc = make_c3_events(voices, period)
        for k in range(period + 2):                        # one life of every voice: the stage mix the GPU leg is timed on
            s.render_events(cyc[k % period], FRAMES, SR, threads=threads, per_voice=False, mix=True)
        s2o.events_seconds(reset=True)
        t0 = time.perf_counter()
        for k in range(buffers):
            s.render_events(cyc[(period + 2 + k) % period], FRAMES, SR, threads=threads, per_voice=False, mix=True)
        dt = time.perf_counter() - t0
        policy_s, render_s = s2o.events_seconds(reset=True)
        rate = voices * FRAMES * buffers / render_s
        # the same population on ONE thread (two buffers): what the all-cores rate is a multiple of
        one = None
        if threads > 1:
            for k in range(2):
                s.render_events(cyc[(period + 2 + buffers + k) % period], FRAMES, SR, threads=1, per_voice=False, mix=True)
            _p1, r1 = s2o.events_seconds(reset=True)
            one = voices * FRAMES * 2 / r1
        return rate, dt, policy_s, one
    for v in range(voices):
        s.note_on(36 + v % 61)
    s.sample_mt(FRAMES, SR, threads)          # warm-up buffer
    t0 = time.perf_counter()
    for _ in range(buffers):
        s.sample_mt(FRAMES, SR, threads)
    dt = time.perf_counter() - t0
    return voices * FRAMES * buffers / dt, dt, 0.0, None


def cpu_baseline_legs(budget_s):
    """§8(d): (1) one thread at the reference's own pool size (NUM_VOICES = 8, synth.rs:7) and at 1 024 voices
    (cache-resident), (2) one thread per physical core at C3's 65 536 voices on C3's event schedule.  Each leg is sized to
    about budget_s seconds.  The oracle is rebuilt for THIS host with -march=native first."""
    from oracle import s2o
    s2o.use_native_build()
    usable, physical = host_cpus()
    legs = []
    for voices, threads, c3 in ((8, 1, False), (1024, 1, False), (65536, physical, True)):
        if c3:
            probe = 1.6e7 * threads                        # (an estimate sizes the leg: a probe would cost a whole period)
        else:
            probe = cpu_oracle_rate(voices, 1, threads)[0]
        nb = int(max(1, min(20000, budget_s * probe / (voices * FRAMES))))
        pins = one_cpu_per_core(threads) if threads > 1 else []
        old_mask = os.sched_getaffinity(0) if pins else None
        if pins:
            s2o.pool_pin(pins); os.sched_setaffinity(0, {pins[0]})
        try:
            v, secs, policy_s, one_thread = cpu_oracle_rate(voices, nb, threads, c3)
        finally:
            if pins:
                s2o.pool_pin([]); os.sched_setaffinity(0, old_mask)
        legs.append({"value": v, "unit": "samples/s", "cores": threads, "kind": "port", "voices": voices,
                     "physical_cores_available": physical, "hardware_threads_available": usable, "threads_pinned_one_per_core": bool(pins),
                     "ns_per_voice_sample_per_core": 1e9 * threads / v,
                     "sample": ("%d voices x %d frames x %d buffers, default patch, %s, %.1f s%s; oracle built -O2 -march=native -ffp-contract=off" % (
                         voices, FRAMES, nb,
                         "the bench's C3 schedule after one untimed period, events applied between 16-frame sample() calls" if c3 else "all notes held (amp sustain)",
                         secs, (" of which %.1f s in the reference's O(voices) note_on / note_off scans, excluded from the rate" % policy_s) if c3 else ""))})
        if c3 and one_thread:
            legs[-1]["one_thread_same_population"] = one_thread
            legs[-1]["scaling_vs_one_thread"] = v / (one_thread * threads)
    return legs


def small_fill_leg():
    """compiles tools/ubench/small_fill.cpp against the in-tree library (g++, two seconds) and runs it as a child process: median and
    p99 wall time of s2r_fill(16 frames) for 8 voices over 20 000 calls with note events in between, both ways"""
    import shutil
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    src = os.path.join(here, "tools", "ubench", "small_fill.cpp")
    exe = os.path.join(here, "tools", "ubench", "_build", "small_fill")
    gxx = shutil.which("g++")
    if not gxx or not os.path.exists(src):
        return {"error": "no g++ or no source"}
    try:
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        libdir = os.path.join(here, "synth2_amd")
        subprocess.run([gxx, "-O2", "-std=c++17", "-I", os.path.join(here, "include"), src, "-o", exe, "-L", libdir, "-ls2r", "-Wl,-rpath," + libdir],
                       check=True, capture_output=True, timeout=120)
        r = subprocess.run([exe, "--json"], check=True, capture_output=True, timeout=120, text=True)
        leg = json.loads(r.stdout.strip().splitlines()[-1])
        leg["caller"] = "C++ over the C ABI (tools/ubench/small_fill.cpp), median of the calls' wall times"
        leg["real_time_budget_us"] = round(16 / SR * 1e6, 1)
        return leg
    except Exception as e:                                        # (a measurement leg: never the bench's failure)
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def run_config_leg(name, voices, patch_text, steps, warmup, oversampled=False, bytes_per_voice=BYTES_PER_VOICE_FILL):
    """One more configuration of BASELINE.json timed the way the headline is (N = 1, after it, so that it cannot perturb
    it): the C3 event schedule on `voices` voices of `patch_text`, aged for two periods, `warmup` + `steps` buffers through
    the host-buffer API.  oversampled: s2r_fill_oversampled (config [4]: the path at 4 x 48 kHz — 4 096 internal frames per
    1 024-frame buffer, note-offs on the internal rate's 16-frame boundaries — and the decimator), one buffer at a time."""
    import synth2_amd as s2
    internal = FRAMES * (4 if oversampled else 1)
    synth = s2.Synth(voices, max_frames=internal)
    synth.load_patch(patch_text)
    period = PERIOD
    cyc = make_c3_events(voices, period, internal)
    out = np.empty(FRAMES, dtype=np.float32)
    state = {"k": 0, "in_flight": 0}

    def step():
        synth.note_events(cyc[state["k"] % period]); state["k"] += 1
        if oversampled:
            out[:] = synth.sample_oversampled(FRAMES, SR)
            return
        synth.sample_begin(FRAMES, SR); state["in_flight"] += 1
        if state["in_flight"] == 2:
            synth.sample_end(out); state["in_flight"] -= 1

    def drain():
        while state["in_flight"]:
            synth.sample_end(out); state["in_flight"] -= 1

    for _ in range(2 * period + 2 + warmup):
        step()
    drain()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    drain()
    dt = time.perf_counter() - t0
    synth.set_timing(True)
    kms = []
    if not oversampled:                                           # (two fills in flight, as in the timed steps: bench main's kernel_ms_loop)
        synth.note_events(cyc[state["k"] % period]); state["k"] += 1
        synth.sample_begin(FRAMES, SR)
    for _ in range(min(max(steps, 4), 16)):
        synth.note_events(cyc[state["k"] % period]); state["k"] += 1
        if oversampled:
            synth.sample_oversampled(FRAMES, SR)
        else:
            synth.sample_begin(FRAMES, SR); synth.sample_end(out)
        kms.append(synth.last_render_ms())
    if not oversampled:
        synth.sample_end(out)
    synth.set_timing(False)
    kernel_ms = float(np.median(kms))
    hbm = bytes_per_voice * voices / (kernel_ms * 1e-3) / 1e9
    leg = {"name": name, "voices": voices, "patch": " ".join(patch_text.split()), "steps": steps,
           "ms_per_step": dt * 1e3 / steps, "value": voices * FRAMES * steps / dt, "unit": "samples/s (output-rate voice-samples)",
           "kernel_ms": kernel_ms, "value_kernel_only": voices * FRAMES / (kernel_ms * 1e-3),
           "roofline": {"bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_voice_per_fill": bytes_per_voice},
           "mix_checksum": float(np.abs(out).sum())}
    # what the leg's render kernel EXECUTED (committed rocprofv3 summary of this build: tools/profile_leg.sh): issue and fp32 fractions
    leg["roofline_valu"] = executed_roofline(committed_profile("c4_summary.json" if oversampled else "c2_summary.json"), voices * (4 if oversampled else 1), kernel_ms)
    if oversampled:
        leg["internal_rate_voice_samples_per_s"] = leg["value"] * 4.0
        leg["timed_call"] = "s2r_note_events + s2r_fill_oversampled (4 096 internal frames at 192 kHz -> 1 024 at 48 kHz), one buffer at a time"
    else:
        leg["timed_call"] = "s2r_note_events + s2r_fill_begin / s2r_fill_end, two buffers in flight"
    return leg, synth


def executed_roofline(prof, voices, kernel_ms):
    """The VALU roofline of the launches bench.py times, from what they EXECUTED (committed rocprofv3 summary of this
    build of the kernels: tools/profile_gpu.sh + tools/summarize_prof.py): wave-instructions against the issue rate, and
    flop-equivalents (add / mul / other 1, fma 2, transcendental 1; a packed instruction counts for its two halves)
    against the fp32 vector peak."""
    if prof is None:
        return None
    pm = prof.get("pmc_avg_per_dispatch", {})
    dur = prof.get("dispatch", {}).get("avg_ns")
    if not dur or "SQ_INSTS_VALU" not in pm:
        return None
    n = pm["SQ_INSTS_VALU"]
    out = {"bound": "valu-issue", "achieved": n / (dur * 1e-9), "peak": VALU_ISSUE_PEAK_PER_S, "unit": "wave-instructions/s",
           "frac": n / (dur * 1e-9) / VALU_ISSUE_PEAK_PER_S,
           "kernel": prof.get("dispatch", {}).get("Kernel_Name"), "kernel_ns_profiled": dur, "kernel_ms_live": kernel_ms,
           "wave_instructions_per_launch": n, "per_voice_frame": n * 64.0 / (voices * FRAMES),
           "note": "SQ_INSTS_VALU per launch of the TIMED steps / launch duration vs one wave-instruction per SIMD per 2 cycles (1024 SIMDs x 2.4 GHz / 2)"}
    ex = prof.get("executed_flop_eq")
    if ex:
        tf = ex["flop_eq_per_launch"] / (dur * 1e-9) / 1e12
        out["executed"] = {"flop_eq_per_launch": ex["flop_eq_per_launch"], "flop_eq_per_voice_sample": ex["flop_eq_per_launch"] / (voices * FRAMES),
                           "achieved_tflops": tf, "peak_tflops": VALU_PEAK_TFLOPS, "frac": tf / VALU_PEAK_TFLOPS, "model": ex.get("model")}
    return out


def kernel_source_hash():
    """identifies the build a committed profile belongs to (the GPU box has no .git): synth2_amd/build.py's source_hash, the
    first half of what s2r_build_id() of a library built from these sources returns"""
    from synth2_amd import build as _b
    return _b.source_hash()


def committed_profile(name="c3_summary.json"):
    """PMC figures of the render kernel from the committed rocprofv3 summary — only if it was taken on THIS build of
    the kernels (same source hash); otherwise None (a stale profile is not evidence for the line being printed)."""
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND, name)))
    except Exception:
        return None
    if prof.get("kernel_source_hash") != kernel_source_hash():
        return None
    return prof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", choices=["c3", "churn"], default="c3")
    ap.add_argument("--voices-per-gpu", type=int, default=65536)
    ap.add_argument("--voices-total", type=int, default=0, help="fix the pool (strong scaling) instead of the per-GPU share")
    ap.add_argument("--churn", type=int, default=128, help="workload churn: note-ons (and note-offs) per step per 65536 voices")
    ap.add_argument("--block-voices", type=int, default=0)
    ap.add_argument("--no-overlap", action="store_true", help="do not overlap the all-gather with the next render")
    ap.add_argument("--reduce", action="store_true", help="N > 1: combine the partial mixes with one reduce(sum) to rank 0 instead of all-gather + rank-ordered sum")
    ap.add_argument("--device-list", default="", help="ONE process over these devices (comma-separated ordinals, e.g. 0,1,2,3 or 0,0 on a one-GPU box): "
                    "the C ABI's device-list handle instead of one process per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the SVF and 4x-oversampled configurations timed after the headline")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="seconds per CPU-baseline leg (three legs)")
    ap.add_argument("--watchdog", type=float, default=float(os.environ.get("S2R_BENCH_WATCHDOG", "600")),
                    help="seconds after which a run that has not finished dumps every thread's stack and exits (0: never): a rank stuck in a "
                         "collective or a device call must end the job, not hold the box")
    args = ap.parse_args()
    if args.watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)

    import torch
    import torch.distributed as dist
    import synth2_amd as s2

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    # Rehearsal on a one-GPU box (S2R_BENCH_BACKEND=gloo S2R_BENCH_SHARE_GPU=1): the ranks share the card and
    # the 4 KiB partial rows travel through the host over gloo.  Everything but the RCCL collective itself is
    # the code an N-GPU run executes; the numbers of such a run mean nothing.
    backend = os.environ.get("S2R_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if os.environ.get("S2R_BENCH_SHARE_GPU") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # The CPU baseline runs FIRST (rank 0 of N = 1): it is reported beside the GPU number, and a box that has been idle
    # hands the first process a package in its lowest power state — measured, the host side of the timed steps
    # (s2r_note_events) then runs 2x slower and the step is host-bound.  Twenty seconds of all cores busy settle that.
    cpu_legs = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu_legs = cpu_baseline_legs(args.cpu_seconds)

    strong = args.voices_total > 0
    if strong:
        if args.voices_total % (64 * world):
            sys.exit("--voices-total must be a multiple of 64 x the number of GPUs")
        vpg = args.voices_total // world
    else:
        vpg = args.voices_per_gpu
    total = vpg * world
    from synth2_amd.sharded import ShardedSynth
    dev_list = [int(x) for x in args.device_list.split(",")] if args.device_list else None
    if dev_list:
        if world != 1:
            sys.exit("--device-list is the one-process form: do not launch it under torch.distributed.run")
        total = vpg * len(dev_list)
    def make_sharded():
        sh_ = ShardedSynth(vpg, max_frames=FRAMES, rank=rank, world=world, device=dev, block_voices=args.block_voices,
                           overlap=not args.no_overlap, reduce_to_root=args.reduce, devices=dev_list)
        sh_.load_patch("synth mySynth {\n\n}\n")       # example.synth2: empty body == default patch
        return sh_
    sh = make_sharded()
    synth = sh.renderer
    block_voices = synth.block_voices
    # N > 1 (one process per GPU): the partial rows are exchanged INSIDE the render kernels through a block of rank 0's device
    # memory (s2r_exchange_create / _attach: no collective, no torch call per step); S2R_BENCH_EXCHANGE=torch keeps the
    # all-gather of round 3.  The handle travels once, at start-up, over torch.distributed.  This form has never run on two
    # or more DEVICES (its authors had one): it is tried first — set-up, then one silent fill on every rank — and if any rank
    # reports a failure every rank goes back to the torch exchange with a fresh handle, and the line says so.
    use_xg = world > 1 and not dev_list and os.environ.get("S2R_BENCH_EXCHANGE", "ipc") == "ipc"
    xg_note = None
    if use_xg:
        ok, why = 1, ""
        handle = None
        if rank == 0:
            try:
                handle = synth.exchange_create(world)
            except Exception as e:                      # noqa: BLE001
                ok, why = 0, "s2r_exchange_create: %s" % e
        objs = [handle]
        dist.broadcast_object_list(objs, src=0)
        if rank != 0:
            if objs[0] is None:
                ok = 0
            else:
                try:
                    synth.exchange_attach(rank, world, objs[0])
                except Exception as e:                  # noqa: BLE001
                    ok, why = 0, "s2r_exchange_attach: %s" % e
        dist.barrier()
        if ok:
            try:                                        # silent fills through the exchange on every rank: one synchronous, then
                probe = np.empty(FRAMES, dtype=np.float32)          # two pairs in flight (both rows slots, both ring slots)
                synth.sample(probe, SR)
                for _ in range(2):
                    synth.sample_begin(FRAMES, SR); synth.sample_begin(FRAMES, SR)
                    synth.sample_end(probe); synth.sample_end(probe)
                if not np.all(np.isfinite(probe)):
                    raise RuntimeError("the probe fills came back non-finite")
            except Exception as e:                      # noqa: BLE001
                ok, why = 0, "first fills through the exchange: %s" % e
        t_ok = torch.tensor([ok], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
        if int(t_ok.item()) == 0:
            whys = [None] * world
            dist.all_gather_object(whys, why)
            xg_note = "the in-kernel exchange failed on this node (%s): fell back to all-gather over RCCL" % "; ".join(w for w in whys if w)
            use_xg = False
            del synth, sh
            sh = make_sharded()
            synth = sh.renderer
    host_api = world == 1 or use_xg               # the step is s2r_note_events + s2r_fill_begin / s2r_fill_end
    # The pool-resident render kernel (s2r_set_resident: a fill is a posted command, no launch).  On by default where the host's
    # share of a step decides (N > 1: every rank resolves the whole pool's events); at N = 1 the launches' two-stream form is
    # as fast on the GPU side and is what is timed (S2R_BENCH_RESIDENT=1 / 0 force either: tools/ab_modes.sh).
    # (The rehearsal in which the ranks SHARE one card is only sound while all their grids fit the card together: a fill's kernels
    # wait for one another inside the launch — bounded, 50 ms — on the understanding that the handle's grid is alone on its device,
    # which one process per GPU guarantees and two processes on one card do not; with grids that do not fit, the waits run out.)
    if os.environ.get("S2R_BENCH_SHARE_GPU") == "1" and world * (vpg // max(1, block_voices)) > torch.cuda.get_device_properties(dev).multi_processor_count:
        sys.exit("S2R_BENCH_SHARE_GPU=1: %d ranks x %d workgroups do not fit this card's %d compute units together; rehearse with a smaller "
                 "--voices-per-gpu (tests/test_bench_rehearsal.py uses 8192)" % (world, vpg // max(1, block_voices), torch.cuda.get_device_properties(dev).multi_processor_count))
    resident = host_api and os.environ.get("S2R_BENCH_RESIDENT", "1" if world > 1 else "0") != "0"
    if resident:
        synth.set_resident(True)
    if os.environ.get("S2R_COEFF_STREAM_MODE"):       # measurement aid (see s2r_set_coeff_stream); results are bit-identical
        synth.set_coeff_stream(int(os.environ["S2R_COEFF_STREAM_MODE"]))

    # ---- the event batches (every rank sees the same stream) ----
    if args.workload == "c3":
        period = PERIOD if total >= PERIOD else 1
        cyc = make_c3_events(total, period)
        events_of = lambda k: cyc[k % period]
        # six periods: after the first every voice has lived once and the stage mix is stationary; the rest is ~25 ms of
        # back-to-back launches so that the timed steps (the driver times 20 of them: 1.3 ms) see settled clocks and caches
        n_setup = (C3_SETUP - 2) // PERIOD * period + 2
    else:
        init = np.zeros(total, dtype=s2.NOTE_EVENT_DTYPE)
        init["kind"] = 1
        init["note"] = 36 + (np.arange(total) % 61)
        init["velocity"] = 1.0
        sh.note_events(init)
        churn_cache = {}
        def events_of(k):
            if k not in churn_cache:
                churn_cache[k] = make_events(total, args.churn, k)
            return churn_cache[k]
        n_setup = 12                              # past the initial 9 600-frame mod decay of the whole pool
    n_events_per_step = float(np.mean([events_of(k).size for k in range(n_setup, n_setup + max(1, min(args.steps, 64)))]))
    for k in range(n_setup + args.warmup + args.steps + 80):
        events_of(k)                              # generated outside the timed region

    out_host = np.empty(FRAMES, dtype=np.float32)
    pinned = torch.empty(FRAMES, dtype=torch.float32).pin_memory()
    if rank == 0:
        sh.copy_mix_to(pinned)                    # every finished mix is copied to host memory behind its combine (async D2H)

    in_flight = [0]

    host_t = [0.0, 0.0, 0.0]                      # seconds inside s2r_note_events / s2r_fill_begin / s2r_fill_end (N = 1)

    def step(k):
        if host_api:
            ta = time.perf_counter()
            sh.note_events(events_of(k))
            tb = time.perf_counter()
            synth.sample_begin(FRAMES, SR)
            tc = time.perf_counter()
            in_flight[0] += 1
            if in_flight[0] == 2:
                synth.sample_end(out_host)
                in_flight[0] -= 1
            td = time.perf_counter()
            host_t[0] += tb - ta; host_t[1] += tc - tb; host_t[2] += td - tc
            return
        sh.note_events(events_of(k))
        sh.fill(FRAMES, SR)

    def fence():
        while in_flight[0]:
            synth.sample_end(out_host)
            in_flight[0] -= 1
        sh.flush()
        synth.quiesce()                           # (a resident kernel would sit out its patience inside a device-wide synchronize)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    k0 = 0
    for k in range(n_setup):                      # untimed set-up: age the population
        step(k)
    k0 += n_setup
    for k in range(k0, k0 + args.warmup):
        step(k)
    k0 += args.warmup
    # The timed steps measure the path in the state the workload defines — the GPU the bottleneck, the host a fill ahead.  A box that
    # has only just come up can run its HOST side 2-3x slower for its first minute (measured on this pool: s2r_note_events 41 us
    # instead of 14, the oracle's CPU legs 18 % down, the step 0.079 ms and host-bound; the next run on the same box 0.0457 ms): the
    # untimed phase goes on — blocks of 64 more warm-up steps, a quarter of a second apart — until the host's own share of a step
    # (s2r_note_events + s2r_fill_begin) is below 0.62 of the step (0.49-0.52 normally; 0.7 and more when it is the bottleneck), or
    # for 45 s at most, and the line says what it did.  S2R_BENCH_SETTLE=0 switches it off (the profiling scripts do: they count
    # launches).
    settle = None
    if host_api and world == 1 and os.environ.get("S2R_BENCH_SETTLE", "1") != "0":
        t_settle = time.perf_counter()
        shares = []
        while True:
            host_t[:] = [0.0, 0.0, 0.0]
            t_b = time.perf_counter()
            for k in range(k0, k0 + 64):
                step(k)
            k0 += 64
            shares.append((host_t[0] + host_t[1]) / (time.perf_counter() - t_b))
            if shares[-1] < 0.62 or time.perf_counter() - t_settle > float(os.environ.get("S2R_BENCH_SETTLE_S", "45")):
                break
            time.sleep(0.25)
        for k in range(k0, k0 + args.steps + 80):
            events_of(k)                          # (workload churn builds its batches on demand: never inside the timed region)
        settle = {"extra_warmup_steps": 64 * len(shares), "seconds": time.perf_counter() - t_settle, "host_share_first": shares[0], "host_share_last": shares[-1],
                  "note": "untimed: warm-up went on until the host's share of a step fell below 0.62 (the step GPU-bound, as the workload defines it) or 45 s had passed"}
    fence()
    # ---- timed region: exactly K steps ----
    host_t[:] = [0.0, 0.0, 0.0]
    # The fence's blocking synchronize puts this thread to sleep, and it wakes on a core in a low power state: measured
    # (per-step trace of the three calls), the next ~16 steps — most of the 20 the driver times — then run the host
    # side 1.5-2x slower (s2r_note_events 19 -> 30 us, s2r_fill_begin 12 -> 30 us) and the step becomes host-bound.
    # 5 ms of busy waiting between the fence and the timer bring the core back; the timed region is still exactly K
    # steps with a fence on either side.
    t_spin = time.perf_counter() + float(os.environ.get("S2R_BENCH_PREWAIT_MS", "5")) * 1e-3      # (the variable: measurement aid)
    while time.perf_counter() < t_spin:
        pass
    t0 = time.perf_counter()
    for k in range(k0, k0 + args.steps):
        step(k)
    t_loop = time.perf_counter() - t0
    fence()
    dt = time.perf_counter() - t0
    host_split = None if not host_api else {"note_events_us": 1e6 * host_t[0] / args.steps, "fill_begin_us": 1e6 * host_t[1] / args.steps,
                  "fill_end_us": 1e6 * host_t[2] / args.steps, "final_fence_us": 1e6 * (dt - t_loop),
                  "note": "host time per timed step inside the three calls (fill_end includes the wait for the GPU), and the fence behind the last step"}
    k0 += args.steps

    t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())
    mix_host = (out_host.copy() if host_api else pinned.numpy().copy()) if rank == 0 else None

    # ---- the same steps through the synchronous host API (N = 1): s2r_fill returns each buffer in host memory ----
    host_api_sync = None
    if world == 1:
        n_p = min(args.steps, 64)
        fence()
        t1 = time.perf_counter()
        for k in range(k0, k0 + n_p):
            sh.note_events(events_of(k))
            synth.sample(out_host, SR)
        host_api_sync = total * FRAMES * n_p / (time.perf_counter() - t1)
        k0 += n_p

    # ---- per-launch duration of the render kernel on the same workload, HIP events on the launch stream ----
    # (separate steps so the events do not perturb the timed region)
    def kernel_ms_loop(n):
        nonlocal k0
        kms = []
        synth.set_timing(True)
        if world == 1:
            # The launch the timed steps make — s2r_fill_begin's render kernel on its own stream, the mix and the chain heads beside
            # it — with HIP events around it on that stream (s2r_set_timing), and in the timed steps' own rhythm: two fills in
            # flight, so that a fill's chain heads are built while the fill before it renders (one fill at a time, the render
            # kernel spends its first ~3 us waiting for them inside the bracket).
            sh.note_events(events_of(k0)); synth.sample_begin(FRAMES, SR)
            for k in range(k0 + 1, k0 + n + 1):
                sh.note_events(events_of(k)); synth.sample_begin(FRAMES, SR)
                synth.sample_end(out_host)
                kms.append(synth.last_render_ms())            # (the launch just begun; waits for it)
            synth.sample_end(out_host)
            k0 += 1
        else:
            for k in range(k0, k0 + n):
                sh.note_events(events_of(k))
                sh.fill(FRAMES, SR)
                kms.append(synth.last_render_ms())
        synth.set_timing(False)
        fence()
        k0 += n
        return float(np.median(kms)) if kms else float("nan")      # (median: one launch behind a host hiccup reads milliseconds)

    kernel_ms = kernel_ms_loop(min(max(args.steps, 4), 16))
    # the same workload with the flat-envelope coefficient reuse and the coefficient tables switched off: every frame
    # of every voice pays the full pow/exp chain in-lane (all of SURVEY §8(d)'s ~250 flop-equivalents really executed)
    synth.set_flat_shortcut(False)
    kernel_ms_full = kernel_ms_loop(min(max(args.steps, 4), 16))
    synth.set_flat_shortcut(True)

    # ---- every voice inside its mod-envelope decay (N = 1): the whole pool re-triggered, then 8 buffers (the default
    #      patch's mod envelope decays for 9 600 frames = 9.4 buffers), coefficient tables ON (the product path) ----
    all_mod = None
    kernel_ms_full_plain = None
    if world == 1:
        retrig = np.zeros(total, dtype=s2.NOTE_EVENT_DTYPE)
        retrig["kind"] = 1
        retrig["note"] = 36 + (np.arange(total) % 61)
        retrig["velocity"] = 1.0
        fence()
        sh.note_events(retrig)
        sh.fill(FRAMES, SR)                       # (the buffer that applies 65 536 note-ons is not timed)
        fence()
        t2 = time.perf_counter()
        for _ in range(8):
            sh.fill(FRAMES, SR)
        fence()
        all_mod = total * FRAMES * 8 / (time.perf_counter() - t2)
        # ... and the same population with every shortcut off (no tables, no flat-envelope reuse): the arithmetic of
        # SURVEY 8(d) executed in-lane by every voice on every frame, no note events in the way — the cleanest reading of
        # the VALU roofline (the C3 launches above carry their events' wave imbalance on top)
        synth.set_flat_shortcut(False)
        sh.note_events(retrig)
        sh.fill(FRAMES, SR)
        synth.set_timing(True)
        kms = []
        for _ in range(8):
            sh.fill(FRAMES, SR)
            kms.append(synth.last_render_ms())
        synth.set_timing(False)
        synth.set_flat_shortcut(True)
        fence()
        kernel_ms_full_plain = float(np.median(kms))

    # ---- the other single-GPU configurations of BASELINE.json, each timed like the headline, after it (N = 1) ----
    config_legs = None
    if world == 1 and rank == 0 and not args.no_config_legs:
        # the headline's handle is destroyed HERE, not when the collector gets to it (ShardedSynth holds a bound method of itself):
        # one handle per device gets the two streams of s2r_fill_begin, and the legs are timed the way the headline is
        synth.close()
        synth = None; sh = None
        config_legs = []
        leg, _s = run_config_leg("config[2] as written: 65 536 voices, saw + ADSR + SVF (build-defined state-variable filter; the reference has none)",
                                 65536, "synth c2 { lpf.kind = svf_lp; lpf.q = 1.4 }", args.steps, args.warmup, bytes_per_voice=36 + 20)
        config_legs.append(leg); _s.close(); del _s
        leg, _s = run_config_leg("config[4]'s per-GPU share: 32 768 voices, alias-suppressed saw (DPW) + SVF, 4x oversampled (192 kHz internal)",
                                 32768, "synth c4 { osc.kind = dpw_saw; lpf.kind = svf_lp; lpf.q = 1.4 }", max(4, args.steps // 4), max(1, args.warmup // 4),
                                 oversampled=True, bytes_per_voice=40 + 24)
        config_legs.append(leg); _s.close(); del _s

    # ---- the reference's own call pattern (BASELINE configs[0]: 8 voices, Synth::sample per 16 frames, main.rs:138-147): wall time of
    #      one s2r_fill from a C++ caller over the C ABI, a launch per call and through the resident kernel (s2r_set_low_latency) ----
    small_fill = None
    if world == 1 and rank == 0 and not args.no_config_legs:
        small_fill = small_fill_leg()

    if rank == 0:
        value = total * FRAMES * args.steps / dt_max
        kernel_s = kernel_ms * 1e-3
        hbm_gbs = BYTES_PER_VOICE_FILL * vpg / kernel_s / 1e9
        valu_tf = FLOPS_PER_VOICE_SAMPLE * vpg * FRAMES / (kernel_ms_full * 1e-3) / 1e12
        prof = committed_profile()
        traffic = None
        traffic_note = "no rocprofv3 PMC summary of this build of the kernels under profiles/%s (hash %s): omitted rather than quoted from another build" % (PROFILE_ROUND, kernel_source_hash())
        if prof is not None:
            pm = prof.get("pmc_avg_per_dispatch", {})
            if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
                # separate passes, KiB units; dword-per-lane accesses, for which the guide's x2 FETCH_SIZE correction
                # (16 B/lane streams) is not calibrated: raw counter sum
                traffic = (pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0
                traffic_note = "profiles/%s/c3_summary.json, same kernel sources (hash %s)" % (PROFILE_ROUND, kernel_source_hash())
        out = {
            "metric": "voice-samples/sec (mono) at 64k voices per GPU, 48 kHz, 1024-frame buffers",
            "value": value,
            "unit": "samples/s",
            "n_gpus": len(set(dev_list)) if dev_list else world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("C3 (SURVEY 8d), periodic: %d voices per GPU, default patch (example.synth2 empty body: saw + amp/mod ADSR + "
                                    "one-pole LPF), 48 kHz, 1024-frame buffers; every voice lives one C3 life (note-on, note-off after "
                                    "16*(512 + lcg(v) mod 2048) frames, release) per %d buffers, lives staggered 1/%d of the pool per buffer; "
                                    "%.0f events per buffer, note-offs as timed events on their 16-frame boundary; population aged six periods "
                                    "before the warm-up" % (vpg, PERIOD, PERIOD, n_events_per_step)) if args.workload == "c3" else
                                   ("churn: %d voices per GPU, default patch, 48 kHz, 1024-frame buffers, all on at frame 0, then %d note-off + %d "
                                    "note-on per buffer per 64k voices" % (vpg, args.churn, args.churn)),
                       "timed_call": ("s2r_note_events + s2r_fill_begin / s2r_fill_end: the host-buffer API with two buffers in flight (s2_bin's own arrangement); voice state resident in HBM, events H2D and every mix's D2H into the caller's buffer inside the timed region" + ("; render grid resident on the device (s2r_set_resident): a fill is a posted command, no launch" if resident else "; a launch per fill") +
                                      ("; every rank the same calls on its shard, the partial rows exchanged inside the render kernels through rank 0's device memory (s2r_exchange_create / _attach) and added there in rank order" if use_xg else "")) if host_api else
                                     "s2r_note_events + s2r_fill_device per rank, all-gather, rank-ordered sum, async D2H of the mix on rank 0",
                       "voices_total": total, "frames": FRAMES, "sample_rate": SR,
                       "parallelism": ("one process, one handle over the device list %s: policy run once, every shard on its own device's stream, rows added in shard order on the first device" % dev_list) if dev_list else
                                      "voice-shard x%d, %s of partial mixes" % (world, "in-kernel exchange: rows into rank 0's device memory, rank-ordered sum by its last workgroup" if use_xg else "reduce(sum) to rank 0" if args.reduce else "all-gather + rank-ordered sum"),
                       "block_voices": block_voices},
            "msamples_per_s": value / 1e6,
            "realtime_factor_64k_voices": value / (65536.0 * SR),
            "roofline": {"bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": hbm_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "kernel": "s2r_render_kernel", "kernel_ms": kernel_ms,
                         "note": "algorithmic bytes = %d B per voice per fill; the path is VALU-issue-bound, see roofline_valu" % BYTES_PER_VOICE_FILL},
            # The bound that applies: VALU issue.  For the launches that are TIMED (not a shortcuts-off variant): the vector
            # instructions the render kernel really executed per launch (SQ_INSTS_VALU and the per-type counters of the
            # committed rocprofv3 summary of this build) over the launch's duration, against the SIMDs' issue rate and,
            # weighted (fma 2, packed x2), against the fp32 vector peak.  null when no summary of this build is committed.
            "roofline_valu": executed_roofline(prof, vpg, kernel_ms),
            # (diagnostic, not the product path) SURVEY 8(d)'s 250 flop-equivalents per voice-sample are the REFERENCE's
            # arithmetic; the product kernel does not execute them (flat-envelope reuse, coefficient tables, hoisted
            # constants, packed arithmetic).  With those shortcuts OFF every one of them runs in-lane:
            "diagnostic_all_in_lane": (lambda tf_c3, tf_plain: {
                "bound": "valu-fp32", "achieved": tf_plain if tf_plain else tf_c3, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": (tf_plain if tf_plain else tf_c3) / VALU_PEAK_TFLOPS,
                "flops_per_voice_sample": FLOPS_PER_VOICE_SAMPLE,
                "kernel_ms": kernel_ms_full_plain if kernel_ms_full_plain else kernel_ms_full,
                "frac_c3_launches": tf_c3 / VALU_PEAK_TFLOPS, "kernel_ms_c3_launches": kernel_ms_full,
                "note": "launch time with the flat-envelope reuse and the coefficient tables OFF, i.e. all 250 flop-eq per voice-sample executed in-lane. "
                        "`frac`: the whole pool re-triggered and held, every voice inside its mod decay, no note events; "
                        "`frac_c3_launches`: the C3 launches in the same mode"})(
                valu_tf, (FLOPS_PER_VOICE_SAMPLE * vpg * FRAMES / (kernel_ms_full_plain * 1e-3) / 1e12) if kernel_ms_full_plain else None),
            "host_time_per_step": host_split,
            "settle": settle,
               "value_host_api_sync": host_api_sync,
            "value_host_api_sync_note": "the same steps through s2r_fill, which returns every buffer in the caller's host memory before the next events are handed over (host event processing and GPU time add up instead of overlapping)",
            "value_kernel_only": vpg * FRAMES / kernel_s,
            "value_all_voices_modulating": all_mod,
            "value_all_voices_modulating_note": "every voice re-triggered, then 8 buffers inside the 9 600-frame mod decay; device-resident fills queued back to back, product path (tables on)",
            "mix_checksum": float(np.abs(mix_host).sum()),
            # the library that was timed says which sources it was built from (s2r_build_id; synth2_amd.load_library refuses one that
            # does not match the sources on disk): its first half is the hash the committed profiles are keyed by
            "build_id": s2.load_library().s2r_build_id().decode(),
            "kernel_source_hash": kernel_source_hash(),
            "mix_note": "the mix follows the build's documented tree (DESIGN.md 4.3), bit-equal to the oracle's same tree; the reference's sequential order is 46 / 2 887 / 7 278 ULP away at 1 024 / 65 536 / 131 072 voices (profiles/r03/mix_deviation.json)",
        }
        if config_legs is not None:
            out["config_legs"] = config_legs
        if small_fill is not None:
            out["small_fill"] = small_fill
        if world > 1:
            out["multi_gpu_note"] = ("this path has not been run on two or more GPUs by its authors (no such box was available to them): "
                                     "tests cover it with two processes sharing one card (the in-kernel exchange), N ranks under gloo and one rank through RCCL")
            if xg_note:
                out["exchange_fallback"] = xg_note
        if cpu_legs is not None:
            legs = cpu_legs
            out["cpu_baseline"] = dict(legs[-1])                  # all cores at C3's pool size
            out["cpu_baseline"]["sample"] += "; C restatement of s2_lib (oracle/), not rustc output"
            out["cpu_baseline_legs"] = legs
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
