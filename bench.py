#!/usr/bin/env python3
"""bench.py — voice-render throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A *step* is one 1024-frame buffer fill of the whole voice pool: note events for the buffer
are applied, every voice's 1024 frames are rendered (oscillator -> envelopes -> LPF) and
mixed down.  Workload (config.workload): BASELINE config "65 536 voices ... 48 kHz on 1
MI355X" with the reference's own patch (Synth::default_config: saw + amp/mod ADSR + the
one-pole LPF; the reference has no SVF) — per GPU, so N GPUs render N x 65 536 voices (weak
scaling), the pool dealt out to the ranks in runs of 64 voices, with one all-gather of the 4 KiB
partial mixes per buffer over RCCL and a rank-ordered sum on rank 0.

`value` = voice-samples/s = voices x frames x steps / wall time, whole job, with the voice
state resident in HBM and the mix left in HBM (the synchronous host-buffer API rate, which
adds a 4 KiB D2H copy and a stream sync per buffer, is printed as `sync_fill_value`).

Two extra objects ride on the JSON line: `roofline` (the render kernel against the HBM
roof, as BASELINE's north_star asks — this path is NOT HBM-bound, see DESIGN.md — plus
`roofline_valu`, the bound that actually applies) and `cpu_baseline` (the CPU oracle, a C
restatement of s2_lib, timed on this host on a bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR = 48000
FRAMES = 1024
# algorithmic HBM bytes per started voice per fill: read pitch, offset, release, flags, phase,
# lpf_last, seed (7 x 4 B) and write offset, phase, lpf_last (3 x 4 B)        
BYTES_PER_VOICE_FILL = 28 + 12
# fp32-equivalent flops per voice-sample of the x16 path with the default patch
FLOPS_PER_VOICE_SAMPLE = 250.0
HBM_PEAK_GBS = 8000.0
VALU_PEAK_TFLOPS = 157.3


def lcg(x):
    return (1103515245 * x + 12345) % (1 << 31)


def make_events(total_voices, churn_per_64k, step, rng_seed=1):
    """Deterministic churn for one step: `churn` note-offs then `churn` note-ons over the
    whole pool (voice stealing picks the oldest voice), notes 36..96."""
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


def cpu_baseline(voices, buffers, threads):
    """The CPU oracle (test infrastructure) timed as the reported CPU baseline."""
    from oracle import s2o
    s = s2o.OracleSynth(voices)
    for v in range(voices):
        s.note_on(36 + v % 61)
    s.sample_mt(FRAMES, SR, threads)          # warm-up buffer
    t0 = time.perf_counter()
    for _ in range(buffers):
        s.sample_mt(FRAMES, SR, threads)
    dt = time.perf_counter() - t0
    return voices * FRAMES * buffers / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--voices-per-gpu", type=int, default=65536)
    ap.add_argument("--churn", type=int, default=128, help="note-ons (and note-offs) per step per 65536 voices")
    ap.add_argument("--block-voices", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=0, help="GPU lanes per voice (1/2/4, 0 = auto)")
    ap.add_argument("--no-overlap", action="store_true", help="do not overlap the all-gather with the next render")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-voices", type=int, default=16384)
    ap.add_argument("--cpu-buffers", type=int, default=0, help="0 = sized for ~15 s")
    args = ap.parse_args()

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

    vpg = args.voices_per_gpu
    total = vpg * world
    from synth2_amd.sharded import ShardedSynth
    sh = ShardedSynth(vpg, max_frames=FRAMES, rank=rank, world=world, device=dev,
                      block_voices=args.block_voices, lanes_per_voice=args.lanes, overlap=not args.no_overlap)
    synth = sh.renderer
    sh.load_patch("synth mySynth {\n\n}\n")       # example.synth2: empty body == default patch
    if os.environ.get("S2R_COEFF_STREAM_MODE"):       # measurement aid (see s2r_set_coeff_stream); results are bit-identical
        synth.set_coeff_stream(int(os.environ["S2R_COEFF_STREAM_MODE"]))

    # initial population: every voice of the pool gets a note (all ranks see the same stream)
    init = np.zeros(total, dtype=s2.NOTE_EVENT_DTYPE)
    init["kind"] = 1
    init["note"] = 36 + (np.arange(total) % 61)
    init["velocity"] = 1.0
    sh.note_events(init)

    n_steps = args.warmup + args.steps
    events = [make_events(total, args.churn, k) for k in range(n_steps)]

    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream
    partial = sh.partial
    mix = sh.mix

    def step(k):
        sh.note_events(events[k])
        sh.fill(FRAMES, SR)

    def fence():
        sh.flush()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    fence()
    # ---- timed region: exactly K steps ----
    t0 = time.perf_counter()
    for k in range(args.warmup, n_steps):
        step(k)
    fence()
    dt = time.perf_counter() - t0

    t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())
    mix_host = mix.cpu().numpy()

    # ---- per-launch duration of the render kernel, HIP events on the launch stream ----
    # (separate short loops so the events do not perturb the timed region)
    def kernel_ms_loop():
        kms = []
        synth.set_timing(True)
        for k in range(min(args.steps, 16)):
            synth.fill_device(partial[0].data_ptr(), FRAMES, SR, sptr)
            kms.append(synth.last_render_ms())
        synth.set_timing(False)
        fence()
        return float(np.mean(kms)) if kms else float("nan")

    kernel_ms = kernel_ms_loop()
    # the same workload with the flat-envelope coefficient reuse switched off: every frame of
    # every voice pays the full pow/exp chain (what the kernel costs when all voices modulate)
    synth.set_flat_shortcut(False)
    kernel_ms_full = kernel_ms_loop()
    t2 = time.perf_counter()
    n_full = min(args.steps, 32)
    for k in range(n_full):
        sh.fill(FRAMES, SR)
    fence()
    dt_full = time.perf_counter() - t2
    synth.set_flat_shortcut(True)

    # ---- synchronous host-buffer API (s2r_fill): D2H + sync per buffer ----
    sync_rate = None
    if world == 1:
        buf = np.empty(FRAMES, dtype=np.float32)
        synth.sample(buf, SR)
        t1 = time.perf_counter()
        for _ in range(min(args.steps, 32)):
            synth.sample(buf, SR)
        sync_rate = vpg * FRAMES * min(args.steps, 32) / (time.perf_counter() - t1)

    if rank == 0:
        # HBM bytes per launch of the render kernel from the committed rocprofv3 PMC passes
        # (FETCH_SIZE and WRITE_SIZE in separate passes, KiB units); dword-per-lane accesses, for
        # which the guide's x2 FETCH_SIZE correction (16 B/lane streams) is not calibrated
        traffic = None
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r01", "steady_state_summary.json")))
            pm = prof["pmc_avg_per_dispatch"]
            traffic = (pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0
        except Exception:
            pass
        value = total * FRAMES * args.steps / dt_max
        kernel_s = kernel_ms * 1e-3
        hbm_gbs = BYTES_PER_VOICE_FILL * vpg / kernel_s / 1e9
        valu_tf = FLOPS_PER_VOICE_SAMPLE * vpg * FRAMES / (kernel_ms_full * 1e-3) / 1e12
        out = {
            "metric": "voice-samples/sec (mono) at 64k voices per GPU, 48 kHz, 1024-frame buffers",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%d voices per GPU, default patch (example.synth2 empty body: saw + amp/mod ADSR + one-pole LPF), "
                                   "48 kHz, 1024-frame buffers, %d note-on + %d note-off per buffer per 64k voices" % (vpg, args.churn, args.churn),
                       "voices_total": total, "frames": FRAMES, "sample_rate": SR,
                       "parallelism": "voice-shard x%d, all-gather of partial mixes" % world,
                       "block_voices": synth.block_voices, "lanes_per_voice": synth.lanes_per_voice},
            "msamples_per_s": value / 1e6,
            "realtime_factor_64k_voices": value / (65536.0 * SR),
            "roofline": {"bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": hbm_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "s2r_render_kernel", "kernel_ms": kernel_ms,
                         "note": "algorithmic bytes = %d B per voice per fill; the path is VALU-bound, see roofline_valu" % BYTES_PER_VOICE_FILL},
            "roofline_valu": {"bound": "valu-fp32", "achieved": valu_tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": valu_tf / VALU_PEAK_TFLOPS,
                              "flops_per_voice_sample": FLOPS_PER_VOICE_SAMPLE, "kernel_ms": kernel_ms_full,
                              "note": "launch time with the flat-envelope coefficient reuse OFF, i.e. all 250 flop-eq per voice-sample executed"},
            "value_all_voices_modulating": total * FRAMES * n_full / dt_full if world == 1 else None,
            "mix_checksum": float(np.abs(mix_host).sum()),
        }
        if sync_rate is not None:
            out["sync_fill_value"] = sync_rate
        if world == 1 and not args.no_cpu_baseline:
            threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            threads = max(1, min(threads, 64))
            if args.cpu_buffers:
                nb = args.cpu_buffers
            else:
                probe, _ = cpu_baseline(args.cpu_voices, 1, threads)
                nb = int(max(2, min(400, 15.0 * probe / (args.cpu_voices * FRAMES))))
            v, secs = cpu_baseline(args.cpu_voices, nb, threads)
            out["cpu_baseline"] = {"value": v, "unit": "samples/s", "cores": threads, "kind": "port",
                                   "sample": "%d voices x %d frames x %d buffers, default patch, same note map; "
                                             "C restatement of s2_lib (oracle/), not rustc output; %.1f s" % (
                                                 args.cpu_voices, FRAMES, nb, secs)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
