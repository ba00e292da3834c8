#!/usr/bin/env python3
"""Headline benchmark: input MS/s (complex) channelized, with the achieved HBM GB/s of the
channelizer kernel against the MI355X roofline (BASELINE.json metric).

Workload (configs[1], the one the metric is quoted on): M=64 channels, 12 taps/branch,
D=64, int16 I/Q in the blade_record_iq_12bit format (12-bit in int16), a 2^30-sample
synthetic pulsed stream per GPU, already resident in HBM when the clock starts.  A "step" is
one pass of the channelizer over that batch.  At N>1 the stream is time-sharded: every rank
owns one contiguous 2^30-sample segment (weak scaling) and, each step, hands the last
history_samples() raw samples of its segment to the next rank over RCCL (the only
data-path communication; SURVEY.md section 8e: one all_gather of the tails by default,
--halo p2p for neighbour send/recv) before running its kernel.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (M, P, D, fmt, bit_width, bytes_in_per_sample)
    "cfg2": (64, 12, 64, "int16", 12, 4),
    "cfg3": (256, 8, 256, "int8", 8, 2),
    "cfg4": (1024, 16, 1024, "int16", 16, 4),
    "cfg5": (128, 12, 64, "int16", 12, 4),
    "ref56": (56, 12, 56, "int16", 12, 4),     # the reference's own band counts at fs = 56 MHz
    "ref560": (560, 12, 560, "int16", 12, 4),
}


def cpu_baseline(iq_prefix: np.ndarray, taps: np.ndarray, M: int, P: int, D: int, bw: int, budget_s: float):
    """Time the oracle's fp32 OpenMP port (kind="port": no MATLAB exists for the reference's
    dsp.Channelizer call) on a bounded prefix of the same stream, on this box's host cores."""
    from oracle.pfb_oracle import COracle
    src = os.path.join(ROOT, "oracle", "pfb_oracle.c")
    path = None
    try:  # tune for this host; fall back to the prebuilt portable build
        out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"libpfb_oracle_native_{os.getpid()}.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-fopenmp", "-shared", "-o", out, src, "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        path = out
    except Exception:
        path = None
    o = COracle(path)
    cores = o.max_threads()
    probe = iq_prefix[: 1 << 22]
    o.channelize_f32_i16(probe, bw, taps, M, P, D, threads=cores)  # warm the thread pool and the caches
    t0 = time.perf_counter()
    o.channelize_f32_i16(probe, bw, taps, M, P, D, threads=cores)
    rate = probe.shape[0] / (time.perf_counter() - t0)
    # bounded sample: about budget_s core-seconds per core, never more than the prefix we were handed
    n = int(min(iq_prefix.shape[0], max(1 << 22, rate * budget_s / 8)))
    n -= n % D
    reps, dt = 0, 0.0
    while dt < 1.0 and reps < 8:  # at least ~1 s of wall time so the number is stable
        t0 = time.perf_counter()
        o.channelize_f32_i16(iq_prefix[:n], bw, taps, M, P, D, threads=cores)
        dt += time.perf_counter() - t0
        reps += 1
    return {"value": round(reps * n / dt / 1e6, 3), "unit": "MS/s", "cores": cores, "kind": "port",
            "sample": f"first {n} samples of the same synthetic stream x{reps} passes, fp32 OpenMP polyphase+FFT "
                      f"port of the oracle (oracle/pfb_oracle.c, built -O3 -march=native on this host), "
                      f"{dt:.2f} s wall = {dt * cores:.0f} core-seconds"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the first ~15 launches after an idle GPU run 3-25 % slow (clock ramp, tools/launch_series.py),
    # so warm up past that transient and time a steady-state window; the whole run is still < 1 s of GPU time
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--log2-samples", type=int, default=30, help="samples per GPU per step (2^k)")
    ap.add_argument("--frames-per-block", type=int, default=0)
    ap.add_argument("--nontemporal", type=int, default=-1)
    ap.add_argument("--xcd-remap", type=int, default=-2)
    ap.add_argument("--schedule", type=int, default=-1)
    ap.add_argument("--grid", type=int, default=-1)
    ap.add_argument("--tile-waves", type=int, default=-1)
    ap.add_argument("--cpu-budget-s", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--channel-major", action="store_true",
                    help="write MATLAB's column-major F x M matrix instead of frame-major rows (not the headline line)")
    ap.add_argument("--halo", default="allgather", choices=("allgather", "p2p"),
                    help="how the ring of halos moves: one all_gather of the tails, or send/recv between neighbours")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from sdr_channelizer_amd import Channelizer, design_prototype, synth
    from sdr_channelizer_amd import _lib as L
    from sdr_channelizer_amd.sharded import exchange_halo, exchange_halo_allgather

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if os.environ.get("PFB_BENCH_ONE_DEVICE"):  # rehearsal on a 1-GPU box: every rank shares device 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    M, P, D, fmt, bw, bytes_in = WORKLOADS[args.workload]
    n = 1 << args.log2_samples
    taps = design_prototype(M, P, 80.0)
    tdtype = torch.int8 if fmt == "int8" else torch.int16
    # rank r owns stream samples [r*n, (r+1)*n): generate that slice of the pulse train in HBM
    iq = synth.pulsed_iq_torch(n, bw, tdtype, seed=synth.SEED + rank, device=dev)
    F = n // D + 1  # +1: with M not a power of two the carried tail completes an extra frame every few steps
    out = torch.empty((M, F) if args.channel_major else (F, M), dtype=torch.complex64, device=dev)

    ch = Channelizer(M, taps=taps, decimation=D, sample_format=fmt, bit_width=bw, device=local_rank,
                     channel_major=args.channel_major)
    stream = torch.cuda.current_stream(dev)
    ch.set_stream(stream.cuda_stream)
    ch.set_option(L.PFB_OPT_KERNEL, 2)  # the hand-written fast kernel or nothing
    if args.frames_per_block:
        ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, args.frames_per_block)
    if args.nontemporal >= 0:
        ch.set_option(L.PFB_OPT_NONTEMPORAL, args.nontemporal)
    if args.xcd_remap >= -1:
        ch.set_option(L.PFB_OPT_XCD_REMAP, args.xcd_remap)
    if args.schedule >= 0:
        ch.set_option(L.PFB_OPT_SCHEDULE, args.schedule)
    if args.grid >= 0:
        ch.set_option(L.PFB_OPT_GRID, args.grid)
    if args.tile_waves >= 0:
        ch.set_option(L.PFB_OPT_TILE_WAVES, args.tile_waves)
    hist = ch.history_samples
    halo = torch.zeros((hist, 2), dtype=tdtype, device=dev)

    def step():
        if world > 1:
            # time shard: my history is the tail of the previous rank's segment (ring, so rank 0
            # continues from the last rank's previous batch)
            if args.halo == "p2p":
                exchange_halo(iq[n - hist:], halo, rank, world)
            else:
                exchange_halo_allgather(iq[n - hist:], halo, rank, world)
            ch.prime(halo)
        ch(iq, out=out, sync=False)

    # The first ~15 launches after an idle GPU run 3-25 % slow (clock ramp).  If the caller asks
    # for fewer warm-up steps than that, run the difference as extra untimed steps first; the W warm-up steps and the
    # K timed steps that follow are exactly what was asked for.
    prewarm = max(0, 25 - args.warmup)
    for _ in range(prewarm + args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    ch.set_option(L.PFB_OPT_PROFILE, 1)  # hipEvents right around each kernel launch, on its stream
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ch.kernel_times_ms()
    ch.set_option(L.PFB_OPT_PROFILE, 0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        value = world * n * args.steps / elapsed / 1e6
        bytes_per_sample = bytes_in + 8 * (M // D)            # B = bytes_in + 8*(M/D)
        k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
        achieved = n * bytes_per_sample / (k_ms * 1e-3) / 1e9  # algorithmic bytes per launch / avg launch time
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tfile) and args.workload == "cfg2" and args.log2_samples == 30 and not args.channel_major:
            try:
                traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        copy_bps = L.C.c_double()
        copy_rc = L.load().pfb_measure_stream_copy(local_rank, 1 << 30, 5, L.C.byref(copy_bps))
        res = {
            "metric": "input MS/s (complex) channelized",
            "value": round(value, 1),
            "unit": "MS/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: M={M} channels, {P} taps/branch, D={D}, {fmt} I/Q "
                                   f"({bw}-bit), 2^{args.log2_samples} samples per GPU per step, {'channel-major' if args.channel_major else 'frame-major'} complex64 out",
                       "kernel": ch.last_kernel,
                       "schedule": args.schedule if args.schedule >= 0 else
                       ("default (channel-major route of the shape, DESIGN.md section 5.2)" if args.channel_major else
                        "default (4: FIR/FFT wave pairs, 8x64-frame workgroups)" if M == 64 else "default (0: sliding runs)"),
                       "samples_per_gpu": n, "prewarm_steps": prewarm,
                       "parallelism": f"time-sharded x{world}, halo {hist} samples/rank over RCCL" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel_ms": round(k_ms, 4), "launches_timed": len(kernel_ms),
                         "algorithmic_bytes_per_sample": bytes_per_sample,
                         "measured_stream_copy_gbs": round(copy_bps.value / 1e9, 1) if copy_rc == 0 else None},
        }
        if world == 1 and not args.no_cpu_baseline and fmt == "int16":
            prefix = iq[: 1 << 28].cpu().numpy()
            res["cpu_baseline"] = cpu_baseline(prefix, taps, M, P, D, bw, args.cpu_budget_s)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    ch.release()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
