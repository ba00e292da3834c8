#!/usr/bin/env python3
"""Headline benchmark: input MS/s (complex) channelized, with the achieved HBM GB/s of the
channelizer kernel against the MI355X roofline (BASELINE.json metric).

Workload (configs[1], the one the metric is quoted on): M=64 channels, 12 taps/branch,
D=64, int16 I/Q in the blade_record_iq_12bit format (12-bit in int16), a 2^30-sample
synthetic pulsed stream per GPU, already resident in HBM when the clock starts.  A "step" is
one pass of the channelizer over that batch.

N > 1: the stream is time-sharded.  Rank r owns samples [r*n, (r+1)*n) of ONE
counter-based stream (same seed everywhere) and every step runs pfb_process_shard_async: the last
(P-1)*M raw samples of its segment go to rank r+1 over RCCL on a side stream while the frames that do
not touch the halo run at once; the first frames follow when the halo has landed.  Weak scaling, no
other communication.  `python bench.py --gpus N` starts its own N ranks (fresh child processes, before
this process touches torch or the GPU); under torch.distributed.run it runs as the rank it is given.

Prints ONE JSON line on rank 0.  At N = 1 the line also carries short timed passes of the other
BASELINE configurations (`other_workloads`), a >= 2 s sustained window (`roofline.sustained_frac`) and
the CPU baseline (the oracle's naive fp32 port, all host threads and one thread).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (M, P, D, fmt, bit_width, bytes_in_per_sample, log2 samples of the BASELINE.json configuration)
    # cfg1 is channelizer_example.m's own shape (M = 8, the 96-tap prototype, complex float input); BASELINE runs it on
    # the CPU at 10^6 samples -- here a 2^28-sample device-resident stream, so that it is a bandwidth figure
    "cfg1": (8, 12, 8, "cf32", 0, 8, 28),
    "cfg2": (64, 12, 64, "int16", 12, 4, 30),
    "cfg3": (256, 8, 256, "int8", 8, 2, 30),
    "cfg4": (1024, 16, 1024, "int16", 16, 4, 30),
    "cfg5": (128, 12, 64, "int16", 12, 4, 28),
    "ref56": (56, 12, 56, "int16", 12, 4, 28),     # the reference's own band counts at fs = 56 MHz
    "ref560": (560, 12, 560, "int16", 12, 4, 28),
}
# what the default N = 1 run times after the headline: (workload, channel_major)
OTHER_WORKLOADS = [("cfg1", False), ("cfg3", False), ("cfg4", False), ("cfg5", False), ("ref56", False), ("ref560", False),
                   ("cfg2", True), ("cfg3", True), ("cfg4", True), ("cfg5", True)]
TRAFFIC_SOURCE = "profiles/r03_pmc_traffic.json"  # rocprofv3 --pmc passes of this command (never measured in-run)


def parse_args():
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
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--cpu-budget-s", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the other BASELINE configurations")
    ap.add_argument("--sustained-s", type=float, default=2.2, help="length of the sustained window (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--channel-major", action="store_true",
                    help="write MATLAB's column-major F x M matrix instead of frame-major rows (not the headline line)")
    ap.add_argument("--halo", default="p2p", choices=("p2p", "allgather"),
                    help="how the halos move: send/recv between neighbours (ncclSend/ncclRecv), or one all_gather of the tails")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start N fresh rank processes.  Nothing in THIS process
    has imported torch or touched the GPU, and no child re-execs: each is a new interpreter that finds RANK / WORLD_SIZE
    in its environment.  Rank 0's stdout (the JSON line) passes through; any child failing fails the run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(args.gpus))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [None] * len(procs)
    try:
        while any(c is None for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.poll()
            if any(c not in (None, 0) for c in codes):  # one rank died: the others would wait in a collective forever
                break
            time.sleep(0.2)
    finally:
        for i, p in enumerate(procs):  # exactly the PIDs started above
            if codes[i] is None and p.poll() is None:
                p.terminate()
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
                codes[i] = p.returncode if p.returncode is not None else 1
    bad = [(i, c) for i, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def cpu_baseline(iq_prefix, taps, M: int, P: int, D: int, bw: int, budget_s: float):
    """Time the oracle's fp32 OpenMP port (kind="port": no MATLAB exists for the reference's
    dsp.Channelizer call) on a bounded prefix of the same stream, on this box's host cores: all hardware
    threads and one thread (BASELINE.md's plan).  The port is NAIVE -- scalar, one branch per tap, stride-M
    gathers -- a baseline for orientation, not a tuned CPU channelizer."""
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

    def timed(threads: int, budget: float):
        probe = iq_prefix[: 1 << (22 if threads > 1 else 20)]
        o.channelize_f32_i16(probe, bw, taps, M, P, D, threads=threads)  # warm the thread pool and the caches
        t0 = time.perf_counter()
        o.channelize_f32_i16(probe, bw, taps, M, P, D, threads=threads)
        rate = probe.shape[0] / (time.perf_counter() - t0)
        # bounded sample: about `budget` seconds of wall time, never more than the prefix we were handed
        n = int(min(iq_prefix.shape[0], max(probe.shape[0], rate * budget)))
        n -= n % D
        reps, dt = 0, 0.0
        while dt < min(1.0, budget) and reps < 8:  # at least ~1 s of wall time so the number is stable
            t0 = time.perf_counter()
            o.channelize_f32_i16(iq_prefix[:n], bw, taps, M, P, D, threads=threads)
            dt += time.perf_counter() - t0
            reps += 1
        return round(reps * n / dt / 1e6, 3), n, reps, dt

    v_all, n_all, r_all, dt_all = timed(cores, budget_s / 8)
    v_one, n_one, r_one, dt_one = timed(1, min(4.0, budget_s / 3))
    return {"value": v_all, "unit": "MS/s", "cores": cores, "kind": "port",
            "sample": f"first {n_all} samples of the same synthetic stream x{r_all} passes, NAIVE fp32 OpenMP polyphase+FFT "
                      f"port of the oracle (oracle/pfb_oracle.c: scalar, branch per tap, stride-M gathers; built -O3 "
                      f"-march=native on this host), {dt_all:.2f} s wall = {dt_all * cores:.0f} core-seconds",
            "one_thread": {"value": v_one, "unit": "MS/s", "cores": 1,
                           "sample": f"first {n_one} samples x{r_one} passes of the same port on one thread, {dt_one:.2f} s"}}


def main() -> None:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))  # nothing above imported torch or touched the GPU

    import numpy as np
    import torch
    import torch.distributed as dist

    from sdr_channelizer_amd import Channelizer, design_prototype, synth
    from sdr_channelizer_amd import _lib as L
    from sdr_channelizer_amd.sharded import make_exchange

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("PFB_BENCH_ONE_DEVICE"):  # rehearsal on a 1-GPU box: every rank shares device 0 (gloo only)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    def make_handle(name: str, channel_major: bool, tuned: bool):
        M, P, D, fmt, bw, _bytes_in, _ = WORKLOADS[name]
        taps = design_prototype(M, P, 80.0)
        ch = Channelizer(M, taps=taps, decimation=D, sample_format=fmt, bit_width=max(bw, 1), device=local_rank,
                         channel_major=channel_major)
        ch.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ch.set_option(L.PFB_OPT_KERNEL, 2)  # the hand-written fast kernel or nothing
        if tuned:  # command-line tuning knobs apply to the headline handle only
            if args.variant >= 0:
                ch.set_option(L.PFB_OPT_VARIANT, args.variant)
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
        return ch, taps

    def make_buffers(name: str, log2n: int, channel_major: bool, start: int = 0):
        M, P, D, fmt, bw, _bytes_in, _ = WORKLOADS[name]
        n = 1 << log2n
        tdtype = torch.int8 if fmt == "int8" else torch.int16
        iq = synth.pulsed_iq_torch(n, bw if bw else 12, tdtype, seed=synth.SEED, device=dev, start=start)
        if fmt == "cf32":  # what the example script holds after its normalisation (channelizer_example.m:18-21)
            iq = iq.to(torch.float32) / 2048.0
        F = n // D + 1  # +1: with M not a power of two the carried tail completes an extra frame every few steps
        out = torch.empty((M, F) if channel_major else (F, M), dtype=torch.complex64, device=dev)
        return iq, out

    def kernel_ms_per_step(ch, launches_per_step: int):
        t = ch.kernel_times_ms()
        k = launches_per_step
        return [sum(t[i:i + k]) for i in range(0, len(t) - len(t) % k, k)]

    M, P, D, fmt, bw, bytes_in, _ = WORKLOADS[args.workload]
    n = 1 << args.log2_samples
    ch, taps = make_handle(args.workload, args.channel_major, tuned=True)
    # ONE stream for the whole job: rank r owns samples [r*n, (r+1)*n) of it (counter-based generator, same seed)
    iq, out = make_buffers(args.workload, args.log2_samples, args.channel_major, start=rank * n)
    halo = ch.halo_samples
    launches_per_step = 1
    if world > 1:
        if n % D:
            raise SystemExit("time shards are cut on frame boundaries")
        # ring: rank 0 continues from the last rank's tail of the previous step, so step after step is one endless stream
        ch.attach_shard(rank, world, make_exchange(rank, world, None, args.halo, local_rank), ring=True)
        launches_per_step = 2  # interior frames, then the head frames behind the halo event
        F_shard = n // D
        out = out.reshape(-1)[: F_shard * M].reshape((M, F_shard) if args.channel_major else (F_shard, M))

    def step():
        if world > 1:
            ch.process_shard(iq, out=out)  # exchange on the side stream, interior frames now, head frames after the halo
        else:
            ch(iq, out=out, sync=False)

    halo_mode = args.halo
    if world > 1:
        # One probing step before anything is timed: if the neighbour exchange (ncclSend / ncclRecv) fails on ANY rank -- the
        # callback reports it, the step returns PFB_ERR_COMM -- every rank falls back to the all_gather form of the same
        # exchange, and the line says so.  (PFB_BENCH_FAIL_P2P=1 injects the failure: how the fallback is rehearsed.)
        failed = 0
        try:
            if os.environ.get("PFB_BENCH_FAIL_P2P") and args.halo == "p2p":
                raise L.PfbError(L.PFB_ERR_COMM, "injected")
            step()
            ch.sync()
        except L.PfbError as e:
            print(f"[rank {rank}] halo exchange ({args.halo}) failed: {e}", file=sys.stderr)
            failed = 1
        flag = torch.tensor([failed], dtype=torch.int32, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) and args.halo == "p2p":
            halo_mode = "allgather"
            ch.reset()
            ch.attach_shard(rank, world, make_exchange(rank, world, None, halo_mode, local_rank), ring=True)
        elif int(flag.item()):
            raise SystemExit("halo exchange failed")

    # The first ~15 launches after an idle GPU run 3-25 % slow (clock ramp).  If the caller asks
    # for fewer warm-up steps than that, run the difference as extra untimed steps first; the W warm-up steps and the
    # K timed steps that follow are exactly what was asked for.
    prewarm = max(0, 25 - args.warmup)
    for _ in range(prewarm + args.warmup):
        step()
    ch.sync()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    ch.set_option(L.PFB_OPT_PROFILE, 1)  # hipEvents right around each kernel launch, on its stream
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ch.sync()  # the handle's stream: kernels and, behind the halo event, the transfers
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = kernel_ms_per_step(ch, launches_per_step)
    ch.set_option(L.PFB_OPT_PROFILE, 0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if args.backend != "nccl":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    def gather_floats(vals):
        """every rank's list of floats -> [world][len] on every rank (one small all_gather)"""
        if world == 1:
            return [list(vals)]
        t = torch.tensor(list(vals), dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        return [x.cpu().tolist() for x in parts]

    headline_kernel = ch.last_kernel
    my_k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    per_rank_kernel_ms = [v[0] for v in gather_floats([my_k_ms])]

    # ---- N > 1: BASELINE's 8-GPU configuration is cfg4 (M = 1024, 16 taps, 2^30 samples per GPU, halo 61 440 B): a short
    # sharded pass of it after the headline, every rank taking part
    sharded_cfg4 = None
    if world > 1 and not args.no_other_workloads and args.workload != "cfg4":
        ch.release()
        del iq, out
        torch.cuda.empty_cache()
        ch = iq = out = None
        try:
            c4M, c4P, c4D, c4fmt, c4bw, c4bytes, c4log2 = WORKLOADS["cfg4"]
            l2 = min(c4log2, args.log2_samples)
            n4 = 1 << l2
            c4, _ = make_handle("cfg4", False, tuned=False)
            iq4, out4 = make_buffers("cfg4", l2, False, start=rank * n4)
            c4.attach_shard(rank, world, make_exchange(rank, world, None, halo_mode, local_rank), ring=True)
            out4 = out4.reshape(-1)[: (n4 // c4D) * c4M].reshape(n4 // c4D, c4M)
            halo4 = c4.halo_samples
            for _ in range(6):
                c4.process_shard(iq4, out=out4)
            c4.sync()
            torch.cuda.synchronize(dev)
            dist.barrier()
            c4.set_option(L.PFB_OPT_PROFILE, 1)
            k4 = 10
            tq = time.perf_counter()
            for _ in range(k4):
                c4.process_shard(iq4, out=out4)
            c4.sync()
            torch.cuda.synchronize(dev)
            dist.barrier()
            wall4 = time.perf_counter() - tq
            ms4 = kernel_ms_per_step(c4, 2)
            c4.set_option(L.PFB_OPT_PROFILE, 0)
            kname4 = c4.last_kernel
            c4.release()
            del iq4, out4
            torch.cuda.empty_cache()
            g = gather_floats([float(np.mean(ms4)), wall4])
            k_all = [v[0] for v in g]
            wall_max = max(v[1] for v in g)
            b4 = c4bytes + 8 * (c4M // c4D)
            sharded_cfg4 = {"workload": "cfg4", "layout": "frame-major", "sharded": f"time-sharded x{world} (ring), {args.backend} ({halo_mode})",
                            "shape": f"M={c4M} P={c4P} D={c4D} {c4fmt}, 2^{l2} samples per GPU per step", "kernel": kname4,
                            "halo_samples": int(halo4), "halo_bytes": int(halo4 * c4bytes), "launches_per_step": 2, "steps": k4,
                            "kernel_ms_per_rank": {"min": round(min(k_all), 4), "max": round(max(k_all), 4),
                                                   "all": [round(v, 4) for v in k_all]},
                            "step_wall_ms": round(wall_max / k4 * 1e3, 4),
                            "ms_value": round(world * n4 * k4 / wall_max / 1e6, 1),
                            "frac_slowest_rank": round(n4 * b4 / (max(k_all) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "algorithmic_bytes_per_sample": b4}
        except Exception as e:  # a failing side pass must not cost the headline line (every rank fails alike or not at all)
            sharded_cfg4 = {"workload": "cfg4", "sharded": f"x{world}", "error": repr(e)}

    if rank == 0:
        value = world * n * args.steps / elapsed / 1e6
        bytes_per_sample = bytes_in + 8 * (M // D)            # B = bytes_in + 8*(M/D)
        k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
        achieved = n * bytes_per_sample / (k_ms * 1e-3) / 1e9  # algorithmic bytes per launch / avg launch time
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
                       "kernel": headline_kernel,
                       "schedule": args.schedule if args.schedule >= 0 else
                       ("default (channel-major route of the shape, DESIGN.md section 5.2)" if args.channel_major else
                        "default (4: FIR/FFT wave pairs, 8x64-frame workgroups)" if M == 64 else "default for the shape (DESIGN.md section 5.2)"),
                       "samples_per_gpu": n, "prewarm_steps": prewarm,
                       "parallelism": (f"time-sharded x{world}: one stream, rank r owns samples [r*n, (r+1)*n); halo {halo} raw samples "
                                       f"({halo * bytes_in} B) per rank per step over {args.backend} ({halo_mode}{' after p2p failed' if halo_mode != args.halo else ''}) on a side stream, "
                                       f"interior frames first") if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         # HBM bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE x2 per the guide's
                         # gfx950 correction, WRITE_SIZE), never from this run: null here, the citation beside it
                         "traffic": None, "traffic_source": TRAFFIC_SOURCE,
                         "kernel_ms": round(k_ms, 4), "launches_timed": len(kernel_ms),
                         "launches_per_step": launches_per_step,
                         "algorithmic_bytes_per_sample": bytes_per_sample},
        }
        if world > 1:  # `frac` above is rank 0's kernels; the spread over ranks (one all_gather of the means):
            res["roofline"]["kernel_ms_per_rank"] = {"min": round(min(per_rank_kernel_ms), 4), "max": round(max(per_rank_kernel_ms), 4),
                                                     "all": [round(v, 4) for v in per_rank_kernel_ms]}
            res["roofline"]["frac_slowest_rank"] = round(n * bytes_per_sample / (max(per_rank_kernel_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            if sharded_cfg4 is not None:
                res["other_workloads"] = [sharded_cfg4]
        tfile = os.path.join(ROOT, TRAFFIC_SOURCE)
        if os.path.exists(tfile) and args.workload == "cfg2" and args.log2_samples == 30 and not args.channel_major:
            try:
                res["roofline"]["traffic_cited"] = json.load(open(tfile)).get("hbm_bytes_per_launch")
            except Exception:
                pass

        if world == 1:
            # ---- sustained window: >= sustained_s seconds of back-to-back launches (clocks settle at sustained power)
            if args.sustained_s > 0:
                m = int(min(4000, max(100, args.sustained_s / (k_ms * 1e-3)))) if k_ms == k_ms else 0
                if m:
                    ch.set_option(L.PFB_OPT_PROFILE, 1)
                    ts0 = time.perf_counter()
                    for _ in range(m):
                        step()
                    ch.sync()
                    wall = time.perf_counter() - ts0
                    s_ms = float(np.mean(kernel_ms_per_step(ch, 1)))
                    ch.set_option(L.PFB_OPT_PROFILE, 0)
                    res["roofline"]["sustained_frac"] = round(n * bytes_per_sample / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                    res["roofline"]["sustained_window_s"] = round(wall, 3)
                    res["roofline"]["sustained_kernel_ms"] = round(s_ms, 4)
            # ---- the yardstick: the fastest 1:2 streaming shape known on this part, same process, same box
            copy_bps = L.C.c_double()
            copy_rc = L.load().pfb_measure_stream_copy(local_rank, 1 << 30, 10, L.C.byref(copy_bps))
            res["roofline"]["measured_stream_copy_gbs"] = round(copy_bps.value / 1e9, 1) if copy_rc == 0 else None
            # ---- what a copy kernel with NO arithmetic reaches for each byte mix, by wave lifetime (rows of 256 B per wave):
            # the bound the fractions above and below are to be read against
            bounds = {}
            for ratio in (2, 4):
                for spw in (2, 8, 512):
                    bps = L.C.c_double()
                    if L.load().pfb_measure_mix_copy(local_rank, 1 << 30, ratio, spw, 5, L.C.byref(bps)) == 0:
                        bounds[f"1_read_{ratio}_written_{spw}_rows_per_wave"] = round(bps.value / 1e9 / HBM_PEAK_GBS, 4)
            res["roofline"]["copy_kernel_frac_by_byte_mix"] = bounds
        prefix = iq[: 1 << 28].cpu().numpy() if (world == 1 and not args.no_cpu_baseline and fmt == "int16") else None
        if ch is not None:
            ch.release()
        del iq, out
        torch.cuda.empty_cache()

        # ---- the other BASELINE configurations, short timed passes (N = 1 only; cfg4 at its per-GPU size)
        if world == 1 and not args.no_other_workloads and args.workload == "cfg2" and not args.channel_major:
            others = []
            for name, cm in OTHER_WORKLOADS:
                oM, oP, oD, ofmt, obw, obytes, olog2 = WORKLOADS[name]
                try:
                    och, _ = make_handle(name, cm, tuned=False)
                    oiq, oout = make_buffers(name, olog2, cm)
                    for _ in range(12):
                        och(oiq, out=oout, sync=False)
                    och.sync()
                    och.set_option(L.PFB_OPT_PROFILE, 1)
                    for _ in range(20):
                        och(oiq, out=oout, sync=False)
                    och.sync()
                    oms = float(np.mean(kernel_ms_per_step(och, 1)))
                    ob = obytes + 8 * (oM // oD)
                    ogbs = (1 << olog2) * ob / (oms * 1e-3) / 1e9
                    others.append({"workload": name, "layout": "channel-major" if cm else "frame-major",
                                   "shape": f"M={oM} P={oP} D={oD} {ofmt}, 2^{olog2} samples", "kernel": och.last_kernel,
                                   "kernel_ms": round(oms, 4), "achieved": round(ogbs, 1), "frac": round(ogbs / HBM_PEAK_GBS, 4),
                                   "ms_value": round((1 << olog2) / (oms * 1e-3) / 1e6, 1),
                                   "algorithmic_bytes_per_sample": ob, "launches_timed": 20})
                    if name == "cfg5" and not cm:
                        # config 5 end to end (create_pdws_channelized.m:57-143): the matrix just written -> PDWs, device-resident
                        from sdr_channelizer_amd.pdw import extract_pdws
                        F5 = (1 << olog2) // oD
                        y5 = oout[:F5]
                        extract_pdws(y5, synth.FS, 915e6, 0.0, decimation=oD)  # warm-up: scratch arenas
                        tp = []
                        for _ in range(3):
                            torch.cuda.synchronize(dev)
                            tq = time.perf_counter()
                            pd = extract_pdws(y5, synth.FS, 915e6, 0.0, decimation=oD)
                            tp.append((time.perf_counter() - tq) * 1e3)
                        others[-1]["end_to_end"] = {"what": "channelizer kernel + PDW extraction of its matrix (host wall, results on the host)",
                                                    "pdw_extraction_ms": round(min(tp), 3), "pulses": int(len(pd)),
                                                    "total_ms": round(oms + min(tp), 3)}
                        # the raw-stream extractor (create_pdws.m:30-105) on the same recorder stream, device-resident
                        from sdr_channelizer_amd.pdw import extract_pdws_raw
                        extract_pdws_raw(oiq, synth.FS, 915e6, 0.0, snr_threshold_db=12.0)
                        tr = []
                        for _ in range(3):
                            torch.cuda.synchronize(dev)
                            tq = time.perf_counter()
                            pr = extract_pdws_raw(oiq, synth.FS, 915e6, 0.0, snr_threshold_db=12.0)
                            tr.append((time.perf_counter() - tq) * 1e3)
                        others[-1]["raw_stream_pdws"] = {"what": "PDW extraction straight from the recorder stream (create_pdws.m), host wall, results on the host",
                                                         "samples": int(oiq.shape[0]), "ms": round(min(tr), 3), "pulses": int(len(pr))}
                    och.release()
                    del oiq, oout
                    torch.cuda.empty_cache()
                except Exception as e:  # a failing side workload must not cost the headline line
                    others.append({"workload": name, "layout": "channel-major" if cm else "frame-major", "error": repr(e)})
            res["other_workloads"] = others

        if prefix is not None:
            res["cpu_baseline"] = cpu_baseline(prefix, taps, M, P, D, bw, args.cpu_budget_s)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    elif ch is not None:
        ch.release()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
