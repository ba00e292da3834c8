#!/usr/bin/env python3
"""Interleaved A/B timing of channelizer option sets in ONE process (same data, same clocks).

usage: tools/ab.py [--log2-samples 30] [--rounds 5] [--steps 5] "name:opt=val,opt=val" ...
options: kernel, fpb, nt, remap, sched, grid, tw  (see PFB_OPT_* in include/pfb_channelizer.h)
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from sdr_channelizer_amd import Channelizer, design_prototype, synth  # noqa: E402
from sdr_channelizer_amd import _lib as L  # noqa: E402

OPT = {"kernel": L.PFB_OPT_KERNEL, "fpb": L.PFB_OPT_FRAMES_PER_BLOCK, "nt": L.PFB_OPT_NONTEMPORAL,
       "remap": L.PFB_OPT_XCD_REMAP, "sched": L.PFB_OPT_SCHEDULE, "grid": L.PFB_OPT_GRID, "tw": L.PFB_OPT_TILE_WAVES, "exp": L.PFB_OPT_EXPERIMENT, "var": L.PFB_OPT_VARIANT,
       "slab": L.PFB_OPT_SLAB_FRAMES}
DEFAULTS = {"kernel": 2, "fpb": 0, "nt": 0, "remap": -1, "sched": -1, "grid": 0, "tw": 8, "exp": 0, "var": 0, "slab": 0}

ap = argparse.ArgumentParser()
ap.add_argument("--log2-samples", type=int, default=30)
ap.add_argument("--samples", type=int, default=0, help="exact sample count (overrides --log2-samples)")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--workload", default="64,12,64,int16,12")
ap.add_argument("--magnitude", action="store_true")
ap.add_argument("--channel-major", action="store_true", help="MATLAB's column-major F x M output")
ap.add_argument("--out-offset-kib", type=int, default=0, help="shift the output buffer start by this many KiB")
ap.add_argument("--exact-out", action="store_true", help="allocate exactly frames*M outputs like bench.py")
ap.add_argument("cases", nargs="+")
a = ap.parse_args()

M, P, D, fmt, bw = a.workload.split(",")
M, P, D, bw = int(M), int(P), int(D), int(bw)
n = a.samples if a.samples > 0 else 1 << a.log2_samples
dev = torch.device("cuda", 0)
if fmt == "cf32":
    iq = torch.randn((n, 2), dtype=torch.float32, device=dev)
else:
    iq = synth.pulsed_iq_torch(n, bw, torch.int8 if fmt == "int8" else torch.int16, device=dev)
odt = torch.float32 if a.magnitude else torch.complex64
if a.channel_major:
    out = torch.empty((M, n // D + 1), dtype=odt, device=dev)  # +1: a carried tail can complete one more frame
elif a.exact_out:
    out = torch.empty((n // D, M), dtype=odt, device=dev)
else:
    pad_rows = (a.out_offset_kib * 1024) // (M * (4 if a.magnitude else 8)) + 1
    big = torch.empty((n // D + 1 + pad_rows, M), dtype=odt, device=dev)
    out = big[pad_rows - 1:]
print(f"# in ptr {iq.data_ptr():#x} out ptr {out.data_ptr():#x} delta {(out.data_ptr() - iq.data_ptr()) / 2**20:.3f} MiB")
ch = Channelizer(M, taps=design_prototype(M, P), decimation=D, sample_format=fmt, bit_width=max(bw, 1), magnitude=a.magnitude,
                 channel_major=a.channel_major)
ch.set_stream(torch.cuda.current_stream(dev).cuda_stream)
cases = []
for c in a.cases:
    name, _, rest = c.partition(":")
    opts = dict(DEFAULTS)
    for kv in filter(None, rest.split(",")):
        k, v = kv.split("=")
        opts[k] = int(v)
    cases.append((name, opts))
times = {name: [] for name, _ in cases}
bytes_per_sample = {"int8": 2, "int16": 4, "cf32": 8}[fmt] + (4 if a.magnitude else 8) * (M // D)
ch.set_option(L.PFB_OPT_PROFILE, 1)
for r in range(a.rounds + 1):
    for name, opts in cases:
        for k, v in opts.items():
            ch.set_option(OPT[k], v)
        for _ in range(a.steps):
            ch(iq, out=out, sync=False)
        t = ch.kernel_times_ms()
        if r:  # round 0 is warm-up
            times[name] += t
cp = L.C.c_double()
L.load().pfb_measure_stream_copy(0, 1 << 30, 10, L.C.byref(cp))
print(f"# box yardstick: 1:2 stream copy {cp.value / 1e9:.1f} GB/s")
print(f"{'case':34s} {'min ms':>8s} {'med ms':>8s} {'max ms':>8s} {'GB/s(med)':>10s} {'frac':>6s}  kernel")
for name, opts in cases:
    t = np.array(times[name])
    for k, v in opts.items():
        ch.set_option(OPT[k], v)
    ch(iq, out=out, sync=True)
    gbs = n * bytes_per_sample / (np.median(t) * 1e-3) / 1e9
    print(f"{name:34s} {t.min():8.4f} {np.median(t):8.4f} {t.max():8.4f} {gbs:10.1f} {gbs / 8000:6.3f}  {ch.last_kernel}")
