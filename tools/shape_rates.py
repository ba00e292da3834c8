#!/usr/bin/env python3
"""Kernel time and roofline fraction of every frame-major fused shape with its default plan at 2^28 samples
(device-resident, PFB_OPT_PROFILE events, median of 12 calls): usage  tools/shape_rates.py [M,P,D,fmt,bits ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_channelizer_amd import Channelizer, design_prototype, synth  # noqa: E402
from sdr_channelizer_amd import _lib as L  # noqa: E402

SHAPES = [(8, 12, 8, "cf32", 1), (8, 12, 8, "int16", 12), (10, 12, 10, "int16", 12), (16, 12, 16, "int16", 12), (20, 12, 20, "int16", 12),
          (32, 12, 32, "int16", 12), (40, 12, 40, "int16", 12), (56, 12, 56, "int16", 12), (64, 12, 64, "int16", 12), (64, 12, 64, "int8", 8),
          (128, 12, 64, "int16", 12), (256, 8, 256, "int8", 8), (256, 8, 256, "int16", 12), (560, 12, 560, "int16", 12),
          (560, 12, 560, "int8", 8), (1024, 16, 1024, "int16", 12), (1024, 16, 1024, "cf32", 1)]
SHAPES += [(M, 12, M, "int16", 12) for M in (12, 24, 25, 30, 48, 50, 80, 96, 100, 112, 120, 160, 200, 250, 280, 320, 400, 500, 512)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) if v.isdigit() else v for v in a.split(",")) for a in sys.argv[1:]]
n = 1 << 28
bufs = {}
for M, P, D, fmt, bw in SHAPES:
    if fmt not in bufs:
        bufs[fmt] = (torch.randn((n, 2), dtype=torch.float32, device="cuda") if fmt == "cf32" else
                     synth.pulsed_iq_torch(n, bw, torch.int8 if fmt == "int8" else torch.int16, device="cuda"))
    iq = bufs[fmt]
    bytes_in = {"int8": 2, "int16": 4, "cf32": 8}[fmt]
    with Channelizer(M, taps=design_prototype(M, P), decimation=D, sample_format=fmt, bit_width=bw) as ch:
        ch.set_option(L.PFB_OPT_KERNEL, 2)
        out = torch.empty((n // D + 1, M), dtype=torch.complex64, device="cuda")
        for _ in range(6):
            ch.reset()
            ch(iq, out=out, sync=False)
        ch.sync()
        ch.set_option(L.PFB_OPT_PROFILE, 1)
        for _ in range(12):
            ch.reset()
            ch(iq, out=out, sync=False)
        t = float(np.median(ch.kernel_times_ms()))
        gbs = (n // D) * (D * bytes_in + M * 8) / (t * 1e-3) / 1e9
        print(f"M={M:4d} P={P:2d} D={D:4d} {fmt:5s} {ch.last_kernel:44s} {t:7.4f} ms  {gbs:7.1f} GB/s  frac {gbs / 8000:.3f}", flush=True)
    del out
