#!/usr/bin/env python3
"""Variants of the slower mixed-radix shapes: tools/mixed_probe2.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_channelizer_amd import Channelizer, PfbError, design_prototype, synth  # noqa: E402
from sdr_channelizer_amd import _lib as L  # noqa: E402

dev = torch.device("cuda", 0)
for M in [int(a) for a in sys.argv[1:]] or [25, 48, 50, 320, 400]:
    h = design_prototype(M, 12)
    n = M * 3000 + 7
    iq = synth.pulsed_iq_torch(n, 12, torch.int16, device=dev)
    nbig = (1 << 28) // M * M
    big = synth.pulsed_iq_torch(nbig, 12, torch.int16, device=dev)
    out = torch.empty((nbig // M + 1, M), dtype=torch.complex64, device=dev)
    with Channelizer(M, taps=h, bit_width=12) as ch:
        ch.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ch.set_option(L.PFB_OPT_KERNEL, 1)
        gen = ch(iq).clone()
        ch.set_option(L.PFB_OPT_KERNEL, 2)
        for v in range(8):
            try:
                ch.set_option(L.PFB_OPT_VARIANT, v)
            except PfbError:
                break
            ch.set_option(L.PFB_OPT_SCHEDULE, -1)
            ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, 0)
            ch.reset()
            err = ((ch(iq) - gen).abs().max() / gen.abs().max()).item()
            res = []
            for sched, fpb in ((-1, 0), (0, 64), (0, 128), (0, 256), (0, 512), (7, 256), (7, 512), (6, 128), (6, 256), (11, 64), (11, 128)):
                try:
                    ch.set_option(L.PFB_OPT_SCHEDULE, sched)
                    ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, fpb)
                    ch.reset()
                    for _ in range(3):
                        ch(big, out=out, sync=False)
                    ch.sync()
                    ch.set_option(L.PFB_OPT_PROFILE, 1)
                    for _ in range(6):
                        ch(big, out=out, sync=False)
                    t = np.median(ch.kernel_times_ms())
                    ch.set_option(L.PFB_OPT_PROFILE, 0)
                    res.append((sched, fpb, nbig * 12 / (t * 1e-3) / 8e12))
                except PfbError:
                    pass
            best = max(res, key=lambda r: r[2])
            print(f"M={M:4d} v{v} {ch.last_kernel:40s} err={err:.1e} " + " ".join(f"s{r[0]}/{r[1]}={r[2]:.3f}" for r in res) + f"  BEST s{best[0]}/{best[1]} {best[2]:.3f}", flush=True)
    del big, out
    torch.cuda.empty_cache()
