#!/usr/bin/env python3
"""The band counts of pfb_kernels_mixed.hip on the GPU: parity of the fused kernel against the generic one on a short
stream, then the fused kernel's rate at 2^28 samples for the schedules the shape has.
usage: tools/mixed_probe.py [M ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_channelizer_amd import Channelizer, PfbError, design_prototype, synth  # noqa: E402
from sdr_channelizer_amd import _lib as L  # noqa: E402

Ms = [int(a) for a in sys.argv[1:]] or [12, 24, 25, 30, 48, 50, 80, 96, 100, 112, 120, 160, 200, 250, 280, 320, 400, 500, 512]
dev = torch.device("cuda", 0)
for M in Ms:
    P = 12
    n = M * 3000 + 7
    iq = synth.pulsed_iq_torch(n, 12, torch.int16, device=dev)
    h = design_prototype(M, P)
    with Channelizer(M, taps=h, bit_width=12) as ch:
        ch.set_option(L.PFB_OPT_KERNEL, 1)
        gen = ch(iq).clone()
        ch.set_option(L.PFB_OPT_KERNEL, 2)
        ch.reset()
        fast = ch(iq)
        name = ch.last_kernel
        err = ((fast - gen).abs().max() / gen.abs().max()).item()
        ch.reset()
        cut = M * 1001 + 3
        two = torch.cat([ch(iq[:cut]), ch(iq[cut:])])
        chunked = torch.equal(two, fast)
    nbig = (1 << 28) // M * M
    big = synth.pulsed_iq_torch(nbig, 12, torch.int16, device=dev)
    out = torch.empty((nbig // M + 1, M), dtype=torch.complex64, device=dev)
    res = []
    with Channelizer(M, taps=h, bit_width=12) as ch:
        ch.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ch.set_option(L.PFB_OPT_KERNEL, 2)
        for sched, fpb in ((-1, 0), (0, 64), (0, 128), (0, 512), (11, 64), (11, 128), (7, 256), (7, 512), (6, 128), (6, 256), (6, 512)):
            try:
                ch.set_option(L.PFB_OPT_SCHEDULE, sched)
                ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, fpb)
                for _ in range(3):
                    ch(big, out=out, sync=False)
                ch.sync()
                ch.set_option(L.PFB_OPT_PROFILE, 1)
                for _ in range(6):
                    ch(big, out=out, sync=False)
                t = np.median(ch.kernel_times_ms())
                ch.set_option(L.PFB_OPT_PROFILE, 0)
                # did the schedule really change the result?  (a schedule the shape lacks falls through to the default)
                res.append((sched, fpb, nbig * 12 / (t * 1e-3) / 8e12))
            except PfbError as e:
                res.append((sched, fpb, float("nan")))
    best = max(res, key=lambda r: (r[2] if r[2] == r[2] else 0))
    print(f"M={M:4d} {name:34s} err_vs_generic={err:.1e} chunked_bit_identical={chunked}  "
          + " ".join(f"s{r[0]}/{r[1]}={r[2]:.3f}" for r in res) + f"  BEST s{best[0]}/{best[1]} {best[2]:.3f}", flush=True)
    del big, out
    torch.cuda.empty_cache()
