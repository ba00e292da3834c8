#!/usr/bin/env python3
"""Short calls on the plans whose tuned runs are long: the default run length (shrunk until the call fills the chip,
pfb_api.cpp launch_frames) against the plan's tuned length forced with PFB_OPT_FRAMES_PER_BLOCK.  Kernel time per call
(PFB_OPT_PROFILE events), median of `reps` calls after a warm-up; outputs compared bit for bit."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_channelizer_amd import Channelizer, design_prototype, synth  # noqa: E402
from sdr_channelizer_amd import _lib as L  # noqa: E402

SHAPES = [  # M, P, D, fmt, bit width, tuned run length
    (8, 12, 8, "cf32", 1, 1024), (16, 12, 16, "int16", 12, 1024), (32, 12, 32, "int16", 12, 512), (56, 12, 56, "int16", 12, 512),
    (64, 12, 64, "int8", 8, 256), (120, 12, 120, "int16", 12, 512), (512, 12, 512, "int16", 12, 512),
    (560, 12, 560, "int8", 8, 252), (1024, 16, 1024, "cf32", 1, 256), (1024, 16, 1024, "int16", 12, 512),
]
reps = 30
for M, P, D, fmt, bw, tuned in SHAPES:
    for frames in (5000, 50000, 500000):
        n = frames * D
        if fmt == "cf32":
            iq = torch.randn((n, 2), dtype=torch.float32, device="cuda")
        else:
            iq = synth.pulsed_iq_torch(n, bw, torch.int8 if fmt == "int8" else torch.int16, device="cuda")
        res = {}
        outs = {}
        for name, fpb in (("auto", 0), ("tuned", tuned)):
            with Channelizer(M, taps=design_prototype(M, P), decimation=D, sample_format=fmt, bit_width=bw) as ch:
                ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, fpb)
                out = torch.empty((frames, M), dtype=torch.complex64, device="cuda")
                for _ in range(5):
                    ch.reset()
                    ch(iq, out=out, sync=False)
                ch.sync()
                ch.set_option(L.PFB_OPT_PROFILE, 1)
                for _ in range(reps):
                    ch.reset()
                    ch(iq, out=out, sync=False)
                t = np.array(ch.kernel_times_ms())
                res[name] = float(np.median(t)) * 1e3
                outs[name] = out.clone()
        same = torch.equal(outs["auto"], outs["tuned"])
        print(f"M={M:5d} {fmt:5s} frames={frames:7d}: auto {res['auto']:8.1f} us   tuned runs ({tuned:4d}) {res['tuned']:8.1f} us   "
              f"x{res['tuned'] / res['auto']:.2f}   bits {'same' if same else 'DIFFER'}")
