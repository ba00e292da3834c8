#!/usr/bin/env python3
"""Bit-identity of every registered plan of a shape against plan 0 (several call shapes: one shot with a ragged tail,
chunked calls, conjugate / fftshift / magnitude), then hands over to tools/ab.py-style timing via the command line.
usage: tools/twin_probe.py [M P]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_channelizer_amd import Channelizer, PfbError, design_prototype, synth  # noqa: E402
from sdr_channelizer_amd import _lib as L  # noqa: E402

M, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 16)
dev = torch.device("cuda", 0)
n = M * 3000 + 5
iq = synth.pulsed_iq_torch(n, 12, torch.int16, device=dev)
bad = 0
for kw in ({}, {"fftshift": True, "conjugate_input": True}, {"magnitude": True}):
    with Channelizer(M, taps=design_prototype(M, P), decimation=M, sample_format="int16", bit_width=12, **kw) as ch:
        ch.set_option(L.PFB_OPT_KERNEL, 2)
        ref = ch(iq).clone()
        for v in range(1, 16):
            try:
                ch.set_option(L.PFB_OPT_VARIANT, v)
            except PfbError:
                break
            for fpb in (0, 8, 24, 200):
                ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, fpb)
                ch.reset()
                y = ch(iq)
                same = torch.equal(y, ref)
                ch.reset()
                cut = M * 1111 + 3  # odd carried phase: the second call is misaligned for the vector loads
                y2 = torch.cat([ch(iq[:cut]), ch(iq[cut:])])
                same2 = torch.equal(y2, ref)
                if not (same and same2):
                    bad += 1
                    d = (y - ref).abs().max().item() / ref.abs().max().item()
                    d2 = (y2 - ref).abs().max().item() / ref.abs().max().item()
                    print("MISMATCH", kw, v, ch.last_kernel, "fpb", fpb, "one-shot", same, d, "chunked", same2, d2)
            print(kw, v, ch.last_kernel, "ok" if not bad else "")
print("bit-identity:", "ALL OK" if bad == 0 else f"{bad} FAILURES")
sys.exit(1 if bad else 0)
