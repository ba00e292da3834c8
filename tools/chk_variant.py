#!/usr/bin/env python3
"""Every fused plan registered for a critically sampled int16 shape (PFB_OPT_VARIANT 0, 1, ...) against the generic
kernel on the same stream: usage  tools/chk_variant.py [M P]   (default 1024 16)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_channelizer_amd import Channelizer, PfbError, design_prototype, synth  # noqa: E402
from sdr_channelizer_amd import _lib as L  # noqa: E402

M, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 16)
n = M * 4000 + 5
iq = synth.pulsed_iq_torch(n, 12, torch.int16, device=torch.device("cuda", 0))
with Channelizer(M, taps=design_prototype(M, P), decimation=M, sample_format="int16", bit_width=12) as ch:
    ch.set_option(L.PFB_OPT_KERNEL, 1)
    generic = ch(iq).clone()
    peak = generic.abs().max().item()
    ch.set_option(L.PFB_OPT_KERNEL, 2)
    for v in range(16):
        try:
            ch.set_option(L.PFB_OPT_VARIANT, v)
        except PfbError:
            break
        ch.reset()
        y = ch(iq)
        print(v, ch.last_kernel, "max err vs generic / max|y| =", (y - generic).abs().max().item() / peak)
