#!/usr/bin/env python3
"""max |y_gpu - y_oracle| / max |y_oracle| per fused shape on a 2^20-sample prefix of the benchmark stream
(SURVEY.md section 8d's error metric; the bound is 1e-5)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pfb_oracle import COracle, OracleConfig  # noqa: E402  (lives under tests/: only test infrastructure may use the oracle)
from sdr_channelizer_amd import Channelizer, design_prototype, synth  # noqa: E402

SHAPES = [(64, 12, 64, "int16", 12), (256, 8, 256, "int8", 8), (1024, 16, 1024, "int16", 16), (128, 12, 64, "int16", 12),
          (56, 12, 56, "int16", 12), (560, 12, 560, "int16", 12), (8, 12, 8, "int16", 12), (16, 12, 16, "int16", 12),
          (20, 12, 20, "int16", 12), (40, 12, 40, "int16", 12)]
o = COracle()
n = 1 << 20
for M, P, D, fmt, bw in SHAPES:
    m = (n // D) * D
    iq = synth.pulsed_iq_torch(m, bw, torch.int8 if fmt == "int8" else torch.int16, seed=synth.SEED, device="cuda")
    h = design_prototype(M, P, 80.0)
    with Channelizer(M, taps=h, decimation=D, sample_format=fmt, bit_width=bw) as ch:
        y = ch(iq).cpu().numpy()
        name = ch.last_kernel
    x = o.unpack(iq.cpu().numpy(), bw)
    want = o.channelize(x, h.astype(np.float64), OracleConfig(M, P, D), "fft" if (M & (M - 1)) == 0 else "polyphase")
    print(f"{name:40s} max rel err {np.abs(y - want).max() / np.abs(want).max():.2e}")
