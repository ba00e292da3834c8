#!/usr/bin/env python3
"""roofline fraction of the cf32 fused shapes (B = 8 + 8 M/D bytes per sample)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdr_channelizer_amd import Channelizer, design_prototype

for M, P, D, log2n in ((56, 12, 56, 27), (64, 12, 64, 28), (128, 12, 64, 27), (256, 8, 256, 28), (560, 12, 560, 27), (1024, 16, 1024, 28)):
    n = (1 << log2n) // D * D
    iq = (torch.randn(n, 2, device="cuda") * 0.3)
    with Channelizer(M, taps=design_prototype(M, P), decimation=D, sample_format="cf32") as ch:
        y = ch(iq)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            ch.reset(); y = ch(iq)
        e0.record()
        for _ in range(10):
            ch.reset(); y = ch(iq)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        B = 8 + 8 * M / D
        print(f"M={M} P={P} D={D} cf32 {ch.last_kernel}: {ms:.3f} ms per {n} samples = {n * B / ms / 1e9:.2f} TB/s = {n * B / ms / 1e9 / 8 * 100:.1f} %")
    del iq, y
