#!/usr/bin/env python3
"""PDW extraction rate on small banks (the bracket pass packs 64 / LPR rows into a wave-load when M <= 32)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdr_channelizer_amd.pdw import extract_pdws
for M in [int(a) for a in sys.argv[1:]] or (8, 16, 32, 56, 64):
    F = (1 << 28) // M
    y = torch.view_as_complex((0.01 * torch.randn(F, M, 2, device="cuda")).contiguous())
    y[1000:1100, 3] += 0.5
    for _ in range(2):
        extract_pdws(y, M * 1e6, 1e9, 0.0)
    t0 = time.perf_counter()
    for _ in range(5):
        got = extract_pdws(y, M * 1e6, 1e9, 0.0)
    dt = (time.perf_counter() - t0) / 5
    print(f"M={M:3d} F={F} ({F * M * 8 >> 20} MB): {dt * 1e3:.3f} ms per extraction = {F * M * 8 / dt / 1e12:.2f} TB/s of matrix, {len(got)} pulses")
    del y
