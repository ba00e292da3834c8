#!/usr/bin/env python3
"""Does the PDW bracket pass run faster on a matrix that fits the 256 MB Infinity Cache?  Same columns (M = 32),
F = 2^19 frames (134 MB, re-read call after call) against F = 2^23 (2.1 GB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdr_channelizer_amd.pdw import extract_pdws

for log2f in (19, 20, 23):
    F, M = 1 << log2f, 32
    y = (0.01 * torch.randn(F, M, 2, device="cuda")).view(torch.float32)
    y = torch.view_as_complex(y.reshape(F, M, 2).contiguous())
    y[1000:1100, 3] += 0.5
    for _ in range(3):
        extract_pdws(y, 32e6, 1e9, 0.0)
    t0 = time.perf_counter()
    for _ in range(10):
        got = extract_pdws(y, 32e6, 1e9, 0.0)
    dt = (time.perf_counter() - t0) / 10
    print(f"F=2^{log2f} M={M} ({F * M * 8 >> 20} MB): {dt * 1e3:.3f} ms per extraction, {len(got)} pulses")
    del y
