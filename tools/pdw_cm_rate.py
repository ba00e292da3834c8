#!/usr/bin/env python3
"""PDW extraction from MATLAB's own layout (M x F channel-major) against frame-major, config 5's matrix"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdr_channelizer_amd import Channelizer, design_prototype, synth
from sdr_channelizer_amd.pdw import extract_pdws
M, P, D = 128, 12, 64
iq = synth.pulsed_iq_torch(1 << 28, 12, device="cuda")
ch = Channelizer(M, taps=design_prototype(M, P), decimation=D, bit_width=12, fftshift=True)
y = ch(iq)
ycm = y.t().contiguous()
torch.cuda.synchronize()
for name, arr, cm in (("frame-major", y, False), ("channel-major", ycm, True)):
    for _ in range(2):
        extract_pdws(arr, 56e6, 915e6, 0.0, decimation=D, channel_major=cm)
    t0 = time.perf_counter()
    for _ in range(5):
        got = extract_pdws(arr, 56e6, 915e6, 0.0, decimation=D, channel_major=cm)
    print(f"{name}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms, {len(got)} pulses")
