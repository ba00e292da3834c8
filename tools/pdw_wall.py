#!/usr/bin/env python3
"""host-side wall of extract_pdws on config 5's matrix, many repetitions (PFB_PDW_DEBUG=1 prints the C side's phases)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdr_channelizer_amd import Channelizer, design_prototype, synth
from sdr_channelizer_amd.pdw import extract_pdws
M, P, D = 128, 12, 64
iq = synth.pulsed_iq_torch(1 << 28, 12, device="cuda")
ch = Channelizer(M, taps=design_prototype(M, P), decimation=D, bit_width=12, fftshift=True)
y = ch(iq); torch.cuda.synchronize()
ts = []
for rep in range(12):
    t0 = time.perf_counter(); got = extract_pdws(y, 56e6, 915e6, 0.0, decimation=D); ts.append((time.perf_counter() - t0) * 1e3)
print("python wall ms:", " ".join(f"{t:.3f}" for t in ts))
