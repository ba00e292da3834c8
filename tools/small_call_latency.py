#!/usr/bin/env python3
"""Per-call cost of pfb_process on device-resident dwell buffers of 2^k samples (the recorder-loop shape: one call per
dwell): synchronous calls, and asynchronous calls with one sync at the end."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_channelizer_amd import Channelizer, design_prototype, synth  # noqa: E402

M, P = 64, 12
ch = Channelizer(M, taps=design_prototype(M, P), bit_width=12)
for k in (12, 16, 20, 24):
    n = 1 << k
    iq = synth.pulsed_iq_torch(n, 12, device="cuda")
    out = torch.empty((n // M + 1, M), dtype=torch.complex64, device="cuda")
    reps = 200
    for sync in (True, False):
        for _ in range(20):
            ch(iq, out=out, sync=sync)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ch(iq, out=out, sync=sync)
        ch.sync()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"2^{k:2d} samples per call, {'sync ' if sync else 'async'}: {dt * 1e6:8.1f} us per call = {n / dt / 1e6:9.1f} MS/s")
