#!/usr/bin/env python3
"""config 5 end to end on the GPU: channelize a 2^28-sample stream (M=128, D=64), then PDW extraction."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sdr_channelizer_amd import Channelizer, design_prototype, synth  # noqa: E402
from sdr_channelizer_amd.pdw import extract_pdws  # noqa: E402

M, P, D = 128, 12, 64
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
iq = synth.pulsed_iq_torch(n, 12, device="cuda")
ch = Channelizer(M, taps=design_prototype(M, P), decimation=D, bit_width=12, fftshift=True)
y = ch(iq)
torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter()
    got = extract_pdws(y, 56e6, 915e6, 0.0, decimation=D)
    dt = time.perf_counter() - t0
    print(f"PDW extraction: F={y.shape[0]} M={M}: {len(got)} pulses in {dt * 1e3:.1f} ms")

if os.environ.get("PDW_BENCH_SKIP_RAW"):
    sys.exit(0)
# raw-stream extraction (create_pdws.m) on the same recorder stream; its pulses stand 14 dB (dB/10) above the floor
from sdr_channelizer_amd.pdw import extract_pdws_raw  # noqa: E402

for rep in range(4):
    t0 = time.perf_counter()
    got = extract_pdws_raw(iq, 56e6, 915e6, 0.0, snr_threshold_db=12.0)
    dt = time.perf_counter() - t0
    print(f"raw PDW extraction: n={n}: {len(got)} pulses in {dt * 1e3:.1f} ms")
