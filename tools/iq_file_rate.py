#!/usr/bin/env python3
"""Rate of the .iq front end (pfb_process_iq_file): a 2^k-sample int16 record on tmpfs -> channels in host memory."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_channelizer_amd import Channelizer, design_prototype, iqfile, pinned_empty, synth  # noqa: E402

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 27)
M, P = 64, 12
base = synth.pulsed_iq_numpy(1 << 22, 12, np.int16)
iq = np.ascontiguousarray(np.tile(base, (n // base.shape[0], 1)))
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, iqfile.filename_for(1_700_000_000_000))
iqfile.write_iq(path, iq, fs=56e6, fc=915e6, bit_width=12)
try:
    with Channelizer(M, taps=design_prototype(M, P), bit_width=12, fftshift=True) as ch:
        out = pinned_empty((n // M, M), np.complex64)
        if len(sys.argv) > 2:
            from sdr_channelizer_amd import _lib as L
            ch.set_option(L.PFB_OPT_HOST_CHUNK_SAMPLES, 1 << int(sys.argv[2]))
        for rep in range(3):
            ch.reset()
            t0 = time.perf_counter()
            y, info = ch.process_iq_file(path, out=out)
            dt = time.perf_counter() - t0
            print(f".iq file -> channels: {n / dt / 1e6:8.1f} MS/s ({n * 4 / dt / 1e9:5.2f} GB/s of file, {dt * 1e3:.1f} ms for 2^{n.bit_length() - 1} samples)")
        from sdr_channelizer_amd.pdw import pdws_from_iq_file
        for rep in range(3):
            t0 = time.perf_counter()
            pdws, info = pdws_from_iq_file(ch, path)
            dt = time.perf_counter() - t0
            print(f".iq file -> PDWs (matrix stays on the GPU): {n / dt / 1e6:8.1f} MS/s ({n * 4 / dt / 1e9:5.2f} GB/s of file, {dt * 1e3:.1f} ms, {len(pdws)} pulses)")
finally:
    os.remove(path)
    os.rmdir(d)
