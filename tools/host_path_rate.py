#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer path (pfb_process with PFB_MEM_HOST): numpy in, numpy out."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_channelizer_amd import Channelizer, design_prototype, pinned_empty, synth  # noqa: E402
from sdr_channelizer_amd import _lib as L  # noqa: E402

n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 27
M, P = 64, 12
iq = synth.pulsed_iq_numpy(1 << 22, 12, np.int16)
iq = np.ascontiguousarray(np.tile(iq, (n // iq.shape[0], 1)))
out = np.empty((n // M, M), np.complex64)
out[:] = 0  # touch the pages
ch = Channelizer(M, taps=design_prototype(M, P), bit_width=12)
iq_p = pinned_empty(iq.shape, iq.dtype)
iq_p[:] = iq
out_p = pinned_empty(out.shape, out.dtype)
out_p[:] = 0
for name, a, o in (("pageable", iq, out), ("page-locked", iq_p, out_p)):
    for chunk in (1 << 22, 1 << 24, 1 << 26):
        ch.set_option(L.PFB_OPT_HOST_CHUNK_SAMPLES, chunk)
        ch.reset(); ch(a, out=o)
        t0 = time.perf_counter(); ch.reset(); ch(a, out=o); dt = time.perf_counter() - t0
        print(f"host path, {name:11s} chunk 2^{chunk.bit_length() - 1}: {n / dt / 1e6:9.1f} MS/s  ({n * 12 / dt / 1e9:6.2f} GB/s over PCIe, {dt * 1e3:.1f} ms for 2^{n.bit_length() - 1} samples)")
assert np.array_equal(out, out_p)
