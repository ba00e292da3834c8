#!/usr/bin/env python3
"""The two launches of one sharded step (interior frames at once, head frames behind the halo) timed separately: one handle
attached as rank 1 of 2 with a transport that moves nothing (the halo zone stays zero: timing only)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_channelizer_amd import Channelizer, design_prototype, synth  # noqa: E402
from sdr_channelizer_amd import _lib as L  # noqa: E402

for M, P, l2 in ((64, 12, 30), (1024, 16, 30), (560, 12, 28)):
    n = (1 << l2) // M * M
    iq = synth.pulsed_iq_torch(n, 12, torch.int16, device="cuda")
    with Channelizer(M, taps=design_prototype(M, P), bit_width=12) as ch:
        ch.attach_shard(1, 2, lambda *a: 0, ring=False)
        out = torch.empty((n // M, M), dtype=torch.complex64, device="cuda")
        for _ in range(5):
            ch.process_shard(iq, out=out)
        ch.sync()
        ch.set_option(L.PFB_OPT_PROFILE, 1)
        for _ in range(10):
            ch.process_shard(iq, out=out)
        ch.sync()
        t = np.array(ch.kernel_times_ms()).reshape(-1, 2)
        print(f"M={M:5d}: interior launch {np.median(t[:, 0]):7.4f} ms, head launch ({ch.shard_head_frames} frames) {np.median(t[:, 1]) * 1e3:7.1f} us")
    del out, iq
