import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_channelizer_amd import Channelizer, design_prototype, synth
from sdr_channelizer_amd import _lib as L
dev = torch.device("cuda", 0)
for M in (36, 90, 360, 560, 600, 22):
    n = (1 << 26) // M * M
    iq = synth.pulsed_iq_torch(n, 12, torch.int16, device=dev)
    out = torch.empty((n // M + 1, M), dtype=torch.complex64, device=dev)
    with Channelizer(M, taps=design_prototype(M, 12), bit_width=12) as ch:
        ch.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ch.set_option(L.PFB_OPT_KERNEL, 1)
        ch(iq, out=out, sync=True)
        ch.set_option(L.PFB_OPT_PROFILE, 1)
        for _ in range(3):
            ch(iq, out=out, sync=False)
        t = np.median(ch.kernel_times_ms())
        print(f"generic M={M}: {t:.3f} ms per 2^26 samples = {n * 12 / (t * 1e-3) / 8e12:.3f} of roofline", flush=True)
