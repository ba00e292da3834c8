#!/usr/bin/env python3
"""What the PDW screen costs inside the channelizer's last pass (schedule 12, pfb_probe_pdw_fused) at config 5:
thresholds from the true per-channel medians of |y|^2 (bracket +-3 %, threshold = 31.6^2 x median, band +-3 %)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sdr_channelizer_amd import Channelizer, design_prototype, synth
from sdr_channelizer_amd import _lib as L

M, P, D = 128, 12, 64
n = 1 << 28
iq = synth.pulsed_iq_torch(n, 12, device="cuda")
ch = Channelizer(M, taps=design_prototype(M, P), decimation=D, bit_width=12, fftshift=True)
y = ch(iq)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    ch.reset(); ch(iq)
e0.record()
for _ in range(10):
    ch.reset(); ch(iq)
e1.record(); torch.cuda.synchronize()
print(f"plain channelizer (schedule 11): {e0.elapsed_time(e1) / 10:.3f} ms")
ch.set_option(L.PFB_OPT_SCHEDULE, 0)
e0.record()
for _ in range(10):
    ch.reset(); ch(iq)
e1.record(); torch.cuda.synchronize()
print(f"plain channelizer (schedule 0, 64-frame runs): {e0.elapsed_time(e1) / 10:.3f} ms")
ch.set_option(L.PFB_OPT_SCHEDULE, -1)
m2 = (y.real.double() ** 2 + y.imag.double() ** 2)
med = m2[:: 64].median(dim=0).values.cpu().numpy()   # a 1-in-64 row sample is plenty for the probe
lib = L.load()
for name, w in (("no candidates (limits at 0)", None), ("bracket +-1 %", 0.01), ("bracket +-3 %", 0.03)):
    thr = np.zeros((M, 4), np.float32)
    if w is None:
        thr[:, 0] = 0.0; thr[:, 1] = -1.0   # nothing below, everything "above": no zone
        thr[:, 2] = 3e38; thr[:, 3] = 3e38
    else:
        thr[:, 0] = med * (1 - w); thr[:, 1] = med * (1 + w)
        t2 = med * 31.6227766 ** 2
        thr[:, 2] = t2 * (1 - w); thr[:, 3] = t2 * (1 + w)
    ms = C.c_double(0.0)
    cnt = (C.c_uint64 * 3)()
    F = n // D
    rc = lib.pfb_probe_pdw_fused(ch._h, C.c_void_p(iq.data_ptr()), n, C.c_void_p(y.data_ptr()), F,
                                 thr.ctypes.data_as(C.POINTER(C.c_float)), 10, C.byref(ms), cnt)
    print(f"fused screen, {name}: rc={rc} {ms.value:.3f} ms  candidates {cnt[0]} ({cnt[0] / (F * M) * 100:.2f} %)  undecided {cnt[1]}  flags {cnt[2]}")
