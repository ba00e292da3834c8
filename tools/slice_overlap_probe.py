#!/usr/bin/env python3
"""Config 5 end to end, route (2) of DESIGN.md section 9, priced before it is built: channelize the stream slice by slice on
one stream while a second stream reads every finished slice back (a torch reduction standing in for the PDW bracket
pass: a plain streaming read of the slice).  If fresh slices come from the 256 MB Infinity Cache and the two kernels share
the chip, the total approaches the channelizer alone; if not, slicing only adds launch tails."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_channelizer_amd import Channelizer, design_prototype, synth  # noqa: E402

M, P, D, n = 128, 12, 64, 1 << 28
dev = torch.device("cuda", 0)
iq = synth.pulsed_iq_torch(n, 12, torch.int16, device=dev)
F = n // D
y = torch.empty((F, M), dtype=torch.complex64, device=dev)
yf = y.view(torch.float32)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ch = Channelizer(M, taps=design_prototype(M, P), decimation=D, bit_width=12, fftshift=True, derotate=True)
ch.set_stream(sa.cuda_stream)
acc = torch.zeros((), dtype=torch.float32, device=dev)


def run(slice_frames, overlap):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ch.reset()
    torch.cuda.synchronize()
    e0.record(sa)
    evs = []
    for f0 in range(0, F, slice_frames):
        f1 = min(F, f0 + slice_frames)
        ch(iq[f0 * D:f1 * D], out=y[f0:f1], sync=False)
        ev = torch.cuda.Event()
        ev.record(sa)
        evs.append((f0, f1, ev))
        if overlap:
            sb.wait_event(ev)
            with torch.cuda.stream(sb):
                acc.add_(yf[f0:f1].amax())
    if not overlap:
        sb.wait_event(evs[-1][2])
        with torch.cuda.stream(sb):
            for f0, f1, _ in evs:
                acc.add_(yf[f0:f1].amax())
    done = torch.cuda.Event()
    done.record(sb)
    sa.wait_event(done)
    e1.record(sa)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for frames_mb in (None, 4096, 1024, 512, 256, 128, 64):
    sf = F if frames_mb is None else frames_mb * (1 << 20) // (M * 8)
    for overlap in (False, True):
        ts = sorted(run(sf, overlap) for _ in range(5))
        print(f"slices of {('the whole matrix' if frames_mb is None else str(frames_mb) + ' MB'):>16s} ({(F + sf - 1) // sf:4d} slices)  "
              f"{'reader behind every slice' if overlap else 'reader after the last slice':28s}  median {ts[2]:7.3f} ms  min {ts[0]:7.3f} ms", flush=True)
