#!/usr/bin/env python3
"""Writes tests/golden/matlab_in_<cfg>.mat: the fixed integer input of every committed golden fixture as a MATLAB v5
file, for tools/export_matlab_golden.m (the one-file pin of the arithmetic against dsp.Channelizer; no MATLAB is
needed to run THIS script).  Input = the `iq` array already held by tests/golden/<cfg>.npz, so a MATLAB export and the
oracle's `expected` refer to the same samples.  Run from the repo root: python tests/golden/make_matlab_inputs.py"""
import os

import numpy as np
import scipy.io

HERE = os.path.dirname(os.path.abspath(__file__))
FS = 56e6

for name in ("cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "ref56", "ref560"):
    g = np.load(os.path.join(HERE, f"{name}.npz"))
    iq = g["iq"]
    scipy.io.savemat(os.path.join(HERE, f"matlab_in_{name}.mat"),
                     dict(iq=iq, M=int(g["M"]), P=int(g["P"]), D=int(g["D"]), bit_width=int(g["bit_width"]), fs=FS,
                          is_float=bool(iq.dtype.kind == "f")), do_compression=True)
    print(name, iq.shape, iq.dtype)
