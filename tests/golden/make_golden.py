"""Generates tests/golden/*.npz.

The reference ships no fixtures, tests or sample data and its
channelizer arithmetic is MathWorks' closed dsp.Channelizer, so these vectors come
from OUR float64 restatement (oracle/pfb_oracle.py::channelize_numpy, the numpy
formulation; tests check the independent C formulations against them).  They pin the
oracle against drift and give the GPU tests fixed known answers -- they are NOT
MATLAB outputs ("parity unpinned", see oracle/pfb_oracle.h).

One fixture per BASELINE.json config shape at reduced N:
  cfg1  M=8    P=12 D=8    cf32            (channelizer_example.m defaults at M=8)
  cfg2  M=64   P=12 D=64   int16, 12-bit   (blade_record_iq_12bit format)
  cfg3  M=256  P=8  D=256  int8,  8-bit    (usrp_record_iq_08bit format)
  cfg4  M=1024 P=16 D=1024 int16, 16-bit   (usrp_record_iq_12bit: sc16 host format)
  cfg5  M=128  P=12 D=64   int16, 12-bit   (2x oversampled)
  ref56 M=56   P=12 D=56   int16, 12-bit   (the reference's own M = fs*1e-6)
  ref560 M=560 P=12 D=560  int16, 12-bit   (its training-set M = round(fs/0.1e6),
                                            generate_channelized_training_iq.m:95-96)

Run from the repo root:  python tests/golden/make_golden.py [name ...]   (default: all)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pfb_oracle import COracle, OracleConfig, channelize_numpy, unpack_numpy  # noqa: E402
from sdr_channelizer_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (M, P, D, fmt, bit_width, frames)
    "cfg1": (8, 12, 8, "cf32", 0, 512),
    "cfg2": (64, 12, 64, "int16", 12, 160),
    "cfg3": (256, 8, 256, "int8", 8, 48),
    "cfg4": (1024, 16, 1024, "int16", 16, 40),
    "cfg5": (128, 12, 64, "int16", 12, 160),
    "ref56": (56, 12, 56, "int16", 12, 96),
    "ref560": (560, 12, 560, "int16", 12, 24),
}


def main():
    o = COracle()
    for name in (sys.argv[1:] or list(CASES)):
        M, P, D, fmt, bw, frames = CASES[name]
        n = frames * D
        h = o.design_prototype(M, P, 80.0).astype(np.float32)  # what the product is handed
        if fmt == "cf32":
            iq16 = synth.pulsed_iq_numpy(n, 16, np.int16, seed=synth.SEED + M)
            iq = (iq16.astype(np.float32) / 32768.0).astype(np.float32)
            x = iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)
        else:
            dt = np.int8 if fmt == "int8" else np.int16
            iq = synth.pulsed_iq_numpy(n, bw, dt, seed=synth.SEED + M)
            x = unpack_numpy(iq, bw)
        y = channelize_numpy(x, h.astype(np.float64), OracleConfig(M, P, D))
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), M=M, P=P, D=D, fmt=fmt, bit_width=bw, iq=iq,
                            taps=h, expected=y)
        print(name, iq.shape, iq.dtype, y.shape, f"max|y|={abs(y).max():.4f}")


if __name__ == "__main__":
    main()
