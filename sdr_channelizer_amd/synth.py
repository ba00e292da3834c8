"""Synthetic pulsed I/Q in the recorders' integer formats.

Recipe restated from /root/reference/matlab/generate_pulsed_iq.m:12-19,63-77,83 and
/root/reference/matlab/generate_training_iq.m:12-26,95-98:
unit rectangular pulses (PW 100 us, PRI 1 ms at fs 56 Msps) on a complex carrier
f0 drawn uniformly from (-fs/2, fs/2), the carrier phase restarting at every pulse
(``phi(idx:idx+pw-1) = my_phi``), amplitude 0.5 full scale, plus complex Gaussian
noise of sigma 2^-6 full scale (the MATLAB generators are noise-free; noise makes the
downstream median/threshold logic meaningful), quantised round-to-nearest and saturated
to ``bit_width`` bits inside int8/int16 (``int16(x*2^15)`` saturates, generate_training_iq.m:95-98).

``pulsed_iq_numpy`` serves tests/fixtures; ``pulsed_iq_torch`` builds the benchmark
stream directly in HBM (chunked so temporaries stay small).
"""
from __future__ import annotations

import numpy as np

FS = 56e6
PW_S = 100e-6
PRI_S = 1000e-6
AMPLITUDE = 0.5
NOISE_SIGMA = 2.0 ** -6
SEED = 0x5D12C4A7


def _carrier(seed: int, fs: float) -> float:
    return float(-(fs / 2) + fs * np.random.default_rng(seed).random())


def pulsed_iq_numpy(n: int, bit_width: int = 12, dtype=np.int16, seed: int = SEED, fs: float = FS,
                    start: int = 0) -> np.ndarray:
    """(n, 2) integer I/Q; sample index start..start+n-1 of the infinite stream."""
    f0 = _carrier(seed, fs)
    pw, pri = int(round(fs * PW_S)), int(round(fs * PRI_S))
    idx = np.arange(start, start + n, dtype=np.int64)
    k = idx % pri
    on = k < pw
    phase = 2.0 * np.pi * f0 * (k + 1) / fs
    rng = np.random.default_rng([seed, start])
    x = AMPLITUDE * on * np.exp(1j * phase) + NOISE_SIGMA * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    full = float(2 ** (bit_width - 1))
    q = np.stack([np.rint(x.real * full), np.rint(x.imag * full)], axis=1)
    return np.clip(q, -full, full - 1).astype(dtype)


def pulsed_iq_torch(n: int, bit_width: int = 12, dtype=None, seed: int = SEED, fs: float = FS, device="cuda",
                    chunk: int = 1 << 25):
    """(n, 2) integer I/Q tensor generated on ``device``."""
    import torch
    dtype = dtype or torch.int16
    f0 = _carrier(seed, fs)
    pw, pri = int(round(fs * PW_S)), int(round(fs * PRI_S))
    out = torch.empty((n, 2), dtype=dtype, device=device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    full = float(2 ** (bit_width - 1))
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        k = torch.arange(s, s + m, dtype=torch.int64, device=device) % pri
        on = (k < pw).to(torch.float32) * AMPLITUDE
        ph = (2.0 * np.pi * f0 / fs) * (k + 1).to(torch.float64)
        ph = torch.remainder(ph, 2.0 * np.pi).to(torch.float32)
        noise = torch.randn((m, 2), generator=gen, device=device, dtype=torch.float32) * NOISE_SIGMA
        x = torch.stack([on * torch.cos(ph), on * torch.sin(ph)], dim=1) + noise
        out[s:s + m] = torch.clamp(torch.round(x * full), -full, full - 1).to(dtype)
        del k, on, ph, noise, x
    return out
