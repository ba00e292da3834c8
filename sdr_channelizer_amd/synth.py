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


BARKER_13 = (+1, +1, +1, +1, +1, -1, -1, +1, +1, -1, +1, -1, +1)  # generate_pulsed_iq.m:49-56 (+-90 degrees)


def pulse_phase(k: np.ndarray, f0: float, fs: float, pw: int, lfm_extent_hz: float = 0.0, barker13: bool = False):
    """Phase (radians) of sample k (0-based) inside a pulse of pw samples, as generate_pulsed_iq.m builds
    it: f_lfm = linspace(f_start, f_stop, pw); phi = cumsum(2 pi f_lfm / Fs) (:43-47), optionally plus the
    Barker-13 +-90 degree chips (:49-59; the chip length is round(pw/13) and pw is adjusted to 13 chips)."""
    kk = np.clip(k, 0, max(pw - 1, 0)).astype(np.float64)
    if pw > 1:
        # cumsum of a linear ramp: sum_{i<=k} (f0 + i*df) with df = extent/(pw-1)
        df = lfm_extent_hz / (pw - 1)
        phase = 2.0 * np.pi / fs * (f0 * (kk + 1) + df * kk * (kk + 1) / 2.0)
    else:
        phase = 2.0 * np.pi * f0 * (kk + 1) / fs
    if barker13:
        chip = max(int(round(pw / 13)), 1)
        code = np.asarray(BARKER_13, dtype=np.float64)[np.clip((kk // chip).astype(np.int64), 0, 12)]
        phase = phase + np.deg2rad(90.0) * code
    return phase


def pulsed_iq_numpy(n: int, bit_width: int = 12, dtype=np.int16, seed: int = SEED, fs: float = FS,
                    start: int = 0, lfm_extent_hz: float = 0.0, barker13: bool = False,
                    pw_s: float = PW_S, pri_s: float = PRI_S) -> np.ndarray:
    """(n, 2) integer I/Q; sample index start..start+n-1 of the infinite stream.
    lfm_extent_hz / barker13 are the LFM_EXTENT and BARKER_13 switches of generate_pulsed_iq.m:17-19."""
    f0 = _carrier(seed, fs)
    pw, pri = int(round(fs * pw_s)), int(round(fs * pri_s))
    if barker13:
        pw = max(int(round(pw / 13)), 1) * 13  # "Pulse width adjusted", generate_pulsed_iq.m:33-40
    idx = np.arange(start, start + n, dtype=np.int64)
    k = idx % pri
    on = k < pw
    phase = pulse_phase(k, f0, fs, pw, lfm_extent_hz, barker13)
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
