"""Synthetic pulsed I/Q in the recorders' integer formats.

Recipe restated from /root/reference/matlab/generate_pulsed_iq.m:12-19,63-77,83 and
/root/reference/matlab/generate_training_iq.m:12-26,95-98:
unit rectangular pulses (PW 100 us, PRI 1 ms at fs 56 Msps) on a complex carrier
f0 drawn uniformly from (-fs/2, fs/2), the carrier phase restarting at every pulse
(``phi(idx:idx+pw-1) = my_phi``), amplitude 0.5 full scale, plus complex Gaussian
noise of sigma 2^-6 full scale (the MATLAB generators are noise-free; noise makes the
downstream median/threshold logic meaningful), quantised round-to-nearest and saturated
to ``bit_width`` bits inside int8/int16 (``int16(x*2^15)`` saturates, generate_training_iq.m:95-98).

``pulsed_iq_numpy`` is the full-featured float64 generator (LFM sweep, Barker-13 chips) for tests and
fixtures.  ``pulsed_iq_counter_numpy`` / ``pulsed_iq_torch`` are the benchmark stream:
a COUNTER-BASED generator -- sample index -> value, integer arithmetic only -- so the CPU and the GPU produce
identical integers and any shard of the stream can be generated on its own (``start``).
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


def pulse_phase(k: np.ndarray, f0: float, fs: float, pw: int, lfm_extent_hz: float = 0.0, barker13: bool = False,
                matlab_quirks: bool = True):
    """Phase (radians) of sample k (0-based) inside a pulse of pw samples, as generate_pulsed_iq.m builds
    it: f_lfm = linspace(f_start, f_stop, pw); phi = cumsum(2 pi f_lfm / Fs) (:43-47), optionally plus the
    Barker-13 chips (:49-59; the chip length is round(pw/13) and pw is adjusted to 13 chips).

    matlab_quirks (default = reference behaviour): the script adds ``+90`` / ``-90`` to ``phi``, which is in
    RADIANS (it goes straight into exp(1j*phi), :63), so the two chip states differ by 180 rad = 233.5 degrees
    (mod 360), not by the 180 degrees of a Barker code.  matlab_quirks=False adds +-90 DEGREES instead."""
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
        phase = phase + (90.0 if matlab_quirks else np.deg2rad(90.0)) * code
    return phase


def pulsed_iq_numpy(n: int, bit_width: int = 12, dtype=np.int16, seed: int = SEED, fs: float = FS,
                    start: int = 0, lfm_extent_hz: float = 0.0, barker13: bool = False,
                    pw_s: float = PW_S, pri_s: float = PRI_S, matlab_quirks: bool = True) -> np.ndarray:
    """(n, 2) integer I/Q; sample index start..start+n-1 of the infinite stream.
    lfm_extent_hz / barker13 are the LFM_EXTENT and BARKER_13 switches of generate_pulsed_iq.m:17-19;
    matlab_quirks: see pulse_phase."""
    f0 = _carrier(seed, fs)
    pw, pri = int(round(fs * pw_s)), int(round(fs * pri_s))
    if barker13:
        pw = max(int(round(pw / 13)), 1) * 13  # "Pulse width adjusted", generate_pulsed_iq.m:33-40
    idx = np.arange(start, start + n, dtype=np.int64)
    k = idx % pri
    on = k < pw
    phase = pulse_phase(k, f0, fs, pw, lfm_extent_hz, barker13, matlab_quirks)
    rng = np.random.default_rng([seed, start])
    x = AMPLITUDE * on * np.exp(1j * phase) + NOISE_SIGMA * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    full = float(2 ** (bit_width - 1))
    q = np.stack([np.rint(x.real * full), np.rint(x.imag * full)], axis=1)
    return np.clip(q, -full, full - 1).astype(dtype)


# ---------------------------------------------------------------------------------------------------
# Counter-based stream: integer arithmetic only, so numpy on the host and torch on the GPU agree bit for bit.
#   carrier  a 32-bit phase accumulator restarting at every pulse (phi = 2 pi f0 (k+1) / fs, k = index inside the
#            period, like ``my_phi`` above), looked up in a 2^16-entry cos table held as fixed-point integers
#            (built once on the host in float64; the GPU gets the same integers)
#   noise    per component the sum of four 16-bit uniforms (Irwin-Hall, sigma = 37837.2 counts) from a 32-bit
#            mixing hash of (seed, sample index, lane), scaled to sigma = 2^-6 full scale
#   output   (carrier + noise) in 2^-20 LSB fixed point, rounded half up, saturated to bit_width bits

_FX = 20                    # fractional bits of the fixed-point sum
_TAB_BITS = 16
_IH_SIGMA = float(np.sqrt(4 * (65536.0 ** 2 - 1) / 12.0))


def _cos_table(bit_width: int) -> np.ndarray:
    j = np.arange(1 << _TAB_BITS, dtype=np.float64)
    full = float(2 ** (bit_width - 1))
    return np.rint(AMPLITUDE * full * (1 << _FX) * np.cos(2.0 * np.pi * j / (1 << _TAB_BITS))).astype(np.int64)


def _mix32(x):
    """32-bit avalanche hash on values held in int64 (products stay below 2^63): same code for numpy and torch."""
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    return x ^ (x >> 16)


def _counter_stream(idx, tab, seed: int, fs: float, bit_width: int, clamp):
    """idx: int64 sample indices (numpy array or torch tensor); tab: the cos table in the same array library;
    returns (I, Q) int64 after rounding, before the dtype cast."""
    f0 = _carrier(seed, fs)
    pw, pri = int(round(fs * PW_S)), int(round(fs * PRI_S))
    fcw = int(round(f0 / fs * 4294967296.0)) % 4294967296  # cycles per sample in 2^-32 turns (negative f0 wraps)
    full = 1 << (bit_width - 1)
    gain = int(round(NOISE_SIGMA * full * (1 << _FX) / _IH_SIGMA))  # noise counts -> 2^-20 LSB
    k = idx % pri
    on = k < pw
    ph = (((k + 1) & 0xFFFFFFFF) * fcw) & 0xFFFFFFFF  # (k+1) < 2^17 and fcw < 2^32: the product fits int64
    ci = ph >> (32 - _TAB_BITS)
    si = (ci - (1 << (_TAB_BITS - 2))) & ((1 << _TAB_BITS) - 1)  # sin(x) = cos(x - pi/2)
    lo, hi = idx & 0xFFFFFFFF, (idx >> 32) & 0xFFFFFFFF
    base = lo ^ ((hi * 0x9E3779B1) & 0xFFFFFFFF) ^ (seed & 0xFFFFFFFF)
    out = []
    for lane, ti in ((0, ci), (1, si)):
        a = _mix32(base ^ (0x68E31DA4 * (2 * lane + 1) & 0xFFFFFFFF))
        b = _mix32(a ^ 0x5BD1E995)
        noise = (a & 0xFFFF) + (a >> 16) + (b & 0xFFFF) + (b >> 16) - 2 * 65535
        v = tab[ti] * on + noise * gain
        out.append(clamp((v + (1 << (_FX - 1))) >> _FX, -full, full - 1))  # arithmetic shift = floor: round half up
    return out


def pulsed_iq_counter_numpy(n: int, bit_width: int = 12, dtype=np.int16, seed: int = SEED, fs: float = FS,
                            start: int = 0) -> np.ndarray:
    """(n, 2) integer I/Q: samples start..start+n-1 of the counter-based benchmark stream (host twin of
    ``pulsed_iq_torch``: identical integers)."""
    idx = np.arange(start, start + n, dtype=np.int64)
    i, q = _counter_stream(idx, _cos_table(bit_width), seed, fs, bit_width, np.clip)
    return np.stack([i, q], axis=1).astype(dtype)


def pulsed_iq_torch(n: int, bit_width: int = 12, dtype=None, seed: int = SEED, fs: float = FS, device="cuda",
                    chunk: int = 1 << 25, start: int = 0):
    """(n, 2) integer I/Q tensor generated on ``device``: samples start..start+n-1 of the counter-based stream, so
    rank r of a time-sharded run generates its own segment with start = r * n (chunked: temporaries stay small)."""
    import torch
    dtype = dtype or torch.int16
    tab = torch.from_numpy(_cos_table(bit_width)).to(device)
    out = torch.empty((n, 2), dtype=dtype, device=device)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        idx = torch.arange(start + s, start + s + m, dtype=torch.int64, device=device)
        i, q = _counter_stream(idx, tab, seed, fs, bit_width, torch.clamp)
        out[s:s + m, 0] = i.to(dtype)
        out[s:s + m, 1] = q.to(dtype)
        del idx, i, q
    return out
