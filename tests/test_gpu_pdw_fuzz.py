"""Randomised PDW extraction against the oracle: random shapes, pulse trains, amplitude grids (ties), thresholds and
quirk flags, channelized (create_pdws_channelized.m) and raw (create_pdws.m).  Seeded."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from sdr_channelizer_amd.pdw import extract_pdws, extract_pdws_raw  # noqa: E402
from test_gpu_pdw import compare  # noqa: E402

CASES = int(os.environ.get("PFB_FUZZ_CASES", "40"))


def pulses_into(rng, x, amp_lo, amp_hi):
    n = x.shape[0]
    for _ in range(int(rng.integers(0, 12))):
        a = int(rng.integers(0, n))
        m = int(rng.integers(1, max(2, n // 8)))
        amp = rng.uniform(amp_lo, amp_hi)
        dphi = rng.uniform(-170.0, 170.0)
        seg = x[a:a + m]
        seg += amp * np.exp(1j * np.deg2rad(dphi) * np.arange(seg.shape[0]))


@pytest.mark.parametrize("case", range(CASES))
def test_channelized_pdws_random(oracle, case):
    rng = np.random.default_rng(7000 + case)
    F, M = int(rng.integers(2, 3000)), int(rng.integers(1, 70))
    y = 0.01 * (rng.standard_normal((F, M)) + 1j * rng.standard_normal((F, M)))
    for b in rng.integers(0, M, size=int(rng.integers(0, 5))):
        col = y[:, b].copy()
        pulses_into(rng, col, 0.2, 1.5)
        y[:, b] = col
    if rng.random() < 0.3:  # coarse amplitude grid: heavily tied magnitudes
        g = float(rng.choice([64.0, 256.0, 1024.0]))
        y = np.round(y.real * g) / g + 1j * np.round(y.imag * g) / g
    y = y.astype(np.complex64)
    snr = float(rng.choice([6.0, 10.0, 15.0, 20.0]))
    quirks, unshifted = bool(rng.integers(2)), bool(rng.integers(2))
    fs_in, fc, t0 = 56e6, 915e6, float(rng.uniform(0, 2e9))
    got, nf = extract_pdws(y, fs_in, fc, t0, snr_threshold_db=snr, matlab_quirks=quirks, binfreq_unshifted=unshifted,
                           return_noise_floor=True)
    want = oracle.extract_pdws(y.astype(np.complex128), fs_in, fc, t0, snr, matlab_quirks=quirks, max_out=1 << 18,
                               binfreq_unshifted=unshifted)
    assert np.allclose(nf, np.median(np.abs(y.astype(np.complex128)), axis=0), rtol=1e-12, atol=0)

    def phase_col(i):  # the samples pulse i spans, located with t0 = 0 (toa + t0 has no sample resolution left)
        p = oracle.extract_pdws(y.astype(np.complex128), fs_in, fc, 0.0, snr, matlab_quirks=quirks, max_out=1 << 18)[i]
        a = int(round(p["toa"] * fs_in / M)) - 1
        return y[a:a + int(round(p["pw"] * fs_in / M)) + 1, 0 if quirks else p["bin"]]

    compare(got, want, fs_in / M, phase_col=phase_col)


@pytest.mark.parametrize("case", range(CASES))
def test_raw_pdws_random(oracle, case):
    rng = np.random.default_rng(9000 + case)
    n = int(rng.integers(2, 200000))
    x = rng.uniform(0.001, 0.02) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    pulses_into(rng, x, 0.1, 1.4)
    kind = int(rng.integers(3))
    if kind == 2:
        iq = x.astype(np.complex64)
        xs, bw = iq.astype(np.complex128), 0
    else:
        dt, bw = (np.int8, 8) if kind == 0 else (np.int16, int(rng.choice([12, 16])))
        full = 2 ** (bw - 1)
        iq = np.stack([np.clip(np.round(x.real * full), -full, full - 1), np.clip(np.round(x.imag * full), -full, full - 1)],
                      axis=1).astype(dt)
        xs = (iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)) / full
    lead = float(rng.choice([6.0, 12.0, 18.0]))
    trail = float(rng.choice([0.0, 3.0, lead]))
    fs, fc, t0 = 56e6, 2.4e9, float(rng.uniform(0, 2e9))
    got, nf = extract_pdws_raw(iq, fs, fc, t0, bit_width=max(bw, 1), snr_threshold_db=lead, trailing_threshold_db=trail,
                               return_noise_floor=True)
    want, want_nf = oracle.extract_pdws_raw(xs, fs, fc, t0, snr_db=lead, trail_db=trail, max_out=1 << 18)
    assert nf == pytest.approx(want_nf, rel=1e-14, abs=0)

    def phase_col(i):
        p = oracle.extract_pdws_raw(xs, fs, fc, 0.0, snr_db=lead, trail_db=trail, max_out=1 << 18)[0][i]
        a = int(round(p["toa"] * fs)) - 1
        return xs[a:a + int(round(p["pw"] * fs)) + 1]

    compare(got, want, fs, phase_col=phase_col)


LONG_CASES = max(6, CASES // 5)


@pytest.mark.parametrize("case", range(LONG_CASES))
def test_channelized_pdws_random_long_streams(oracle, case):
    """F >= 8 * 65536 frames: the sampled-bracket route (float32 screen, parked candidates, bracket finish) on random
    shapes, amplitude scales from 1e-18 to 1e15 (the screen hands very small and very large magnitudes to the exact
    route), coarse grids (ties), non-stationary noise and low thresholds (many samples near the threshold).  Medians exact,
    PDWs the oracle's; whichever route the data force (1 / 4 sampled, 3 full select after a failed proof)."""
    from sdr_channelizer_amd import _lib as L
    rng = np.random.default_rng(11000 + case)
    F = int(rng.integers(8 * 65536, 8 * 65536 + 300000))
    M = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 11, 12, 16, 17, 24, 32, 33]))   # every lane packing of the bracket pass: 8 / 16 / 32 / 64 lanes per row
    scale = float(10.0 ** rng.choice([-18.0, -6.0, -2.0, -2.0, -2.0, 0.0, 3.0, 15.0]))
    y = rng.standard_normal((F, M), dtype=np.float32) + 1j * rng.standard_normal((F, M), dtype=np.float32)
    if rng.random() < 0.4:  # the noise level drifts along the record
        y *= np.linspace(1.0, float(rng.uniform(1.0, 3.0)), F, dtype=np.float32)[:, None]
    if rng.random() < 0.3:  # moderately tied magnitudes
        g = float(rng.choice([50.0, 400.0, 3000.0]))
        y = np.round(y.real * g) / g + 1j * np.round(y.imag * g) / g
    y = (y * np.float32(0.01)).astype(np.complex64)
    for b in rng.integers(0, M, size=int(rng.integers(1, 4))):
        for _ in range(int(rng.integers(1, 5))):
            a, n = int(rng.integers(0, F - 5000)), int(rng.integers(1, 4000))
            y[a:a + n, b] += (rng.uniform(0.2, 1.0) * np.exp(1j * np.deg2rad(rng.uniform(-170, 170)) * np.arange(n))).astype(np.complex64)
    y = (y.astype(np.complex128) * scale).astype(np.complex64)
    snr = float(rng.choice([3.0, 6.0, 10.0, 15.0]))
    got, nf = extract_pdws(y, 8e6, 1e9, 0.0, snr_threshold_db=snr, return_noise_floor=True, capacity=1 << 22)
    assert L.load().pfb_pdw_last_noise_floor_path() in (1, 3, 4)
    yd = y.astype(np.complex128)
    assert np.allclose(nf, np.median(np.abs(yd), axis=0), rtol=1e-12, atol=0)
    want = oracle.extract_pdws(yd, 8e6, 1e9, 0.0, snr, max_out=1 << 22)

    def phase_col(i):  # the samples pulse i spans (t0 = 0 here); the scripts' quirk reads the phases of column 1
        a = int(round(want[i]["toa"] * 8e6 / M)) - 1
        return y[a:a + int(round(want[i]["pw"] * 8e6 / M)) + 1, 0]

    compare(got, want, 8e6 / M, phase_col=phase_col)
    L.load().pfb_pdw_release_workspace(-1)
