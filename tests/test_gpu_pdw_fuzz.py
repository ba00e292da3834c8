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
