"""CPU tests of the oracle itself (no GPU): the two independent C formulations and the
numpy twin must agree, analytic known answers must hold, and the committed golden
fixtures must reproduce.  The reference has no tests of its own to mirror
(SURVEY.md section 4); the KATs below are the ones section 8c prescribes."""
import os

import numpy as np
import pytest

from oracle.pfb_oracle import OracleConfig, channelize_numpy, unpack_numpy

SHAPES = [(8, 12, 8), (64, 12, 64), (128, 12, 64), (16, 3, 4), (6, 5, 3), (7, 4, 7), (56, 12, 56)]


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("M,P,D", SHAPES)
def test_formulations_agree(oracle, M, P, D):
    rng = np.random.default_rng(M * 1000 + P)
    x = rng.standard_normal(D * 24) + 1j * rng.standard_normal(D * 24)
    h = rng.standard_normal(M * P)
    cfg = OracleConfig(M, P, D)
    a = oracle.channelize(x, h, cfg, "direct")
    b = oracle.channelize(x, h, cfg, "polyphase")
    c = channelize_numpy(x, h, cfg)
    assert rel(b, a) < 1e-12 and rel(c, a) < 1e-12
    if M & (M - 1) == 0:
        assert rel(oracle.channelize(x, h, cfg, "fft"), a) < 1e-12


@pytest.mark.parametrize("kw", [dict(off=0), dict(off=2), dict(conj_input=True), dict(derotate=True),
                                dict(fftshift=True), dict(fftshift=True, derotate=True, conj_input=True)])
@pytest.mark.parametrize("M,P,D", [(16, 3, 4), (8, 12, 8), (7, 4, 7), (128, 4, 64)])
def test_switches_agree(oracle, M, P, D, kw):
    rng = np.random.default_rng(7)
    x = rng.standard_normal(D * 20) + 1j * rng.standard_normal(D * 20)
    h = rng.standard_normal(M * P)
    cfg = OracleConfig(M, P, D, **kw)
    a = oracle.channelize(x, h, cfg, "direct")
    assert rel(oracle.channelize(x, h, cfg, "polyphase"), a) < 1e-12
    assert rel(channelize_numpy(x, h, cfg), a) < 1e-12


def test_m1024_fft_matches_plain_dft(oracle):
    rng = np.random.default_rng(3)
    M, P = 1024, 16
    x = rng.standard_normal(M * 20) + 1j * rng.standard_normal(M * 20)
    h = oracle.design_prototype(M, P)
    cfg = OracleConfig(M, P, M)
    assert rel(oracle.channelize(x, h, cfg, "fft"), oracle.channelize(x, h, cfg, "polyphase")) < 1e-12


# ---- analytic known answers -------------------------------------

@pytest.mark.parametrize("M,P,D,q0", [(8, 12, 8, 0), (8, 12, 8, 5), (64, 12, 64, 11), (128, 12, 64, 3), (16, 3, 4, 2)])
def test_one_hot_tap_is_a_pure_delay_bit_exact(oracle, M, P, D, q0):
    """h = delta[n - M q0]  =>  y_k[m] = x[mD + off - M q0] for every k, exactly."""
    rng = np.random.default_rng(11)
    iq = rng.integers(-2048, 2048, size=(D * 40, 2)).astype(np.int16)
    x = oracle.unpack(iq, 12)
    h = np.zeros(M * P)
    h[M * q0] = 1.0
    cfg = OracleConfig(M, P, D)
    for method in ("direct", "polyphase"):
        y = oracle.channelize(x, h, cfg, method)
        for m in range(y.shape[0]):
            s = m * D + (D - 1) - M * q0
            want = x[s] if s >= 0 else 0.0
            assert np.all(y[m] == want), (method, m)


def test_dc_input_steady_state(oracle):
    M, P = 16, 6
    h = oracle.design_prototype(M, P)
    y = oracle.channelize(np.ones(M * 30, dtype=complex), h, OracleConfig(M, P, M))
    steady = y[P:]
    assert np.allclose(steady[:, 0], h.sum(), atol=1e-13)
    # other bins see the prototype's response at k*fs/M: deep in the stop band
    assert np.abs(steady[:, 1:]).max() < 1e-3


@pytest.mark.parametrize("k0", [1, 5, 13])
def test_on_bin_tone(oracle, k0):
    """x = e^{j2pi k0 n/M} => y_{k0}[m] = (sum h) e^{j2pi k0 (mD+off)/M} in steady state."""
    M, P, D = 16, 6, 16
    h = oracle.design_prototype(M, P)
    n = np.arange(M * 30)
    y = oracle.channelize(np.exp(2j * np.pi * k0 * n / M), h, OracleConfig(M, P, D))
    m = np.arange(P, y.shape[0])
    want = h.sum() * np.exp(2j * np.pi * k0 * (m * D + D - 1) / M)
    assert np.allclose(y[P:, k0], want, atol=1e-12)


def test_impulse_gives_dft_of_polyphase_columns(oracle):
    """x = delta[n - s0]: frame m row = sum_p e^{j2pi kp/M} h[p + Mq] over the single (p,q) hit set."""
    M, P, D = 8, 4, 8
    rng = np.random.default_rng(5)
    h = rng.standard_normal(M * P)
    x = np.zeros(M * 12, dtype=complex)
    s0 = 19
    x[s0] = 1.0
    y = oracle.channelize(x, h, OracleConfig(M, P, D))
    for m in range(y.shape[0]):
        n = m * D + (D - 1) - s0  # tap index that meets the impulse
        want = h[n] * np.exp(2j * np.pi * np.arange(M) * n / M) if 0 <= n < M * P else np.zeros(M)
        assert np.allclose(y[m], want, atol=1e-13)


def test_linearity_and_frame_shift(oracle):
    M, P, D = 32, 5, 16
    rng = np.random.default_rng(9)
    h = rng.standard_normal(M * P)
    a = rng.standard_normal(D * 40) + 1j * rng.standard_normal(D * 40)
    b = rng.standard_normal(D * 40) + 1j * rng.standard_normal(D * 40)
    cfg = OracleConfig(M, P, D)
    ya, yb = oracle.channelize(a, h, cfg), oracle.channelize(b, h, cfg)
    assert rel(oracle.channelize(2 * a - 3j * b, h, cfg), 2 * ya - 3j * yb) < 1e-12
    shifted = np.concatenate([np.zeros(D * 3, dtype=complex), a])[: a.size]
    assert rel(oracle.channelize(shifted, h, cfg)[3:], ya[:-3]) < 1e-12


def test_unpack_matches_reference_rule(oracle):
    iq = np.array([[-2048, 2047], [1, -1], [0, 1024]], dtype=np.int16)
    assert np.array_equal(oracle.unpack(iq, 12), unpack_numpy(iq, 12))
    assert oracle.unpack(iq, 12)[0] == complex(-1.0, 2047 / 2048)
    i8 = np.array([[-128, 127]], dtype=np.int8)
    assert oracle.unpack(i8, 8)[0] == complex(-1.0, 127 / 128)


def test_center_frequencies(oracle):
    assert np.array_equal(oracle.center_frequencies(8, 8e6), np.array([0, 1, 2, 3, -4, -3, -2, -1]) * 1e6)
    assert np.array_equal(oracle.center_frequencies(5, 5.0), [0, 1, 2, -2, -1])
    f = oracle.center_frequencies(56, 56e6)
    assert f[0] == 0 and f[27] == 27e6 and f[28] == -28e6 and f[55] == -1e6


def test_prototype_design(oracle):
    h = oracle.design_prototype(8, 12)
    assert h.size == 96 and h.argmax() == 48 and abs(h.sum() - 1.0) < 1e-3


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "ref56", "ref560"])
def test_golden_fixtures_reproduce(oracle, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    M, P, D, bw = int(g["M"]), int(g["P"]), int(g["D"]), int(g["bit_width"])
    iq = g["iq"]
    x = (iq[:, 0].astype(np.float64) + 1j * iq[:, 1]) if str(g["fmt"]) == "cf32" else oracle.unpack(iq, bw)
    method = "fft" if (M >= 256 and (M & (M - 1)) == 0) else "polyphase"
    y = oracle.channelize(x, g["taps"].astype(np.float64), OracleConfig(M, P, D), method)
    assert y.shape == g["expected"].shape
    assert rel(y, g["expected"]) < 1e-12


def test_fp32_cpu_port_matches_oracle(oracle):
    """the cpu_baseline leg of bench.py times this port; make sure it computes the same thing"""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cfg2.npz"))
    y = oracle.channelize_f32_i16(g["iq"], 12, g["taps"], 64, 12, threads=2)
    assert rel(y, g["expected"]) < 2e-6


def test_pdw_extraction_on_a_synthetic_pulse(oracle):
    """One rectangular pulse in one bin: restated create_pdws_channelized.m:64-143 finds it."""
    F, M = 400, 8
    rng = np.random.default_rng(2)
    y = 0.01 * (rng.standard_normal((F, M)) + 1j * rng.standard_normal((F, M)))
    fs_in, fc, t0 = 8e6, 1e9, 1000.0
    fs = fs_in / M
    dphi = 20.0  # degrees per frame
    y[100:151, 5] += 0.5 * np.exp(1j * np.deg2rad(dphi) * np.arange(51))
    pdws = oracle.extract_pdws(y, fs_in, fc, t0, 15.0, matlab_quirks=False)
    hits = [p for p in pdws if p["bin"] == 5]
    assert len(hits) == 1
    p = hits[0]
    assert abs(p["toa"] - (t0 + 101 / fs)) < 1e-9            # 1-based index / fs + start (:98)
    assert abs(p["pw"] - 51 / fs) <= 1 / fs + 1e-12            # (jj - toa)/fs (:110)
    unshifted = oracle.center_frequencies(M, fs_in)
    want_f = fc + unshifted[(5 + M // 2) % M] + fs / (360.0 / dphi)
    assert abs(p["freq"] - want_f) < 0.05 * fs
    assert not p["sat"]
    # the reference's own indexing choices change freq, nothing else
    q = [p for p in oracle.extract_pdws(y, fs_in, fc, t0, 15.0, matlab_quirks=True) if p["bin"] == 5][0]
    assert q["toa"] == p["toa"] and q["pw"] == p["pw"] and q["snr"] == p["snr"]
    # binFreqs as an FFT-ordered list indexed by the shifted column (:42 / :80 if centerFrequencies is unshifted):
    # exactly half the band away from the column's true centre, everything else equal
    u = [p for p in oracle.extract_pdws(y, fs_in, fc, t0, 15.0, matlab_quirks=False, binfreq_unshifted=True) if p["bin"] == 5][0]
    assert abs((u["freq"] - p["freq"]) - (unshifted[5] - unshifted[(5 + M // 2) % M])) < 1e-6
    assert u["toa"] == p["toa"] and u["pw"] == p["pw"] and u["snr"] == p["snr"]


def test_raw_pdw_extraction_known_answer(oracle):
    """create_pdws.m:44-105 restated: one rectangular pulse on a noise floor, with a dip that stays above the 3 dB
    trailing threshold (hysteresis) and a second pulse; fields follow the script's formulas."""
    n, fs, fc, t0 = 4000, 1e6, 2e9, 50.0
    rng = np.random.default_rng(5)
    x = 0.001 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    k = np.arange(300)
    x[1000:1300] = 0.5 * np.exp(1j * np.deg2rad(30.0) * k) * np.where((k >= 100) & (k < 110), 0.06, 1.0)
    x[2000:2050] = 0.5 * np.exp(-1j * np.deg2rad(45.0) * np.arange(50))
    pdws, nf = oracle.extract_pdws_raw(x, fs, fc, t0)
    mag = np.abs(x)
    assert nf == pytest.approx(np.median(mag), rel=1e-14)   # libm hypot vs numpy's: last-bit differences
    assert len(pdws) == 2
    p, q = pdws
    assert p["toa"] == pytest.approx(1001 / fs + t0, abs=1e-9)      # 1-based index of the first sample (:67)
    assert p["pw"] == pytest.approx(300 / fs, abs=1.5 / fs)          # the dip (0.03 > trail = 2 nf = 0.0024) does not end it
    assert p["freq"] == pytest.approx(fc + fs * 30.0 / 360.0, rel=1e-9)
    assert q["freq"] == pytest.approx(fc - fs * 45.0 / 360.0, rel=1e-9)
    assert p["mag"] == pytest.approx(0.5, rel=0.05) and p["snr"] == pytest.approx(10 * np.log10(p["mag"] / nf))
    assert not p["sat"] and p["bin"] == 0
    # with the trailing threshold raised to the leading one the dip splits the first pulse
    split, _ = oracle.extract_pdws_raw(x, fs, fc, t0, snr_db=18.0, trail_db=18.0)
    assert len(split) == 3


@pytest.mark.parametrize("M,P,D", [(8, 12, 8), (64, 12, 64), (128, 12, 64), (56, 12, 56), (12, 5, 4)])
def test_against_scipy_polyphase_decimator(oracle, M, P, D):
    """An implementation nobody here wrote: scipy.signal.upfirdn (polyphase FIR + decimate, SciPy's C code) applied per
    channel to the band-pass filter h[n] e^{+j 2 pi k n / M}.  With input_offset = D-1 frame m of channel k is sample
    m*D + D-1 of the full-rate convolution, i.e. upfirdn(..., down=D) after dropping the first D-1 outputs.  This does
    not pin dsp.Channelizer (nothing available can), but it does pin the FIR / decimation indexing of the oracle to a
    third-party library."""
    from scipy.signal import upfirdn
    rng = np.random.default_rng(M + P + D)
    n = D * 50
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    h = rng.standard_normal(M * P)
    got = oracle.channelize(x, h, OracleConfig(M, P, D), "polyphase")
    want = np.empty_like(got)
    nn = np.arange(M * P)
    for k in range(M):
        hk = h * np.exp(2j * np.pi * k * nn / M)
        full = upfirdn(hk, x, up=1, down=1)            # full-rate convolution, full[i] = sum_n hk[n] x[i-n]
        want[:, k] = full[D - 1:n:D]
    assert got.shape == want.shape == (n // D, M)
    assert rel(got, want) < 1e-12
