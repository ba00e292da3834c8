"""The one-file MATLAB pin.  Skipped until tests/golden/matlab_<cfg>.mat exists
(tools/export_matlab_golden.m writes it from the committed tests/golden/matlab_in_<cfg>.mat); then it pins the float64
ORACLE against dsp.Channelizer on CPU and -- with -m gpu -- the HIP path against the same file, <= 1e-5, under the
alignment the search finds.  The search itself is exercised here without MATLAB, on a stand-in export built by the
oracle with a hidden alignment."""
import os

import numpy as np
import pytest

from oracle.pfb_oracle import OracleConfig, unpack_numpy
from matlab_pin import CONFIGS, GOLDEN, export_path, find_alignment, load_export

REL_TOL = 1e-5  # BASELINE.json north star: <= 1e-5 relative error vs channelizer_example.m


def golden_input(cfg):
    g = np.load(os.path.join(GOLDEN, f"{cfg}.npz"))
    iq, bw = g["iq"], int(g["bit_width"])
    x = (iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)) if iq.dtype.kind == "f" else unpack_numpy(iq, bw)
    M = int(g["M"])
    return g, x[: x.size // M * M]


def test_committed_matlab_inputs_are_the_golden_inputs():
    """matlab_in_<cfg>.mat (what the export script feeds dsp.Channelizer) holds the same samples as <cfg>.npz."""
    import scipy.io
    for cfg in CONFIGS:
        g = np.load(os.path.join(GOLDEN, f"{cfg}.npz"))
        m = scipy.io.loadmat(os.path.join(GOLDEN, f"matlab_in_{cfg}.mat"))
        assert np.array_equal(m["iq"], g["iq"]) and m["iq"].dtype == g["iq"].dtype
        assert (int(m["M"].item()), int(m["P"].item()), int(m["D"].item())) == (int(g["M"]), int(g["P"]), int(g["D"]))


@pytest.mark.parametrize("hidden", [dict(off=1, conj=False, derot=False, extra_tap="last", scale=1.0 / 8),
                                    dict(off=0, conj=True, derot=True, extra_tap=None, scale=1.0),
                                    dict(off=2, conj=False, derot=True, extra_tap="first", scale=4.0)])
def test_alignment_search_recovers_a_hidden_alignment(oracle, tmp_path, hidden):
    """A stand-in for the MATLAB export: the oracle run under an alignment the search is not told, saved the way the
    export script saves, loaded back with the test's loader."""
    import scipy.io
    M, P, D = 8, 6, 4
    rng = np.random.default_rng(11)
    x = rng.standard_normal(M * 60) + 1j * rng.standard_normal(M * 60)
    h = rng.standard_normal(M * P)
    y = hidden["scale"] * oracle.channelize(x, h, OracleConfig(M, P, D, off=hidden["off"], conj_input=hidden["conj"],
                                                               derotate=hidden["derot"]))
    exported = {"last": np.append(h, 0.0), "first": np.append(0.0, h), None: h}[hidden["extra_tap"]]
    path = os.path.join(tmp_path, "matlab_test.mat")
    scipy.io.savemat(path, dict(out=y, taps=exported[None, :], M=float(M), P=float(P), D=float(D), fs=8e6,
                                center_frequencies=np.arange(-4, 4) * 1e6, matlab_version="stand-in"))
    ex = load_export(path)
    best = find_alignment(oracle, x, ex)
    assert best["rel_err"] < 1e-12
    assert (best["input_offset"], best["conjugate_input"], best["derotate"]) == (hidden["off"], hidden["conj"], hidden["derot"])
    assert abs(best["scale"] - hidden["scale"]) < 1e-12 and best["frame_lag"] == 0 and not best["channel_flip"]


@pytest.mark.parametrize("cfg", CONFIGS)
def test_oracle_against_matlab_export(oracle, cfg):
    if not os.path.exists(export_path(cfg)):
        pytest.skip(f"parity unpinned: no tests/golden/matlab_{cfg}.mat (run tools/export_matlab_golden.m in MATLAB)")
    ex = load_export(export_path(cfg))
    g, x = golden_input(cfg)
    best = find_alignment(oracle, x, ex)
    print(f"{cfg}: MATLAB {ex['matlab_version']}: " + ", ".join(f"{k}={v}" for k, v in best.items() if k != "h"))
    assert best["rel_err"] <= 1e-9, best          # float64 against float64
    assert abs(best["scale"].imag) < 1e-9          # a real, fixed gain (1, 1/M, M ...), not a rotation
    if "center_frequencies" in ex:                 # settles pfb_center_frequencies' order question
        f = ex["center_frequencies"]
        fft_order = oracle.center_frequencies(ex["M"], ex["fs"])
        order = "fft" if np.allclose(f, fft_order) else "centered" if np.allclose(f, np.fft.fftshift(fft_order)) else None
        assert order is not None, f
        print(f"{cfg}: centerFrequencies order = {order}")


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", CONFIGS)
def test_hip_path_against_matlab_export(oracle, cfg):
    if not os.path.exists(export_path(cfg)):
        pytest.skip(f"parity unpinned: no tests/golden/matlab_{cfg}.mat (run tools/export_matlab_golden.m in MATLAB)")
    from sdr_channelizer_amd import Channelizer
    ex = load_export(export_path(cfg))
    g, x = golden_input(cfg)
    best = find_alignment(oracle, x, ex)
    M, D, P = ex["M"], ex["D"], best["P"]
    fmt = {"i": {1: "int8", 2: "int16"}.get(g["iq"].dtype.itemsize), "f": "cf32"}[g["iq"].dtype.kind]
    taps = (best["h"] * best["scale"].real).astype(np.float32)  # the fixed output gain folds into the taps (linear)
    with Channelizer(M, taps=taps, decimation=D, sample_format=fmt, bit_width=int(g["bit_width"]) or 16,
                     input_offset=best["input_offset"], conjugate_input=best["conjugate_input"],
                     derotate=best["derotate"]) as ch:
        y = ch(g["iq"])
    want = ex["out"]
    lag = best["frame_lag"]
    a, b = (want[-lag:], y) if lag < 0 else (want, y[lag:])
    k = min(a.shape[0], b.shape[0])
    if best["channel_flip"]:
        b = b[:, (-np.arange(M)) % M]
    assert float(np.abs(a[:k] - b[:k]).max() / np.abs(a[:k]).max()) <= REL_TOL
