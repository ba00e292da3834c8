"""Loader + alignment search for a MATLAB export of the channelizer (tools/export_matlab_golden.m).

The reference's arithmetic is MathWorks' closed dsp.Channelizer (matlab/channelizer_example.m:31,56): nothing in this
image can run it, so parity is UNPINNED until somebody runs the export script in MATLAB and drops
tests/golden/matlab_<cfg>.mat next to the committed inputs.  What is not known in advance -- which sample of a block
is "newest" (input_offset), whether oversampled outputs are derotated, whether the example's conjugation is inside,
output scale, whether the designed prototype carries an extra end tap, a frame of latency -- is searched here, so the
day the file exists the pin is `pytest tests/test_matlab_pin.py` and a configuration, not a code change."""
from __future__ import annotations

import itertools
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CONFIGS = ("cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "ref56", "ref560")


def export_path(cfg: str, folder: str = GOLDEN) -> str:
    return os.path.join(folder, f"matlab_{cfg}.mat")


def load_export(path: str) -> dict:
    """scipy.io.loadmat of the -v7 file the export script saves (a data loader: nothing in the file is executed)."""
    import scipy.io
    m = scipy.io.loadmat(path, squeeze_me=True, struct_as_record=False)
    out = np.asarray(m["out"], dtype=np.complex128)
    if out.ndim == 1:
        out = out[None, :]
    d = dict(out=out, taps=np.asarray(m["taps"], dtype=np.float64).reshape(-1), M=int(m["M"]), P=int(m["P"]), D=int(m["D"]),
             fs=float(m["fs"]))
    if "center_frequencies" in m:
        d["center_frequencies"] = np.asarray(m["center_frequencies"], dtype=np.float64).reshape(-1)
    d["matlab_version"] = str(m.get("matlab_version", ""))
    return d


def tap_candidates(taps: np.ndarray, M: int):
    """Arrangements of the exported prototype as an M*P-tap filter: as is, an extra end tap dropped at either side,
    or zero-padded to whole branches."""
    L = taps.size
    P = -(-L // M)
    cands = []
    if L % M == 0:
        cands.append(("as exported", taps))
    if L % M == 1 and L > M:
        cands.append(("last tap dropped", taps[:-1]))
        cands.append(("first tap dropped", taps[1:]))
    if L % M:
        pad = P * M - L
        cands.append((f"{pad} zeros appended", np.concatenate([taps, np.zeros(pad)])))
        cands.append((f"{pad} zeros prepended", np.concatenate([np.zeros(pad), taps])))
    return cands


def find_alignment(oracle, x: np.ndarray, export: dict, max_lag: int = 2):
    """Search the alignment switches for the combination whose float64 oracle output is closest to MATLAB's.
    x: the normalised complex input MATLAB was given.  Returns the best candidate as a dict (rel_err first)."""
    from oracle.pfb_oracle import OracleConfig
    M, D, want = export["M"], export["D"], export["out"]
    best = None
    for tap_name, h in tap_candidates(export["taps"], M):
        P = h.size // M
        offsets = [D - 1, 0] + [o for o in range(1, D - 1)]
        for conj, derot in itertools.product((False, True), (False, True) if D != M else (False,)):
            for off in offsets:
                y = oracle.channelize(x, h, OracleConfig(M, P, D, off=off, conj_input=conj, derotate=derot))
                for lag in range(-max_lag, max_lag + 1):  # MATLAB frame m = our frame m + lag
                    a, b = (want[-lag:], y) if lag < 0 else (want, y[lag:])
                    k = min(a.shape[0], b.shape[0])
                    if k < max(4, want.shape[0] // 2):
                        continue
                    a, b = a[:k], b[:k]
                    for flip in (False, True):  # channel k <-> M - k (the other sign convention of the modulation)
                        bb = b[:, (-np.arange(M)) % M] if flip else b
                        denom = np.vdot(bb, bb).real
                        if denom == 0:
                            continue
                        scale = np.vdot(bb, a) / denom  # least-squares complex gain; a pin needs it real and fixed
                        err = float(np.abs(a - scale * bb).max() / max(np.abs(a).max(), 1e-300))
                        cand = dict(rel_err=err, taps=tap_name, P=P, input_offset=off, conjugate_input=conj, derotate=derot,
                                    frame_lag=lag, channel_flip=flip, scale=complex(scale), h=h)
                        if best is None or err < best["rel_err"]:
                            best = cand
                if best is not None and best["rel_err"] < 1e-9:
                    return best  # exact hit: stop searching
    return best
