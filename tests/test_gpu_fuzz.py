"""Randomised cross-check of every fused shape against the generic kernel (same library, independent code path:
branches from global memory, radix-2 or plain DFT): random flags, input offsets, layouts, run lengths, schedules and
call boundaries.  Seeded, so a failure reproduces."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from sdr_channelizer_amd import Channelizer, synth  # noqa: E402
from sdr_channelizer_amd import _lib as L  # noqa: E402

# (M, P, D, formats, schedules worth forcing besides the default)
SHAPES = [
    (64, 12, 64, ("int16", "int8", "cf32"), (0, 2, 3, 4, 7, 8, 11)),
    (64, 16, 64, ("int16",), (0, 4, 7, 8)),
    (128, 12, 64, ("int16", "cf32"), (0, 2, 3, 7, 8, 11)),
    (256, 8, 256, ("int8", "int16", "cf32"), (0, 2, 8, 11)),
    (1024, 16, 1024, ("int16", "cf32"), (0, 6)),
    (56, 12, 56, ("int16", "int8", "cf32"), (0, 2, 3, 7, 8)),
    (560, 12, 560, ("int16", "int8", "cf32"), (0, 6)),
    (32, 12, 32, ("int16", "int8"), (0, 2, 7, 8)),
    (16, 12, 16, ("int16", "int8"), (0,)),
    (8, 12, 8, ("int16", "int8", "cf32"), (0,)),
    (10, 12, 10, ("int16",), (0,)),
    (20, 12, 20, ("int16",), (0,)),
    (40, 12, 40, ("int16",), (0,)),
    # csrc/pfb_kernels_mixed.hip: SegKernel shapes, single-wave two-pass shapes, multi-wave three-pass shapes (lockstep / teams)
    (12, 12, 12, ("int16",), (0,)), (24, 12, 24, ("int16",), (0,)), (25, 12, 25, ("int16",), (0,)), (30, 12, 30, ("int16",), (0,)),
    (48, 12, 48, ("int16",), (0, 7, 11, 8)), (50, 12, 50, ("int16",), (0, 7, 11)),
    (80, 12, 80, ("int16",), (0, 7, 11, 8)), (96, 12, 96, ("int16",), (0, 7, 11, 8)), (100, 12, 100, ("int16",), (0, 7, 11)),
    (112, 12, 112, ("int16",), (0, 7, 11, 8)), (120, 12, 120, ("int16",), (0, 7, 11)), (160, 12, 160, ("int16",), (0,)),
    (200, 12, 200, ("int16",), (0, 6)), (250, 12, 250, ("int16",), (0, 6)), (280, 12, 280, ("int16",), (0, 6)),
    (320, 12, 320, ("int16",), (0, 6)), (400, 12, 400, ("int16",), (0, 6)), (500, 12, 500, ("int16",), (0, 6)),
    (512, 12, 512, ("int16",), (0, 6)),
]


def make_input(rng, n, fmt):
    if fmt == "cf32":
        return (rng.standard_normal((n, 2)) * 0.3).astype(np.float32), 1
    bw = 8 if fmt == "int8" else int(rng.choice([12, 16]))
    return synth.pulsed_iq_numpy(n, bw, np.int8 if fmt == "int8" else np.int16, seed=int(rng.integers(1 << 30))), bw


@pytest.mark.parametrize("case", range(int(os.environ.get("PFB_FUZZ_CASES", "384"))))  # 12 per shape; more with PFB_FUZZ_CASES=N
def test_fused_kernels_agree_with_the_generic_kernel(case):
    rng = np.random.default_rng(1000 + case)
    M, P, D, fmts, scheds = SHAPES[case % len(SHAPES)]
    fmt = fmts[int(rng.integers(len(fmts)))]
    frames = int(rng.integers(40, 2500))
    n = frames * D + int(rng.integers(0, D))
    iq, bw = make_input(rng, n, fmt)
    p_used = P if rng.random() < 0.7 else int(rng.integers(max(1, P - 5), P + 1))   # sometimes a shorter prototype
    h = (rng.standard_normal(M * p_used) / M).astype(np.float32)
    kw = dict(fftshift=bool(rng.integers(2)), conjugate_input=bool(rng.integers(2)), derotate=bool(rng.integers(2)),
              magnitude=bool(rng.integers(2)), channel_major=bool(rng.integers(2)),
              input_offset=int(rng.integers(-1, D)))
    kw["power"] = kw["magnitude"] and bool(rng.integers(2))   # |y|^2 instead of |y|
    cuts = sorted({0, n, *(int(c) for c in rng.integers(0, n, size=int(rng.integers(0, 3))))})
    with Channelizer(M, taps=h, decimation=D, sample_format=fmt, bit_width=bw, **kw) as ch:
        ch.set_option(L.PFB_OPT_KERNEL, 1)
        want = ch(iq)
        assert ch.last_kernel == "pfb_generic"
        ch.reset()
        ch.set_option(L.PFB_OPT_KERNEL, 2)
        if rng.random() < 0.6:
            ch.set_option(L.PFB_OPT_SCHEDULE, int(scheds[int(rng.integers(len(scheds)))]))
        if rng.random() < 0.5:
            ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, int(rng.choice([8, 24, 40, 64, 100, 256, 1000])))
        if rng.random() < 0.3:
            ch.set_option(L.PFB_OPT_TILE_WAVES, int(rng.choice([4, 6, 8, 16])))
        if rng.random() < 0.3:
            ch.set_option(L.PFB_OPT_XCD_REMAP, int(rng.integers(0, 2)))
        if kw["channel_major"] and rng.random() < 0.3:   # frame-major slabs + transpose instead of the fused stores
            ch.set_option(L.PFB_OPT_SCHEDULE, 9)
            ch.set_option(L.PFB_OPT_SLAB_FRAMES, int(rng.choice([0, 64, 192, 1024])))
        parts = [ch(iq[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
        assert ch.last_kernel.startswith("pfb_fast")
        axis = 1 if kw["channel_major"] else 0
        got = np.concatenate([q for q in parts if q.size], axis=axis) if any(q.size for q in parts) else parts[0]
    assert got.shape == want.shape, (case, kw)
    scale = max(float(np.abs(want).max()), 1e-30)
    assert float(np.abs(got - want).max()) / scale < 3e-6, (case, M, fmt, kw)


@pytest.mark.parametrize("case", range(int(os.environ.get("PFB_ORACLE_FUZZ_CASES", "460"))))  # 10 per shape
def test_fused_kernels_agree_with_the_oracle(oracle, case):
    """The same seeded walk over shapes, formats, switches, schedules, run lengths and call cuts -- against the float64 CPU
    oracle (oracle/pfb_oracle.c) instead of the generic kernel, at lengths the oracle finishes in a moment: the fast and
    the generic kernel share tables and host code, the oracle shares nothing with either."""
    from oracle.pfb_oracle import OracleConfig
    rng = np.random.default_rng(7000 + case)
    M, P, D, fmts, scheds = SHAPES[case % len(SHAPES)]
    fmt = fmts[int(rng.integers(len(fmts)))]
    frames = int(rng.integers(30, 400 if M <= 256 else 120))
    n = frames * D + int(rng.integers(0, D))
    iq, bw = make_input(rng, n, fmt)
    h = (rng.standard_normal(M * P) / M).astype(np.float32)
    kw = dict(fftshift=bool(rng.integers(2)), conjugate_input=bool(rng.integers(2)), derotate=bool(rng.integers(2)),
              input_offset=int(rng.integers(-1, D)))
    channel_major = bool(rng.integers(2))
    cuts = sorted({0, n, *(int(c) for c in rng.integers(0, n, size=int(rng.integers(0, 3))))})
    with Channelizer(M, taps=h, decimation=D, sample_format=fmt, bit_width=bw, channel_major=channel_major, **kw) as ch:
        ch.set_option(L.PFB_OPT_KERNEL, 2)
        if rng.random() < 0.6:
            ch.set_option(L.PFB_OPT_SCHEDULE, int(scheds[int(rng.integers(len(scheds)))]))
        if rng.random() < 0.5:
            ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, int(rng.choice([8, 12, 24, 32, 40, 64, 100, 256])))
        parts = [ch(iq[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
        assert ch.last_kernel.startswith("pfb_fast")
        got = np.concatenate([q for q in parts if q.size], axis=1 if channel_major else 0) if any(q.size for q in parts) else parts[0]
    x = (iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)) if fmt == "cf32" else oracle.unpack(iq, bw)
    cfg = OracleConfig(M, P, D, fftshift=kw["fftshift"], conj_input=kw["conjugate_input"], derotate=kw["derotate"], off=kw["input_offset"])
    want = oracle.channelize(x, h.astype(np.float64), cfg, "fft" if (M & (M - 1)) == 0 else "polyphase")
    if channel_major:
        want = want.T
    assert got.shape == want.shape, (case, M, fmt, kw)
    scale = max(float(np.abs(want).max()), 1e-30)
    assert float(np.abs(got - want).max()) / scale < 1e-5, (case, M, fmt, kw)   # the fp32 tolerance of the parity tests
