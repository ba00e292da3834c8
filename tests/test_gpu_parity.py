"""Parity of the HIP path against the CPU oracle, through the C ABI (run on the MI355X
box with ``-m gpu``).  Tolerances: the int->complex unpack and the commutator indexing
are BIT-EXACT (one-hot-tap KAT); the full fp32 path must stay within 1e-5 of the float64
oracle relative to the output's peak magnitude (BASELINE.json north star)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.pfb_oracle import OracleConfig  # noqa: E402
from sdr_channelizer_amd import Channelizer, synth  # noqa: E402
from sdr_channelizer_amd import _lib as L  # noqa: E402

REL_TOL = 1e-5  # north star: <= 1e-5 relative error


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def oracle_run(oracle, iq, h, M, P, D, bw, fmt="int", **kw):
    if fmt == "cf32":
        x = iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)
    else:
        x = oracle.unpack(iq, bw)
    method = "fft" if (M & (M - 1)) == 0 else "polyphase"
    return oracle.channelize(x, np.asarray(h, dtype=np.float64), OracleConfig(M, P, D, **kw), method)


FMT_NAME = {"int8": "int8", "int16": "int16", "cf32": "cf32"}
# band counts M = 2^a 3^b 5^c 7^d that are plausible radio rates (numBands = fs * 1e-6, channelizer_example.m:29;
# round(fs / 0.1e6), generate_channelized_training_iq.m:95-96): the fused shapes of csrc/pfb_kernels_mixed.hip
MIXED_RADIX_BANDS = (12, 24, 25, 30, 48, 50, 80, 96, 100, 112, 120, 160, 200, 250, 280, 320, 400, 500, 512)


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "ref56", "ref560"])
@pytest.mark.parametrize("kernel", [0, 1])
def test_golden_fixtures(golden_dir, name, kernel):
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    M, P, D, bw, fmt = int(g["M"]), int(g["P"]), int(g["D"]), int(g["bit_width"]), str(g["fmt"])
    with Channelizer(M, taps=g["taps"], decimation=D, sample_format=fmt, bit_width=max(bw, 1)) as ch:
        ch.set_option(L.PFB_OPT_KERNEL, kernel)
        y = ch(g["iq"])
        assert y.shape == g["expected"].shape and y.dtype == np.complex64
        assert rel(y, g["expected"]) < REL_TOL, (name, ch.last_kernel)
        if kernel == 1:
            assert ch.last_kernel == "pfb_generic"


def test_cfg2_uses_the_fast_kernel(golden_dir):
    g = np.load(os.path.join(golden_dir, "cfg2.npz"))
    with Channelizer(64, taps=g["taps"], bit_width=12) as ch:
        ch.set_option(L.PFB_OPT_KERNEL, 2)  # require the hand-written fast path
        y = ch(g["iq"])
        assert ch.last_kernel.startswith("pfb_fast<M64,P12,D64,int16>")
        assert rel(y, g["expected"]) < REL_TOL


@pytest.mark.parametrize("M,P,D,fmt,bw", [(64, 12, 64, "int16", 12), (64, 12, 64, "int8", 8), (256, 8, 256, "int8", 8),
                                          (128, 12, 64, "int16", 12), (1024, 16, 1024, "int16", 16),
                                          (56, 12, 56, "int16", 12), (8, 12, 8, "int16", 12),
                                          (560, 12, 560, "int16", 12), (560, 12, 560, "int8", 8), (56, 12, 56, "int8", 8),
                                          (32, 12, 32, "int16", 12), (16, 12, 16, "int16", 12), (20, 12, 20, "int16", 12),
                                          (10, 12, 10, "int16", 12), (40, 12, 40, "int16", 12)]
                         + [(m, 12, m, "int16", 12) for m in MIXED_RADIX_BANDS])
@pytest.mark.parametrize("q0", [0, 3])
def test_one_hot_tap_bit_exact(M, P, D, fmt, bw, q0):
    """h = delta[n - M q0] => every channel of frame m equals x[mD + D-1 - M q0] exactly:
    pins the integer unpack, the 2^-(bw-1) scale and the commutator indexing bit for bit."""
    rng = np.random.default_rng(M + q0)
    full = 2 ** (bw - 1)
    dt = np.int8 if fmt == "int8" else np.int16
    iq = rng.integers(-full, full, size=(D * 96 + 0, 2)).astype(dt)
    h = np.zeros(M * P, np.float32)
    h[M * q0] = 1.0
    with Channelizer(M, taps=h, decimation=D, sample_format=fmt, bit_width=bw) as ch:
        y = ch(iq)
    x = (iq[:, 0].astype(np.float32) + 1j * iq[:, 1].astype(np.float32)) / np.float32(full)
    F = iq.shape[0] // D
    s = np.arange(F) * D + (D - 1) - M * q0
    want = np.where(s >= 0, x[np.clip(s, 0, None)], 0).astype(np.complex64)
    assert np.array_equal(y, np.repeat(want[:, None], M, axis=1))


@pytest.mark.parametrize("M,P,D,fmt,bw,n", [
    (64, 12, 64, "int16", 12, 1 << 18), (64, 12, 64, "int8", 8, 1 << 16), (64, 12, 64, "cf32", 0, 1 << 16),
    (256, 8, 256, "int8", 8, 1 << 17), (128, 12, 64, "int16", 12, 1 << 16), (1024, 16, 1024, "int16", 16, 1 << 18),
    (32, 12, 32, "int16", 12, 1 << 14), (16, 4, 8, "int16", 16, 1 << 12), (56, 12, 56, "int16", 12, 56 * 300),
    (12, 5, 4, "int8", 8, 4 * 500), (560, 12, 560, "int16", 12, 560 * 150 + 31), (560, 12, 560, "int8", 8, 560 * 64),
    (56, 12, 56, "int8", 8, 56 * 1500 + 3), (16, 12, 16, "int16", 12, 16 * 4000 + 9), (8, 12, 8, "int16", 12, 8 * 9000 + 3),
    (20, 12, 20, "int16", 12, 20 * 3000 + 7), (10, 12, 10, "int16", 12, 10 * 5000 + 3), (40, 12, 40, "int16", 12, 40 * 2100 + 11),
    (56, 12, 56, "cf32", 0, 56 * 1200 + 5), (128, 12, 64, "cf32", 0, (1 << 16) + 17), (256, 8, 256, "cf32", 0, (1 << 17) + 100),
    (560, 12, 560, "cf32", 0, 560 * 130 + 77), (1024, 16, 1024, "cf32", 0, (1 << 18) + 300)]
    + [(m, 12, m, "int16", 12, m * (700 if m < 100 else 260) + 11) for m in MIXED_RADIX_BANDS])
def test_random_stream_vs_oracle(oracle, M, P, D, fmt, bw, n):
    rng = np.random.default_rng(n + M)
    if fmt == "cf32":
        iq = rng.standard_normal((n, 2)).astype(np.float32)
    else:
        iq = synth.pulsed_iq_numpy(n, bw, np.int8 if fmt == "int8" else np.int16, seed=n + M)
    h = rng.standard_normal(M * P).astype(np.float32) / np.float32(M)
    with Channelizer(M, taps=h, decimation=D, sample_format=fmt, bit_width=max(bw, 1)) as ch:
        y = ch(iq)
        if (M, P, D) in ((64, 12, 64), (56, 12, 56), (128, 12, 64), (256, 8, 256), (8, 12, 8), (560, 12, 560),
                         (1024, 16, 1024)) or (M in MIXED_RADIX_BANDS and P == 12 and D == M and fmt == "int16"):
            assert ch.last_kernel.startswith("pfb_fast<M%d," % M), ch.last_kernel  # fused (the first group in every format)
    want = oracle_run(oracle, iq, h, M, P, D, bw, fmt)
    assert rel(y, want) < REL_TOL


@pytest.mark.parametrize("M,P,D", [(6, 4, 6), (36, 12, 36), (45, 12, 45), (63, 6, 63), (75, 12, 25), (90, 12, 90), (126, 8, 63),
                                   (360, 12, 360), (600, 12, 600), (22, 12, 22), (34, 5, 17), (121, 3, 121), (1000, 4, 1000)])
def test_generic_kernel_any_band_count(oracle, M, P, D):
    """Band counts without a fused shape: the generic kernel runs a run-time mixed-radix FFT (radices 2 ... 7) when
    M = 2^a 3^b 5^c 7^d -- any other radio rate in MHz -- and a plain DFT when M has a larger prime factor (22, 34, 121)."""
    n = D * 150 + 5
    iq = synth.pulsed_iq_numpy(n, 12, np.int16, seed=M)
    h = np.random.default_rng(M).standard_normal(M * P).astype(np.float32) / np.float32(M)
    with Channelizer(M, taps=h, decimation=D, bit_width=12) as ch:
        y = ch(iq)
        assert ch.last_kernel == "pfb_generic"
    assert rel(y, oracle_run(oracle, iq, h, M, P, D, 12)) < REL_TOL


@pytest.mark.parametrize("M,P,D", [(64, 12, 64), (128, 12, 64), (56, 12, 56), (560, 12, 560), (32, 12, 32), (8, 12, 8),
                                   (20, 12, 20), (10, 12, 10), (24, 12, 24), (25, 12, 25), (96, 12, 96), (250, 12, 250),
                                   (320, 12, 320)])
@pytest.mark.parametrize("kw", [dict(fftshift=True), dict(conjugate_input=True), dict(derotate=True),
                                dict(input_offset=0), dict(input_offset=5),
                                dict(fftshift=True, conjugate_input=True, derotate=True)])
def test_switches(oracle, M, P, D, kw):
    n = D * 200
    iq = synth.pulsed_iq_numpy(n, 12, np.int16, seed=99)
    h = oracle.design_prototype(M, P).astype(np.float32)
    okw = dict(fftshift=kw.get("fftshift", False), conj_input=kw.get("conjugate_input", False),
               derotate=kw.get("derotate", False), off=kw.get("input_offset", -1))
    with Channelizer(M, taps=h, decimation=D, bit_width=12, **kw) as ch:
        y = ch(iq)
    want = oracle_run(oracle, iq, h, M, P, D, 12, **okw)
    assert rel(y, want) < REL_TOL


@pytest.mark.parametrize("M,P,D,fmt,bw", [(64, 12, 64, "int16", 12), (16, 4, 8, "int16", 12), (256, 8, 256, "int8", 8),
                                          (1024, 16, 1024, "int16", 16), (128, 12, 64, "int16", 12),
                                          (56, 12, 56, "int16", 12), (560, 12, 560, "int16", 12), (8, 12, 8, "int16", 12),
                                          (20, 12, 20, "int16", 12)])
def test_channel_major_layout(oracle, M, P, D, fmt, bw):
    """PFB_LAYOUT_CHANNEL_MAJOR = MATLAB's column-major F x M.  Every fused shape writes it: transposed
    in LDS inside the kernel, by the kernel's own stores, or from frame-major slabs (M = 1024, 560); (16, 4, 8) has
    no fused path and takes the generic kernel."""
    iq = synth.pulsed_iq_numpy(D * 333, bw, np.int8 if fmt == "int8" else np.int16, seed=5)
    h = oracle.design_prototype(M, P).astype(np.float32)
    with Channelizer(M, taps=h, decimation=D, sample_format=fmt, bit_width=bw, channel_major=True, fftshift=True,
                     derotate=(D != M)) as ch:
        ch.set_option(L.PFB_OPT_HOST_CHUNK_SAMPLES, D * 100)  # several staging chunks
        y = ch(iq)
        assert ch.last_kernel.startswith("pfb_fast") == (M != 16)
        ch.reset()
        ch.set_option(L.PFB_OPT_KERNEL, 1)
        y_generic = ch(iq)
    want = oracle_run(oracle, iq, h, M, P, D, bw, fftshift=True, derotate=(D != M))
    assert y.shape == (M, want.shape[0])
    assert rel(y.T, want) < REL_TOL
    assert rel(y, y_generic) < 2e-6


def test_channel_major_device_path_is_bit_identical_to_frame_major():
    """same arithmetic, different store addresses: the transposed frame-major result, bit for bit; cut into calls too"""
    import torch
    M, P = 64, 12
    n = M * 5000 + 13
    iq = synth.pulsed_iq_torch(n, 12, device="cuda")
    h = np.random.default_rng(4).standard_normal(M * P).astype(np.float32) / M
    with Channelizer(M, taps=h, bit_width=12) as a, Channelizer(M, taps=h, bit_width=12, channel_major=True) as b:
        a.set_option(L.PFB_OPT_SCHEDULE, 0)
        fm = a(iq)
        cm = b(iq)
        assert torch.equal(cm, fm.T.contiguous()) and cm.shape == (M, n // M)
        b.reset()
        parts = [b(iq[s:e]) for s, e in ((0, M * 1000 + 7), (M * 1000 + 7, M * 1001), (M * 1001, n))]
        assert torch.equal(torch.cat([p_ for p_ in parts if p_.numel()], dim=1), cm)


@pytest.mark.parametrize("M,P,D,fmt,bw,routes", [
    (64, 12, 64, "int16", 12, ((0, 0), (2, 0), (8, 8), (8, 16), (8, 32), (9, 64), (9, 0))),
    (64, 12, 64, "int8", 8, ((8, 16), (9, 192))),
    (56, 12, 56, "int16", 12, ((0, 0), (8, 16), (9, 128))),
    (128, 12, 64, "int16", 12, ((2, 0), (8, 8), (8, 16), (9, 64))),
    (256, 8, 256, "int8", 8, ((0, 0), (8, 8), (9, 64))),
    (32, 12, 32, "int16", 12, ((8, 32), (9, 64))),
    (1024, 16, 1024, "int16", 12, ((-1, 0), (9, 64), (9, 192))),
    (560, 12, 560, "int16", 12, ((-1, 0), (9, 128))),
    (20, 12, 20, "int16", 12, ((0, 0), (9, 64)))])
def test_channel_major_routes_are_bit_identical(M, P, D, fmt, bw, routes):
    """Channel-major output by every route -- the kernel's own stores (schedules 0 and 2), short runs transposed in LDS
    (schedule 8; frames_per_block = frames per wave), frame-major slabs + the transpose kernel (schedule 9; the default
    of the team plans) -- with and without fused abs(), over a stream cut into calls: the transposed frame-major result,
    bit for bit."""
    import torch
    n = D * 1500 + 11
    iq = synth.pulsed_iq_torch(n, bw, torch.int8 if fmt == "int8" else torch.int16, device="cuda")
    h = np.random.default_rng(6).standard_normal(M * P).astype(np.float32) / M
    cuts = [0, D * 700 + 5, D * 701, n]
    for mag in (False, True):
        kw = dict(taps=h, decimation=D, sample_format=fmt, bit_width=bw, magnitude=mag, fftshift=True, derotate=(D != M))
        with Channelizer(M, **kw) as a:
            want = a(iq).T.contiguous()
        for sched, arg in routes:
            with Channelizer(M, channel_major=True, **kw) as b:
                if sched >= 0:
                    b.set_option(L.PFB_OPT_SCHEDULE, sched)
                if sched == 8:
                    b.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, arg)
                    b.set_option(L.PFB_OPT_TILE_WAVES, {8: 4, 16: 4, 32: 2}[arg])
                if sched == 9:
                    b.set_option(L.PFB_OPT_SLAB_FRAMES, arg)
                got = b(iq)
                assert torch.equal(got, want), (M, mag, sched, arg)
                b.reset()
                parts = [b(iq[s:e]) for s, e in zip(cuts[:-1], cuts[1:])]
                assert torch.equal(torch.cat([q for q in parts if q.numel()], dim=1), want), (M, mag, sched, arg)


@pytest.mark.parametrize("M,P,log2n", [(64, 12, 28), (1024, 16, 28), (560, 12, 27)])
def test_channel_major_at_size_equals_frame_major(M, P, log2n):
    """The default channel-major route at BASELINE-like sizes (millions of frames: thousands of transposed tiles, or
    several default-length slabs for the team plans) against the frame-major result, bit for bit, column by column."""
    import torch
    n = (1 << log2n) + 3 * M + 5
    iq = synth.pulsed_iq_torch(n, 12, device="cuda")
    h = np.random.default_rng(8).standard_normal(M * P).astype(np.float32) / M
    with Channelizer(M, taps=h, bit_width=12, fftshift=True) as a, \
            Channelizer(M, taps=h, bit_width=12, fftshift=True, channel_major=True) as b:
        fm = a(iq)
        cm = b(iq)
        assert cm.shape == (M, n // M) and fm.shape == (n // M, M)
        for c0 in range(0, M, 64):   # compare in column blocks: no third full-size buffer
            assert torch.equal(cm[c0:c0 + 64], fm[:, c0:c0 + 64].T), (M, c0)


@pytest.mark.parametrize("M,P,D,fmt,bw", [(64, 12, 64, "int16", 12), (128, 12, 64, "int16", 12), (56, 12, 56, "int16", 12),
                                          (256, 8, 256, "int8", 8), (560, 12, 560, "int16", 12), (16, 12, 16, "int16", 12),
                                          (20, 12, 20, "int16", 12), (40, 12, 40, "int16", 12)])
def test_chunked_equals_one_shot_bit_exact(M, P, D, fmt, bw):
    """The handle is stateful like the System object (channelizer_example.m:50-56): any split of
    the stream -- including pieces that are not multiples of D -- gives identical bits."""
    n = D * 700 + 17
    iq = synth.pulsed_iq_numpy(n, bw, np.int8 if fmt == "int8" else np.int16, seed=3)
    h = np.random.default_rng(1).standard_normal(M * P).astype(np.float32)
    with Channelizer(M, taps=h, decimation=D, sample_format=fmt, bit_width=bw) as ch:
        one = ch(iq)
        ch.reset()
        cuts = [0, D * 3, D * 3 + 5, D * 120 + 1, D * 121, D * 400 + 63, n]
        parts = [ch(iq[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
        assert ch.frames_for(0) == 0
    many = np.concatenate([p for p in parts if p.size], axis=0)
    assert many.shape == one.shape == (n // D, M)
    assert np.array_equal(many, one)


def test_state_blob_round_trip():
    M, P = 64, 12
    iq = synth.pulsed_iq_numpy(M * 500, 12, np.int16, seed=8)
    h = np.random.default_rng(2).standard_normal(M * P).astype(np.float32)
    with Channelizer(M, taps=h, bit_width=12) as a, Channelizer(M, taps=h, bit_width=12) as b:
        full = a(iq)
        a.reset()
        a(iq[: M * 200 + 9])
        b.set_state(a.get_state())  # resume in a fresh handle
        tail = b(iq[M * 200 + 9:])
    assert np.array_equal(tail, full[200:])


def test_time_sharded_equals_single_stream_bit_exact():
    """shard g is primed with the last history_samples() raw samples of shard g-1."""
    M, P, D = 64, 12, 64
    n = D * 1024
    iq = synth.pulsed_iq_numpy(n, 12, np.int16, seed=21)
    h = np.random.default_rng(4).standard_normal(M * P).astype(np.float32)
    with Channelizer(M, taps=h, bit_width=12) as ch:
        one = ch(iq)
        G = 4
        seg = n // G
        outs = []
        for g in range(G):
            ch.reset()
            if g:
                ch.prime(iq[g * seg - ch.history_samples: g * seg])
            outs.append(ch(iq[g * seg:(g + 1) * seg]))
    assert np.array_equal(np.concatenate(outs, axis=0), one)


def test_fast_and_generic_kernels_agree():
    M, P = 64, 12
    iq = synth.pulsed_iq_numpy(M * 4096 + 1000, 12, np.int16, seed=77)
    h = np.random.default_rng(6).standard_normal(M * P).astype(np.float32) / M
    with Channelizer(M, taps=h, bit_width=12) as ch:
        ch.set_option(L.PFB_OPT_SCHEDULE, 0)
        ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, 104)  # not a multiple of the grid: partial last block
        fast = ch(iq)
        assert ch.last_kernel.startswith("pfb_fast")
        ch.reset()
        ch.set_option(L.PFB_OPT_KERNEL, 1)
        gen = ch(iq)
    assert rel(fast, gen) < 2e-6


@pytest.mark.parametrize("M,P,nvar", [(1024, 16, 4), (560, 12, 3)])
def test_every_registered_plan_variant_matches_the_oracle(oracle, M, P, nvar):
    """PFB_OPT_VARIANT: the alternative fused plans kept for a shape (cfg4: the FIR-team / FFT-team kernel, 16 waves x
    1 column 8 x 8 x 16, 8 waves x 2 columns 16 x 16 x 4, independent 4-wave workgroups whose waves transform whole
    frames; M = 560: teams on 2-frame chunks with two workgroups per CU, teams on 4-frame chunks, 9 waves in lockstep) all meet the fp32 tolerance, also on a stream that is not a whole number
    of workgroups; an index past the last registered plan is refused and leaves the handle usable."""
    n = M * 700 + 17
    iq = synth.pulsed_iq_numpy(n, 12, np.int16, seed=31)
    h = np.random.default_rng(8).standard_normal(M * P).astype(np.float32) / M
    want = oracle_run(oracle, iq, h, M, P, M, 12)
    names = set()
    with Channelizer(M, taps=h, bit_width=12) as ch:
        for v in range(nvar):
            ch.set_option(L.PFB_OPT_VARIANT, v)
            ch.reset()
            got = ch(iq)
            names.add(ch.last_kernel)
            assert ch.last_kernel.startswith("pfb_fast")
            assert rel(got, want) < 1e-5
        with pytest.raises(Exception):
            ch.set_option(L.PFB_OPT_VARIANT, 9)
        ch.reset()
        assert rel(ch(iq), want) < 1e-5
    assert len(names) == nvar


@pytest.mark.parametrize("kw", [{}, {"fftshift": True, "conjugate_input": True}, {"magnitude": True}])
def test_wave_frame_plan_gives_the_team_plan_s_bits(kw):
    """cfg4's alternative plan (schedule 13: unsynchronised 4-wave workgroups, packed raw window, every wave transforms whole
    frames by itself; one workgroup per run or resident workgroups chaining short runs) against the default team plan: same
    taps in the same order through the same passes, so the outputs are bit-identical -- one shot with a ragged tail (the
    careful path takes the run that touches the history and the partial last chunk), two calls cut at an odd sample (the
    second call is misaligned for the vector loads: every run takes the careful path), several run lengths."""
    M, P = 1024, 16
    n = M * 1500 + 5
    iq = synth.pulsed_iq_numpy(n, 12, np.int16, seed=77)
    h = np.random.default_rng(3).standard_normal(M * P).astype(np.float32) / M
    with Channelizer(M, taps=h, bit_width=12, **kw) as ch:
        ref = ch(iq)
        assert "int16>" in ch.last_kernel
        ch.set_option(L.PFB_OPT_VARIANT, 3)
        for fpb, grid in ((0, 0), (8, 0), (24, 0), (200, 0), (16, 24), (64, 7)):
            ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, fpb)
            ch.set_option(L.PFB_OPT_GRID, grid)
            ch.reset()
            got = ch(iq)
            assert ch.last_kernel.endswith("duo>")
            assert np.array_equal(got, ref), (fpb, grid)
            ch.reset()
            cut = M * 611 + 3
            assert np.array_equal(np.concatenate([ch(iq[:cut]), ch(iq[cut:])]), ref), (fpb, grid, "two calls")


@pytest.mark.parametrize("fmt,bw", [("int16", 12), ("int8", 8)])
def test_m560_team_plans_give_the_same_bits(fmt, bw):
    """M = 560: the default (teams on 2-frame chunks, two unsynchronised workgroups per CU) and variant 1 (one workgroup
    per CU on 4-frame chunks) run the same taps in the same order through the same 14 x 10 x 4 passes: bit-identical, one
    shot with a ragged tail, cut at an odd sample, several run lengths, with fftshift and fused abs()."""
    M, P = 560, 12
    n = M * 2100 + 7
    iq = synth.pulsed_iq_numpy(n, bw, np.int8 if fmt == "int8" else np.int16, seed=78)
    h = np.random.default_rng(4).standard_normal(M * P).astype(np.float32) / M
    for kw in ({}, {"fftshift": True, "magnitude": True}):
        with Channelizer(M, taps=h, sample_format=fmt, bit_width=bw, **kw) as ch:
            ch.set_option(L.PFB_OPT_VARIANT, 1)
            ref = ch(iq)
            assert ch.last_kernel.endswith(",4f>")
            ch.set_option(L.PFB_OPT_VARIANT, 0)
            for fpb in (0, 4, 36, 512):
                ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, fpb)
                ch.reset()
                assert np.array_equal(ch(iq), ref), (kw, fpb)
                assert ch.last_kernel.endswith(fmt + ">")
                ch.reset()
                cut = M * 611 + 3
                assert np.array_equal(np.concatenate([ch(iq[:cut]), ch(iq[cut:])]), ref), (kw, fpb, "two calls")


@pytest.mark.parametrize("M,P,D,fmt,bw,P_fused", [(64, 8, 64, "int16", 12, 12), (64, 5, 64, "int8", 8, 12), (256, 6, 256, "int8", 8, 8),
                                                  (1024, 10, 1024, "int16", 16, 16), (128, 7, 64, "int16", 12, 12),
                                                  (8, 3, 8, "int16", 12, 12), (64, 16, 64, "int16", 12, 16), (64, 14, 64, "int16", 12, 16)])
def test_shorter_prototypes_run_on_the_fused_shapes(oracle, M, P, D, fmt, bw, P_fused):
    """taps_per_channel below a fused shape's: the same filter with zero taps appended, so the fused kernel serves it
    (the oracle is evaluated with the ORIGINAL P); the stream state grows to the padded length."""
    n = D * 900 + 13
    iq = synth.pulsed_iq_numpy(n, bw, np.int8 if fmt == "int8" else np.int16, seed=M + P)
    h = np.random.default_rng(P).standard_normal(M * P).astype(np.float32) / M
    with Channelizer(M, taps=h, decimation=D, sample_format=fmt, bit_width=bw) as ch:
        y = ch(iq)
        assert ch.last_kernel.startswith("pfb_fast<M%d,P%d," % (M, P_fused))
        assert ch.history_samples == M * P_fused + D
        ch.reset()
        cut = D * 333 + 5
        assert np.array_equal(np.concatenate([ch(iq[:cut]), ch(iq[cut:])]), y)
    assert rel(y, oracle_run(oracle, iq, h, M, P, D, bw)) < REL_TOL
    with Channelizer(M, taps=np.ones(M * 17, np.float32), decimation=D, sample_format=fmt, bit_width=bw) as ch:
        ch(iq[: D * 40])
        assert ch.last_kernel == "pfb_generic"      # longer than any fused shape: no padding possible


def test_empty_and_tiny_inputs():
    M, P = 64, 12
    h = np.ones(M * P, np.float32)
    with Channelizer(M, taps=h, bit_width=12) as ch:
        assert ch(np.zeros((0, 2), np.int16)).shape == (0, M)
        assert ch(np.ones((63, 2), np.int16)).shape == (0, M)    # carried, no frame yet
        y = ch(np.ones((1, 2), np.int16))                          # completes frame 0
        assert y.shape == (1, M)
        # frame 0 = 64 ones through an all-ones filter: channel 0 sums 64 samples, scaled by 2^-11
        assert y[0, 0] == np.complex64(complex(64 / 2048, 64 / 2048))


def test_capacity_error_leaves_state_untouched():
    import ctypes as C
    M, P = 64, 12
    iq = synth.pulsed_iq_numpy(M * 10, 12, np.int16, seed=1)
    h = np.ones(M * P, np.float32)
    with Channelizer(M, taps=h, bit_width=12) as ch:
        out = np.empty((4, M), np.complex64)
        f = C.c_uint64()
        rc = ch._lib.pfb_process(ch._h, C.c_void_p(iq.ctypes.data), 10 * M, C.c_void_p(out.ctypes.data), 4,
                                 C.byref(f), L.PFB_MEM_HOST)
        assert rc == L.PFB_ERR_CAPACITY and f.value == 10
        assert ch(iq).shape == (10, M)  # nothing was consumed by the failed call


def test_device_tensor_path_matches_host_path():
    import torch
    M, P = 64, 12
    iq = synth.pulsed_iq_numpy(M * 2048, 12, np.int16, seed=31)
    h = np.random.default_rng(3).standard_normal(M * P).astype(np.float32)
    with Channelizer(M, taps=h, bit_width=12) as ch:
        host = ch(iq)
        ch.reset()
        s = torch.cuda.Stream()
        ch.set_stream(s.cuda_stream)
        with torch.cuda.stream(s):
            d = torch.from_numpy(iq).cuda()
            y = ch(d)
        assert y.is_cuda and tuple(y.shape) == host.shape
        assert np.array_equal(y.cpu().numpy(), host)


@pytest.mark.parametrize("M,P,D,fmt,bw", [(64, 12, 64, "int16", 12), (256, 8, 256, "int8", 8), (1024, 16, 1024, "int16", 16),
                                          (128, 12, 64, "int16", 12)])
def test_bench_stream_prefix_of_2e20_samples(oracle, M, P, D, fmt, bw):
    """SURVEY.md section 8d's error metric: max|y_gpu - y_oracle| / max|y_oracle| <= 1e-5 on a 2^20-sample prefix of
    the benchmark's own synthetic stream (same generator, same seed, same prototype as bench.py)."""
    import torch
    from sdr_channelizer_amd import design_prototype
    n = 1 << 20
    iq = synth.pulsed_iq_torch(n, bw, torch.int8 if fmt == "int8" else torch.int16, seed=synth.SEED, device="cuda")
    h = design_prototype(M, P, 80.0)
    with Channelizer(M, taps=h, decimation=D, sample_format=fmt, bit_width=bw) as ch:
        ch.set_option(L.PFB_OPT_KERNEL, 2)
        y = ch(iq).cpu().numpy()
    want = oracle_run(oracle, iq.cpu().numpy(), h, M, P, D, bw)
    assert y.shape == want.shape == (n // D, M)
    assert rel(y, want) < REL_TOL


@pytest.mark.parametrize("M,P,D,fmt,bw,log2n", [(64, 12, 64, "int16", 12, 30), (64, 12, 64, "int16", 12, 31), (64, 12, 64, "int16", 12, 33),
                                                (256, 8, 256, "int8", 8, 30),
                                                (1024, 16, 1024, "int16", 16, 30), (128, 12, 64, "int16", 12, 28),
                                                (56, 12, 56, "int16", 12, 26), (560, 12, 560, "int16", 12, 26)])
def test_large_stream_interior_windows(oracle, M, P, D, fmt, bw, log2n):
    """BASELINE.json's full sizes through size-independent checks: run the whole synthetic stream
    (generated in HBM), then compare random interior frame windows -- plus the first and last frames and
    windows on pulses -- with the oracle evaluated on just the samples those frames depend on."""
    import torch
    n = 1 << log2n
    need = n * (2 if fmt == "int8" else 4) + (n // D) * M * 8
    if torch.cuda.mem_get_info()[0] < need * 1.3:
        pytest.skip(f"needs {need >> 30} GiB of HBM")
    h = oracle.design_prototype(M, P).astype(np.float32)
    iq = synth.pulsed_iq_torch(n, bw, torch.int8 if fmt == "int8" else torch.int16, device="cuda")
    with Channelizer(M, taps=h, decimation=D, sample_format=fmt, bit_width=bw) as ch:
        y = ch(iq)
        assert ch.last_kernel.startswith("pfb_fast")
    F = n // D
    assert tuple(y.shape) == (F, M)
    W = M * P // D  # rows a frame depends on
    nwin = 16
    rng = np.random.default_rng(0)
    pulse_frames = (np.arange(0, n, 56000) // D)[1:]
    starts = list(rng.integers(W, F - nwin, size=6)) + [0, F - nwin] + list(pulse_frames[rng.integers(0, len(pulse_frames), 4)])
    for f0 in starts:
        f0 = int(min(f0, F - nwin))
        lo = max(0, (f0 - W) * D)
        seg = iq[lo:(f0 + nwin) * D].cpu().numpy()
        want = oracle_run(oracle, seg, h, M, P, D, bw)[-nwin:]
        got = y[f0:f0 + nwin].cpu().numpy()
        assert np.abs(got - want).max() / max(np.abs(want).max(), 0.05) < REL_TOL, f0
    for part in y.split(max(1, (1 << 28) // M)):
        assert bool(torch.isfinite(torch.view_as_real(part)).all())
    del y, iq
    torch.cuda.empty_cache()


def test_cfg1_as_the_example_script_calls_it(oracle):
    """BASELINE config 1 the way /root/reference/matlab/channelizer_example.m runs it: M = 8 (12 taps per band = the
    96-tap prototype), complex float input, `iq'` (conjugated, :23), ONE stateful System object (:31) fed overlapping
    windows of 5 ms = 5000 frames hopped by 100 frames (:33-34, :50-53), `abs` (:56) and `fftshift(., 2)` (:58) -- 10^6
    samples, every window compared.  The object keeps its filter state across the overlapping calls (the first 11 frames
    of every window are computed from the PREVIOUS window's tail): the oracle, which has no state,
    is fed [previous window's last M*P samples | window] and its first P frames are dropped -- the same call sequence."""
    M, P, N = 8, 12, 10 ** 6
    win, hop = 5000 * M, 100 * M
    q = synth.pulsed_iq_numpy(N, 12, np.int16, seed=5)
    iq = (q.astype(np.float32) / np.float32(2048.0))  # (N, 2) float32: what the script holds after :18-21
    x = iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)
    h = oracle.design_prototype(M, P).astype(np.float32)
    cfg = OracleConfig(M, P, M, conj_input=True, fftshift=True)
    worst, calls = 0.0, 0
    with Channelizer(M, taps=h, sample_format="cf32", conjugate_input=True, magnitude=True, fftshift=True) as ch:
        tail = np.zeros(M * P, dtype=np.complex128)  # a fresh object: zero state
        for ii in range(0, N - win + 1, hop):
            got = ch(iq[ii:ii + win])
            assert got.shape == (win // M, M) and got.dtype == np.float32
            want = np.abs(oracle.channelize(np.concatenate([tail, x[ii:ii + win]]), h.astype(np.float64), cfg, "fft"))[P:]
            worst = max(worst, float(np.abs(got - want).max() / np.abs(want).max()))
            tail = x[ii + win - M * P:ii + win]
            calls += 1
        assert ch.last_kernel == "pfb_fast<M8,P12,D8,cf32>"
    assert calls == (N - win) // hop + 1 == 1201
    assert worst < REL_TOL, worst


@pytest.mark.parametrize("opts", [{L.PFB_OPT_SCHEDULE: 0}, {L.PFB_OPT_SCHEDULE: 0, L.PFB_OPT_FRAMES_PER_BLOCK: 40},
                                  {L.PFB_OPT_SCHEDULE: 2, L.PFB_OPT_TILE_WAVES: 1},
                                  {L.PFB_OPT_SCHEDULE: 2, L.PFB_OPT_TILE_WAVES: 8, L.PFB_OPT_XCD_REMAP: 0},
                                  {L.PFB_OPT_SCHEDULE: 2, L.PFB_OPT_TILE_WAVES: 8, L.PFB_OPT_NONTEMPORAL: 1},
                                  {L.PFB_OPT_SCHEDULE: 3, L.PFB_OPT_TILE_WAVES: 8, L.PFB_OPT_FRAMES_PER_BLOCK: 32},
                                  {L.PFB_OPT_SCHEDULE: 3, L.PFB_OPT_TILE_WAVES: 4, L.PFB_OPT_FRAMES_PER_BLOCK: 32},
                                  {L.PFB_OPT_SCHEDULE: 3, L.PFB_OPT_TILE_WAVES: 16, L.PFB_OPT_FRAMES_PER_BLOCK: 24,
                                   L.PFB_OPT_XCD_REMAP: 0},
                                  {L.PFB_OPT_SCHEDULE: 3, L.PFB_OPT_TILE_WAVES: 8, L.PFB_OPT_FRAMES_PER_BLOCK: 24},
                                  {L.PFB_OPT_SCHEDULE: 11}, {L.PFB_OPT_SCHEDULE: 11, L.PFB_OPT_FRAMES_PER_BLOCK: 40}])
def test_every_schedule_gives_identical_bits(oracle, opts):
    """The schedules only change which wave computes which frames, never the arithmetic."""
    M, P = 64, 12
    n = M * 5003 + 11                      # partial last chunk, carried tail
    iq = synth.pulsed_iq_numpy(n, 12, np.int16, seed=13)
    h = oracle.design_prototype(M, P).astype(np.float32)
    with Channelizer(M, taps=h, bit_width=12) as ch:
        ch.set_option(L.PFB_OPT_KERNEL, 2)
        ch.set_option(L.PFB_OPT_SCHEDULE, 0)
        ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, 512)
        ref = ch(iq)
        ch.reset()
        for k, v in opts.items():
            ch.set_option(k, v)
        got = ch(iq[:M * 1000 + 5])          # second call starts mid-frame: unaligned, history rows
        got2 = ch(iq[M * 1000 + 5:])
    want = oracle_run(oracle, iq, h, M, P, M, 12)
    assert rel(ref, want) < REL_TOL
    assert np.array_equal(np.concatenate([got, got2]), ref)


@pytest.mark.parametrize("M,P,D,fmt,bw,kw", [(64, 12, 64, "int16", 12, {}), (128, 12, 64, "int16", 12, {}),
                                             (256, 8, 256, "int8", 8, {}), (1024, 16, 1024, "int16", 16, {}),
                                             (56, 12, 56, "int16", 12, {}), (64, 12, 64, "int16", 12, dict(channel_major=True)),
                                             (560, 12, 560, "int16", 12, {})])
def test_fused_magnitude_output(oracle, M, P, D, fmt, bw, kw):
    """PFB_FLAG_MAGNITUDE = abs(channelizer(x)) of channelizer_example.m:56, fused into the store."""
    n = D * 300
    iq = synth.pulsed_iq_numpy(n, bw, np.int8 if fmt == "int8" else np.int16, seed=17)
    h = oracle.design_prototype(M, P).astype(np.float32)
    with Channelizer(M, taps=h, decimation=D, sample_format=fmt, bit_width=bw, magnitude=True, fftshift=True, **kw) as ch:
        m = ch(iq)
    want = np.abs(oracle_run(oracle, iq, h, M, P, D, bw, fftshift=True))
    if kw.get("channel_major"):
        m = m.T
    assert m.dtype == np.float32 and m.shape == want.shape
    assert np.abs(m - want).max() / want.max() < REL_TOL
    # PFB_FLAG_POWER: |y|^2 through the same stores, no square root -- and exactly the square the magnitude was the root of
    with Channelizer(M, taps=h, decimation=D, sample_format=fmt, bit_width=bw, power=True, fftshift=True, **kw) as ch:
        pw = ch(iq)
        assert ch.last_kernel.startswith("pfb_fast")
        ch.set_option(L.PFB_OPT_KERNEL, 1)
        ch.reset()
        pw_generic = ch(iq)
    if kw.get("channel_major"):
        pw, pw_generic = pw.T, pw_generic.T
    assert pw.dtype == np.float32 and pw.shape == want.shape
    assert np.abs(pw - want ** 2).max() / (want ** 2).max() < 2 * REL_TOL
    assert np.abs(pw_generic - want ** 2).max() / (want ** 2).max() < 2 * REL_TOL
    assert np.all(np.abs(np.sqrt(pw) - m) <= np.spacing(m))


def test_misaligned_device_buffer_and_cf32(oracle):
    """A device pointer that is only sample-aligned takes the checked load path; cf32 input has its own kernel."""
    import torch
    M, P = 256, 8
    n = M * 400
    iq = synth.pulsed_iq_numpy(n + 3, 8, np.int8, seed=23)
    h = oracle.design_prototype(M, P).astype(np.float32)
    d = torch.from_numpy(iq).cuda()
    with Channelizer(M, taps=h, sample_format="int8", bit_width=8) as ch:
        y = ch(d[3:])                      # byte offset 6: not 8-byte aligned for the 4-column vector loads
        assert ch.last_kernel.startswith("pfb_fast<M256")
    want = oracle_run(oracle, iq[3:], h, M, P, M, 8)
    assert rel(y.cpu().numpy(), want) < REL_TOL
    x = (np.random.default_rng(5).standard_normal((64 * 500, 2)) * 0.3).astype(np.float32)
    h64 = oracle.design_prototype(64, 12).astype(np.float32)
    with Channelizer(64, taps=h64, sample_format="cf32") as ch:
        y = ch(x)
        assert ch.last_kernel == "pfb_fast<M64,P12,D64,cf32>"
    assert rel(y, oracle_run(oracle, x, h64, 64, 12, 64, 0, "cf32")) < REL_TOL


def test_largest_supported_shapes(oracle):
    """limits of pfb_create: M up to 4096, P up to 64 (generic kernel)"""
    for M, P, D in ((4096, 2, 4096), (16, 64, 16)):
        iq = synth.pulsed_iq_numpy(D * 40, 12, np.int16, seed=M)
        h = np.random.default_rng(M).standard_normal(M * P).astype(np.float32) / M
        with Channelizer(M, taps=h, decimation=D, bit_width=12) as ch:
            y = ch(iq)
        assert rel(y, oracle_run(oracle, iq, h, M, P, D, 12)) < REL_TOL
    with pytest.raises(Exception):
        Channelizer(8192, taps=np.zeros(8192 * 2, np.float32))


@pytest.mark.parametrize("tw,fpb", [(8, 64), (4, 64), (5, 64), (8, 48), (8, 128)])
def test_paired_schedule_gives_identical_bits(oracle, tw, fpb):
    """schedule 4: FIR and FFT on different waves of a workgroup, LDS double buffer in between"""
    M, P = 64, 12
    n = M * 7001 + 3
    iq = synth.pulsed_iq_numpy(n, 12, np.int16, seed=29)
    h = oracle.design_prototype(M, P).astype(np.float32)
    with Channelizer(M, taps=h, bit_width=12) as ch:
        ch.set_option(L.PFB_OPT_SCHEDULE, 0)
        ref = ch(iq)
        ch.reset()
        ch.set_option(L.PFB_OPT_SCHEDULE, 4)
        ch.set_option(L.PFB_OPT_TILE_WAVES, tw)
        ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, fpb)
        got = np.concatenate([ch(iq[:M * 2000 + 7]), ch(iq[M * 2000 + 7:])])
    assert np.array_equal(got, ref)


def test_page_locked_host_buffers_and_pipelined_staging():
    """pfb_host_alloc buffers through the host path: five staging chunks in flight over three streams give the bits of
    one device-resident call, in both layouts"""
    import torch
    from sdr_channelizer_amd import pinned_empty
    M, P = 64, 12
    n = M * 9000 + 21
    iq = synth.pulsed_iq_numpy(n, 12, np.int16, seed=13)
    h = np.random.default_rng(3).standard_normal(M * P).astype(np.float32) / M
    iq_p = pinned_empty(iq.shape, iq.dtype)
    iq_p[:] = iq
    for cm in (False, True):
        with Channelizer(M, taps=h, bit_width=12, channel_major=cm) as ch:
            ref = ch(torch.from_numpy(iq).cuda()).cpu().numpy()
            ch.reset()
            ch.set_option(L.PFB_OPT_HOST_CHUNK_SAMPLES, M * 2000)
            out_p = pinned_empty(ref.shape, ref.dtype)
            got = ch(iq_p, out=out_p)
            assert np.array_equal(got, ref)
            ch.reset()
            assert np.array_equal(ch(iq), ref)   # pageable memory, same pipeline


@pytest.mark.parametrize("M,P,D,fmt,bw,pairs", [(128, 12, 64, "int16", 12, 6), (128, 12, 64, "int16", 12, 4), (56, 12, 56, "int16", 12, 8),
                                                (256, 8, 256, "int8", 8, 4), (64, 12, 64, "int16", 12, 8)])
def test_pairs_over_sliding_runs_give_identical_bits(M, P, D, fmt, bw, pairs):
    """schedule 7: a FIR wave and an FFT wave per long sliding run; streams that end mid-run and mid-workgroup (whole
    pairs idle), calls cut mid-run"""
    n = D * 5003 + 11
    iq = synth.pulsed_iq_numpy(n, bw, np.int8 if fmt == "int8" else np.int16, seed=17)
    h = np.random.default_rng(12).standard_normal(M * P).astype(np.float32) / M
    with Channelizer(M, taps=h, decimation=D, sample_format=fmt, bit_width=bw, derotate=(D != M)) as ch:
        ch.set_option(L.PFB_OPT_SCHEDULE, 0)
        ref = ch(iq)
        for fpb in (64, 256):
            ch.reset()
            ch.set_option(L.PFB_OPT_SCHEDULE, 7)
            ch.set_option(L.PFB_OPT_TILE_WAVES, pairs)
            ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, fpb)
            cut = D * 1777 + 5
            got = np.concatenate([ch(iq[:cut]), ch(iq[cut:])])
            assert np.array_equal(got, ref), fpb


@pytest.mark.parametrize("M,P,D,fmt,bw", [(128, 12, 64, "int16", 12), (256, 8, 256, "int8", 8), (256, 8, 256, "int16", 12),
                                          (56, 12, 56, "int16", 12), (56, 12, 56, "int8", 8), (32, 12, 32, "int16", 12),
                                          (64, 12, 64, "int8", 8), (64, 16, 64, "int16", 12)])
def test_software_pipelined_runs_give_identical_bits(M, P, D, fmt, bw):
    """schedule 11: sliding runs with the next chunk's FIR scheduled into this chunk's FFT (two LDS chunk buffers, rows two
    chunks ahead): runs of one chunk, runs that end mid-chunk, calls cut mid-frame -- the bits of schedule 0.  Runs of exactly
    one PERIOD of chunks (12, 24 or 32 frames depending on the shape) take the straight-line variant whose window is a ring
    of registers that never moves (run_overlap_ring); with fused abs() too."""
    n = D * 5003 + 11
    iq = synth.pulsed_iq_numpy(n, bw, np.int8 if fmt == "int8" else np.int16, seed=19)
    h = np.random.default_rng(14).standard_normal(M * P).astype(np.float32) / M
    with Channelizer(M, taps=h, decimation=D, sample_format=fmt, bit_width=bw, derotate=(D != M), fftshift=True) as ch:
        ch.set_option(L.PFB_OPT_SCHEDULE, 0)
        ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, 512)
        ref = ch(iq)
        for fpb in (8, 12, 24, 32, 52, 512):
            ch.reset()
            ch.set_option(L.PFB_OPT_SCHEDULE, 11)
            ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, fpb)
            cut = D * 1777 + 5
            got = np.concatenate([ch(iq[:cut]), ch(iq[cut:])])
            assert np.array_equal(got, ref), fpb
    with Channelizer(M, taps=h, decimation=D, sample_format=fmt, bit_width=bw, magnitude=True) as ch:
        ch.set_option(L.PFB_OPT_SCHEDULE, 0)
        ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, 512)
        ref = ch(iq)
        for fpb in (12, 24, 32):
            ch.reset()
            ch.set_option(L.PFB_OPT_SCHEDULE, 11)
            ch.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, fpb)
            assert np.array_equal(ch(iq), ref), ("magnitude", fpb)


def test_iq_file_front_end(oracle, tmp_path):
    """pfb_process_iq_file: record from disk -> channels, and its checks (format mismatch, truncated payload)."""
    import os
    from sdr_channelizer_amd import iqfile, PfbError
    M, P = 64, 12
    iq = synth.pulsed_iq_numpy(M * 3000 + 9, 12, np.int16, seed=41)
    path = os.path.join(tmp_path, iqfile.filename_for(1_700_000_000_123))
    iqfile.write_iq(path, iq, fs=56e6, fc=2.4e9, bit_width=12, marker=0x02020202)
    h = oracle.design_prototype(M, P).astype(np.float32)
    with Channelizer(M, taps=h, bit_width=12, fftshift=True) as ch:
        y, info = ch.process_iq_file(path)
        assert info.packet.numSamples == iq.shape[0] and info.file_format == 2
        assert rel(y, oracle_run(oracle, iq, h, M, P, M, 12, fftshift=True)) < REL_TOL
    with Channelizer(M, taps=h, bit_width=12, fftshift=True, channel_major=True) as cm:   # one M x F matrix per record
        y_cm, _ = cm.process_iq_file(path)
        assert y_cm.shape == (M, y.shape[0]) and np.array_equal(y_cm, y.T)
    big = np.tile(iq[: M * 3000], (200, 1))                 # 38 M samples: three 2^24-sample file chunks, four staging steps each
    big_path = os.path.join(tmp_path, iqfile.filename_for(1_700_000_000_456))
    iqfile.write_iq(big_path, big, fs=56e6, fc=2.4e9, bit_width=12)
    with Channelizer(M, taps=h, bit_width=12) as ch, Channelizer(M, taps=h, bit_width=12, channel_major=True) as cm:
        y_big, _ = ch.process_iq_file(big_path)
        ch.reset()
        assert np.array_equal(y_big, ch(big))               # the same stream in one call from memory
        y_big_cm, _ = cm.process_iq_file(big_path)
        assert np.array_equal(y_big_cm, y_big.T)
    with Channelizer(M, taps=h, bit_width=16) as ch16:      # record says 12-bit
        with pytest.raises(PfbError) as e:
            ch16.process_iq_file(path)
        assert e.value.status == L.PFB_ERR_BAD_FORMAT
    with open(path, "r+b") as f:                             # drop the last sample: length(iq) ~= numSamples
        f.truncate(os.path.getsize(path) - 4)
    with Channelizer(M, taps=h, bit_width=12) as ch:
        with pytest.raises(PfbError) as e:
            ch.process_iq_file(path)
        assert e.value.status == L.PFB_ERR_BAD_FORMAT


def test_cpp_host_loop_without_python(oracle, tmp_path):
    """examples/iq_channelize.cpp -- the recorder-shaped C++ loop over the C ABI -- built with g++ and run as its own
    process (no torch, no Python in it): two records of different formats, output files equal to the Python path's."""
    import shutil
    import subprocess
    from sdr_channelizer_amd import LIB_PATH, iqfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cxx = shutil.which("g++") or shutil.which("hipcc")
    if cxx is None:
        pytest.skip("no C++ compiler on this box")
    exe = os.path.join(tmp_path, "iq_channelize")
    libdir = os.path.dirname(LIB_PATH)
    subprocess.run([cxx, "-std=c++17", "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "iq_channelize.cpp"),
                    "-o", exe, "-L" + libdir, "-lpfb_channelizer", "-Wl,-rpath," + libdir], check=True)
    M, P = 64, 12
    recs = []
    for name, dt, bw, marker in (("a.iq", np.int16, 12, 0x03030303), ("b.iq", np.int8, 8, 0x02020202)):
        iq = synth.pulsed_iq_numpy(M * 2500 + 5, bw, dt, seed=bw)
        path = os.path.join(tmp_path, name)
        iqfile.write_iq(path, iq, fs=56e6, fc=915e6, bit_width=bw, marker=marker)
        recs.append((path, iq, "int8" if dt == np.int8 else "int16", bw))
    out = subprocess.run([exe, str(M), str(P)] + [r[0] for r in recs], check=True, capture_output=True, text=True).stdout
    assert out.count("frames x 64 channels") == 2 and "pfb_fast<M64" in out and out.count(" PDWs") == 2
    h = oracle.design_prototype(M, P).astype(np.float32)   # the same Kaiser design pfb_design_prototype makes
    for path, iq, fmt, bw in recs:
        got = np.fromfile(path + ".chan", dtype=np.complex64).reshape(-1, M)
        with Channelizer(M, taps=h, sample_format=fmt, bit_width=bw, fftshift=True) as ch:
            want = ch(iq)
        assert got.shape == want.shape
        assert rel(got, want) < 2e-6   # taps designed in C (float) vs by the oracle (double, then rounded)
        # ... and the PDWs of the same record, extracted by the C++ process in one call
        from sdr_channelizer_amd.pdw import PDW_DTYPE, extract_pdws
        pdws = np.fromfile(path + ".pdw", dtype=PDW_DTYPE)
        ref_pdws = extract_pdws(got, 56e6, 915e6, 0.0)
        assert len(pdws) == len(ref_pdws) > 0 and np.array_equal(pdws, ref_pdws)


def test_removed_study_schedules_are_refused():
    """schedules 1, 5, 10 and 12 were studies that lost (DESIGN.md sections 5 and 9); their numbers are not reused"""
    from sdr_channelizer_amd import design_prototype
    with Channelizer(64, taps=design_prototype(64, 12)) as ch:
        for sched in (1, 5, 10, 12, 14):
            with pytest.raises(L.PfbError):
                ch.set_option(L.PFB_OPT_SCHEDULE, sched)
        ch.set_option(L.PFB_OPT_SCHEDULE, 4)
        with pytest.raises(L.PfbError):
            ch.set_option(L.PFB_OPT_EXPERIMENT, 1 << 16)



@pytest.mark.parametrize("M,P,D,fmt,bw,tuned", [(1024, 16, 1024, "int16", 12, 512), (560, 12, 560, "int8", 8, 512), (56, 12, 56, "int16", 12, 512),
                                               (8, 12, 8, "cf32", 1, 1024), (512, 12, 512, "int16", 12, 512), (320, 12, 320, "int16", 12, 256)])
def test_default_run_length_follows_the_call_and_keeps_the_bits(M, P, D, fmt, bw, tuned):
    """Unless PFB_OPT_FRAMES_PER_BLOCK fixes it, the run length of the long-run plans is chosen per call (pfb_api.cpp
    launch_frames): whole rounds of the CUs for the team kernels, shorter runs when the tuned length would leave the chip
    partly idle.  Which frames a workgroup computes never changes a bit: calls of very different lengths, also as a
    stream of unequal calls, against the tuned length forced."""
    import torch
    from sdr_channelizer_amd import design_prototype
    lens = [37, 5000, 683, 50001, 3 * tuned * 256 // 7]
    n = sum(lens) * D + 5
    if fmt == "cf32":
        iq = torch.randn((n, 2), dtype=torch.float32, device="cuda")
    else:
        iq = synth.pulsed_iq_torch(n, bw, torch.int8 if fmt == "int8" else torch.int16, device="cuda")
    kw = dict(taps=design_prototype(M, P), decimation=D, sample_format=fmt, bit_width=bw, fftshift=True)
    with Channelizer(M, **kw) as a, Channelizer(M, **kw) as b:
        b.set_option(L.PFB_OPT_FRAMES_PER_BLOCK, tuned)
        want = b(iq)
        assert torch.equal(a(iq), want)
        a.reset()
        cuts = np.concatenate([[0], np.cumsum(lens) * D - 3, [n]])
        parts = [a(iq[s:e]) for s, e in zip(cuts[:-1], cuts[1:])]
        assert torch.equal(torch.cat([q for q in parts if q.numel()]), want)


@pytest.mark.parametrize("M,P", [(8, 12), (64, 12), (256, 8), (560, 12), (1024, 16)])
def test_non_finite_samples_stay_inside_their_window(M, P):
    """complex float32 input with one NaN and one Inf: exactly the frames whose M*P-sample window covers them are non-finite
    (every channel of them: the DFT mixes all branches), every other frame is bit-identical to the clean stream's -- nothing
    leaks through the carried state, the fused kernels and the generic one agree on which frames those are."""
    D = M
    n = D * 400
    rng = np.random.default_rng(5)
    iq = (rng.standard_normal((n, 2)) * 0.25).astype(np.float32)
    bad = iq.copy()
    i_nan, i_inf = D * 101 + 3, D * 250 + D // 2
    bad[i_nan, 0] = np.nan
    bad[i_inf, 1] = np.inf
    h = (rng.standard_normal(M * P) / M).astype(np.float32)
    h[h == 0] = 1e-3   # every tap touches its sample
    with Channelizer(M, taps=h, sample_format="cf32", bit_width=1) as ch:
        clean = ch(iq)
        ch.reset()
        cut = D * 180 + 1
        got = np.concatenate([ch(bad[:cut]), ch(bad[cut:])])   # the Inf arrives in the second call
        assert ch.last_kernel.startswith("pfb_fast")
        ch.set_option(L.PFB_OPT_KERNEL, 1)
        ch.reset()
        gen = ch(bad)
    hit = np.zeros(n // D, bool)
    for i in (i_nan, i_inf):   # frame m reads samples [m*D + D - M*P, m*D + D - 1]: frames i // D ... (i + M*P - D) // D
        hit[i // D: (i + M * P - D) // D + 1] = True
    finite = np.isfinite(got).all(axis=1)
    assert np.array_equal(~finite, hit), (np.flatnonzero(~finite)[[0, -1]], np.flatnonzero(hit)[[0, -1]])
    assert not np.isfinite(got[hit]).any()
    assert np.array_equal(got[~hit], clean[~hit])
    assert np.array_equal(np.isfinite(gen).all(axis=1), finite)
