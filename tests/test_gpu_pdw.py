"""GPU PDW extraction (pfb_pdw_extract) against the oracle's restatement of
matlab/create_pdws_channelized.m:64-143, both evaluated on the SAME F x M complex64 matrix.
Integer outcomes (pulse count, column, time of arrival, width, saturation) must be identical;
float fields may differ in the last bits (device vs host hypot/atan2/log10): rtol 1e-9."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from sdr_channelizer_amd import Channelizer, synth  # noqa: E402
from sdr_channelizer_amd.pdw import extract_pdws, extract_pdws_raw  # noqa: E402


def compare(got, want, fs, phase_col=None):
    """phase_col(i) -> the complex samples pulse i's phase steps are taken over: lets the caller's data excuse the one
    ill-conditioned point of the reference algorithm (see antipodal_slack)."""
    assert len(got) == len(want), (len(got), len(want))
    w = {k: np.array([p[k] for p in want]) for k in ("toa", "freq", "pw", "snr", "sat", "bin")}
    assert np.array_equal(got["bin"], w["bin"])
    assert np.array_equal(got["sat"] != 0, w["sat"].astype(bool))
    assert np.allclose(got["toa"], w["toa"], rtol=0, atol=1e-9 / fs + 1e-12 * np.abs(w["toa"]).max(initial=1.0))
    assert np.allclose(got["pw"], w["pw"], rtol=1e-12, atol=0)
    assert np.allclose(got["snr"], w["snr"], rtol=1e-9, atol=1e-9, equal_nan=True)
    bad = ~np.isclose(got["freq"], w["freq"], rtol=1e-9, atol=1e-6, equal_nan=True)
    if phase_col is not None:
        for i in np.flatnonzero(bad):
            bad[i] = not antipodal_slack(phase_col(i), float(got["freq"][i]), float(w["freq"][i]), fs)
    assert not bad.any(), (np.flatnonzero(bad), got["freq"][bad], w["freq"][bad])
    assert np.allclose(got["mag"], np.array([p["mag"] for p in want]), rtol=1e-12, atol=0)


def antipodal_slack(col, got_freq, want_freq, fs):
    """create_pdws_channelized.m:114-117 wraps each phase step at +-180 degrees and takes the median.  Two consecutive
    samples that are exact negative multiples of each other (quantised data has them) step by 180 +- 1 ulp, so the
    last bit of atan2 decides between +180 and -180 there; the device's libm, this host's (the oracle) and numpy's all
    differ in that bit (so would MATLAB's).  Every assignment of +-180 to those steps gives one legitimate median:
    accept a device result whose distance from the oracle's is the distance between two of them, and nothing else."""
    c = np.asarray(col, np.complex128)
    d = np.diff(np.arctan2(c.imag, c.real) * (180.0 / np.pi))
    anti = np.flatnonzero(np.abs(np.abs(d) - 180.0) < 1e-9)
    if len(anti) == 0:
        return False
    d[d < -180.0] += 360.0
    d[d > 180.0] -= 360.0
    delta = 360.0 * (got_freq - want_freq) / fs   # freq = base + fs * med / 360
    tol = 360.0 * (1e-9 * abs(want_freq) + 1e-6) / fs
    if len(anti) > 10:  # too many assignments to list: the median is monotone in every step, so bound it
        lo, hi = d.copy(), d.copy()
        lo[anti], hi[anti] = -180.0, 180.0
        return abs(delta) <= np.median(hi) - np.median(lo) + tol
    meds = []
    for bits in range(1 << len(anti)):
        e = d.copy()
        e[anti] = [180.0 if (bits >> j) & 1 else -180.0 for j in range(len(anti))]
        meds.append(np.median(e))
    meds = np.array(meds)
    return bool((np.abs((meds[:, None] - meds[None, :]) - delta) <= tol).any())


def synthetic_matrix(F=6000, M=16, seed=0):
    rng = np.random.default_rng(seed)
    y = 0.01 * (rng.standard_normal((F, M)) + 1j * rng.standard_normal((F, M)))
    def pulse(b, a, n, amp=0.5, dphi=25.0):
        y[a:a + n, b] += amp * np.exp(1j * np.deg2rad(dphi) * np.arange(n))
    pulse(3, 100, 51)
    pulse(3, 400, 7, dphi=-140.0)        # wraps past +-180 degrees
    pulse(3, 500, 1)                     # single-frame pulse
    pulse(5, 480, 80)                    # crosses the 512-frame tile boundary
    pulse(5, 1000, 1500, amp=0.3)        # longer than the LDS cache, crosses several tiles
    pulse(0, 2000, 40, amp=1.2)          # saturates (|re| or |im| >= 0.9999 inside)
    pulse(M - 1, 0, 30)                  # starts on the very first frame
    pulse(M - 1, F - 20, 20)             # still active at the end of the data: no PDW
    pulse(9, 3000, 64); pulse(9, 3064 + 1, 10)  # one-frame gap between pulses
    return y.astype(np.complex64)


@pytest.mark.parametrize("quirks,unshifted", [(True, False), (False, False), (True, True), (False, True)])
def test_synthetic_pulses_match_oracle(oracle, quirks, unshifted):
    """quirks = the :114 column-1 phase indexing; unshifted = binFreqs as an FFT-ordered list (:42/:80, unpinned)"""
    y = synthetic_matrix()
    fs_in, fc, t0 = 16e6, 2.4e9, 1.7e9
    got, nf = extract_pdws(y, fs_in, fc, t0, matlab_quirks=quirks, binfreq_unshifted=unshifted, return_noise_floor=True)
    want = oracle.extract_pdws(y.astype(np.complex128), fs_in, fc, t0, 15.0, matlab_quirks=quirks,
                               binfreq_unshifted=unshifted)
    compare(got, want, fs_in / y.shape[1])
    mag = np.abs(y.astype(np.complex128))
    assert np.allclose(nf, np.median(mag, axis=0), rtol=1e-12, atol=0)
    assert {int(p) for p in got["bin"]} == {0, 3, 5, 9, y.shape[1] - 1}
    assert got["sat"][got["bin"] == 0].all() and not got["sat"][got["bin"] == 3].any()


def test_channel_major_input_gives_the_same_pdws():
    """PFB_PDW_CHANNEL_MAJOR: MATLAB's own layout of the matrix (M columns of F frames), host and device input."""
    import torch
    y = synthetic_matrix(F=5003, M=70, seed=5)
    want, nf = extract_pdws(y, 7e6, 1e9, 5.0, return_noise_floor=True)
    for ycm in (np.ascontiguousarray(y.T), torch.from_numpy(y).cuda().T.contiguous()):
        got, nf_cm = extract_pdws(ycm, 7e6, 1e9, 5.0, return_noise_floor=True, channel_major=True)
        assert len(got) > 5 and np.array_equal(got, want) and np.array_equal(nf_cm, nf)


def test_odd_frame_count_and_device_input(oracle):
    import torch
    y = synthetic_matrix(F=4097, M=70, seed=3)   # odd F (plain median), M not a multiple of 64
    got = extract_pdws(torch.from_numpy(y).cuda(), 7e6, 1e9, 5.0, matlab_quirks=True)
    want = oracle.extract_pdws(y.astype(np.complex128), 7e6, 1e9, 5.0, 15.0, matlab_quirks=True)
    compare(got, want, 7e6 / 70)


def test_degenerate_all_equal_magnitudes(oracle):
    """every |y| identical: the radix select runs all 8 digit passes and the bucket never shrinks"""
    y = np.full((64, 4), 0.25 + 0.0j, dtype=np.complex64)
    got, nf = extract_pdws(y, 4e6, 0.0, 0.0, return_noise_floor=True)
    want = oracle.extract_pdws(y.astype(np.complex128), 4e6, 0.0, 0.0, 15.0)
    assert np.array_equal(nf, np.full(4, 0.25))
    compare(got, want, 1e6)


def big_matrix(F, M, seed, levels=0):
    rng = np.random.default_rng(seed)
    y = (0.01 * (rng.standard_normal((F, M), dtype=np.float32) + 1j * rng.standard_normal((F, M), dtype=np.float32)))
    if levels:  # magnitudes on a coarse grid: most of a channel's values are tied
        y = (np.round(y.real * levels) / levels + 1j * np.round(y.imag * levels) / levels)
    y = y.astype(np.complex64)
    for b, a, n in ((1, 1000, 90), (1, 300000, 90), (M - 2, 500000, 2000), (M // 2, F - 3000, 100)):
        y[a:a + n, b] += (0.5 * np.exp(1j * np.deg2rad(33.0) * np.arange(n))).astype(np.complex64)
    return y


@pytest.mark.parametrize("F,M,levels,path", [(600000, 24, 0, 1), (600001, 70, 0, 1), (600000, 8, 200, 3), (3000000, 6, 0, 1), (2999999, 3, 0, 1)])
def test_long_streams_take_the_sampled_bracket_and_stay_exact(oracle, F, M, levels, path):
    """F >= 8 * 65536 frames: the noise floor comes from the sampled bracket + one pass (path 1); data too tied
    for the bracket (a coarse amplitude grid) fail its count check and fall back to the full select (path 3).
    Either way the medians are the exact ones."""
    from sdr_channelizer_amd import _lib as L
    y = big_matrix(F, M, seed=F % 97 + M, levels=levels)
    got, nf = extract_pdws(y, 8e6, 1e9, 0.0, return_noise_floor=True)
    assert L.load().pfb_pdw_last_noise_floor_path() == path
    mag = np.abs(y.astype(np.complex128))
    assert np.allclose(nf, np.median(mag, axis=0), rtol=1e-12, atol=0)
    want = oracle.extract_pdws(y.astype(np.complex128), 8e6, 1e9, 0.0, 15.0)
    compare(got, want, 8e6 / M)
    assert len(want) >= 3
    L.load().pfb_pdw_release_workspace(-1)


def test_capacity_overflow_is_reported():
    y = synthetic_matrix()
    with pytest.raises(OverflowError):
        extract_pdws(y, 16e6, 0.0, 0.0, capacity=2)


def test_config5_end_to_end(oracle):
    """BASELINE config 5: 2x oversampled M=128 bank feeding PDW extraction, all on the GPU."""
    import torch
    M, P, D, fs, fc, t0 = 128, 12, 64, 56e6, 915e6, 1.7e9
    n = D * 60000                                  # ~68 ms at 56 Msps: ~68 pulses of 100 us
    iq = synth.pulsed_iq_torch(n, 12, device="cuda")
    h = oracle.design_prototype(M, P).astype(np.float32)
    with Channelizer(M, taps=h, decimation=D, bit_width=12, fftshift=True) as ch:
        y = ch(iq)                                 # (F, M) complex64 on the device
        assert ch.last_kernel.startswith("pfb_fast<M128")
    got = extract_pdws(y, fs, fc, t0, decimation=D, matlab_quirks=True)
    yh = y.cpu().numpy()
    want = oracle.extract_pdws(yh.astype(np.complex128), fs, fc, t0, 15.0, matlab_quirks=True, decim=D)
    assert len(want) >= 60
    compare(got, want, fs / D)
    # the pulse train is 100 us wide every 1 ms: the carrier's channel sees ~68 pulses of ~100 us
    # (the many short detections are the rectangular pulses' edge transients in the other channels)
    assert np.count_nonzero(np.abs(got["pw"] - 100e-6) < 15e-6) >= 60
    assert bool(torch.isfinite(torch.view_as_real(y)).all())


def test_record_to_pdws_in_one_call(oracle, tmp_path):
    """pfb_pdw_from_iq_file = one iteration of create_pdws_channelized.m:22-143 (record -> channelizer -> fftshift ->
    PDWs) with the channel matrix left on the GPU: the same PDWs, bit for bit, as the two-step path (record -> host
    matrix -> pfb_pdw_extract), and the oracle's; fs / fc / start time come from the record's header."""
    import os
    from sdr_channelizer_amd import iqfile, PfbError
    from sdr_channelizer_amd import _lib as L
    from sdr_channelizer_amd.pdw import pdws_from_iq_file
    M, P, fs, fc, t0 = 56, 12, 56_000_000, 915_000_000, 1.7e9 + 0.25
    n = M * 400_000 + 17                            # 22.4 M samples: two file chunks, several staging steps each
    iq = synth.pulsed_iq_numpy(n, 12, np.int16, seed=12)
    path = os.path.join(tmp_path, iqfile.filename_for(1_700_000_000_250))
    iqfile.write_iq(path, iq, fs=fs, fc=fc, bit_width=12, start_time=t0)
    h = oracle.design_prototype(M, P).astype(np.float32)
    with Channelizer(M, taps=h, bit_width=12, fftshift=True) as ch:
        got, nf, info = pdws_from_iq_file(ch, path, return_noise_floor=True)
        assert info.packet.sampleRateSps == fs and info.packet.sampleStartTime == t0
        y, _ = ch.process_iq_file(path)
        two_step, nf2 = extract_pdws(y, fs, fc, t0, matlab_quirks=True, return_noise_floor=True)
        assert len(got) >= 300 and np.array_equal(got, two_step) and np.array_equal(nf, nf2)
        again, _ = pdws_from_iq_file(ch, path)      # scratch reused, state reset per file
        assert np.array_equal(again, got)
        want = oracle.extract_pdws(y.astype(np.complex128), fs, fc, t0, 15.0, matlab_quirks=True, max_out=1 << 18)
        compare(got, want, fs / M)
    with Channelizer(M, taps=h, bit_width=12, magnitude=True) as chm:      # needs the complex matrix
        with pytest.raises(PfbError) as e:
            pdws_from_iq_file(chm, path)
        assert e.value.status == L.PFB_ERR_UNSUPPORTED
    with Channelizer(M, taps=h, bit_width=16) as ch16:                     # record says 12-bit
        with pytest.raises(PfbError) as e:
            pdws_from_iq_file(ch16, path)
        assert e.value.status == L.PFB_ERR_BAD_FORMAT


def test_folder_of_records_like_the_scripts_loop(tmp_path):
    """sharded.pdws_from_folder on one rank = create_pdws_channelized.m's loop: per-record PDWs concatenated in file
    order (what the script accumulates); a two-way split of the same folder holds the same PDWs."""
    import os
    from sdr_channelizer_amd import iqfile
    from sdr_channelizer_amd.pdw import pdws_from_iq_file
    from sdr_channelizer_amd.sharded import pdws_from_folder
    M, fs, fc = 56, 56_000_000, 915_000_000
    paths = []
    for i in range(3):
        iq = synth.pulsed_iq_numpy(M * 30_000, 12, np.int16, seed=50 + i)
        paths.append(os.path.join(tmp_path, iqfile.filename_for(1_700_000_000_000 + 1000 * i)))
        iqfile.write_iq(paths[-1], iq, fs=fs, fc=fc, bit_width=12, start_time=1.7e9 + i)
    with Channelizer(M, bit_width=12, fftshift=True) as ch:
        allp, listing = pdws_from_folder(ch, paths[::-1])
        each = [pdws_from_iq_file(ch, p)[0] for p in paths]
        assert [p for p, _ in listing] == paths and [c for _, c in listing] == [len(e) for e in each]
        assert len(allp) > 30 and np.array_equal(allp, np.concatenate(each))
        halves = [pdws_from_folder(ch, paths, r, 2, gather=False)[0] for r in range(2)]
        assert sum(len(h_) for h_ in halves) == len(allp)
        assert np.array_equal(halves[0], np.concatenate([each[0], each[2]])) and np.array_equal(halves[1], each[1])


# ---- raw stream (matlab/create_pdws.m) ------------------------------------------------------------------

def raw_stream(n, dtype, bw, seed, cf32=False, noise=0.004):
    """noise + rectangular pulses, quantised like the recorders' payload; returns the (n, 2) integer array (or the
    complex64 vector) and the normalised complex128 samples the script would see"""
    rng = np.random.default_rng(seed)
    x = noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    def pulse(a, m, amp=0.4, dphi=21.0):
        x[a:a + m] += amp * np.exp(1j * np.deg2rad(dphi) * np.arange(m))
    pulse(300, 700)
    pulse(5000, 40, dphi=-150.0)        # wraps past +-180 degrees
    pulse(9000, 3000, amp=0.35)         # longer than the LDS cache, many tiles
    pulse(20000, 60, amp=1.3)           # clips: saturated
    pulse(30000 + 33, 64 * 9)           # word- and tile-crossing
    k = np.arange(400)                   # dips to a tenth mid-pulse: below the leading, above the 3 dB trailing threshold
    x[40000:40400] = 0.4 * np.exp(1j * 0.3 * k) * np.where((k > 180) & (k < 220), 0.1, 1.0)
    pulse(n - 50, 50)                   # still active at the end: no PDW
    if cf32:
        v = x.astype(np.complex64)
        return v, v.astype(np.complex128)
    full = 2 ** (bw - 1)
    iq = np.stack([np.clip(np.round(x.real * full), -full, full - 1), np.clip(np.round(x.imag * full), -full, full - 1)],
                  axis=1).astype(dtype)
    return iq, (iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)) / full


@pytest.mark.parametrize("dtype,bw,n,noise,snr_db", [(np.int16, 12, 60000, 0.004, 18.0), (np.int16, 16, 65537, 0.004, 18.0),
                                                     (np.int8, 8, 70001, 0.012, 12.0)])
def test_raw_stream_pdws_match_oracle(oracle, dtype, bw, n, noise, snr_db):
    """int8 has too little dynamic range for the script's 18 dB leading threshold (1.5 LSB of noise puts it at
    ~100 LSB), so that case runs at 12 dB; its magnitudes are heavily tied, which the integer-key select handles"""
    iq, x = raw_stream(n, dtype, bw, seed=bw + n % 7, noise=noise)
    fs, fc, t0 = 56e6, 915e6, 1.7e9
    got, nf = extract_pdws_raw(iq, fs, fc, t0, bit_width=bw, snr_threshold_db=snr_db, return_noise_floor=True)
    want, want_nf = oracle.extract_pdws_raw(x, fs, fc, t0, snr_db=snr_db)
    assert nf == pytest.approx(want_nf, rel=1e-14) and nf == pytest.approx(np.median(np.abs(x)), rel=1e-14)
    assert len(want) >= 6
    compare(got, want, fs)
    assert (got["bin"] == 0).all()
    assert got["sat"].any() and not got["sat"].all()
    # the dip inside the 400-sample pulse stays above the trailing threshold: one PDW, not two
    assert np.count_nonzero(np.abs(got["pw"] - 399 / fs) < 3 / fs) == 1


def test_raw_stream_cf32_device_input_and_custom_thresholds(oracle):
    import torch
    v, x = raw_stream(50001, None, 0, seed=4, cf32=True)
    got = extract_pdws_raw(torch.from_numpy(v).cuda(), 8e6, 0.0, 10.0, snr_threshold_db=12.0, trailing_threshold_db=6.0)
    want, _ = oracle.extract_pdws_raw(x, 8e6, 0.0, 10.0, snr_db=12.0, trail_db=6.0)
    assert len(want) >= 6
    compare(got, want, 8e6)


def test_raw_stream_misaligned_device_buffer(oracle):
    """a device pointer that is not 16-byte aligned takes the narrow loads; same PDWs"""
    import torch
    iq, x = raw_stream(60000, np.int16, 12, seed=11)
    d = torch.from_numpy(iq).cuda()
    got = extract_pdws_raw(d[3:], 56e6, 0.0, 0.0)
    want, _ = oracle.extract_pdws_raw(x[3:], 56e6, 0.0, 0.0)
    assert len(want) >= 6
    compare(got, want, 56e6)


def test_raw_stream_at_recorder_size(oracle):
    """2^24 samples on the device (a 0.3 s dwell at 56 Msps): pulse train of synth.pulsed_iq_torch.  Its pulses stand
    14 dB (the script's dB/10 convention) above the noise floor, so the leading threshold is set to 12 dB."""
    import torch
    n = 1 << 24
    iq = synth.pulsed_iq_torch(n, 12, device="cuda")
    got, nf = extract_pdws_raw(iq, 56e6, 915e6, 0.0, snr_threshold_db=12.0, return_noise_floor=True)
    h = iq.cpu().numpy()
    x = (h[:, 0].astype(np.float64) + 1j * h[:, 1].astype(np.float64)) / 2048.0
    want, want_nf = oracle.extract_pdws_raw(x, 56e6, 915e6, 0.0, snr_db=12.0, max_out=1 << 18)
    assert nf == pytest.approx(want_nf, rel=1e-14)
    assert len(want) >= 250           # 1 ms PRI over 0.3 s
    compare(got, want, 56e6)


def test_raw_stream_long_tiles(oracle):
    """2^26 samples and a ragged tail: the edge scan's tiles are 64 words long, which the one-column stream hands to a
    wave per tile (pdw_tilefn_wave_kernel / pdw_edges_wave_kernel); a threshold low enough for tens of thousands of
    detections spread over every tile."""
    n = (1 << 26) + 12345
    iq = synth.pulsed_iq_torch(n, 12, device="cuda")
    got, nf = extract_pdws_raw(iq, 56e6, 915e6, 0.0, snr_threshold_db=4.0, trailing_threshold_db=2.0, return_noise_floor=True)
    h = iq.cpu().numpy()
    x = (h[:, 0].astype(np.float64) + 1j * h[:, 1].astype(np.float64)) / 2048.0
    want, want_nf = oracle.extract_pdws_raw(x, 56e6, 915e6, 0.0, snr_db=4.0, trail_db=2.0, max_out=1 << 20)
    assert nf == pytest.approx(want_nf, rel=1e-14)
    assert len(want) >= 10000
    compare(got, want, 56e6)


def _sample_row(q, stride):
    """the library's hashed sample position (pfb_pdw.hip: sample_row)"""
    h = (q * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    h ^= h >> 29
    return q * stride + (((h >> 40) * stride) >> 24)


@pytest.mark.parametrize("fmt", ["int16", "cf32"])
def test_raw_median_prediction_is_checked_not_trusted(oracle, fmt):
    """Streams of >= 2^22 samples start their radix select below the digits a 4096-sample bracket shares.  Here the
    sampled positions (the hash is deterministic) hold strong samples and everything else weak noise, so the prediction
    is wrong in every leading digit: the counting pass must notice (the rank falls outside the predicted bucket) and the
    select must start over -- same median, same PDWs as the oracle.  A second stream, ordinary, takes the shortcut."""
    n = 1 << 22
    rng = np.random.default_rng(3)
    pos = np.array([_sample_row(q, n // 4096) for q in range(4096)])
    for adversarial in (True, False):
        if fmt == "int16":
            iq = np.round(rng.standard_normal((n, 2)) * 12).astype(np.int16)
            iq[200000:203000] += 900
            if adversarial:
                iq[pos] = 1500
            x = (iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)) / 2048.0
        else:
            iq = (rng.standard_normal((n, 2)) * 0.006).astype(np.float32)
            iq[200000:203000] += 0.45
            if adversarial:
                iq[pos] = 0.7
            x = iq[:, 0].astype(np.float64) + 1j * iq[:, 1].astype(np.float64)
            iq = (iq[:, 0] + 1j * iq[:, 1]).astype(np.complex64)
        got, nf = extract_pdws_raw(iq, 56e6, 915e6, 0.0, return_noise_floor=True)
        want, want_nf = oracle.extract_pdws_raw(x, 56e6, 915e6, 0.0, max_out=1 << 18)
        assert nf == pytest.approx(want_nf, rel=1e-14), (fmt, adversarial)
        assert len(want) >= 1
        compare(got, want, 56e6)


def test_raw_record_to_pdws_in_one_call(tmp_path):
    """pfb_pdw_raw_from_iq_file = one iteration of create_pdws.m's loop: the same PDWs as extract_pdws_raw on the
    record's payload with the header's fs / fc / bit width / start time; int16 and int8 records."""
    import os
    from sdr_channelizer_amd import iqfile, PfbError
    from sdr_channelizer_amd import _lib as L
    from sdr_channelizer_amd.pdw import raw_pdws_from_iq_file
    for dt, bw, n, noise, snr in ((np.int16, 12, 40_000_003, 0.004, 18.0), (np.int8, 8, 3_000_001, 0.012, 12.0)):   # 160 MB: three 64 MB chunks
        iq, _ = raw_stream(n, dt, bw, seed=bw, noise=noise)
        path = os.path.join(tmp_path, iqfile.filename_for(1_700_000_000_000 + bw))
        iqfile.write_iq(path, iq, fs=20e6, fc=2.4e9, bit_width=bw, start_time=1.7e9 + 0.5)
        got, nf, info = raw_pdws_from_iq_file(path, snr_threshold_db=snr, return_noise_floor=True)
        want, want_nf = extract_pdws_raw(iq, 20e6, 2.4e9, 1.7e9 + 0.5, bit_width=bw, snr_threshold_db=snr,
                                         return_noise_floor=True)
        assert info.packet.numSamples == n and len(got) >= 5
        assert np.array_equal(got, want) and nf == want_nf
    with open(path, "r+b") as f:
        f.truncate(os.path.getsize(path) - 2)
    with pytest.raises(PfbError) as e:
        raw_pdws_from_iq_file(path)
    assert e.value.status == L.PFB_ERR_BAD_FORMAT


def test_raw_stream_argument_checks_and_silence(oracle):
    from sdr_channelizer_amd import _lib as L
    iq = np.zeros((1000, 2), np.int16)
    with pytest.raises(L.PfbError):
        extract_pdws_raw(iq, 1e6, 0.0, 0.0, snr_threshold_db=3.0, trailing_threshold_db=18.0)
    with pytest.raises(L.PfbError):
        extract_pdws_raw(iq, 1e6, 0.0, 0.0, bit_width=17)
    # all-zero stream: noise floor 0, both thresholds 0, so the script toggles on every sample (0 >= 0 starts a
    # pulse, 0 <= 0 ends it): 500 two-sample "pulses" with NaN snr -- reproduced, not patched
    got = extract_pdws_raw(iq, 1e6, 0.0, 0.0)
    want, _ = oracle.extract_pdws_raw(np.zeros(1000, np.complex128), 1e6, 0.0, 0.0)
    assert len(want) == 500
    compare(got, want, 1e6)


def test_extraction_past_2_31_matrix_elements():
    """64-bit indexing through the whole PDW pipeline: a 2^27 x 20 matrix (2.7e9 elements, 21 GB) built as 128 copies of
    a 2^20 x 20 one.  The multiset of magnitudes is the small one's with every value 128 times, so the medians are
    identical, and with quiet margins at both ends of the small matrix the pulses of the big one are the small one's,
    128 times, shifted by whole copies."""
    import torch
    Fs, M, reps = 1 << 20, 20, 128
    need = reps * Fs * M * 8
    if torch.cuda.mem_get_info()[0] < need * 1.5:
        pytest.skip(f"needs {need >> 30} GiB of HBM")
    rng = np.random.default_rng(5)
    y = (0.01 * (rng.standard_normal((Fs, M), dtype=np.float32) + 1j * rng.standard_normal((Fs, M), dtype=np.float32))).astype(np.complex64)
    for _ in range(40):
        b, a, n = int(rng.integers(0, M)), int(rng.integers(1000, Fs - 6000)), int(rng.integers(1, 3000))
        y[a:a + n, b] += (rng.uniform(0.3, 0.9) * np.exp(1j * np.deg2rad(rng.uniform(-170, 170)) * np.arange(n))).astype(np.complex64)
    fs_in, fc = 20e6, 1e9
    small, nf_small = extract_pdws(y, fs_in, fc, 0.0, return_noise_floor=True)
    assert len(small) >= 20
    yd = torch.from_numpy(y).cuda()
    big_y = yd.repeat(reps, 1)
    del yd
    big, nf_big = extract_pdws(big_y, fs_in, fc, 0.0, return_noise_floor=True, capacity=1 << 18)
    del big_y
    torch.cuda.empty_cache()
    assert np.array_equal(nf_big, nf_small)
    assert len(big) == reps * len(small)
    # the reference's order: channels outermost, time within a channel
    order = np.lexsort((big["toa"], big["bin"]))
    assert np.array_equal(order, np.arange(len(big)))
    fs = fs_in / M
    for c in np.unique(small["bin"]):
        s, bg = small[small["bin"] == c], big[big["bin"] == c]
        want_toa = (s["toa"][None, :] + (np.arange(reps) * Fs / fs)[:, None]).ravel()
        assert np.allclose(bg["toa"], want_toa, rtol=0, atol=1e-9)
        for f in ("pw", "snr", "mag", "freq"):
            assert np.array_equal(bg[f], np.tile(s[f], reps)), f
        assert np.array_equal(bg["sat"], np.tile(s["sat"], reps))
    from sdr_channelizer_amd import _lib as L
    L.load().pfb_pdw_release_workspace(-1)
