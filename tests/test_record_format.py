"""The .iq record format: our restatements (include/pfb_iq_packet.h via libpfb_channelizer.so,
oracle/pfb_oracle.c, sdr_channelizer_amd/iqfile.py) pinned against the reference's own
cpp/IqPacket.h + cpp/Helper.cpp compiled into oracle/_ref."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle.pfb_oracle import RefIqPacket
from sdr_channelizer_amd import _lib as L
from sdr_channelizer_amd import iqfile

needs_ref = pytest.mark.skipif(not RefIqPacket.available(), reason="oracle/_ref not built")

EXPECTED_OFFSETS = dict(endianness=0, linkSpeed=4, frequencyHz=8, bandwidthHz=16, sampleRateSps=20, rxGainDb=24,
                        numSamples=28, bitWidth=32, spare0=36, boardName=40, serialNumber=56, fpgaVersion=72,
                        fwVersion=88, sampleStartTime=104)


def test_our_struct_layout():
    assert C.sizeof(L.PfbIqPacket) == 112
    for name, off in EXPECTED_OFFSETS.items():
        assert getattr(L.PfbIqPacket, name).offset == off, name


@needs_ref
def test_layout_matches_reference_build():
    ref = RefIqPacket()
    assert ref.sizeof() == C.sizeof(L.PfbIqPacket) == 112
    assert ref.offsets() == EXPECTED_OFFSETS


@needs_ref
@pytest.mark.parametrize("marker,bw", [(0x02020202, 12), (0x03030303, 16), (0x03030303, 8), (0x02020202, 8)])
def test_parse_reference_written_header(oracle, marker, bw):
    ref = RefIqPacket()
    raw = ref.make_header(marker, 5000, 2_412_000_000, 56_000_000, 56_000_000, 31.5, 123456, bw,
                          b"bladerf", b"0123456789abcde", b"v0.15.0", b"v2.4.0", 1.7e9 + 0.25)
    assert len(raw) == 112
    info = iqfile.parse_header(raw)
    p = info.packet
    assert (p.endianness, p.linkSpeed, p.frequencyHz, p.bandwidthHz, p.sampleRateSps) == \
        (marker, 5000, 2_412_000_000, 56_000_000, 56_000_000)
    assert (p.numSamples, p.bitWidth, p.sampleStartTime) == (123456, bw, 1.7e9 + 0.25)
    assert p.boardName == b"bladerf" and p.serialNumber == b"0123456789abcde"
    assert info.header_bytes == 112 and info.file_format == (2 if marker == 0x02020202 else 3)
    assert info.bytes_per_sample == (2 if bw <= 8 else 4)
    # convert_my_iq_to_mat.m:73-77 reads the gain word as uint32 unless fmt >= 3 (bladeRF quirk kept)
    want_gain = 31.5 if marker == 0x03030303 else float(np.float32(31.5).view(np.uint32))
    assert info.rx_gain_as_read == want_gain
    o = oracle.parse_iq_header(raw)  # the oracle's own parser agrees field by field
    assert o["frequency_hz"] == p.frequencyHz and o["num_samples"] == p.numSamples
    assert o["rx_gain_db"] == want_gain and o["header_bytes"] == 112 and o["bytes_per_sample"] == info.bytes_per_sample


def test_parse_fmt1_header_as_generate_training_iq_writes_it(oracle):
    """matlab/generate_training_iq.m:109-123: 8 x uint32, 64 bytes of strings, one double = 104 bytes."""
    words = np.array([0x01010101, 1, 0, 56_000_000, 56_000_000, 0, 5_600_000, 16], dtype="<u4").tobytes()
    strings = b"simulated" + bytes(64 - 9)
    raw = words + strings + np.array([1.6e9], dtype="<f8").tobytes()
    assert len(raw) == 104
    info = iqfile.parse_header(raw)
    assert info.file_format == 1 and info.header_bytes == 104 and info.bytes_per_sample == 4
    assert info.packet.sampleRateSps == 56_000_000 and info.packet.numSamples == 5_600_000
    assert info.packet.bitWidth == 16 and info.packet.boardName == b"simulated"
    assert info.packet.sampleStartTime == 1.6e9
    assert oracle.parse_iq_header(raw)["header_bytes"] == 104


def test_bad_headers_are_rejected():
    lib = L.load()
    info = L.PfbIqInfo()
    bad = np.array([0xdeadbeef] + [0] * 27, dtype="<u4").tobytes()
    assert lib.pfb_iq_parse_header(bad, len(bad), C.byref(info)) == L.PFB_ERR_BAD_FORMAT
    all_zero = bytes(112)  # marker 0 is accepted as format 2 (below), but bitWidth 0 is not a sample format (:96-98)
    assert lib.pfb_iq_parse_header(all_zero, 112, C.byref(info)) == L.PFB_ERR_BAD_FORMAT
    short = np.array([0x03030303], dtype="<u4").tobytes() + bytes(20)
    assert lib.pfb_iq_parse_header(short, len(short), C.byref(info)) == L.PFB_ERR_BAD_ARG
    hdr = bytearray(112)
    hdr[0:4] = (0x03030303).to_bytes(4, "little")
    hdr[32:36] = (24).to_bytes(4, "little")  # bitWidth 24: "Unsupported bit width" (:96-98)
    assert lib.pfb_iq_parse_header(bytes(hdr), 112, C.byref(info)) == L.PFB_ERR_BAD_FORMAT


def test_marker_zero_reads_as_format_2_in_native_byte_order():
    """convert_my_iq_to_mat.m:42-45: marker 0x00000000 prints "big endian", sets fileFormat = 2 -- and keeps reading with
    the byte order the file was opened with (native), so the fields are those of a little-endian format-2 header."""
    hdr = bytearray(112)
    hdr[4:8] = (5000).to_bytes(4, "little")
    hdr[8:16] = (2_412_000_000).to_bytes(8, "little")
    hdr[20:24] = (56_000_000).to_bytes(4, "little")
    hdr[28:32] = (1234).to_bytes(4, "little")
    hdr[32:36] = (12).to_bytes(4, "little")
    info = iqfile.parse_header(bytes(hdr))
    assert info.file_format == 2 and info.header_bytes == 112 and info.bytes_per_sample == 4
    assert (info.packet.frequencyHz, info.packet.sampleRateSps, info.packet.numSamples, info.packet.bitWidth) == \
        (2_412_000_000, 56_000_000, 1234, 12)


@needs_ref
@pytest.mark.parametrize("ms", [0, 1_700_000_000_123, 951_782_400_000, 1_709_251_199_999])
def test_filename_matches_reference_helper(ms):
    assert iqfile.filename_for(ms) == RefIqPacket().filename(ms)


def test_record_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    for dt, bw in ((np.int16, 12), (np.int8, 8), (np.int16, 16)):
        iq = rng.integers(-100, 100, size=(1000, 2)).astype(dt)
        path = os.path.join(tmp_path, f"r{bw}.iq")
        iqfile.write_iq(path, iq, fs=56e6, fc=2.4e9, bit_width=bw, gain_db=30.0, start_time=123.5)
        r = iqfile.read_iq(path)
        assert np.array_equal(r.iq, iq) and r.iq.dtype == dt
        assert (r.fs, r.fc, r.bitWidth, r.sampleStartTime, r.gain, r.fileFormat) == (56e6, 2.4e9, bw, 123.5, 30.0, 3)
        assert os.path.getsize(path) == 112 + iq.nbytes  # header then (n)*sizeof(complex<intX>) (:321-322)


@needs_ref
def test_our_writer_emits_the_reference_bytes(tmp_path):
    ref = RefIqPacket().make_header(0x03030303, 0, 915_000_000, 20_000_000, 20_000_000, 12.0, 10, 16,
                                    b"usrp", b"s1", b"", b"", 99.0)
    path = os.path.join(tmp_path, "w.iq")
    iqfile.write_iq(path, np.zeros((10, 2), np.int16), fs=20e6, fc=915e6, bit_width=16, gain_db=12.0,
                    start_time=99.0, board="usrp", serial="s1")
    assert open(path, "rb").read()[:112] == ref


def test_fmt1_writer_matches_generate_training_iq_layout(tmp_path):
    iq = np.arange(40, dtype=np.int16).reshape(-1, 2)
    path = os.path.join(tmp_path, "sim.iq")
    iqfile.write_iq_fmt1(path, iq, fs=56e6, start_time=1.6e9)
    raw = open(path, "rb").read()
    assert len(raw) == 104 + iq.nbytes
    r = iqfile.read_iq(path)
    assert (r.fileFormat, r.fs, r.bw, r.fc, r.bitWidth, r.boardName, r.sampleStartTime) == \
        (1, 56e6, 56e6, 0.0, 16, "simulated", 1.6e9)
    assert np.array_equal(r.iq, iq)


def test_pulse_generator_switches():
    """generate_pulsed_iq.m:17-19,43-59: LFM sweep and Barker-13 phase code."""
    from sdr_channelizer_amd import synth
    fs, pw = 56e6, 5600
    k = np.arange(pw)
    plain = synth.pulse_phase(k, 1e6, fs, pw)
    assert np.allclose(np.diff(plain), 2 * np.pi * 1e6 / fs)                     # constant carrier
    lfm = synth.pulse_phase(k, 1e6, fs, pw, lfm_extent_hz=2e6)
    inst = np.diff(lfm) * fs / (2 * np.pi)                                       # instantaneous frequency
    assert abs(inst[0] - 1e6) < 1e3 and abs(inst[-1] - 3e6) < 1e3 and np.all(np.diff(inst) > 0)
    # reference behaviour (default): generate_pulsed_iq.m:49-59 adds +-90 to a phase held in RADIANS (it goes straight
    # into exp(1j*my_phi), :62), so the chip states are 180 rad = 233.5 degrees (mod 360) apart
    bk = synth.pulse_phase(k, 0.0, fs, 13 * 431, barker13=True)
    chips = bk[::431][:13]
    assert np.allclose(np.sign(chips), synth.BARKER_13) and np.allclose(np.abs(chips), 90.0, atol=1e-9)
    step = np.rad2deg(np.angle(np.exp(1j * (chips[5] - chips[4]))))              # a +1 -> -1 chip boundary
    assert abs(abs(step) - (360.0 - np.rad2deg(180.0) % 360.0)) < 1e-6 or abs(abs(step) - np.rad2deg(180.0) % 360.0) < 1e-6
    # the textbook code, behind the switch: +-90 degrees, chips 180 degrees apart
    fixed = np.rad2deg(synth.pulse_phase(k, 0.0, fs, 13 * 431, barker13=True, matlab_quirks=False)[::431][:13])
    assert np.allclose(np.sign(fixed), synth.BARKER_13) and np.allclose(np.abs(fixed), 90.0, atol=1e-6)
    a = synth.pulsed_iq_numpy(60000, 12, np.int16, lfm_extent_hz=5e6, barker13=True)
    assert a.shape == (60000, 2) and np.abs(a).max() <= 2048


def test_counter_based_stream_is_identical_on_numpy_and_torch():
    """sample index -> value, so the host twin and the device generator agree bit for bit and any
    shard can be generated on its own.  (torch on the CPU device here; tests/test_gpu_parity.py repeats it on the GPU.)"""
    import torch
    from sdr_channelizer_amd import synth
    for bw, dt, tdt in ((12, np.int16, torch.int16), (8, np.int8, torch.int8), (16, np.int16, torch.int16)):
        for start, n in ((0, 120_000), ((1 << 33) + 12345, 60_000), (55_990, 200)):
            a = synth.pulsed_iq_counter_numpy(n, bw, dt, start=start)
            b = synth.pulsed_iq_torch(n, bw, tdt, device="cpu", start=start, chunk=50_001).numpy()
            assert np.array_equal(a, b), (bw, start)
    whole = synth.pulsed_iq_counter_numpy(3000, 12, np.int16)
    assert np.array_equal(synth.pulsed_iq_counter_numpy(1000, 12, np.int16, start=700), whole[700:1700])
    x = synth.pulsed_iq_counter_numpy(1 << 18, 12, np.int16).astype(np.float64)
    on = (np.arange(1 << 18) % 56000) < 5600
    assert abs(np.hypot(x[on, 0], x[on, 1]).mean() - 1024) < 5           # 0.5 full scale
    assert abs(x[~on].std() - 32.0) < 0.5 and abs(x[~on].mean()) < 0.5    # sigma = 2^-6 full scale, zero mean
