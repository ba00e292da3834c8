"""The reference's .iq record format on the host side (numpy in/out).

Mirrors /root/reference/matlab/convert_my_iq_to_mat.m:40-118 (reader: returns the
same variables that script saves -- iq, fs, fc, dur, bw, gain, bitWidth,
sampleStartTime, linkSpeed, boardName, serialNo, fpgaVersion, fwVersion), the
recorders' writer (/root/reference/cpp/blade_record_iq_12bit.cpp:318-323) and
getFilenameStr (/root/reference/cpp/Helper.cpp:6-23).  Header parsing itself is
done by the C library (pfb_iq_parse_header) so the C++ host and Python agree.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L


@dataclass
class IqRecord:
    iq: np.ndarray          # (numSamples, 2) int8 or int16; column 0 = I, column 1 = Q
    fs: float
    fc: float
    bw: float
    gain: float             # as convert_my_iq_to_mat.m reads it (uint32 for fmt < 3)
    bitWidth: int
    sampleStartTime: float
    linkSpeed: int
    boardName: str
    serialNo: str
    fpgaVersion: str
    fwVersion: str
    fileFormat: int

    @property
    def dur(self) -> float:
        return self.iq.shape[0] / self.fs

    @property
    def sample_format(self) -> int:
        return L.PFB_FMT_INT8_IQ if self.iq.dtype == np.int8 else L.PFB_FMT_INT16_IQ


def parse_header(data: bytes) -> L.PfbIqInfo:
    info = L.PfbIqInfo()
    buf = C.create_string_buffer(bytes(data[:128]), 128)
    L.check(L.load().pfb_iq_parse_header(buf, min(len(data), 128), C.byref(info)), "pfb_iq_parse_header")
    return info


def read_iq(path: str) -> IqRecord:
    with open(path, "rb") as f:
        data = f.read()
    info = parse_header(data)
    p = info.packet
    dt = np.int8 if info.bytes_per_sample == 2 else np.int16
    payload = np.frombuffer(data, dtype=dt, offset=info.header_bytes)
    iq = payload[: (payload.size // 2) * 2].reshape(-1, 2)
    if iq.shape[0] != p.numSamples:  # assert(length(iq) == numSamples), convert_my_iq_to_mat.m:102
        raise ValueError(f"{path}: header says {p.numSamples} samples, payload has {iq.shape[0]}")
    s = lambda b: b.split(b"\0", 1)[0].decode("latin1")
    return IqRecord(iq=iq, fs=float(p.sampleRateSps), fc=float(p.frequencyHz), bw=float(p.bandwidthHz),
                    gain=float(info.rx_gain_as_read), bitWidth=int(p.bitWidth),
                    sampleStartTime=float(p.sampleStartTime), linkSpeed=int(p.linkSpeed),
                    boardName=s(p.boardName), serialNo=s(p.serialNumber), fpgaVersion=s(p.fpgaVersion),
                    fwVersion=s(p.fwVersion), fileFormat=int(info.file_format))


def write_iq(path: str, iq: np.ndarray, fs: float, fc: float, bit_width: int, *, bw: float | None = None,
             gain_db: float = 0.0, start_time: float = 0.0, marker: int = 0x03030303,
             board: str = "simulated", serial: str = "") -> None:
    """Write a fmt-2/3 record exactly as the recorders do: header then interleaved payload."""
    iq = np.ascontiguousarray(iq)
    assert iq.ndim == 2 and iq.shape[1] == 2 and iq.dtype in (np.int8, np.int16)
    pk = L.PfbIqPacket()
    L.load().pfb_iq_fill_packet(C.byref(pk), marker, int(fc), int(fs if bw is None else bw), int(fs),
                                float(gain_db), iq.shape[0], bit_width, board.encode(), serial.encode(),
                                float(start_time))
    with open(path, "wb") as f:
        f.write(bytes(pk))
        f.write(iq.tobytes())


def filename_for(epoch_ms: int) -> str:
    buf = C.create_string_buffer(80)
    L.load().pfb_iq_filename(int(epoch_ms), buf, 80)
    return buf.value.decode()


def write_iq_fmt1(path: str, iq: np.ndarray, fs: float, start_time: float = 0.0, board: str = "simulated") -> None:
    """The 104-byte format-1 record matlab/generate_training_iq.m:107-127 writes by hand: marker 0x01010101,
    link speed 1, 32-bit frequency 0, bandwidth = sample rate = fs, integer gain 0, numSamples, bitWidth 16,
    64 bytes of strings ("simulated"), start time; then the int16 payload."""
    iq = np.ascontiguousarray(iq, dtype=np.int16)
    assert iq.ndim == 2 and iq.shape[1] == 2
    words = np.array([0x01010101, 1, 0, int(fs), int(fs), 0, iq.shape[0], 16], dtype="<u4").tobytes()
    name = board.encode()[:64]
    with open(path, "wb") as f:
        f.write(words + name + bytes(64 - len(name)) + np.array([start_time], dtype="<f8").tobytes())
        f.write(iq.tobytes())
