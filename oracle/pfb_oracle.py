"""CPU ORACLE bindings + numpy twin (test infrastructure, NOT the product).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  See ``oracle/pfb_oracle.h`` for the reference
citations and the "parity unpinned" statement (the filterbank arithmetic of
/root/reference lives in MathWorks' closed ``dsp.Channelizer``; what is pinned
is the record format, via ``oracle/_ref``).

Two things live here:

* ``COracle`` -- ctypes view of ``libpfb_oracle.so`` (the plain-C float64
  restatement, formulations A / B / B').
* ``channelize_numpy`` -- an independent vectorised numpy formulation
  (frame-matrix FIR + ``numpy.fft.ifft``) used to generate the committed golden
  fixtures and to cross-check the C code.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libpfb_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libref_iqpacket.so")


def build(force: bool = False) -> None:
    """Compile the C oracle (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(_LIB) or (
        os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "pfb_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "--no-print-directory"])


class _Cfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("M", "P", "D", "off", "conj_input", "derotate", "fftshift")]


class _IqHeader(C.Structure):
    _fields_ = [
        ("marker", C.c_uint32), ("file_format", C.c_int), ("link_speed", C.c_uint32),
        ("frequency_hz", C.c_uint64), ("bandwidth_hz", C.c_uint32), ("sample_rate_sps", C.c_uint32),
        ("rx_gain_db", C.c_double), ("num_samples", C.c_uint32), ("bit_width", C.c_uint32),
        ("spare0", C.c_uint32), ("board_name", C.c_char * 17), ("serial_number", C.c_char * 17),
        ("fpga_version", C.c_char * 17), ("fw_version", C.c_char * 17),
        ("sample_start_time", C.c_double), ("header_bytes", C.c_uint32),
        ("bytes_per_sample", C.c_uint32),
    ]


class _Pdw(C.Structure):
    _fields_ = [("toa", C.c_double), ("freq", C.c_double), ("pw", C.c_double),
                ("snr", C.c_double), ("sat", C.c_int), ("bin", C.c_int), ("mag", C.c_double)]


@dataclass
class OracleConfig:
    M: int
    P: int
    D: int | None = None
    off: int = -1
    conj_input: bool = False
    derotate: bool = False
    fftshift: bool = False

    def c(self) -> _Cfg:
        D = self.M if self.D is None else self.D
        return _Cfg(self.M, self.P, D, self.off, int(self.conj_input), int(self.derotate), int(self.fftshift))

    @property
    def decim(self) -> int:
        return self.M if self.D is None else self.D

    @property
    def offset(self) -> int:
        return self.decim - 1 if self.off < 0 else self.off


_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


class COracle:
    def __init__(self, path: str | None = None):
        if path is None:
            build()
            path = _LIB
        self.lib = lib = C.CDLL(path)
        for name in ("pfbo_channelize_direct", "pfbo_channelize_polyphase", "pfbo_channelize_polyphase_fft"):
            f = getattr(lib, name)
            f.restype = C.c_size_t
            f.argtypes = [_dp, _dp, C.c_size_t, _dp, C.POINTER(_Cfg), _dp, _dp]
        lib.pfbo_unpack_int16.argtypes = [C.c_void_p, C.c_size_t, C.c_int, _dp, _dp]
        lib.pfbo_unpack_int8.argtypes = [C.c_void_p, C.c_size_t, C.c_int, _dp, _dp]
        lib.pfbo_center_frequencies.argtypes = [C.c_int, C.c_double, _dp]
        lib.pfbo_design_prototype.argtypes = [C.c_int, C.c_int, C.c_double, _dp]
        lib.pfbo_channelize_f32_i16.restype = C.c_size_t
        lib.pfbo_channelize_f32_i16.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_int,
                                                C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        lib.pfbo_max_threads.restype = C.c_int
        lib.pfbo_parse_iq_header.restype = C.c_int
        lib.pfbo_parse_iq_header.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_IqHeader)]
        lib.pfbo_extract_pdws.restype = C.c_size_t
        lib.pfbo_extract_pdws.argtypes = [_dp, _dp, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                          C.c_double, C.c_int, C.POINTER(_Pdw), C.c_size_t]
        lib.pfbo_extract_pdws_raw.restype = C.c_size_t
        lib.pfbo_extract_pdws_raw.argtypes = [_dp, _dp, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double,
                                              C.c_double, C.POINTER(_Pdw), C.c_size_t, C.POINTER(C.c_double)]

    # -- a3 -------------------------------------------------------------------
    def unpack(self, iq: np.ndarray, bit_width: int) -> np.ndarray:
        """iq: int8/int16 array of shape (N, 2) or (2N,) interleaved I,Q -> complex128."""
        flat = np.ascontiguousarray(iq).reshape(-1)
        n = flat.size // 2
        xr = np.empty(n)
        xi = np.empty(n)
        if flat.dtype == np.int16:
            self.lib.pfbo_unpack_int16(flat.ctypes.data, n, bit_width, xr, xi)
        elif flat.dtype == np.int8:
            self.lib.pfbo_unpack_int8(flat.ctypes.data, n, bit_width, xr, xi)
        else:
            raise TypeError(flat.dtype)
        return xr + 1j * xi

    # -- a8 -------------------------------------------------------------------
    def channelize(self, x: np.ndarray, h: np.ndarray, cfg: OracleConfig, method: str = "polyphase") -> np.ndarray:
        x = np.asarray(x, dtype=np.complex128)
        xr = np.ascontiguousarray(x.real)
        xi = np.ascontiguousarray(x.imag)
        h = np.ascontiguousarray(h, dtype=np.float64)
        assert h.size == cfg.M * cfg.P
        F = x.size // cfg.decim
        yr = np.zeros(F * cfg.M)
        yi = np.zeros(F * cfg.M)
        fn = {"direct": self.lib.pfbo_channelize_direct,
              "polyphase": self.lib.pfbo_channelize_polyphase,
              "fft": self.lib.pfbo_channelize_polyphase_fft}[method]
        c = cfg.c()
        got = fn(xr, xi, x.size, h, C.byref(c), yr, yi)
        assert got == F, (got, F)
        return (yr + 1j * yi).reshape(F, cfg.M)

    def center_frequencies(self, M: int, fs: float) -> np.ndarray:
        out = np.empty(M)
        self.lib.pfbo_center_frequencies(M, fs, out)
        return out

    def design_prototype(self, M: int, P: int, atten_db: float = 80.0) -> np.ndarray:
        h = np.empty(M * P)
        self.lib.pfbo_design_prototype(M, P, atten_db, h)
        return h

    def channelize_f32_i16(self, iq: np.ndarray, bit_width: int, h: np.ndarray, M: int, P: int,
                           D: int | None = None, off: int = -1, threads: int = 0) -> np.ndarray:
        flat = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1)
        n = flat.size // 2
        D = M if D is None else D
        F = n // D
        h32 = np.ascontiguousarray(h, dtype=np.float32)
        out = np.empty((F, M), dtype=np.complex64)
        got = self.lib.pfbo_channelize_f32_i16(flat.ctypes.data, n, bit_width, h32.ctypes.data, M, P, D, off,
                                               out.ctypes.data, threads)
        assert got == F
        return out

    def max_threads(self) -> int:
        return int(self.lib.pfbo_max_threads())

    def parse_iq_header(self, data: bytes) -> dict:
        hdr = _IqHeader()
        rc = self.lib.pfbo_parse_iq_header(data, len(data), C.byref(hdr))
        if rc != 0:
            raise ValueError(f"pfbo_parse_iq_header rc={rc}")
        d = {}
        for name, _ in _IqHeader._fields_:
            v = getattr(hdr, name)
            d[name] = v.decode("latin1") if isinstance(v, bytes) else v
        return d

    def extract_pdws(self, y: np.ndarray, fs_in: float, fc: float, start_time: float,
                     snr_db: float = 15.0, matlab_quirks: bool = True, max_out: int = 1 << 16, decim: int | None = None,
                     binfreq_unshifted: bool = False):
        y = np.asarray(y, dtype=np.complex128)
        F, M = y.shape
        yr = np.ascontiguousarray(y.real).reshape(-1)
        yi = np.ascontiguousarray(y.imag).reshape(-1)
        buf = (_Pdw * max_out)()
        n = self.lib.pfbo_extract_pdws(yr, yi, F, M, M if decim is None else decim, fs_in, fc, start_time, snr_db,
                                       int(bool(matlab_quirks)) | (2 if binfreq_unshifted else 0), buf, max_out)
        n = min(int(n), max_out)
        return [dict(toa=b.toa, freq=b.freq, pw=b.pw, snr=b.snr, sat=bool(b.sat), bin=b.bin, mag=b.mag) for b in buf[:n]]

    def extract_pdws_raw(self, x: np.ndarray, fs: float, fc: float, start_time: float, snr_db: float = 18.0,
                         trail_db: float = 3.0, max_out: int = 1 << 16):
        """matlab/create_pdws.m:30-105 on normalised complex samples x.  Returns (pdws, noise_floor)."""
        x = np.asarray(x, dtype=np.complex128).reshape(-1)
        xr = np.ascontiguousarray(x.real)
        xi = np.ascontiguousarray(x.imag)
        buf = (_Pdw * max_out)()
        nf = C.c_double(0.0)
        n = self.lib.pfbo_extract_pdws_raw(xr, xi, x.size, fs, fc, start_time, snr_db, trail_db, buf, max_out,
                                           C.byref(nf))
        n = min(int(n), max_out)
        return ([dict(toa=b.toa, freq=b.freq, pw=b.pw, snr=b.snr, sat=bool(b.sat), bin=b.bin, mag=b.mag)
                 for b in buf[:n]], nf.value)


class RefIqPacket:
    """ctypes view of oracle/_ref/libref_iqpacket.so (the reference's own
    cpp/IqPacket.h + cpp/Helper.cpp compiled where they lie)."""

    FIELDS = ("endianness", "linkSpeed", "frequencyHz", "bandwidthHz", "sampleRateSps", "rxGainDb",
              "numSamples", "bitWidth", "spare0", "boardName", "serialNumber", "fpgaVersion", "fwVersion",
              "sampleStartTime")

    def __init__(self):
        if not os.path.exists(_REF):
            raise FileNotFoundError(_REF)
        self.lib = lib = C.CDLL(_REF)
        lib.ref_sizeof_iqpacket.restype = C.c_uint
        lib.ref_iqpacket_offsets.restype = C.c_uint
        lib.ref_make_header.restype = C.c_uint
        lib.ref_make_header.argtypes = [C.c_uint, C.c_uint, C.c_ulonglong, C.c_uint, C.c_uint, C.c_float, C.c_uint,
                                        C.c_uint, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_double,
                                        C.c_char_p, C.c_uint]
        lib.ref_get_filename_str.argtypes = [C.c_longlong, C.c_char_p, C.c_int]

    @staticmethod
    def available() -> bool:
        return os.path.exists(_REF)

    def sizeof(self) -> int:
        return int(self.lib.ref_sizeof_iqpacket())

    def offsets(self) -> dict:
        arr = (C.c_uint * 32)()
        n = self.lib.ref_iqpacket_offsets(arr)
        return dict(zip(self.FIELDS, list(arr[:n])))

    def make_header(self, marker, link_speed, freq_hz, bw_hz, fs_sps, gain_db, num_samples, bit_width,
                    board=b"", serial=b"", fpga=b"", fw=b"", start_time=0.0) -> bytes:
        buf = C.create_string_buffer(256)
        n = self.lib.ref_make_header(marker, link_speed, freq_hz, bw_hz, fs_sps, gain_db, num_samples, bit_width,
                                     board, serial, fpga, fw, start_time, buf, 256)
        return buf.raw[:n]

    def filename(self, epoch_ms: int) -> str:
        buf = C.create_string_buffer(int(self.lib.ref_filename_length()))
        self.lib.ref_get_filename_str(epoch_ms, buf, len(buf))
        return buf.value.decode()


# ---------------------------------------------------------------------------------------------
# numpy twin (independent formulation; generates tests/golden/*)


def channelize_numpy(x: np.ndarray, h: np.ndarray, cfg: OracleConfig) -> np.ndarray:
    """y[m,k] = sum_n h[n] e^{+j2pi kn/M} x[mD+off-n], via a frame matrix and numpy.fft.ifft."""
    M, P, D, off = cfg.M, cfg.P, cfg.decim, cfg.offset
    x = np.asarray(x, dtype=np.complex128)
    if cfg.conj_input:
        x = np.conj(x)
    h = np.asarray(h, dtype=np.float64)
    F = x.size // D
    L = M * P
    xp = np.concatenate([np.zeros(L, dtype=np.complex128), x])  # x[s] at xp[s + L]
    m = np.arange(F)[:, None]
    n = np.arange(L)[None, :]
    seg = xp[m * D + off - n + L] * h[None, :]                  # (F, L): h[n] x[mD+off-n]
    u = seg.reshape(F, P, M).sum(axis=1)                        # u[m,p] = sum_q seg[m, p+Mq]
    y = np.fft.ifft(u, axis=1) * M                              # sum_p u_p e^{+j2pi kp/M}
    if cfg.derotate:
        k = np.arange(M)[None, :]
        ph = (k * ((m * D) % M)) % M
        y = y * np.exp(-2j * np.pi * ph / M)
    if cfg.fftshift:
        y = np.fft.fftshift(y, axes=1)
    return y


def unpack_numpy(iq: np.ndarray, bit_width: int) -> np.ndarray:
    a = np.asarray(iq).reshape(-1, 2).astype(np.float64) / float(2 ** (bit_width - 1))
    return a[:, 0] + 1j * a[:, 1]
