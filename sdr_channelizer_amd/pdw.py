"""PDW extraction over the C ABI: the channelized one (the second half of
matlab/create_pdws_channelized.m, lines 64-143, pfb_pdw_extract) and the raw-stream one
(matlab/create_pdws.m:30-105, pfb_pdw_extract_raw).  Both return the fields the scripts accumulate in
their ``pdw`` struct -- toa, freq, pw, snr, sat, and mag (create_pdws.m:96) -- plus the column each pulse
was found in."""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import _lib as L

PDW_DTYPE = np.dtype([("toa", "f8"), ("freq", "f8"), ("pw", "f8"), ("snr", "f8"), ("sat", "i4"), ("bin", "i4"),
                      ("mag", "f8")])


_tls = threading.local()


def _out_buffer(capacity: int):
    """The landing zone of a call's PDWs: one buffer per thread, kept (and its pages touched) between calls -- a fresh
    48 MB np.zeros per call costs more than the extraction's host side; callers get a copy of the records found.
    (Page-locking it was tried: 0.05 ms per call for a 140 ms first call -- not kept.)"""
    buf = getattr(_tls, "out", None)
    if buf is None or len(buf) < capacity:
        buf = _tls.out = np.empty(max(int(capacity), 1), dtype=PDW_DTYPE)
    return buf


def _take(out, n: int):
    """A copy of the first n records.  Through a byte view: numpy copies a structured array field by field (0.25 ms for
    24 000 PDWs), a byte array at memcpy speed."""
    return out[:n].view(np.uint8).copy().view(PDW_DTYPE)


def extract_pdws(y, fs_in: float, fc: float, sample_start_time: float, *, decimation: int | None = None,
                 snr_threshold_db: float = 15.0, matlab_quirks: bool = True, capacity: int = 1 << 20,
                 return_noise_floor: bool = False, device: int = -1, channel_major: bool = False,
                 binfreq_unshifted: bool = False):
    """y: (frames, M) complex64, frame-major, fftshift-ed -- a numpy array or a torch CUDA tensor
    (used in place); with channel_major=True y is (M, frames), MATLAB's own layout of the same matrix (what a
    channel-major Channelizer returns).  Returns a structured numpy array with PDW_DTYPE, in the reference's order
    (channels outermost, time within a channel).

    matlab_quirks: the script's phase(toa:jj) at :114 linear-indexes column 1 whatever channel the pulse is in
    (reference behaviour, default on).  binfreq_unshifted: take fc_chan (:80) from the FFT-ordered centre-frequency list
    indexed with the shifted column -- what the script computes IF MathWorks' centerFrequencies returns the unshifted
    list; that order is not pinned (closed toolbox; channelizer_example.m:58-66 suggests the list is centred), so the
    default is the column's true centre frequency."""
    lib = L.load()
    is_torch = type(y).__module__.startswith("torch")
    if is_torch and y.is_cuda:
        if not y.is_contiguous() or y.dim() != 2:
            raise ValueError("need a contiguous (frames, M) complex64 tensor")
        frames, M = (int(y.shape[1]), int(y.shape[0])) if channel_major else (int(y.shape[0]), int(y.shape[1]))
        ptr, mem, keep = C.c_void_p(y.data_ptr()), L.PFB_MEM_DEVICE, y
        import torch
        stream = C.c_void_p(torch.cuda.current_stream(y.device).cuda_stream)
        device = y.device.index
    else:
        a = np.ascontiguousarray(np.asarray(y), dtype=np.complex64)
        frames, M = a.shape[::-1] if channel_major else a.shape
        ptr, mem, keep, stream = C.c_void_p(a.ctypes.data), L.PFB_MEM_HOST, a, C.c_void_p(0)
    out = _out_buffer(capacity)
    assert out.dtype.itemsize == C.sizeof(L.PfbPdw)
    nf = np.zeros(M, dtype=np.float64)
    count = C.c_uint64(0)
    flags = (L.PFB_PDW_MATLAB_QUIRKS if matlab_quirks else 0) | (L.PFB_PDW_CHANNEL_MAJOR if channel_major else 0) \
        | (L.PFB_PDW_BINFREQ_UNSHIFTED if binfreq_unshifted else 0)
    rc = lib.pfb_pdw_extract(ptr, frames, M, M if decimation is None else int(decimation), float(fs_in), float(fc),
                             float(sample_start_time), float(snr_threshold_db), flags,
                             out.ctypes.data_as(C.POINTER(L.PfbPdw)), capacity, C.byref(count),
                             nf.ctypes.data_as(C.POINTER(C.c_double)), mem, int(device), stream)
    del keep
    if rc != L.PFB_OK:
        detail = lib.pfb_pdw_last_error_detail().decode()
        raise L.PfbError(rc, "pfb_pdw_extract" + (f" [{detail}]" if detail else ""))
    n = int(count.value)
    if n > capacity:
        raise OverflowError(f"{n} pulses found, capacity {capacity}")
    return (_take(out, n), nf) if return_noise_floor else _take(out, n)


def pdws_from_iq_file(channelizer, path: str, *, snr_threshold_db: float = 15.0, matlab_quirks: bool = True,
                      capacity: int = 1 << 20, reset: bool = True, return_noise_floor: bool = False,
                      binfreq_unshifted: bool = False):
    """One iteration of create_pdws_channelized.m:22-143 in one call (pfb_pdw_from_iq_file): the record at ``path``
    is streamed through ``channelizer`` (a frame-major, complex-output Channelizer matching the record), the channel
    matrix stays on the GPU, the PDWs come back.  fs, fc and the start time are the record's.  Returns (pdws, info)
    or (pdws, noise_floor, info)."""
    lib = L.load()
    if reset:
        channelizer.reset()  # a fresh channelizer per file, create_pdws_channelized.m:33
    out = _out_buffer(capacity)
    nf = np.zeros(channelizer.num_bands, dtype=np.float64)
    count = C.c_uint64(0)
    info = L.PfbIqInfo()
    rc = lib.pfb_pdw_from_iq_file(channelizer._h, path.encode(), float(snr_threshold_db),
                                  (L.PFB_PDW_MATLAB_QUIRKS if matlab_quirks else 0)
                                  | (L.PFB_PDW_BINFREQ_UNSHIFTED if binfreq_unshifted else 0),
                                  out.ctypes.data_as(C.POINTER(L.PfbPdw)),
                                  capacity, C.byref(count), nf.ctypes.data_as(C.POINTER(C.c_double)), C.byref(info))
    if rc != L.PFB_OK:
        detail = lib.pfb_pdw_last_error_detail().decode()
        raise L.PfbError(rc, "pfb_pdw_from_iq_file" + (f" [{detail}]" if detail else ""))
    n = int(count.value)
    if n > capacity:
        raise OverflowError(f"{n} pulses found, capacity {capacity}")
    return (_take(out, n), nf, info) if return_noise_floor else (_take(out, n), info)


def raw_pdws_from_iq_file(path: str, *, snr_threshold_db: float = 18.0, trailing_threshold_db: float = 3.0,
                          capacity: int = 1 << 20, return_noise_floor: bool = False, device: int = -1):
    """One iteration of create_pdws.m's loop (pfb_pdw_raw_from_iq_file): the record's raw stream -> PDWs, with fs, fc,
    bit width and start time from its header.  Returns (pdws, info) or (pdws, noise_floor, info)."""
    lib = L.load()
    out = _out_buffer(capacity)
    nf = C.c_double(0.0)
    count = C.c_uint64(0)
    info = L.PfbIqInfo()
    rc = lib.pfb_pdw_raw_from_iq_file(path.encode(), float(snr_threshold_db), float(trailing_threshold_db),
                                      out.ctypes.data_as(C.POINTER(L.PfbPdw)), capacity, C.byref(count), C.byref(nf),
                                      C.byref(info), int(device))
    if rc != L.PFB_OK:
        detail = lib.pfb_pdw_last_error_detail().decode()
        raise L.PfbError(rc, "pfb_pdw_raw_from_iq_file" + (f" [{detail}]" if detail else ""))
    n = int(count.value)
    if n > capacity:
        raise OverflowError(f"{n} pulses found, capacity {capacity}")
    return (_take(out, n), nf.value, info) if return_noise_floor else (_take(out, n), info)


def extract_pdws_raw(iq, fs: float, fc: float, sample_start_time: float, *, bit_width: int = 12,
                     snr_threshold_db: float = 18.0, trailing_threshold_db: float = 3.0, capacity: int = 1 << 20,
                     return_noise_floor: bool = False, device: int = -1):
    """create_pdws.m on the recorder stream: iq is (n, 2) int8 / int16 (I, Q columns; bit_width as in the
    record header) or (n,) complex64 -- a numpy array or a torch CUDA tensor (used in place).  Defaults are
    the script's thresholds (18 dB leading, 3 dB trailing)."""
    lib = L.load()
    is_torch = type(iq).__module__.startswith("torch")
    if is_torch and iq.is_cuda:
        import torch
        if not iq.is_contiguous():
            raise ValueError("need a contiguous tensor")
        fmt = {torch.int8: L.PFB_FMT_INT8_IQ, torch.int16: L.PFB_FMT_INT16_IQ, torch.complex64: L.PFB_FMT_CF32}[iq.dtype]
        n = int(iq.shape[0])
        ptr, mem, keep = C.c_void_p(iq.data_ptr()), L.PFB_MEM_DEVICE, iq
        stream = C.c_void_p(torch.cuda.current_stream(iq.device).cuda_stream)
        device = iq.device.index
    else:
        a = np.ascontiguousarray(np.asarray(iq))
        if a.dtype == np.complex64:
            fmt = L.PFB_FMT_CF32
        elif a.dtype in (np.int8, np.int16) and a.ndim == 2 and a.shape[1] == 2:
            fmt = L.PFB_FMT_INT8_IQ if a.dtype == np.int8 else L.PFB_FMT_INT16_IQ
        else:
            raise ValueError("iq must be (n, 2) int8/int16 or (n,) complex64")
        n = int(a.shape[0])
        ptr, mem, keep, stream = C.c_void_p(a.ctypes.data), L.PFB_MEM_HOST, a, C.c_void_p(0)
    out = _out_buffer(capacity)
    assert out.dtype.itemsize == C.sizeof(L.PfbPdw)
    nf = C.c_double(0.0)
    count = C.c_uint64(0)
    rc = lib.pfb_pdw_extract_raw(ptr, n, fmt, int(bit_width), float(fs), float(fc), float(sample_start_time),
                                 float(snr_threshold_db), float(trailing_threshold_db),
                                 out.ctypes.data_as(C.POINTER(L.PfbPdw)), capacity, C.byref(count), C.byref(nf), mem,
                                 int(device), stream)
    del keep
    if rc != L.PFB_OK:
        detail = lib.pfb_pdw_last_error_detail().decode()
        raise L.PfbError(rc, "pfb_pdw_extract_raw" + (f" [{detail}]" if detail else ""))
    k = int(count.value)
    if k > capacity:
        raise OverflowError(f"{k} pulses found, capacity {capacity}")
    return (_take(out, k), nf.value) if return_noise_floor else _take(out, k)
