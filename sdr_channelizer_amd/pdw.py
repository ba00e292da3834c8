"""Channelized PDW extraction (the second half of matlab/create_pdws_channelized.m, lines 64-143)
over the C ABI's pfb_pdw_extract.  Returns the same fields the script accumulates in its ``pdw``
struct (:16-20,124-128): toa, freq, pw, snr, sat -- plus the column each pulse was found in."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

PDW_DTYPE = np.dtype([("toa", "f8"), ("freq", "f8"), ("pw", "f8"), ("snr", "f8"), ("sat", "i4"), ("bin", "i4")])


def extract_pdws(y, fs_in: float, fc: float, sample_start_time: float, *, decimation: int | None = None,
                 snr_threshold_db: float = 15.0, matlab_quirks: bool = True, capacity: int = 1 << 20,
                 return_noise_floor: bool = False, device: int = -1):
    """y: (frames, M) complex64, frame-major, fftshift-ed -- a numpy array or a torch CUDA tensor
    (used in place).  Returns a structured numpy array with PDW_DTYPE, in the reference's order
    (channels outermost, time within a channel)."""
    lib = L.load()
    is_torch = type(y).__module__.startswith("torch")
    if is_torch and y.is_cuda:
        if not y.is_contiguous() or y.dim() != 2:
            raise ValueError("need a contiguous (frames, M) complex64 tensor")
        frames, M = int(y.shape[0]), int(y.shape[1])
        ptr, mem, keep = C.c_void_p(y.data_ptr()), L.PFB_MEM_DEVICE, y
        import torch
        stream = C.c_void_p(torch.cuda.current_stream(y.device).cuda_stream)
        device = y.device.index
    else:
        a = np.ascontiguousarray(np.asarray(y), dtype=np.complex64)
        frames, M = a.shape
        ptr, mem, keep, stream = C.c_void_p(a.ctypes.data), L.PFB_MEM_HOST, a, C.c_void_p(0)
    out = np.zeros(capacity, dtype=PDW_DTYPE)
    assert out.dtype.itemsize == C.sizeof(L.PfbPdw)
    nf = np.zeros(M, dtype=np.float64)
    count = C.c_uint64(0)
    flags = L.PFB_PDW_MATLAB_QUIRKS if matlab_quirks else 0
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
    return (out[:n], nf) if return_noise_floor else out[:n]
