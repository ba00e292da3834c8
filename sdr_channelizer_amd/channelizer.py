"""Host-side mirror of the reference's channelizer usage, over the C ABI.

The reference drives MathWorks' ``dsp.Channelizer`` from script code:

    channelizer = dsp.Channelizer(numBands)        channelizer_example.m:31
    out = channelizer(iq)                          channelizer_example.m:56
    zeroCenterOut = fftshift(out, 2)               channelizer_example.m:58
    f = centerFrequencies(channelizer, fs)         channelizer_example.m:60

``Channelizer`` keeps those names and meanings (constructor takes the band
count, the object is callable and stateful, ``centerFrequencies(fs)``,
``reset()``/``release()``), but consumes the recorders' raw integer I/Q
directly -- the normalise step (channelizer_example.m:18-21) is fused into the
kernel -- and runs on the GPU through libpfb_channelizer.so.  Nothing here
computes on the CPU: without the library or a HIP device it raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

_FMT = {"int8": L.PFB_FMT_INT8_IQ, "int16": L.PFB_FMT_INT16_IQ, "cf32": L.PFB_FMT_CF32}
_NP_DTYPE = {L.PFB_FMT_INT8_IQ: np.int8, L.PFB_FMT_INT16_IQ: np.int16, L.PFB_FMT_CF32: np.float32}


def design_prototype(num_bands: int, taps_per_band: int = 12, stopband_atten: float = 80.0) -> np.ndarray:
    """Convenience Kaiser-windowed-sinc prototype (NOT verified against MathWorks' design)."""
    h = np.empty(num_bands * taps_per_band, dtype=np.float32)
    L.check(L.load().pfb_design_prototype(num_bands, taps_per_band, stopband_atten,
                                          h.ctypes.data_as(C.POINTER(C.c_float))), "pfb_design_prototype")
    return h


def pinned_empty(shape, dtype):
    """numpy array in page-locked host memory (pfb_host_alloc): buffers like this let the host path overlap the two
    PCIe directions.  The memory is released when the array (and every view of it) is garbage-collected."""
    import weakref
    lib = L.load()
    dt = np.dtype(dtype)
    count = int(np.prod(shape))
    nbytes = max(1, count * dt.itemsize)
    ptr = lib.pfb_host_alloc(nbytes)
    if not ptr:
        raise MemoryError(f"pfb_host_alloc({nbytes}) failed")
    buf = (C.c_char * nbytes).from_address(ptr)
    weakref.finalize(buf, lib.pfb_host_free, C.c_void_p(ptr))
    return np.frombuffer(buf, dtype=dt, count=count).reshape(shape)


def center_frequencies(num_bands: int, fs: float, order: str = "fft") -> np.ndarray:
    """centerFrequencies(channelizer, fs).  order="fft": the frequency of unshifted output column k,
    [0, 1, .., -1] * fs / M; order="centered": of column c of fftshift(out, 2), ascending from -fs/2.  Which of the two
    MathWorks' function returns is not pinned here (closed toolbox): channelizer_example.m:58-66 plots the fftshift-ed
    output against this list, which suggests "centered" (DESIGN.md section 4)."""
    out = np.empty(num_bands, dtype=np.float64)
    code = {"fft": L.PFB_FREQ_ORDER_FFT, "centered": L.PFB_FREQ_ORDER_CENTERED}[order]
    L.check(L.load().pfb_center_frequencies_ordered(num_bands, fs, code, out.ctypes.data_as(C.POINTER(C.c_double))),
            "pfb_center_frequencies_ordered")
    return out


class Channelizer:
    """``dsp.Channelizer``-shaped front end of the MI355X polyphase filterbank.

    Parameters mirror the System object where it has them (NumFrequencyBands,
    NumTapsPerBand, StopbandAttenuation, DecimationFactor) plus what the raw
    I/Q path needs (sample_format / bit_width from the IqPacket header).
    """

    def __init__(self, num_bands: int, *, taps: np.ndarray | None = None, taps_per_band: int = 12,
                 stopband_atten: float = 80.0, decimation: int | None = None, sample_format: str = "int16",
                 bit_width: int = 12, channel_major: bool = False, fftshift: bool = False,
                 conjugate_input: bool = False, derotate: bool = False, magnitude: bool = False, power: bool = False,
                 input_offset: int = -1, device: int = -1):
        self._h = C.c_void_p()
        lib = L.load()
        M = int(num_bands)
        if taps is None:
            taps = design_prototype(M, taps_per_band, stopband_atten)
        taps = np.ascontiguousarray(taps, dtype=np.float32).reshape(-1)
        if taps.size % M:
            raise ValueError("taps must hold num_bands * taps_per_band coefficients")
        self.num_bands = M
        self.taps_per_band = taps.size // M
        self.decimation = M if decimation is None else int(decimation)
        self.taps = taps
        self.fmt = _FMT[sample_format]
        self.bit_width = int(bit_width)
        self.channel_major = bool(channel_major)
        self.device = device
        self.power = bool(power)          # float32 |y|^2 out (implies the fused-magnitude output path, no square root)
        magnitude = bool(magnitude) or self.power
        self.magnitude = bool(magnitude)  # fused abs(channelizer(x)): float32 out
        self.fftshift = bool(fftshift)
        self._shard_cb = None  # keeps the ctypes halo-exchange callback alive while attached
        flags = (L.PFB_FLAG_FFTSHIFT if fftshift else 0) | (L.PFB_FLAG_CONJUGATE_INPUT if conjugate_input else 0) \
            | (L.PFB_FLAG_DEROTATE if derotate else 0) | (L.PFB_FLAG_MAGNITUDE if magnitude else 0) \
            | (L.PFB_FLAG_POWER if self.power else 0)
        cfg = L.PfbConfig(C.sizeof(L.PfbConfig), M, self.taps_per_band, self.decimation,
                          taps.ctypes.data_as(C.POINTER(C.c_float)), self.fmt, self.bit_width,
                          L.PFB_LAYOUT_CHANNEL_MAJOR if channel_major else L.PFB_LAYOUT_FRAME_MAJOR, flags,
                          int(input_offset), int(device))
        L.check(lib.pfb_create(C.byref(cfg), C.byref(self._h)), "pfb_create")
        self._lib = lib
        dev = C.c_int(-1)  # device=-1: the library took the device current at pfb_create; it says which one that was
        L.check(lib.pfb_get_device(self._h, C.byref(dev)), "pfb_get_device")
        self._device_index = int(dev.value)

    # -- lifecycle ---------------------------------------------------------------
    def release(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.pfb_destroy(self._h)
            self._h = C.c_void_p()

    close = release

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.release()

    def reset(self) -> None:
        L.check(self._lib.pfb_reset(self._h), "pfb_reset")

    # -- helpers -----------------------------------------------------------------
    def centerFrequencies(self, fs: float, order: str | None = None) -> np.ndarray:  # noqa: N802 (reference name)
        """Centre frequency of every output column of THIS handle: centred order when it fftshifts its output,
        FFT order otherwise (override with order="fft" / "centered")."""
        return center_frequencies(self.num_bands, fs, order or ("centered" if self.fftshift else "fft"))

    def frames_for(self, num_samples: int) -> int:
        f = C.c_uint64()
        L.check(self._lib.pfb_frames_for(self._h, num_samples, C.byref(f)), "pfb_frames_for")
        return int(f.value)

    @property
    def history_samples(self) -> int:
        return int(self._lib.pfb_history_samples(self._h))

    def set_option(self, option: int, value: int) -> None:
        L.check(self._lib.pfb_set_option(self._h, option, value), "pfb_set_option")

    @property
    def last_kernel(self) -> str:
        return self._lib.pfb_last_kernel(self._h).decode()

    def set_stream(self, hip_stream: int) -> None:
        L.check(self._lib.pfb_set_stream(self._h, C.c_void_p(hip_stream)), "pfb_set_stream")

    def set_frame_index(self, next_frame: int) -> None:
        L.check(self._lib.pfb_set_frame_index(self._h, next_frame), "pfb_set_frame_index")

    def get_state(self) -> bytes:
        n = C.c_size_t(0)
        L.check(self._lib.pfb_get_state(self._h, None, C.byref(n)), "pfb_get_state")
        buf = C.create_string_buffer(n.value)
        L.check(self._lib.pfb_get_state(self._h, buf, C.byref(n)), "pfb_get_state")
        return buf.raw[: n.value]

    def set_state(self, blob: bytes) -> None:
        L.check(self._lib.pfb_set_state(self._h, blob, len(blob)), "pfb_set_state")

    # -- samples -----------------------------------------------------------------
    def _host_samples(self, iq: np.ndarray) -> tuple[np.ndarray, int]:
        want = _NP_DTYPE[self.fmt]
        a = np.asarray(iq)
        if self.fmt == L.PFB_FMT_CF32 and np.iscomplexobj(a):
            a = np.ascontiguousarray(a, dtype=np.complex64).view(np.float32)
        if a.dtype != want:
            raise TypeError(f"expected {np.dtype(want)} I/Q for this channelizer, got {a.dtype}")
        a = np.ascontiguousarray(a).reshape(-1)
        if a.size % 2:
            raise ValueError("interleaved I,Q needs an even element count")
        return a, a.size // 2

    def _is_torch(self, x) -> bool:
        return type(x).__module__.startswith("torch")

    def _device_samples(self, iq) -> int:
        """Validate a CUDA tensor of raw samples against the handle (dtype, device, contiguity) BEFORE its pointer goes
        to the library -- a wrong dtype would be read at the handle's sample size, past the end of the allocation --
        and return its length in complex samples."""
        import torch
        want = {L.PFB_FMT_INT8_IQ: (torch.int8,), L.PFB_FMT_INT16_IQ: (torch.int16,),
                L.PFB_FMT_CF32: (torch.float32, torch.complex64)}[self.fmt]
        if iq.dtype not in want:
            raise TypeError(f"expected {want[0]} I/Q for this channelizer, got {iq.dtype}")
        dev = self.device_index
        if iq.device.index != dev:
            raise ValueError(f"I/Q tensor is on cuda:{iq.device.index}, the channelizer on cuda:{dev}")
        if not iq.is_contiguous():
            raise ValueError("device I/Q must be contiguous")
        if iq.is_complex():
            return iq.numel()
        if iq.numel() % 2 or (iq.dim() >= 2 and iq.shape[-1] != 2):
            raise ValueError("interleaved I,Q: the last dimension must be 2 (or a flat tensor of even length)")
        return iq.numel() // 2

    @property
    def device_index(self) -> int:
        """The HIP device ordinal the handle lives on (device=-1 at construction = the device current at that moment)."""
        return self._device_index

    def prime(self, iq) -> None:
        """Feed history without producing output (time-shard halo, resume)."""
        if self._is_torch(iq) and iq.is_cuda:
            n = self._device_samples(iq)
            L.check(self._lib.pfb_prime(self._h, C.c_void_p(iq.data_ptr()), n, L.PFB_MEM_DEVICE), "pfb_prime")
            return
        a, n = self._host_samples(iq)
        L.check(self._lib.pfb_prime(self._h, C.c_void_p(a.ctypes.data), n, L.PFB_MEM_HOST), "pfb_prime")

    def __call__(self, iq, out=None, sync: bool = True):
        """Channelize one buffer.  numpy in -> numpy out (staged through the GPU);
        torch CUDA tensor in -> torch CUDA tensor out (no copies).
        Returns complex64 of shape (frames, M), or (M, frames) when channel_major."""
        M = self.num_bands
        if self._is_torch(iq) and iq.is_cuda:
            import torch
            n = self._device_samples(iq)
            F = self.frames_for(n)
            shape = (M, F) if self.channel_major else (F, M)
            odt = torch.float32 if self.magnitude else torch.complex64
            if out is None:
                out = torch.empty(shape, dtype=odt, device=iq.device)
            elif (not self._is_torch(out) or not out.is_cuda or out.device != iq.device or out.numel() < F * M
                  or out.dtype != odt or not out.is_contiguous()):
                raise ValueError(f"out must be a contiguous {odt} tensor on {iq.device} with room for frames*M values")
            f = C.c_uint64()
            fn = self._lib.pfb_process if sync else self._lib.pfb_process_async
            args = [self._h, C.c_void_p(iq.data_ptr()), n, C.c_void_p(out.data_ptr()), F, C.byref(f)]
            if sync:
                args.append(L.PFB_MEM_DEVICE)
            L.check(fn(*args), "pfb_process")
            return out if out.shape == shape else out.reshape(-1)[: F * M].reshape(shape)
        a, n = self._host_samples(iq)
        F = self.frames_for(n)
        shape = (M, F) if self.channel_major else (F, M)
        odt = np.float32 if self.magnitude else np.complex64
        if out is None:
            res = np.empty(shape, dtype=odt)
        else:  # the library writes F * M elements through this pointer: check before handing it over
            if not isinstance(out, np.ndarray) or out.dtype != odt or out.size < F * M or not out.flags.c_contiguous:
                raise ValueError("out must be a C-contiguous numpy array of the output dtype with room for frames*M values")
            res = out if out.shape == shape else out.reshape(-1)[: F * M].reshape(shape)
        f = C.c_uint64()
        L.check(self._lib.pfb_process(self._h, C.c_void_p(a.ctypes.data), n, C.c_void_p(res.ctypes.data), F,
                                      C.byref(f), L.PFB_MEM_HOST), "pfb_process")
        return res

    # -- time sharding (pfb_shard_attach / pfb_process_shard_async) -------------------
    @property
    def halo_samples(self) -> int:
        """Raw samples a time shard needs from its predecessor: M*P - 1 - input_offset = (P-1)*M by default."""
        return int(self._lib.pfb_halo_samples(self._h))

    @property
    def shard_head_frames(self) -> int:
        return int(self._lib.pfb_shard_head_frames(self._h))

    def attach_shard(self, rank: int, world: int, exchange=None, ring: bool = False) -> None:
        """Make this handle segment ``rank`` of ``world``.  ``exchange(d_send, d_recv, nbytes, send_to, recv_from,
        hip_stream) -> int`` enqueues the two transfers on ``hip_stream`` (pointers are ints, 0 / rank -1 = absent) and
        returns 0; sdr_channelizer_amd.sharded.make_exchange builds one over torch.distributed."""
        cb = L.HALO_EXCHANGE_FN(0)
        if exchange is not None:
            def _trampoline(_user, d_send, d_recv, nbytes, send_to, recv_from, stream):
                try:
                    return int(exchange(d_send or 0, d_recv or 0, int(nbytes), int(send_to), int(recv_from), stream or 0))
                except Exception:  # an exception must not unwind through the C library
                    import traceback
                    traceback.print_exc()
                    return -1
            cb = L.HALO_EXCHANGE_FN(_trampoline)
        cfg = L.PfbShardConfig(C.sizeof(L.PfbShardConfig), int(rank), int(world), 1 if ring else 0, cb, None)
        L.check(self._lib.pfb_shard_attach(self._h, C.byref(cfg)), "pfb_shard_attach")
        self._shard_cb = cb

    def process_shard(self, segment, out=None):
        """Channelize this rank's segment (CUDA tensor, whole frames): halo exchange on a side stream, interior frames at
        once, head frames when the halo has landed.  Asynchronous; ``sync()`` waits for kernels and transfers."""
        import torch
        n = self._device_samples(segment)
        if n % self.decimation:
            raise ValueError("a shard is cut on frame boundaries: its length must be a multiple of the decimation")
        F, M = n // self.decimation, self.num_bands
        shape = (M, F) if self.channel_major else (F, M)
        odt = torch.float32 if self.magnitude else torch.complex64
        if out is None:
            out = torch.empty(shape, dtype=odt, device=segment.device)
        elif (not out.is_cuda or out.device != segment.device or out.numel() < F * M or out.dtype != odt
              or not out.is_contiguous()):
            raise ValueError(f"out must be a contiguous {odt} tensor on {segment.device} with room for frames*M values")
        f = C.c_uint64()
        L.check(self._lib.pfb_process_shard_async(self._h, C.c_void_p(segment.data_ptr()), n, C.c_void_p(out.data_ptr()),
                                                  F, C.byref(f)), "pfb_process_shard_async")
        return out if out.shape == shape else out.reshape(-1)[: F * M].reshape(shape)

    def kernel_times_ms(self) -> list[float]:
        """Durations of the channelizer kernel launches recorded since PFB_OPT_PROFILE was set."""
        buf = (C.c_float * 4096)()
        n = C.c_int(0)
        L.check(self._lib.pfb_get_kernel_times(self._h, buf, 4096, C.byref(n)), "pfb_get_kernel_times")
        return list(buf[: n.value])

    def process_iq_file(self, path: str, reset: bool = True, out=None):
        """Channelize one .iq record straight from disk (header parsed and checked by the library).
        Returns (y, info): y is (frames, M) complex64 (or float32 with magnitude=True), (M, frames) for a
        channel-major handle.  ``out``: optional numpy
        buffer with room for frames * M values (page-locked, see pinned_empty, for the full rate)."""
        from . import iqfile
        with open(path, "rb") as f:
            info = iqfile.parse_header(f.read(128))
        if reset:
            self.reset()  # a fresh channelizer per file, create_pdws_channelized.m:33
        F = self.frames_for(int(info.packet.numSamples))
        dt = np.float32 if self.magnitude else np.complex64
        shape = (self.num_bands, F) if self.channel_major else (F, self.num_bands)
        if out is None:
            res = np.empty(shape, dtype=dt)
        else:
            if out.dtype != dt or out.size < F * self.num_bands or not out.flags.c_contiguous:
                raise ValueError("out must be a C-contiguous array of the output dtype with room for frames*M values")
            res = out.reshape(-1)[: F * self.num_bands].reshape(shape)
        f_out = C.c_uint64()
        got = L.PfbIqInfo()
        L.check(self._lib.pfb_process_iq_file(self._h, path.encode(), C.c_void_p(res.ctypes.data), F, C.byref(f_out),
                                              C.byref(got)), "pfb_process_iq_file")
        return (res if self.channel_major else res[: f_out.value]), got

    def sync(self) -> None:
        L.check(self._lib.pfb_sync(self._h), "pfb_sync")
