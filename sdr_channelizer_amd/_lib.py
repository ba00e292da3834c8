"""ctypes binding of libpfb_channelizer.so (the C ABI in include/pfb_channelizer.h).

The library is the product: if it is missing this module raises -- there is no
Python or CPU fallback for the channelizer arithmetic.
"""
from __future__ import annotations

import ctypes as C  # noqa: N811
import os
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libpfb_channelizer.so")

PFB_OK = 0
PFB_ERR_BAD_ARG = -1
PFB_ERR_BAD_FORMAT = -2
PFB_ERR_UNSUPPORTED = -3
PFB_ERR_NO_DEVICE = -4
PFB_ERR_HIP = -5
PFB_ERR_NO_MEMORY = -6
PFB_ERR_CAPACITY = -7
PFB_ERR_INTERNAL = -8
PFB_ERR_COMM = -9
PFB_ABI_VERSION = 2
PFB_FREQ_ORDER_FFT, PFB_FREQ_ORDER_CENTERED = 0, 1

PFB_FMT_INT8_IQ, PFB_FMT_INT16_IQ, PFB_FMT_CF32 = 0, 1, 2
PFB_LAYOUT_FRAME_MAJOR, PFB_LAYOUT_CHANNEL_MAJOR = 0, 1
PFB_FLAG_FFTSHIFT, PFB_FLAG_CONJUGATE_INPUT, PFB_FLAG_DEROTATE, PFB_FLAG_MAGNITUDE = 1, 2, 4, 8
PFB_FLAG_POWER = 16  # with PFB_FLAG_MAGNITUDE: |y|^2 instead of |y|
PFB_MEM_HOST, PFB_MEM_DEVICE = 0, 1
PFB_OPT_KERNEL, PFB_OPT_FRAMES_PER_BLOCK, PFB_OPT_HOST_CHUNK_SAMPLES, PFB_OPT_NONTEMPORAL, PFB_OPT_PROFILE, PFB_OPT_XCD_REMAP = 0, 1, 2, 3, 4, 5
PFB_OPT_SCHEDULE, PFB_OPT_GRID, PFB_OPT_TILE_WAVES, PFB_OPT_EXPERIMENT, PFB_OPT_VARIANT = 6, 7, 8, 9, 10
PFB_OPT_SLAB_FRAMES = 11


class PfbConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("num_channels", C.c_uint32), ("taps_per_channel", C.c_uint32),
        ("decimation", C.c_uint32), ("taps", C.POINTER(C.c_float)), ("sample_format", C.c_uint32),
        ("bit_width", C.c_uint32), ("output_layout", C.c_uint32), ("flags", C.c_uint32),
        ("input_offset", C.c_int32), ("device_id", C.c_int32),
    ]


class PfbIqPacket(C.Structure):
    _fields_ = [
        ("endianness", C.c_uint32), ("linkSpeed", C.c_uint32), ("frequencyHz", C.c_uint64),
        ("bandwidthHz", C.c_uint32), ("sampleRateSps", C.c_uint32), ("rxGainDb", C.c_float),
        ("numSamples", C.c_uint32), ("bitWidth", C.c_uint32), ("spare0", C.c_uint32),
        ("boardName", C.c_char * 16), ("serialNumber", C.c_char * 16), ("fpgaVersion", C.c_char * 16),
        ("fwVersion", C.c_char * 16), ("sampleStartTime", C.c_double),
    ]


class PfbPdw(C.Structure):
    _fields_ = [("toa", C.c_double), ("freq", C.c_double), ("pw", C.c_double), ("snr", C.c_double),
                ("sat", C.c_int32), ("bin", C.c_int32), ("mag", C.c_double)]


PFB_PDW_MATLAB_QUIRKS = 1        # phase(toa:jj) linear-indexes column 1 (create_pdws_channelized.m:114)
PFB_PDW_CHANNEL_MAJOR = 2
PFB_PDW_BINFREQ_UNSHIFTED = 4    # binFreqs(bin) from the FFT-ordered list (:42/:80 if centerFrequencies is unshifted; unpinned)

# int exchange(void* user, const void* d_send, void* d_recv, size_t bytes, int send_to, int recv_from, void* hip_stream)
HALO_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p)


class PfbShardConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("rank", C.c_int32), ("world", C.c_int32), ("ring", C.c_uint32),
                ("exchange", HALO_EXCHANGE_FN), ("user", C.c_void_p)]


class PfbIqInfo(C.Structure):
    _fields_ = [
        ("packet", PfbIqPacket), ("file_format", C.c_int32), ("header_bytes", C.c_uint32),
        ("bytes_per_sample", C.c_uint32), ("sample_format", C.c_uint32), ("rx_gain_as_read", C.c_double),
    ]


# every symbol include/pfb_channelizer.h + include/pfb_iq_packet.h declare: the drop-in boundary
EXPORTS = (
    "pfb_create", "pfb_destroy", "pfb_reset", "pfb_set_stream", "pfb_process", "pfb_process_async", "pfb_sync", "pfb_process_iq_file",
    "pfb_frames_for", "pfb_history_samples", "pfb_prime", "pfb_get_state", "pfb_set_state", "pfb_set_frame_index",
    "pfb_get_frame_index", "pfb_center_frequencies", "pfb_design_prototype", "pfb_strerror",
    "pfb_last_error_detail", "pfb_abi_version", "pfb_device_count", "pfb_set_option", "pfb_last_kernel", "pfb_get_device",
    "pfb_get_kernel_times", "pfb_host_alloc", "pfb_host_free", "pfb_pdw_extract", "pfb_pdw_extract_raw", "pfb_pdw_from_iq_file", "pfb_pdw_raw_from_iq_file", "pfb_pdw_last_error_detail", "pfb_pdw_release_workspace", "pfb_pdw_last_noise_floor_path",
    "pfb_iq_parse_header", "pfb_iq_fill_packet", "pfb_iq_filename",
    "pfb_shard_attach", "pfb_halo_samples", "pfb_shard_head_frames", "pfb_halo_recv_buffer", "pfb_process_shard_async",
    "pfb_center_frequencies_ordered",
)
# include/pfb_channelizer_dev.h: measurement yardsticks and the ABI self test (bench.py, tools/, tests/)
DEV_EXPORTS = ("pfb_measure_stream_copy", "pfb_measure_mix_copy", "pfb_selftest_exception_guard")

_lib = None


class PfbError(RuntimeError):
    def __init__(self, status: int, where: str):
        lib = load()
        msg = lib.pfb_strerror(status).decode()
        detail = lib.pfb_last_error_detail().decode()
        super().__init__(f"{where}: {msg} ({status})" + (f" [{detail}]" if detail else ""))
        self.status = status


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch wheels bundle their own libamdhip64; two HIP runtimes in one process cannot both own the
    # GPU ("No HIP GPUs are available" for whichever comes second).  If PyTorch is installed, let it
    # load its runtime first: our library's libamdhip64.so.7 dependency then binds to that same copy,
    # and torch tensors / streams can be handed straight to the C ABI.
    if "torch" not in sys.modules and os.environ.get("PFB_NO_TORCH_PRELOAD") is None:
        import importlib.util
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m sdr_channelizer_amd.build` "
            "(hipcc, gfx950). The channelizer has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, u64, u32, i32, i64 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_int64
    lib.pfb_create.argtypes = [C.POINTER(PfbConfig), C.POINTER(vp)]
    lib.pfb_destroy.argtypes = [vp]
    lib.pfb_reset.argtypes = [vp]
    lib.pfb_set_stream.argtypes = [vp, vp]
    lib.pfb_process.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64), u32]
    lib.pfb_process_async.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64)]
    lib.pfb_sync.argtypes = [vp]
    lib.pfb_process_iq_file.argtypes = [vp, C.c_char_p, vp, u64, C.POINTER(u64), C.POINTER(PfbIqInfo)]
    lib.pfb_frames_for.argtypes = [vp, u64, C.POINTER(u64)]
    lib.pfb_history_samples.argtypes = [vp]
    lib.pfb_history_samples.restype = u64
    lib.pfb_prime.argtypes = [vp, vp, u64, u32]
    lib.pfb_get_state.argtypes = [vp, vp, C.POINTER(C.c_size_t)]
    lib.pfb_set_state.argtypes = [vp, vp, C.c_size_t]
    lib.pfb_set_frame_index.argtypes = [vp, u64]
    lib.pfb_get_frame_index.argtypes = [vp, C.POINTER(u64)]
    lib.pfb_center_frequencies.argtypes = [u32, C.c_double, C.POINTER(C.c_double)]
    lib.pfb_center_frequencies_ordered.argtypes = [u32, C.c_double, u32, C.POINTER(C.c_double)]
    lib.pfb_selftest_exception_guard.argtypes = [C.c_int]
    lib.pfb_shard_attach.argtypes = [vp, C.POINTER(PfbShardConfig)]
    lib.pfb_halo_samples.argtypes = [vp]
    lib.pfb_halo_samples.restype = u64
    lib.pfb_shard_head_frames.argtypes = [vp]
    lib.pfb_shard_head_frames.restype = u64
    lib.pfb_halo_recv_buffer.argtypes = [vp]
    lib.pfb_halo_recv_buffer.restype = C.c_void_p
    lib.pfb_process_shard_async.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64)]
    lib.pfb_design_prototype.argtypes = [u32, u32, C.c_double, C.POINTER(C.c_float)]
    lib.pfb_strerror.argtypes = [C.c_int]
    lib.pfb_strerror.restype = C.c_char_p
    lib.pfb_last_error_detail.restype = C.c_char_p
    lib.pfb_set_option.argtypes = [vp, C.c_int, i64]
    lib.pfb_last_kernel.argtypes = [vp]
    lib.pfb_last_kernel.restype = C.c_char_p
    lib.pfb_get_device.argtypes = [vp, C.POINTER(C.c_int)]
    lib.pfb_get_kernel_times.argtypes = [vp, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
    lib.pfb_host_alloc.argtypes = [C.c_size_t]
    lib.pfb_host_alloc.restype = C.c_void_p
    lib.pfb_host_free.argtypes = [vp]
    lib.pfb_host_free.restype = None
    lib.pfb_pdw_extract.argtypes = [vp, u64, u32, u32, C.c_double, C.c_double, C.c_double, C.c_double, u32,
                                    C.POINTER(PfbPdw), u64, C.POINTER(u64), C.POINTER(C.c_double), u32, i32, vp]
    lib.pfb_pdw_from_iq_file.argtypes = [vp, C.c_char_p, C.c_double, u32, C.POINTER(PfbPdw), u64, C.POINTER(u64),
                                         C.POINTER(C.c_double), C.POINTER(PfbIqInfo)]
    lib.pfb_pdw_from_iq_file.restype = C.c_int
    lib.pfb_pdw_raw_from_iq_file.argtypes = [C.c_char_p, C.c_double, C.c_double, C.POINTER(PfbPdw), u64, C.POINTER(u64),
                                             C.POINTER(C.c_double), C.POINTER(PfbIqInfo), C.c_int32]
    lib.pfb_pdw_raw_from_iq_file.restype = C.c_int
    lib.pfb_pdw_extract_raw.argtypes = [vp, u64, u32, u32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                        C.POINTER(PfbPdw), u64, C.POINTER(u64), C.POINTER(C.c_double), u32, i32, vp]
    lib.pfb_pdw_last_error_detail.restype = C.c_char_p
    lib.pfb_pdw_release_workspace.argtypes = [C.c_int32]
    lib.pfb_pdw_release_workspace.restype = C.c_int
    lib.pfb_pdw_last_noise_floor_path.restype = C.c_int
    lib.pfb_measure_stream_copy.argtypes = [C.c_int, u64, C.c_int, C.POINTER(C.c_double)]
    lib.pfb_measure_mix_copy.argtypes = [C.c_int, u64, u32, u32, C.c_int, C.POINTER(C.c_double)]
    lib.pfb_iq_parse_header.argtypes = [vp, C.c_size_t, C.POINTER(PfbIqInfo)]
    lib.pfb_iq_fill_packet.argtypes = [C.POINTER(PfbIqPacket), u32, u64, u32, u32, C.c_float, u32, u32,
                                       C.c_char_p, C.c_char_p, C.c_double]
    lib.pfb_iq_fill_packet.restype = None
    lib.pfb_iq_filename.argtypes = [i64, C.c_char_p, C.c_int]
    for name in EXPORTS + DEV_EXPORTS:
        getattr(lib, name)  # AttributeError here = header/library mismatch
    _lib = lib
    return lib


def check(status: int, where: str) -> None:
    if status != PFB_OK:
        raise PfbError(status, where)
