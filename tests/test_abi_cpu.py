"""The C-ABI library without a GPU: it loads, exports every symbol the headers declare,
the pure-host helpers work, and compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import sdr_channelizer_amd as pkg
from sdr_channelizer_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(headers=("pfb_channelizer.h", "pfb_iq_packet.h")):
    names = set()
    for hdr in headers:
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(pfb_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol():
    lib = L.load()
    declared = declared_symbols()
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    # measurement yardsticks and the ABI self test: their own header, outside the drop-in boundary
    dev = declared_symbols(("pfb_channelizer_dev.h",))
    assert dev == set(L.DEV_EXPORTS) and not (dev & declared), dev ^ set(L.DEV_EXPORTS)
    for name in declared | dev:
        assert hasattr(lib, name), name
    pub = open(os.path.join(ROOT, "include", "pfb_channelizer.h")).read()
    for word in ("PFB_OPT_EXPERIMENT", "PFB_OPT_VARIANT", "PFB_OPT_GRID", "PFB_OPT_TILE_WAVES", "pfb_measure_", "pfb_probe_", "pfb_selftest_"):
        assert word not in pub.replace("PFB_OPT_GRID / TILE_WAVES / EXPERIMENT / VARIANT", ""), word
    assert lib.pfb_abi_version() == L.PFB_ABI_VERSION == 2


def test_strerror_and_center_frequencies(oracle):
    lib = L.load()
    assert lib.pfb_strerror(0) == b"ok" and lib.pfb_strerror(L.PFB_ERR_NO_DEVICE) == b"no HIP device"
    assert lib.pfb_strerror(L.PFB_ERR_INTERNAL).startswith(b"internal") and lib.pfb_strerror(L.PFB_ERR_COMM).startswith(b"halo")
    for M, fs in ((8, 8e6), (56, 56e6), (5, 5.0)):
        fft = pkg.center_frequencies(M, fs)
        assert np.array_equal(fft, oracle.center_frequencies(M, fs))
        # centred order = the frequency of every column of fftshift(out, 2): ascending from -floor(M/2) * fs / M
        centered = pkg.center_frequencies(M, fs, order="centered")
        assert np.array_equal(centered, np.fft.fftshift(fft)) and np.all(np.diff(centered) > 0)
    out = np.zeros(4)
    assert lib.pfb_center_frequencies_ordered(4, 1.0, 2, out.ctypes.data_as(C.POINTER(C.c_double))) == L.PFB_ERR_BAD_ARG


def test_nothing_thrown_crosses_the_c_abi():
    """Every entry point that can allocate or throw runs under one guard (pfb_common.h, abi_guard); the diagnostic entry
    throws inside that same guard: bad_alloc -> PFB_ERR_NO_MEMORY, anything else -> PFB_ERR_INTERNAL, never an
    exception unwinding into the (C / ctypes / MATLAB) caller."""
    lib = L.load()
    assert lib.pfb_selftest_exception_guard(0) == L.PFB_ERR_NO_MEMORY
    assert lib.pfb_selftest_exception_guard(1) == L.PFB_ERR_INTERNAL
    assert lib.pfb_selftest_exception_guard(2) == L.PFB_ERR_INTERNAL
    assert lib.pfb_selftest_exception_guard(3) == L.PFB_OK
    # and every extern "C" definition in the product sources that builds C++ objects sits under the guard
    csrc = os.path.join(ROOT, "sdr_channelizer_amd", "csrc")
    api = open(os.path.join(csrc, "pfb_api.cpp")).read()
    for name in ("pfb_create", "pfb_process", "pfb_process_async", "pfb_process_iq_file", "pfb_pdw_from_iq_file",
                 "pfb_pdw_raw_from_iq_file", "pfb_prime", "pfb_set_option", "pfb_get_kernel_times", "pfb_shard_attach",
                 "pfb_process_shard_async"):
        body = api[api.index(f"\nint {name}("):]
        assert "abi_guard" in body[: body.index("\n}\n")].split("{", 1)[1][:80], name
    pdw = open(os.path.join(csrc, "pfb_pdw.hip")).read()
    for name in ("pfb_pdw_extract", "pfb_pdw_extract_raw", "pfb_pdw_release_workspace"):
        body = pdw[pdw.index(f'extern "C" int {name}('):]
        assert "abi_guard" in body[: body.index("\n}\n")], name


def test_shard_entry_points_validate_arguments():
    lib = L.load()
    cfg = L.PfbShardConfig(C.sizeof(L.PfbShardConfig), 0, 2, 0, L.HALO_EXCHANGE_FN(0), None)
    assert lib.pfb_shard_attach(None, C.byref(cfg)) == L.PFB_ERR_BAD_ARG
    assert lib.pfb_halo_samples(None) == 0 and lib.pfb_shard_head_frames(None) == 0
    assert lib.pfb_halo_recv_buffer(None) is None
    n = C.c_uint64()
    assert lib.pfb_process_shard_async(None, None, 0, None, 0, C.byref(n)) == L.PFB_ERR_BAD_ARG


def test_prototype_matches_oracle_design(oracle):
    for M, P in ((8, 12), (64, 12), (256, 8)):
        assert np.allclose(pkg.design_prototype(M, P), oracle.design_prototype(M, P), rtol=0, atol=2e-8)


def test_create_validates_arguments():
    lib = L.load()
    h = C.c_void_p()
    taps = np.zeros(64 * 12, np.float32)
    tp = taps.ctypes.data_as(C.POINTER(C.c_float))

    def cfg(**kw):
        d = dict(struct_size=C.sizeof(L.PfbConfig), num_channels=64, taps_per_channel=12, decimation=0, taps=tp,
                 sample_format=L.PFB_FMT_INT16_IQ, bit_width=12, output_layout=0, flags=0, input_offset=-1,
                 device_id=-1)
        d.update(kw)
        return L.PfbConfig(**d)

    assert lib.pfb_create(None, C.byref(h)) == L.PFB_ERR_BAD_ARG
    assert lib.pfb_create(C.byref(cfg(struct_size=8)), C.byref(h)) == L.PFB_ERR_BAD_ARG
    assert lib.pfb_create(C.byref(cfg(num_channels=1)), C.byref(h)) == L.PFB_ERR_BAD_ARG
    assert lib.pfb_create(C.byref(cfg(decimation=65)), C.byref(h)) == L.PFB_ERR_BAD_ARG
    assert lib.pfb_create(C.byref(cfg(input_offset=64)), C.byref(h)) == L.PFB_ERR_BAD_ARG
    assert lib.pfb_create(C.byref(cfg(flags=L.PFB_FLAG_POWER)), C.byref(h)) == L.PFB_ERR_BAD_ARG  # |y|^2 is an option of the magnitude output
    assert lib.pfb_create(C.byref(cfg(sample_format=7)), C.byref(h)) == L.PFB_ERR_BAD_FORMAT
    assert lib.pfb_create(C.byref(cfg(sample_format=L.PFB_FMT_INT8_IQ, bit_width=12)), C.byref(h)) == L.PFB_ERR_BAD_FORMAT
    assert lib.pfb_process(None, None, 0, None, 0, None, 0) == L.PFB_ERR_BAD_ARG


def test_no_cpu_fallback():
    """On a box without a HIP device the product refuses to compute instead of falling back."""
    lib = L.load()
    if lib.pfb_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.PfbError) as e:
        pkg.Channelizer(64)
    assert e.value.status == L.PFB_ERR_NO_DEVICE
    r = C.c_double()
    assert lib.pfb_measure_stream_copy(-1, 1 << 20, 1, C.byref(r)) == L.PFB_ERR_NO_DEVICE


def test_product_sources_do_not_touch_the_oracle():
    """oracle/ is test infrastructure: nothing under the package may import, link or call it."""
    for base, _, files in os.walk(os.path.join(ROOT, "sdr_channelizer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", ".c")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "pfb_oracle" not in text and "pfbo_" not in text and "from oracle" not in text, f
    # the same goes for the tools and the example; bench.py may use it in its cpu_baseline leg only
    for sub in ("tools", "examples", "include"):
        for base, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".sh", ".cpp", ".hip", ".h")):
                    text = open(os.path.join(base, f), errors="ignore").read()
                    assert "pfb_oracle" not in text and "pfbo_" not in text and "from oracle" not in text, f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert bench.count("from oracle") == 1 and bench.index("from oracle") > bench.index("def cpu_baseline")
    assert bench.index("from oracle") < bench.index("def main")


def test_fft_plan_model_covers_the_registered_plans():
    """tools/fft_plan_model.py is the executable statement of the LDS layout formulas (pos_i, twiddles, digit order)
    the fused kernels use: every plan registered in pfb_kernels.hip must reproduce numpy's inverse-type DFT."""
    import importlib.util
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fft_plan_model", os.path.join(root, "tools", "fft_plan_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rng = np.random.default_rng(3)
    modelled = set()
    for name, (M, R, RS, FS, C, NT) in m.PLANS.items():
        x = rng.standard_normal(M) + 1j * rng.standard_normal(M)
        y = m.run_plan(x, R, RS)
        ref = np.fft.ifft(x) * M
        assert np.abs(y - ref).max() / np.abs(ref).max() < 1e-12, name
        assert max(R[i] * RS[i] for i in range(len(R))) <= FS, name
        modelled.add((M, tuple(R), tuple(RS), FS))
    # every FastCfg<...> in the kernel table has its (M, radices, row strides, frame stride) in the model
    csrc = os.path.join(root, "sdr_channelizer_amd", "csrc")
    src = "".join(open(os.path.join(csrc, f)).read() for f in ("pfb_kernels.hip", "pfb_kernels_mid.hip", "pfb_kernels_big.hip",
                                                                       "pfb_kernels_mixed.hip"))
    assert src.count("FastCfg<") >= 40
    for args in re.findall(r"FastCfg<([^>]*)>", src):
        a = [t.strip() for t in args.split(",")]
        M, NP = int(a[0]), int(a[6])
        R = tuple(int(v) for v in a[7:7 + NP])
        RS = tuple(int(v) for v in a[10:10 + NP])
        assert (M, R, RS, int(a[13])) in modelled, args


def test_headers_compile_as_plain_c(tmp_path):
    """The boundary is a C ABI: the headers (pfb_channelizer.h pulls in pfb_iq_packet.h; the dev header too) must be valid C99 (the recorders are C++, MATLAB's loadlibrary parses C) and
    a C program must link against the library without a C++ runtime of its own."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "abi_c.c"
    src.write_text('#include "pfb_channelizer.h"\n#include "pfb_channelizer_dev.h"\n#include <stdio.h>\n'
                   'int main(void) {\n'
                   '  pfb_config cfg; pfb_shard_config sc; pfb_pdw w; pfb_iq_info info; double f[8];\n'
                   '  (void)cfg; (void)sc; (void)w; (void)info;\n'
                   '  if (pfb_abi_version() != PFB_ABI_VERSION) return 1;\n'
                   '  if (pfb_center_frequencies_ordered(8, 8e6, PFB_FREQ_ORDER_CENTERED, f) != PFB_OK || f[0] != -4e6) return 2;\n'
                   '  if (pfb_selftest_exception_guard(0) != PFB_ERR_NO_MEMORY) return 3;\n'
                   '  printf("%s\\n", pfb_strerror(PFB_ERR_COMM));\n  return 0;\n}\n')
    exe = tmp_path / "abi_c"
    libdir = os.path.dirname(L.LIB_PATH)
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L" + libdir, "-lpfb_channelizer", "-Wl,-rpath," + libdir], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "halo exchange" in r.stdout, (r.returncode, r.stdout, r.stderr)
