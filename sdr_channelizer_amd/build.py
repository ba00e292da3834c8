"""In-tree build of libpfb_channelizer.so (HIP kernels + C ABI) for gfx950.

``python -m sdr_channelizer_amd.build`` or ``__graft_entry__.build()``.  hipcc
cross-compiles without a GPU; the resulting .so is git-ignored but travels to
the GPU box with the source snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libpfb_channelizer.so")
SOURCES = ["pfb_api.cpp", "pfb_kernels.hip", "pfb_kernels_mid.hip", "pfb_kernels_big.hip", "pfb_kernels_mixed.hip", "pfb_pdw.hip", "iq_packet.c"]
HEADERS = ["pfb_common.h", "pfb_fast.hpp", "pfb_table.h"]
ARCH = "gfx950"
PUBLIC_HEADERS = ["pfb_channelizer.h", "pfb_channelizer_dev.h", "pfb_iq_packet.h"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the channelizer has no CPU build")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps += [os.path.join(ROOT, "include", f) for f in PUBLIC_HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs = []
    bdir = os.path.join(PKG, "build")
    os.makedirs(bdir, exist_ok=True)
    common = ["-O3", "-fPIC", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}"] + os.environ.get("PFB_EXTRA_CXXFLAGS", "").split()
    jobs = []
    hdr_time = max(os.path.getmtime(h) for h in [os.path.join(CSRC, f) for f in HEADERS] +
                   [os.path.join(ROOT, "include", f) for f in PUBLIC_HEADERS])
    for src in SOURCES:
        obj = os.path.join(bdir, src + ".o")
        path = os.path.join(CSRC, src)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_time):
            continue  # this translation unit is up to date
        if src.endswith(".c"):
            cmd = [hipcc, "-x", "c", "-std=c11"] + common + ["-c", path, "-o", obj]
        else:
            cmd = [hipcc, "-x", "hip", f"--offload-arch={ARCH}", "-std=c++20"] + common + ["-c", path, "-o", obj]
        jobs.append(cmd)

    def compile_one(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    # the kernel translation units dominate (tens of seconds each): compile them side by side
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(5, os.cpu_count() or 1)) as pool:
        list(pool.map(compile_one, jobs))
    tmp = LIB + ".tmp"
    subprocess.check_call([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", tmp] + objs)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
