"""Build the native pieces of disco_amd in-tree (gfx950 only).

  libdisco_hip.so : HIP kernels + C-ABI (disco_amd/csrc/disco_hip.hip)          -> disco_amd/libdisco_hip.so
  buildG          : C++ host executable, drop-in for the reference buildG CLI  -> disco_amd/bin/buildG
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libdisco_hip.so")
BUILDG = os.path.join(HERE, "bin", "buildG")

HIP_SOURCES = [os.path.join(HERE, "csrc", "disco_hip.hip")]
HIP_DEPS = HIP_SOURCES + [os.path.join(HERE, "csrc", f) for f in ("disco_kernels.h", "disco_device.h", "readgen.h")] + [
    os.path.join(ROOT, "include", "disco_hip.h")]
HOST_SOURCES = [os.path.join(HERE, "host", f) for f in ("buildg_main.cpp", "fastx.cpp", "writer.cpp")]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if force or _stale(LIB, HIP_DEPS):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", LIB] + HIP_SOURCES
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_host(force: bool = False, verbose: bool = False) -> str:
    srcs = [s for s in HOST_SOURCES if os.path.exists(s)]
    if not srcs:
        return ""
    deps = srcs + [os.path.join(HERE, "host", f) for f in os.listdir(os.path.join(HERE, "host")) if f.endswith(".h")] + [LIB]
    if force or _stale(BUILDG, deps):
        os.makedirs(os.path.dirname(BUILDG), exist_ok=True)
        cmd = ["g++", "-O2", "-std=c++17", "-fopenmp", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", BUILDG] + srcs + [
            "-L", HERE, "-ldisco_hip", "-lz", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + HERE]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    dump = os.path.join(HERE, "bin", "fastx_dump")
    dsrc = [os.path.join(HERE, "host", "fastx_dump.cpp"), os.path.join(HERE, "host", "fastx.cpp")]
    if force or _stale(dump, dsrc + [os.path.join(HERE, "host", "fastx.h")]):
        cmd = ["g++", "-O2", "-std=c++17", "-fopenmp", "-Wall", "-o", dump] + dsrc + ["-lz"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    gen = os.path.join(HERE, "bin", "readgen")
    gsrc = [os.path.join(HERE, "host", "readgen_main.cpp"), os.path.join(HERE, "csrc", "readgen.h")]
    if force or _stale(gen, gsrc):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-o", gen, gsrc[0]]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return BUILDG


def build_all(force: bool = False, verbose: bool = False):
    build_lib(force, verbose)
    build_host(force, verbose)


if __name__ == "__main__":
    build_all(force=True, verbose=True)
