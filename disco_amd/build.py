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
HIP_DEPS = HIP_SOURCES + [os.path.join(HERE, "csrc", f) for f in sorted(f for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith(".h"))] + [
    os.path.join(ROOT, "include", "disco_hip.h"), os.path.join(ROOT, "include", "disco_hip_test.h")]
HOST_SOURCES = [os.path.join(HERE, "host", f) for f in ("buildg_main.cpp", "fastx.cpp", "writer.cpp", "parsimple.cpp")]


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


def vgpr_spills(remarks: str) -> dict:
    """kernel -> spilled vector registers, from hipcc's -Rpass-analysis=kernel-resource-usage remarks"""
    out, name = {}, None
    for line in remarks.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split("[")[0].strip()
        elif "VGPRs Spill:" in line and name:
            n = int(line.split("VGPRs Spill:")[1].split("[")[0])
            if n:
                out[name] = n
    return out


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """The wave-per-read kernels read values out of lanes that are inactive where the value was last written (v_readlane of
    the chunk registers, DPP row scans). A vector register that is spilled and reloaded under a partial exec mask loses exactly
    those lanes, so a spilling build computes WRONG graphs (measured: verify_kernel<5> forced to 6 waves per SIMD spilled 16
    registers and returned 46.1 M instead of 45.3 M edges). The build therefore fails when any kernel spills vector registers
    (DISCO_ALLOW_SPILLS=1 to experiment)."""
    if force or _stale(LIB, HIP_DEPS):
        tmp = LIB + ".tmp"
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Rpass-analysis=kernel-resource-usage",
               "-o", tmp] + HIP_SOURCES + ["-L/opt/rocm/lib", "-lrccl", "-lrocprofiler-sdk-roctx", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + "\n".join(l for l in r.stderr.splitlines() if "-Rpass-analysis" not in l)[-4000:])
        spills = vgpr_spills(r.stderr)
        if spills and not os.environ.get("DISCO_ALLOW_SPILLS"):
            os.unlink(tmp)
            raise RuntimeError(f"kernels spill vector registers (cross-lane reads would return garbage): {spills}")
        os.replace(tmp, LIB)
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
    ps = os.path.join(HERE, "bin", "parsimplify")
    psrc = [os.path.join(HERE, "host", f) for f in ("parsimplify_main.cpp", "parsimple.cpp", "fastx.cpp", "writer.cpp")]
    if force or _stale(ps, psrc + [os.path.join(HERE, "host", f) for f in ("parsimple.h", "writer.h", "fastx.h")]):
        cmd = ["g++", "-O2", "-std=c++17", "-fopenmp", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", ps] + psrc + ["-lz"]
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
