#!/usr/bin/env python3
"""build variants of libdisco_hip.so with extra -D flags for A/B timing in ONE gpurun call (boxes differ by a few per cent):
   python tools/ab_build.py NAME [-DFOO=1 ...]  ->  gpurun_tmp/lib_NAME.so ;  DISCO_LIB=gpurun_tmp/lib_NAME.so python bench.py ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from disco_amd import build  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
os.makedirs(os.path.join(ROOT, "gpurun_tmp"), exist_ok=True)
out = os.path.join(ROOT, "gpurun_tmp", f"lib_{name}.so")
cmd = [build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Rpass-analysis=kernel-resource-usage", "-o", out] + flags + \
      build.HIP_SOURCES + ["-L/opt/rocm/lib", "-lrccl", "-lrocprofiler-sdk-roctx", "-Wl,-rpath,/opt/rocm/lib"]
r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
if r.returncode:
    sys.exit(r.stderr[-3000:])
print(out, "spills:", build.vgpr_spills(r.stderr))
