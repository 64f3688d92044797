"""CPU: the C-ABI library loads and exports every symbol include/disco_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from disco_amd import build, buildgraph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    h = open(os.path.join(ROOT, "include", "disco_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|void|int64_t|uint32_t|uint64_t|char)\s*\*?\s*(disco_\w+)\s*\(", h, flags=re.M)
    return sorted(set(names))


def test_library_builds_and_exports_every_declared_symbol():
    so = build.build_lib()
    assert os.path.exists(so)
    L = ctypes.CDLL(so)
    decl = declared_functions()
    assert len(decl) >= 30, decl
    missing = [n for n in decl if not hasattr(L, n)]
    assert not missing, f"declared in disco_hip.h but not exported: {missing}"
    assert L.disco_abi_version() == 1


def test_python_mirror_binds_the_same_set():
    bound = sorted(n for n, _r, _a in buildgraph.ABI)
    assert bound == declared_functions()
    buildgraph.load()


def test_pack_ascii_host_helper():
    import numpy as np

    L = buildgraph.load()
    out = np.zeros(2, dtype=np.uint64)
    seq = b"ACGT" * 9 + b"GA"  # 38 bases -> 2 words
    assert L.disco_pack_ascii(seq, len(seq), out.ctypes.data) == 0
    want0 = 0
    for i, c in enumerate(seq[:32]):
        want0 |= b"ACGT".index(c) << (62 - 2 * i)
    assert int(out[0]) == want0
    assert int(out[1]) == (0 << 62 | 1 << 60 | 2 << 58 | 3 << 56 | 2 << 54 | 0 << 52)
    assert L.disco_pack_ascii(b"ACGN", 4, out.ctypes.data) != 0


def test_no_product_code_touches_the_oracle():
    """the product path must never route through the checker"""
    bad = []
    for d, _dirs, files in os.walk(os.path.join(ROOT, "disco_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                txt = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "disco_oracle" in txt or "libdisco_oracle" in txt:
                    bad.append(f)
    assert not bad, bad


def test_a_plain_c_caller_compiles_links_and_runs(tmp_path):
    """the boundary is a C ABI: a C99 program includes include/disco_hip.h, links libdisco_hip.so and calls the host-only entry
    points (no GPU needed: version, packing, error text)"""
    import subprocess

    src = tmp_path / "caller.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "disco_hip.h"
int main(void)
{
    uint64_t w[2] = {0, 0};
    if (disco_abi_version() <= 0) return 2;
    if (disco_pack_ascii("ACGTACGTACGTACGTACGTACGTACGTACGTA", 33, w) != 0) return 3; /* MSB first, A0 C1 G2 T3 */
    if (w[0] != 0x1B1B1B1B1B1B1B1BULL || w[1] != 0) return 4;
    if (disco_pack_ascii("ACGN", 4, w) == 0) return 5;                               /* non-ACGT is an error */
    if (disco_last_error(NULL) == NULL) return 6;
    printf("abi %d\n", disco_abi_version());
    return 0;
}
''')
    exe = tmp_path / "caller"
    libdir = os.path.join(ROOT, "disco_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src),
                           "-L", libdir, "-ldisco_hip", "-Wl,-rpath," + libdir])
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0 and out.stdout.startswith("abi "), (out.returncode, out.stdout)


def test_build_rejects_vector_register_spills():
    """the kernels read lanes that are inactive where a value was written; a spilled vector register loses them, so
    disco_amd/build.py refuses a build whose resource remarks report VGPR spills (parser check; the shipped build has none)"""
    from disco_amd import build

    remarks = "\n".join([
        "x.h:1:1: remark: Function Name: _Z13verify_kernelILi5EEv10VerifyArgs [-Rpass-analysis=kernel-resource-usage]",
        "x.h:1:1: remark:     VGPRs: 80 [-Rpass-analysis=kernel-resource-usage]",
        "x.h:1:1: remark:     SGPRs Spill: 12 [-Rpass-analysis=kernel-resource-usage]",
        "x.h:1:1: remark:     VGPRs Spill: 16 [-Rpass-analysis=kernel-resource-usage]",
        "x.h:1:1: remark: Function Name: _Z12probe_kernelILb0ELb1ELb1EEv9ProbeArgs [-Rpass-analysis=kernel-resource-usage]",
        "x.h:1:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]",
    ])
    assert build.vgpr_spills(remarks) == {"_Z13verify_kernelILi5EEv10VerifyArgs": 16}
