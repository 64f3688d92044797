"""CPU: the C-ABI library loads and exports every symbol include/disco_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from disco_amd import build, buildgraph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ("disco_hip.h", "disco_hip_test.h")  # the boundary a BuildGraph host binds; the bench / test-only entry points of the same library


def declared_functions(headers=HEADERS):
    names = []
    for f in headers:
        h = open(os.path.join(ROOT, "include", f)).read()
        h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
        names += re.findall(r"^\s*(?:const\s+)?(?:int|void|int64_t|uint32_t|uint64_t|char)\s*\*?\s*(disco_\w+)\s*\(", h, flags=re.M)
    return sorted(set(names))


def header_abi_version():
    return int(re.search(r"^#define DISCO_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "disco_hip.h")).read(), flags=re.M).group(1))


def test_library_builds_and_exports_every_declared_symbol():
    so = build.build_lib()
    assert os.path.exists(so)
    L = ctypes.CDLL(so)
    decl = declared_functions()
    assert len(decl) >= 30, decl
    missing = [n for n in decl if not hasattr(L, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    # round 6: version 2 — disco_dist_info grew in the middle in round 5; library, header and the Python mirror carry ONE number, and the
    # mirror / buildG refuse a library that answers another
    assert L.disco_abi_version() == header_abi_version() == buildgraph.ABI_VERSION == 2


def test_test_only_entry_points_live_in_their_own_header():
    """what bench.py and the tests need beyond the reference's interface (synthetic reads, substitutions, bandwidth probes) is declared in
    include/disco_hip_test.h, not in the boundary header a BuildGraph host binds"""
    product, test_only = declared_functions(("disco_hip.h",)), declared_functions(("disco_hip_test.h",))
    assert set(test_only) == {"disco_generate_reads", "disco_substitute_bases", "disco_dist_generate_reads", "disco_measure_hbm", "disco_measure_gather", "disco_probe_run_words"}
    assert not set(product) & set(test_only)
    for host in ("buildg_main.cpp", "writer.cpp", "parsimple.cpp", "fastx.cpp"):  # the drop-in executable binds the product header only
        txt = open(os.path.join(ROOT, "disco_amd", "host", host)).read()
        assert "disco_hip_test.h" not in txt and not any(n + "(" in txt for n in test_only), host


def test_dist_info_mirror_matches_the_header_layout(tmp_path):
    """sizeof / offsetof of disco_dist_info as a C compiler lays it out == the ctypes mirror (the struct grew in the middle once: ADVICE r5)"""
    import subprocess

    src = tmp_path / "lay.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "disco_hip.h"
int main(void)
{
    printf("%zu %zu %zu %zu %zu %zu %d\n", sizeof(disco_dist_info), offsetof(disco_dist_info, bytes_sent), offsetof(disco_dist_info, ms),
           offsetof(disco_dist_info, ms_total), offsetof(disco_dist_info, hbm_peak), offsetof(disco_dist_info, placement), (int)DISCO_X_COUNT);
    return 0;
}
''')
    exe = tmp_path / "lay"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = [int(x) for x in subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True, check=True).stdout.split()]
    D = buildgraph.DistInfo
    assert got == [ctypes.sizeof(D), D.bytes_sent.offset, D.ms.offset, D.ms_total.offset, D.hbm_peak.offset, D.placement.offset, len(buildgraph.XCHG)]


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
