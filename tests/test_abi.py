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
