"""the synthetic read generator: numpy twin == C twin (CPU) == HIP kernel (gpu)"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from disco_amd import readgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SRC = r"""
#include "readgen.h"
void gen_codes(const disco_genspec *s, unsigned long long r0, unsigned long long r1, unsigned char *codes, unsigned long long *off)
{
    unsigned long long o = 0;
    for (unsigned long long r = r0; r < r1; r++) {
        disco_readloc loc = disco_read_location(s, r);
        off[r - r0] = o;
        for (unsigned i = 0; i < loc.len; i++) codes[o++] = (unsigned char)disco_read_base(s, &loc, i);
    }
    off[r1 - r0] = o;
}
"""


class Spec(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("n_reads", ctypes.c_uint64), ("contig_len", ctypes.c_uint64), ("n_contigs", ctypes.c_uint32),
                ("len_min", ctypes.c_uint32), ("len_max", ctypes.c_uint32), ("skew", ctypes.c_uint32)]


@pytest.mark.parametrize("kw", [dict(seed=42, n_reads=700, read_len=150, cov=30.0),
                                dict(seed=3, n_reads=500, read_len=100, cov=10.0, len_max=250, n_contigs=3),
                                dict(seed=9, n_reads=900, read_len=100, cov=10.0, len_max=250, n_contigs=7, skew=1),
                                dict(seed=10, n_reads=2000, read_len=150, cov=20.0, n_contigs=2, long_len=600, long_share=2000)])
def test_numpy_twin_equals_c_twin(tmp_path, kw):
    src = tmp_path / "g.c"
    src.write_text(C_SRC)
    so = str(tmp_path / "g.so")
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-I", os.path.join(ROOT, "disco_amd", "csrc"), "-o", so, str(src)])
    L = ctypes.CDLL(so)
    spec = readgen.GenSpec.coverage(**kw)
    codes, off = readgen.generate_codes(spec)
    s = Spec(spec.seed, spec.n_reads, spec.contig_len, spec.n_contigs, spec.len_min, spec.len_max, spec.skew_word)
    c2 = np.zeros(len(codes), dtype=np.uint8)
    o2 = np.zeros(len(off), dtype=np.uint64)
    L.gen_codes(ctypes.byref(s), ctypes.c_ulonglong(0), ctypes.c_ulonglong(spec.n_reads), c2.ctypes.data_as(ctypes.c_void_p), o2.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(off, o2) and np.array_equal(codes, c2)
    assert codes.max() <= 3 and len(set(np.diff(off).tolist())) >= (1 if spec.len_min == spec.len_max and not spec.long_share else 2)
    if spec.long_share:
        n_long = int((np.diff(off) == spec.long_len).sum())
        assert 0.5 * spec.n_reads * spec.long_share / 65536 < n_long < 2 * spec.n_reads * spec.long_share / 65536


@pytest.mark.gpu
@pytest.mark.parametrize("skew", [0, 1])
def test_hip_generator_equals_numpy_twin(skew):
    from disco_amd import buildgraph

    spec = readgen.GenSpec.coverage(seed=5, n_reads=4000, read_len=100, cov=20.0, len_max=250, n_contigs=5, skew=skew)
    codes, off = readgen.generate_codes(spec)
    with buildgraph.BuildGraph(min_overlap=40) as g:
        g.generate_reads(spec)
        packed, l2 = g.download_reads()
        want, lens = readgen.pack_reads(codes, off, stride_words=g.stride_words)  # device rows are padded to 64 B
    assert np.array_equal(lens, l2) and np.array_equal(want, packed)
