"""ctypes binding of oracle/libdisco_oracle.so (the C restatement, disco_oracle.c) + canonical forms.

TEST INFRASTRUCTURE ONLY — the checker, never the product path.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class ContainedRow(C.Structure):
    _fields_ = [("contained", C.c_uint64), ("super", C.c_uint64), ("orient", C.c_uint32), ("len2", C.c_uint32),
                ("len1", C.c_uint32), ("start", C.c_uint32), ("j", C.c_uint32), ("type", C.c_uint32)]


class Edge(C.Structure):
    _fields_ = [("src", C.c_uint64), ("dst", C.c_uint64), ("orient", C.c_uint32), ("offset", C.c_uint32),
                ("len_src", C.c_uint32), ("len_dst", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_reads", "probes", "kmer_hits", "n_contained", "e_pre", "e_out",
                                          "cap_bind_sites", "asymmetric_pairs")]


class Result(C.Structure):
    _fields_ = [("contained", C.POINTER(ContainedRow)), ("edges", C.POINTER(Edge)), ("c", Counters),
                ("edge_subs", C.POINTER(C.c_uint32))]


CONTAINED_DTYPE = np.dtype([("contained", "<u8"), ("super", "<u8"), ("orient", "<u4"), ("len2", "<u4"),
                            ("len1", "<u4"), ("start", "<u4"), ("j", "<u4"), ("type", "<u4")])
EDGE_DTYPE = np.dtype([("src", "<u8"), ("dst", "<u8"), ("orient", "<u4"), ("offset", "<u4"),
                       ("len_src", "<u4"), ("len_dst", "<u4")])


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libdisco_oracle.so")
    src = os.path.join(_HERE, "disco_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libdisco_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.oracle_build_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(Result)]
        L.oracle_build_graph.restype = C.c_int
        L.oracle_build_graph_inexact.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Result)]
        L.oracle_build_graph_inexact.restype = C.c_int
        L.oracle_free_result.argtypes = [C.POINTER(Result)]
        L.oracle_test_read.argtypes = [C.c_char_p, C.c_size_t]
        L.oracle_test_read.restype = C.c_int
        L.oracle_encode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        L.oracle_encode.restype = C.c_int
        _LIB = L
    return _LIB


_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_char), C.c_size_t)


def parse_records(data: bytes):
    """oracle_parse_records: list of raw sequences (bytes), in file order."""
    L = lib()
    out = []

    def cb(_u, p, n):
        out.append(C.string_at(p, n))

    L.oracle_parse_records.argtypes = [C.c_char_p, C.c_size_t, _CB, C.c_void_p]
    L.oracle_parse_records.restype = C.c_long
    rc = L.oracle_parse_records(data, len(data), _CB(cb), None)
    if rc < 0:
        raise ValueError("Unknown input file format.")
    return out


def test_read(seq: str) -> bool:
    b = seq.encode()
    return bool(lib().oracle_test_read(b, len(b)))


def load_good_reads(paths, min_overlap: int):
    """Dataset ctor semantics (BG/Dataset.cpp:109-128,294,305): returns (reads, file_index) of the good reads;
    file_index is 1-based over ALL records of all files in order."""
    reads, fidx = [], []
    idx = 0
    for p in paths:
        with open(p, "rb") as f:
            data = f.read()
        for rec in parse_records(data):
            idx += 1
            s = rec.decode("latin-1").upper()
            if len(s) > min_overlap and test_read(s):
                reads.append(s)
                fidx.append(idx)
    return reads, np.asarray(fidx, dtype=np.uint64), idx


_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


def encode_reads(reads):
    lens = np.fromiter((len(r) for r in reads), dtype=np.uint64, count=len(reads))
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    codes = _CODE[np.frombuffer("".join(reads).encode(), dtype=np.uint8)]
    assert codes.size == 0 or codes.max() < 4
    return np.ascontiguousarray(codes), off


def build_graph_inexact(codes, off, min_overlap: int, max_subs: int):
    """the inexact-overlap EXTENSION (disco_oracle.h; the reference has no such mode): (rows, edges, counters, substitutions per edge)"""
    return build_graph(codes, off, min_overlap, max_subs=max_subs, with_subs=True)


def build_graph(codes, off, min_overlap: int, count_hits: bool = False, max_subs: int = 0, with_subs: bool = False):
    """returns (contained rows [CONTAINED_DTYPE], edges [EDGE_DTYPE], counters dict); ids are 0-based good-read ranks."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    res = Result()
    if max_subs:
        rc = lib().oracle_build_graph_inexact(codes.ctypes.data, off.ctypes.data, len(off) - 1, min_overlap,
                                              1 if count_hits else 0, max_subs, C.byref(res))
    else:
        rc = lib().oracle_build_graph(codes.ctypes.data, off.ctypes.data, len(off) - 1, min_overlap,
                                      1 if count_hits else 0, C.byref(res))
    if rc != 0:
        raise ValueError(f"oracle_build_graph: {'bad arguments' if rc == -1 else 'an edge exceeds the substitution threshold'}")
    nc, ne = res.c.n_contained, res.c.e_out
    rows = np.ctypeslib.as_array(C.cast(res.contained, C.POINTER(C.c_uint8)), (nc * C.sizeof(ContainedRow),)).copy().view(
        CONTAINED_DTYPE) if nc else np.zeros(0, CONTAINED_DTYPE)
    edges = np.ctypeslib.as_array(C.cast(res.edges, C.POINTER(C.c_uint8)), (ne * C.sizeof(Edge),)).copy().view(
        EDGE_DTYPE) if ne else np.zeros(0, EDGE_DTYPE)
    counters = {n: int(getattr(res.c, n)) for n, _ in Counters._fields_}
    subs = np.ctypeslib.as_array(res.edge_subs, (ne,)).copy() if ne else np.zeros(0, np.uint32)
    lib().oracle_free_result(C.byref(res))
    return (rows, edges, counters, subs) if with_subs else (rows, edges, counters)


# ----------------------------------------------------------------------------------------------------------------
# canonical forms: what "bit-identical edge list" means
# ----------------------------------------------------------------------------------------------------------------
def canonical_edges(src_f, dst_f, orient, offset, len_src, len_dst):
    """edge tuples in FILE indices -> sorted unique array of (src,dst,orient,ovl,len1,start1,len2), src<dst."""
    src_f = np.asarray(src_f, dtype=np.int64)
    dst_f = np.asarray(dst_f, dtype=np.int64)
    orient = np.asarray(orient, dtype=np.int64)
    offset = np.asarray(offset, dtype=np.int64)
    l1 = np.asarray(len_src, dtype=np.int64)
    l2 = np.asarray(len_dst, dtype=np.int64)
    swap = src_f > dst_f
    tw = np.array([3, 1, 2, 0])[orient]
    o2 = np.where(swap, tw, orient)
    off2 = np.where(swap, l2 + offset - l1, offset)
    a = np.where(swap, dst_f, src_f)
    b = np.where(swap, src_f, dst_f)
    la = np.where(swap, l2, l1)
    lb = np.where(swap, l1, l2)
    t = np.stack([a, b, o2, la - off2, la, off2, lb], axis=1) if len(a) else np.zeros((0, 7), np.int64)
    return np.unique(t, axis=0)


def canonical_contained(contained_f, super_f, orient, len2, len1, start):
    t = np.stack([np.asarray(x, dtype=np.int64) for x in (contained_f, super_f, orient, len2, len1, start)], axis=1) \
        if len(contained_f) else np.zeros((0, 6), np.int64)
    return np.unique(t, axis=0)


def edges_text(canon) -> str:
    """reference line format (BG/OverlapGraph.cpp:864-867) with the batch flag omitted."""
    return "".join(f"{s}\t{d}\t{o},{ovl},0,0,{l1},{st},{l1 - 1},{l2},0,{ovl - 1},NA\n" for s, d, o, ovl, l1, st, l2 in canon)


def contained_text(canon) -> str:
    """reference row format (BG/OverlapGraph.cpp:438-447)."""
    return "".join(f"{c}\t{s}\t{o},{l2},0,0,{l2},0,{l2},{l1},{st},{st + l2}\n" for c, s, o, l2, l1, st in canon)


def digest(text: str) -> str:
    return hashlib.sha256(text.encode()).hexdigest()


def digest_array(canon) -> str:
    """digest of a canonical (sorted, unique) int64 array — for cases too large to serialise as text"""
    a = np.ascontiguousarray(canon, dtype=np.int64)
    return hashlib.sha256(a.tobytes()).hexdigest()


def canonical_edges_large(src_f, dst_f, orient, offset, len_src, len_dst):
    """canonical_edges for tens of millions of edges: same result, sorted with lexsort instead of np.unique(axis=0)"""
    src_f = np.asarray(src_f, dtype=np.int64)
    dst_f = np.asarray(dst_f, dtype=np.int64)
    orient = np.asarray(orient, dtype=np.int64)
    offset = np.asarray(offset, dtype=np.int64)
    l1 = np.asarray(len_src, dtype=np.int64)
    l2 = np.asarray(len_dst, dtype=np.int64)
    swap = src_f > dst_f
    tw = np.array([3, 1, 2, 0])[orient]
    o2 = np.where(swap, tw, orient)
    off2 = np.where(swap, l2 + offset - l1, offset)
    a = np.where(swap, dst_f, src_f)
    b = np.where(swap, src_f, dst_f)
    la = np.where(swap, l2, l1)
    lb = np.where(swap, l1, l2)
    t = np.stack([a, b, o2, la - off2, la, off2, lb], axis=1)
    # one edge per read pair is the rule: then (a, b) alone decides the order and one 64-bit key sort does it (seconds at
    # 45 M edges, where the 7-key lexsort takes minutes); pairs that occur twice fall back to the full lexsort
    order = None
    if len(t) and a.min() >= 0 and b.min() >= 0 and a.max() < (1 << 31) and b.max() < (1 << 32):
        key = (a.astype(np.uint64) << np.uint64(32)) | b.astype(np.uint64)
        order = np.argsort(key, kind="stable")
        ks = key[order]
        if len(ks) > 1 and np.any(ks[1:] == ks[:-1]):
            order = None
    if order is None:
        order = np.lexsort((t[:, 6], t[:, 5], t[:, 4], t[:, 3], t[:, 2], t[:, 1], t[:, 0]))
    t = t[order]
    if len(t) > 1:
        keep = np.ones(len(t), dtype=bool)
        keep[1:] = np.any(t[1:] != t[:-1], axis=1)
        t = t[keep]
    return t


def oracle_canonical(reads, file_index, min_overlap: int, count_hits: bool = False):
    """run the C oracle on good reads; returns (canonical edges, canonical contained rows, counters)."""
    codes, off = encode_reads(reads)
    rows, edges, counters = build_graph(codes, off, min_overlap, count_hits)
    fi = np.asarray(file_index, dtype=np.int64)
    ce = canonical_edges(fi[edges["src"].astype(np.int64)], fi[edges["dst"].astype(np.int64)], edges["orient"],
                         edges["offset"], edges["len_src"], edges["len_dst"])
    cc = canonical_contained(fi[rows["contained"].astype(np.int64)], fi[rows["super"].astype(np.int64)], rows["orient"],
                             rows["len2"], rows["len1"], rows["start"])
    return ce, cc, counters
