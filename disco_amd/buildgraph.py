"""ctypes mirror of the C-ABI in include/disco_hip.h (libdisco_hip.so) — the product's Python host side.

The reference has no Python; this mirrors the order in which its main() drives the path
(/root/reference/src/BuildGraph/src/main.cpp:55-62: Dataset -> HashTable::insertDataset -> OverlapGraph) with the
same phase names the C-ABI uses.  There is NO CPU fallback: a missing extension or GPU is a loud error.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("DISCO_LIB") or os.path.join(_HERE, "libdisco_hip.so")  # DISCO_LIB: an experimental build (tools/ab_build.py)
_lib = None


class DiscoError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("min_overlap", C.c_uint32), ("max_edges_per_kmer", C.c_uint32), ("flags", C.c_uint32), ("max_substitutions", C.c_uint32)]


class GenSpecABI(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_reads", C.c_uint64), ("contig_len", C.c_uint64), ("n_contigs", C.c_uint32),
                ("len_min", C.c_uint32), ("len_max", C.c_uint32), ("skew", C.c_uint32)]


CONTAINED_DTYPE = np.dtype([("contained", "<u8"), ("super", "<u8"), ("orient", "<u4"), ("len2", "<u4"), ("len1", "<u4"),
                            ("start", "<u4"), ("j", "<u4"), ("type", "<u4")])
EDGE_DTYPE = np.dtype([("src", "<u8"), ("dst", "<u8"), ("orient", "<u4"), ("offset", "<u4"), ("len_src", "<u4"), ("len_dst", "<u4")])
COUNTER_NAMES = ("n_reads", "probes", "kmer_hits", "n_contained", "raw_hits", "e_pre", "e_out", "cap_bind_sites",
                 "asymmetric_pairs", "big_rows", "index_buckets", "hbm_bytes")


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_NAMES]


# every symbol include/disco_hip.h and include/disco_hip_test.h declare: (name, restype, argtypes)
_P = C.c_void_p
ABI = [
    ("disco_abi_version", C.c_int, []),
    ("disco_create", C.c_int, [C.c_int, C.POINTER(Params), C.POINTER(_P)]),
    ("disco_destroy", None, [_P]),
    ("disco_last_error", C.c_char_p, [_P]),
    ("disco_set_stream", C.c_int, [_P, _P]),
    ("disco_synchronize", C.c_int, [_P]),
    ("disco_pack_ascii", C.c_int, [C.c_char_p, C.c_uint32, _P]),
    ("disco_host_alloc", _P, [C.c_size_t]),
    ("disco_host_free", None, [_P]),
    ("disco_upload_reads", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint64]),
    ("disco_adopt_reads", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint64]),
    ("disco_generate_reads", C.c_int, [_P, C.POINTER(GenSpecABI)]),
    ("disco_substitute_bases", C.c_int, [_P, C.c_uint64, C.c_uint32]),
    ("disco_download_reads", C.c_int, [_P, _P, _P]),
    ("disco_stride_words", C.c_uint32, [_P]),
    ("disco_num_reads", C.c_uint64, [_P]),
    ("disco_long_rows", C.c_uint64, [_P]),
    ("disco_upload_reads_ragged", C.c_int, [_P, _P, _P, C.c_uint64]),
    ("disco_set_query_range", C.c_int, [_P, C.c_uint64, C.c_uint64]),
    ("disco_build_index", C.c_int, [_P]),
    ("disco_probe", C.c_int, [_P]),
    ("disco_mark_contained", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("disco_build_edges", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("disco_select_edges", C.c_int, [_P]),
    ("disco_symmetrize", C.c_int, [_P, C.c_int, C.POINTER(C.c_uint64)]),
    ("disco_merge_extras", C.c_int, [_P]),
    ("disco_transitive_reduce", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("disco_transitive_mark", C.c_int, [_P]),
    ("disco_emit_edges", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("disco_run_graph", C.c_int, [_P]),
    ("disco_adjacency_size", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("disco_export_adjacency", C.c_int, [_P, _P, _P]),
    ("disco_import_adjacency", C.c_int, [_P, _P, _P, C.c_uint64]),
    ("disco_fetch_contained", C.c_int64, [_P, _P, C.c_uint64]),
    ("disco_fetch_contained_grouped", C.c_int64, [_P, _P, C.c_uint64]),
    ("disco_fetch_edges", C.c_int64, [_P, _P, C.c_uint64]),
    ("disco_fetch_edge_substitutions", C.c_int64, [_P, _P, C.c_uint64]),
    ("disco_get_counters", C.c_int, [_P, C.POINTER(Counters)]),
    ("disco_phase_ms", C.c_int, [_P, C.POINTER(C.c_float), C.c_int]),
    ("disco_memcpy_d2d", C.c_int, [_P, _P, _P, C.c_uint64]),
    ("disco_fetch_edge_files", C.c_int64, [_P, C.c_uint32, _P, C.c_uint64]),
    ("disco_partition_edges", C.c_int64, [_P, _P, C.c_uint64, C.c_uint64, C.c_uint32, _P]),
    ("disco_format_edges", C.c_int64, [_P, C.c_uint32, _P, _P, _P]),
    ("disco_fetch_edge_text", C.c_int, [_P, _P, C.c_uint64]),
    ("disco_write_edge_text", C.c_int, [_P, _P, C.c_uint32, C.c_uint32]),
    ("disco_start_contained_rows", C.c_int, [_P, C.c_int]),
    ("disco_contract_chains", C.c_int, [_P, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("disco_contract_chains_of", C.c_int, [_P, _P, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("disco_fetch_chains", C.c_int, [_P, _P, _P, _P]),
    ("disco_set_query_order", C.c_int, [_P, _P]),
    ("disco_get_query_order", C.c_int, [_P, C.POINTER(_P)]),
    ("disco_probe_run_words", C.c_int, [_P]),
    ("disco_measure_hbm", C.c_int, [_P, C.c_uint64, C.c_int, C.POINTER(C.c_double)]),
    ("disco_measure_gather", C.c_int, [_P, C.c_uint64, C.c_int, C.POINTER(C.c_double)]),
    ("disco_comm_unique_id", C.c_int, [_P, C.c_size_t]),
    ("disco_comm_init", C.c_int, [_P, _P, C.c_int, C.c_int]),
    ("disco_comm_init_local", C.c_int, [C.POINTER(_P), C.c_int]),
    ("disco_comm_rank", C.c_int, [_P]),
    ("disco_comm_world", C.c_int, [_P]),
    ("disco_comm_kind", C.c_char_p, [_P]),
    ("disco_dist_range", C.c_int, [_P, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("disco_dist_upload_reads", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint64]),
    ("disco_dist_generate_reads", C.c_int, [_P, C.POINTER(GenSpecABI)]),
    ("disco_dist_run_graph", C.c_int, [_P, C.c_uint32]),
    ("disco_dist_get_info", C.c_int, [_P, _P]),
    ("disco_ingest_fasta", C.c_int, [_P, C.POINTER(C.c_char_p), C.c_int, C.c_uint32, _P, _P]),
    ("disco_ingest_fetch", C.c_int, [_P, _P, _P]),
]

ABI_VERSION = 2  # DISCO_ABI_VERSION of include/disco_hip.h (tests/test_abi.py keeps the two equal)
FLAG_TWO_PASS_VERIFY = 1  # DISCO_FLAG_TWO_PASS_VERIFY
XCHG = ("reads", "index_records", "index_shards", "contain", "row_requests", "row_data", "push", "adjacency", "twins", "queries", "hits", "keys", "reads_dealt", "contain_keys")
UNIQUE_ID_BYTES = 128
DIST_GATHER_READS = 1
DIST_KEEP_INDEX_PARTITIONED = 2


CHAIN_EDGE_DTYPE = np.dtype([("a", "<u8"), ("b", "<u8"), ("offset", "<u8"), ("orient", "<u4"), ("n_links", "<u4"), ("first_link", "<u8")])
CHAIN_LINK_DTYPE = np.dtype([("to", "<u4"), ("offset", "<u4"), ("orient", "<u4")])


class IngestFile(C.Structure):
    _fields_ = [("first_index", C.c_uint64), ("last_index", C.c_uint64), ("good", C.c_uint64), ("bad", C.c_uint64)]


class IngestInfo(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("total_records", C.c_uint64), ("too_long", C.c_uint64), ("stride_words", C.c_uint32), ("shortest", C.c_uint32),
                ("longest", C.c_uint32), ("read_s", C.c_float), ("device_s", C.c_float)]


class DistInfo(C.Structure):
    _fields_ = [("world", C.c_uint32), ("rank", C.c_uint32), ("n_reads", C.c_uint64), ("own_lo", C.c_uint64), ("own_hi", C.c_uint64),
                ("n_contained", C.c_uint64), ("e_pre", C.c_uint64), ("e_out", C.c_uint64), ("e_out_local", C.c_uint64),
                ("n_contained_local", C.c_uint64), ("cap_bind_sites", C.c_uint64), ("asymmetric_pairs", C.c_uint64),
                ("dropped_hits", C.c_uint64), ("probes", C.c_uint64), ("kmer_hits", C.c_uint64), ("regime", C.c_uint32),
                ("tr_rounds", C.c_uint32), ("tr_deferred", C.c_uint64), ("bytes_sent", C.c_uint64 * len(XCHG)),
                ("ms", C.c_float * len(XCHG)), ("ms_total", C.c_float), ("kernel_ms", C.c_float), ("comm_ops", C.c_uint32),
                ("host_syncs", C.c_uint32), ("device_allocs", C.c_uint32), ("device_frees", C.c_uint32), ("arena_bytes", C.c_uint64),
                ("arena_peak", C.c_uint64), ("hbm_peak", C.c_uint64), ("own_reads", C.c_uint64), ("placement", C.c_uint32), ("reserved_", C.c_uint32)]

PHASES = ("index", "probe_kernel", "verify", "contain", "select", "csr", "twin", "trmark", "emit", "order")


def lib_path() -> str:
    return _LIBPATH


def load():
    """dlopen libdisco_hip.so and bind the ABI. Raises DiscoError if the extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            raise DiscoError(f"{_LIBPATH} is missing: build it with `python -m disco_amd.build` (hipcc, gfx950). "
                             "There is no CPU fallback.")
        L = C.CDLL(_LIBPATH)
        L.disco_abi_version.restype = C.c_int
        if L.disco_abi_version() != ABI_VERSION:  # struct layouts below (DistInfo, ...) are those of exactly this version of include/disco_hip.h
            raise DiscoError(f"{_LIBPATH} speaks ABI version {L.disco_abi_version()}, this mirror was written against {ABI_VERSION}: rebuild "
                             "(`python -m disco_amd.build`)")
        for name, res, args in ABI:
            fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class BuildGraph:
    """One context = one GPU. Phases in the order the reference's main() runs them."""

    def __init__(self, min_overlap: int = 40, device: int = 0, max_edges_per_kmer: int = 4, flags: int = 0, max_substitutions: int = 0):
        self.L = load()
        self.min_overlap = min_overlap
        self._h = _P()
        p = Params(min_overlap, max_edges_per_kmer, flags, max_substitutions)
        rc = self.L.disco_create(device, C.byref(p), C.byref(self._h))
        if rc != 0:
            raise DiscoError(f"disco_create failed ({rc}): {self.L.disco_last_error(None).decode()}")

    # -- plumbing ---------------------------------------------------------------------------------------------
    def _chk(self, rc):
        if rc < 0:
            raise DiscoError(f"libdisco_hip error {rc}: {self.L.disco_last_error(self._h).decode()}")
        return rc

    def close(self):
        if self._h:
            self.L.disco_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, stream_ptr: int | None):
        self._chk(self.L.disco_set_stream(self._h, _P(stream_ptr) if stream_ptr else None))

    def synchronize(self):
        self._chk(self.L.disco_synchronize(self._h))

    # -- reads -------------------------------------------------------------------------------------------------
    def upload_reads(self, packed: np.ndarray, lens: np.ndarray):
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint16)
        assert packed.ndim == 2 and packed.shape[0] == lens.shape[0]
        self._keep = (packed, lens)
        self._chk(self.L.disco_upload_reads(self._h, packed.ctypes.data, packed.shape[1], lens.ctypes.data, packed.shape[0]))

    def upload_reads_ragged(self, words: np.ndarray, lens: np.ndarray):
        """reads back to back: read i = the ceil(lens[i] / 32) words behind those of read i - 1 (disco_upload_reads_ragged)"""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint16)
        assert words.ndim == 1 and len(words) == int(((lens.astype(np.int64) + 31) // 32).sum())
        self._chk(self.L.disco_upload_reads_ragged(self._h, words.ctypes.data, lens.ctypes.data, len(lens)))

    def upload_ascii(self, reads, ragged: bool = False):
        """pack upper-case ACGT strings on the host (disco_pack_ascii) and upload them (ragged: back to back, no stride)"""
        n = len(reads)
        lens = np.fromiter((len(r) for r in reads), dtype=np.uint16, count=n)
        if ragged:
            off = np.zeros(n + 1, dtype=np.int64)
            np.cumsum((lens.astype(np.int64) + 31) // 32, out=off[1:])
            words = np.zeros(max(int(off[-1]), 1), dtype=np.uint64)
            for i, r in enumerate(reads):
                b = r.encode()
                if self.L.disco_pack_ascii(b, len(b), words[off[i]:].ctypes.data) != 0:
                    raise DiscoError(f"read {i} contains a non-ACGT base")
            return self.upload_reads_ragged(words[:int(off[-1])], lens)
        stride = int((int(lens.max()) + 31) // 32) if n else 1
        packed = np.zeros((n, stride), dtype=np.uint64)
        for i, r in enumerate(reads):
            b = r.encode()
            if self.L.disco_pack_ascii(b, len(b), packed[i].ctypes.data) != 0:
                raise DiscoError(f"read {i} contains a non-ACGT base")
        self.upload_reads(packed, lens)

    def adopt_reads(self, d_packed_ptr: int, stride_words: int, d_len_ptr: int, n: int):
        self._chk(self.L.disco_adopt_reads(self._h, _P(d_packed_ptr), stride_words, _P(d_len_ptr), n))

    def generate_reads(self, spec):
        s = GenSpecABI(spec.seed, spec.n_reads, spec.contig_len, spec.n_contigs, spec.len_min, spec.len_max, int(spec.skew_word if hasattr(spec, "skew_word") else getattr(spec, "skew", 0)))
        self._chk(self.L.disco_generate_reads(self._h, C.byref(s)))

    def substitute_bases(self, seed: int, rate_ppm: int):
        """substitution errors into the resident reads (readgen.substitute is the numpy twin)"""
        self._chk(self.L.disco_substitute_bases(self._h, seed, rate_ppm))

    def download_reads(self):
        n, s = self.num_reads, self.stride_words
        packed = np.zeros((n, s), dtype=np.uint64)
        lens = np.zeros(n, dtype=np.uint16)
        self._chk(self.L.disco_download_reads(self._h, packed.ctypes.data, lens.ctypes.data))
        return packed, lens

    @property
    def num_reads(self) -> int:
        return int(self.L.disco_num_reads(self._h))

    @property
    def stride_words(self) -> int:
        return int(self.L.disco_stride_words(self._h))

    @property
    def long_rows(self) -> int:
        """reads of more than 256 bases that got rows of their own (two classes of rows, include/disco_hip.h); 0: one stride"""
        return int(self.L.disco_long_rows(self._h))

    def set_query_range(self, lo: int, hi: int):
        self._chk(self.L.disco_set_query_range(self._h, lo, hi))

    # -- phases ------------------------------------------------------------------------------------------------
    def build_index(self):
        self._chk(self.L.disco_build_index(self._h))

    def probe(self):
        self._chk(self.L.disco_probe(self._h))

    def mark_contained(self) -> int:
        n = C.c_uint64()
        self._chk(self.L.disco_mark_contained(self._h, C.byref(n)))
        return n.value

    def build_edges(self) -> int:
        n = C.c_uint64()
        self._chk(self.L.disco_build_edges(self._h, C.byref(n)))
        return n.value

    def select_edges(self):
        self._chk(self.L.disco_select_edges(self._h))

    def symmetrize(self, full: bool) -> int:
        n = C.c_uint64()
        self._chk(self.L.disco_symmetrize(self._h, 1 if full else 0, C.byref(n)))
        return n.value

    def merge_extras(self):
        self._chk(self.L.disco_merge_extras(self._h))

    def transitive_reduce(self) -> int:
        n = C.c_uint64()
        self._chk(self.L.disco_transitive_reduce(self._h, C.byref(n)))
        return n.value

    def transitive_mark(self):
        self._chk(self.L.disco_transitive_mark(self._h))

    def emit_edges(self) -> int:
        n = C.c_uint64()
        self._chk(self.L.disco_emit_edges(self._h, C.byref(n)))
        return n.value

    def run_graph(self):
        self._chk(self.L.disco_run_graph(self._h))

    # -- adjacency in / out ----------------------------------------------------------------------------------------
    def adjacency_size(self) -> int:
        n = C.c_uint64()
        self._chk(self.L.disco_adjacency_size(self._h, C.byref(n)))
        return n.value

    def export_adjacency(self, d_deg_ptr: int, d_entries_ptr: int):
        self._chk(self.L.disco_export_adjacency(self._h, _P(d_deg_ptr), _P(d_entries_ptr)))

    def import_adjacency(self, d_deg_all_ptr: int, d_entries_all_ptr: int, n_entries: int):
        self._chk(self.L.disco_import_adjacency(self._h, _P(d_deg_all_ptr), _P(d_entries_all_ptr), n_entries))

    # -- results -----------------------------------------------------------------------------------------------
    def fetch_contained(self) -> np.ndarray:
        n = self._chk(self.L.disco_fetch_contained(self._h, None, 0))
        out = np.zeros(n, dtype=CONTAINED_DTYPE)
        if n:
            self._chk(self.L.disco_fetch_contained(self._h, out.ctypes.data, n))
        return out

    def fetch_contained_grouped(self):
        """the contained rows in the order of the contained-read files (containing read, j, contained read), sorted on the device
        during the pass; None when that did not happen (DISCO_E_UNSUPPORTED: sort fetch_contained()'s rows instead)"""
        n = self._chk(self.L.disco_fetch_contained_grouped(self._h, None, 0))
        out = np.zeros(n, dtype=CONTAINED_DTYPE)
        if n:
            rc = self.L.disco_fetch_contained_grouped(self._h, out.ctypes.data, n)
            if rc == -6:
                return None
            self._chk(rc)
        return out

    def fetch_edges(self) -> np.ndarray:
        n = self._chk(self.L.disco_fetch_edges(self._h, None, 0))
        out = np.zeros(n, dtype=EDGE_DTYPE)
        if n:
            self._chk(self.L.disco_fetch_edges(self._h, out.ctypes.data, n))
        return out

    def format_edges(self, n_files: int = 1, edge_file=None, file_index=None):
        """the edge lines of the text files, formatted on the GPU (disco_format_edges): (text bytes, offsets [n_files + 1])"""
        off = np.zeros(n_files + 1, dtype=np.uint64)
        ef = None if edge_file is None else np.ascontiguousarray(edge_file, dtype=np.uint16)
        fi = None if file_index is None else np.ascontiguousarray(file_index, dtype=np.uint64)
        nb = self._chk(self.L.disco_format_edges(self._h, n_files, None if ef is None else ef.ctypes.data, None if fi is None else fi.ctypes.data, off.ctypes.data))
        buf = np.empty(max(nb, 1), dtype=np.uint8)
        self._chk(self.L.disco_fetch_edge_text(self._h, buf.ctypes.data, nb))
        return buf[:nb].tobytes(), off

    def write_edge_text(self, fds, threads: int = 8):
        """the text of the last format_edges straight from the device into open files: fds[f] receives file f (disco_write_edge_text)"""
        arr = (C.c_int * len(fds))(*fds)
        self._chk(self.L.disco_write_edge_text(self._h, arr, len(fds), threads))

    def start_contained_rows(self, grouped: bool = False):
        """the contained rows start their way to the host now (between mark_contained and the fetch)"""
        self._chk(self.L.disco_start_contained_rows(self._h, 1 if grouped else 0))

    def contract_chains(self, min_overlap_simplify: int = 0, edges=None):
        """chains of the reduced graph as composite edges (disco_contract_chains; with `edges`: disco_contract_chains_of on that array).
        Returns (composite edges [CHAIN_EDGE_DTYPE], links [CHAIN_LINK_DTYPE], absorbed flag per edge)"""
        nc, nl = C.c_uint64(), C.c_uint64()
        if edges is None:
            self._chk(self.L.disco_contract_chains(self._h, min_overlap_simplify, C.byref(nc), C.byref(nl)))
            ne = int(self.L.disco_fetch_edges(self._h, None, 0))
        else:
            edges = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
            self._chk(self.L.disco_contract_chains_of(self._h, edges.ctypes.data, len(edges), min_overlap_simplify, C.byref(nc), C.byref(nl)))
            ne = len(edges)
        comp = np.zeros(nc.value, dtype=CHAIN_EDGE_DTYPE)
        links = np.zeros(nl.value, dtype=CHAIN_LINK_DTYPE)
        absorbed = np.zeros(ne, dtype=np.uint8)
        self._chk(self.L.disco_fetch_chains(self._h, comp.ctypes.data if len(comp) else None, links.ctypes.data if len(links) else None,
                                            absorbed.ctypes.data if ne else None))
        return comp, links, absorbed

    def fetch_edge_substitutions(self) -> np.ndarray:
        """substitutions of every edge's overlap, in the order of fetch_edges (all 0 unless max_substitutions > 0)"""
        n = self._chk(self.L.disco_fetch_edge_substitutions(self._h, None, 0))
        out = np.zeros(n, dtype=np.uint16)
        if n:
            self._chk(self.L.disco_fetch_edge_substitutions(self._h, out.ctypes.data, n))
        return out

    def phase_ms(self) -> dict:
        a = (C.c_float * len(PHASES))()
        self._chk(self.L.disco_phase_ms(self._h, a, len(PHASES)))
        return {n: float(a[i]) for i, n in enumerate(PHASES)}

    def fetch_edge_files(self, n_files: int):
        """file index of every edge (order of fetch_edges): connected components dealt out to n_files files"""
        n = int(self.L.disco_fetch_edges(self._h, None, 0))
        out = np.zeros(max(n, 1), dtype=np.uint16)
        self._chk(self.L.disco_fetch_edge_files(self._h, n_files, out.ctypes.data, n))
        return out[:n]

    def partition_edges(self, edges: np.ndarray, n_nodes: int, n_files: int):
        """file index of every edge of a host array (EDGE_DTYPE): connected components dealt out to n_files files"""
        edges = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
        out = np.zeros(max(len(edges), 1), dtype=np.uint16)
        self._chk(self.L.disco_partition_edges(self._h, edges.ctypes.data, len(edges), n_nodes, n_files, out.ctypes.data))
        return out[:len(edges)]

    def get_query_order(self) -> int:
        """device pointer of the processing order the last probe walked (0 = file order); entries: read id | length << 32"""
        out = _P()
        self._chk(self.L.disco_get_query_order(self._h, C.byref(out)))
        return out.value or 0

    def set_query_order(self, d_order_ptr: int):
        self._chk(self.L.disco_set_query_order(self._h, _P(d_order_ptr)))

    def probe_run_words(self) -> int:
        """32-bit words of minimizer runs per read the last index build left for the probe (16 / 32), or 0: round 2's probe"""
        return int(self.L.disco_probe_run_words(self._h))

    def measure_hbm(self, nbytes: int = 4 << 30, reps: int = 5) -> float:
        """attainable HBM bandwidth in GB/s (read + write bytes of a streaming copy kernel)"""
        out = C.c_double(0.0)
        self._chk(self.L.disco_measure_hbm(self._h, nbytes, reps, C.byref(out)))
        return out.value

    def measure_gather(self, nbytes: int = 3 << 30, reps: int = 3) -> float:
        """attainable bandwidth in GB/s of random 64-byte row fetches out of a table of nbytes"""
        out = C.c_double(0.0)
        self._chk(self.L.disco_measure_gather(self._h, nbytes, reps, C.byref(out)))
        return out.value

    def memcpy_d2d(self, dst_ptr: int, src_ptr: int, nbytes: int):
        self._chk(self.L.disco_memcpy_d2d(self._h, _P(dst_ptr), _P(src_ptr), nbytes))

    # -- multi-GPU flow (one BuildGraph per rank; every call below is collective) ---------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        rc = load().disco_comm_unique_id(buf, UNIQUE_ID_BYTES)
        if rc != 0:
            raise DiscoError(f"disco_comm_unique_id failed ({rc})")
        return buf.raw

    def comm_init(self, unique_id: bytes, world: int, rank: int):
        """join the RCCL communicator (one rank per GPU)"""
        self._chk(self.L.disco_comm_init(self._h, C.c_char_p(unique_id), world, rank))

    @staticmethod
    def comm_init_local(graphs):
        """the ranks are the given contexts of this process, one host thread each (single-GPU boxes, tests)"""
        arr = (_P * len(graphs))(*[g._h for g in graphs])
        rc = load().disco_comm_init_local(arr, len(graphs))
        if rc != 0:
            raise DiscoError(f"disco_comm_init_local failed ({rc})")

    @property
    def rank(self) -> int:
        return int(self.L.disco_comm_rank(self._h))

    @property
    def world(self) -> int:
        return int(self.L.disco_comm_world(self._h))

    @property
    def transport(self) -> str:
        """the transport behind the communicator: rccl, loop (ranks of one process sharing a device) or none"""
        return self.L.disco_comm_kind(self._h).decode()

    def dist_range(self, n_total: int):
        lo, hi = C.c_uint64(), C.c_uint64()
        self._chk(self.L.disco_dist_range(self._h, n_total, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def dist_upload_reads(self, packed_own: np.ndarray, lens_own: np.ndarray, n_total: int, stride_words: int | None = None):
        packed_own = np.ascontiguousarray(packed_own, dtype=np.uint64)
        lens_own = np.ascontiguousarray(lens_own, dtype=np.uint16)
        self._keep = (packed_own, lens_own)
        stride = int(stride_words if stride_words is not None else packed_own.shape[1])
        self._chk(self.L.disco_dist_upload_reads(self._h, packed_own.ctypes.data, stride, lens_own.ctypes.data, n_total))

    def dist_upload_ascii(self, reads):
        """every rank passes ALL reads of the job; only its own range is packed and uploaded"""
        n = len(reads)
        lo, hi = self.dist_range(n)
        stride = int((max((len(r) for r in reads), default=1) + 31) // 32)
        packed = np.zeros((hi - lo, stride), dtype=np.uint64)
        lens = np.fromiter((len(r) for r in reads[lo:hi]), dtype=np.uint16, count=hi - lo)
        for i, r in enumerate(reads[lo:hi]):
            b = r.encode()
            if self.L.disco_pack_ascii(b, len(b), packed[i].ctypes.data) != 0:
                raise DiscoError(f"read {lo + i} contains a non-ACGT base")
        self.dist_upload_reads(packed, lens, n, stride)

    def dist_generate_reads(self, spec):
        s = GenSpecABI(spec.seed, spec.n_reads, spec.contig_len, spec.n_contigs, spec.len_min, spec.len_max, int(spec.skew_word if hasattr(spec, "skew_word") else getattr(spec, "skew", 0)))
        self._chk(self.L.disco_dist_generate_reads(self._h, C.byref(s)))

    def ingest_fasta(self, paths, threads: int = 16):
        """the input stage on the GPU (disco_ingest_fasta): FASTA files -> the context's read table. Returns (info dict, per-file
        dicts), or None when a file is not of the form the device stage accepts (the caller then takes the host stage)"""
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        info, files = IngestInfo(), (IngestFile * len(paths))()
        rc = self.L.disco_ingest_fasta(self._h, arr, len(paths), threads, C.byref(info), files)
        if rc == -6:  # DISCO_E_UNSUPPORTED
            return None
        self._chk(rc)
        return ({n: getattr(info, n) for n, _ in IngestInfo._fields_}, [{n: getattr(f, n) for n, _ in IngestFile._fields_} for f in files])

    def ingest_fetch(self):
        """(lengths, 1-based file indices) of the reads the last ingest_fasta kept"""
        n = self.num_reads
        ln, fi = np.zeros(n, dtype=np.uint16), np.zeros(n, dtype=np.uint64)
        self._chk(self.L.disco_ingest_fetch(self._h, ln.ctypes.data, fi.ctypes.data))
        return ln, fi

    def dist_run_graph(self, gather_reads: bool = True, partitioned_index: bool = False):
        """one collective pass; partitioned_index: the index stays hash-partitioned, lookups travel to the buckets' owners and the
        matching records back (every rank must pass the same flags)"""
        self._chk(self.L.disco_dist_run_graph(self._h, (DIST_GATHER_READS if gather_reads else 0) | (DIST_KEEP_INDEX_PARTITIONED if partitioned_index else 0)))

    def dist_info(self) -> dict:
        d = DistInfo()
        self._chk(self.L.disco_dist_get_info(self._h, C.byref(d)))
        out = {n: getattr(d, n) for n, _ in DistInfo._fields_ if n not in ("bytes_sent", "ms")}
        out["bytes_sent"] = {x: int(d.bytes_sent[i]) for i, x in enumerate(XCHG)}
        out["ms"] = {x: float(d.ms[i]) for i, x in enumerate(XCHG)}
        return out

    def host_to_host_pass(self, passes: int = 2) -> dict:
        """SURVEY.md §8(d) 'graph' wall: packed reads in (pinned) HOST memory -> upload -> whole graph pass -> contained rows and
        edges in HOST structs. Uses the reads the context currently holds: they are downloaded first (untimed) into pinned
        memory — at the words the rows use, as the input stage hands them over (disco_amd/host/fastx.cpp: 5 words at 150 bp) — and
        uploaded again (a read set of the same shape: the context keeps its buffers, and the caller its result arrays, as a service
        that processes one sample after the other would). Returns milliseconds per part of the LAST of `passes` passes."""
        import time

        n, s = self.num_reads, self.stride_words
        p = self.L.disco_host_alloc(max(n * s * 8, 8))
        if not p:
            raise DiscoError("disco_host_alloc failed")
        q = None
        try:
            lens = np.zeros(n, dtype=np.uint16)
            self._chk(self.L.disco_download_reads(self._h, p, lens.ctypes.data))
            w = max(1, (int(lens.max()) + 31) // 32) if n else 1
            src, sw = p, s
            if w < s:
                q = self.L.disco_host_alloc(max(n * w * 8, 8))
                if not q:
                    raise DiscoError("disco_host_alloc failed")
                full = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n, s))
                np.ctypeslib.as_array(C.cast(q, C.POINTER(C.c_uint64)), shape=(n, w))[:] = full[:, :w]
                src, sw = q, w
            rows = edges = None
            out, first = {}, None
            for ip in range(max(1, passes)):
                t0 = time.perf_counter()
                self._chk(self.L.disco_upload_reads(self._h, src, sw, lens.ctypes.data, n))
                t1 = time.perf_counter()
                self.run_graph()
                self.synchronize()
                t2 = time.perf_counter()
                nc = self._chk(self.L.disco_fetch_contained(self._h, None, 0))
                ne = self._chk(self.L.disco_fetch_edges(self._h, None, 0))
                if rows is None or len(rows) < nc or len(edges) < ne:
                    rows = np.zeros(max(nc, 1), dtype=CONTAINED_DTYPE)
                    edges = np.zeros(max(ne, 1), dtype=EDGE_DTYPE)
                t3 = time.perf_counter()
                if nc:
                    self._chk(self.L.disco_fetch_contained(self._h, rows.ctypes.data, nc))
                t4 = time.perf_counter()
                if ne:
                    self._chk(self.L.disco_fetch_edges(self._h, edges.ctypes.data, ne))
                t5 = time.perf_counter()
                out = {"upload_ms": (t1 - t0) * 1e3, "graph_ms": (t2 - t1) * 1e3, "fetch_contained_ms": (t4 - t3) * 1e3, "fetch_edges_ms": (t5 - t4) * 1e3,
                       "fetch_ms": (t5 - t3) * 1e3, "total_ms": (t2 - t0 + t5 - t3) * 1e3, "upload_bytes": int(n * sw * 8), "fetch_bytes": int(nc * 12 + ne * 12),
                       "result_bytes": int(nc * 40 + ne * 32), "n_contained": int(nc), "e_out": int(ne)}
                if ip == 0:
                    first = out["total_ms"]
            # the first pass pays the page faults of the caller's never-touched result arrays and the context's first allocations:
            # what a cold caller sees; total_ms is the steady state of a service that processes one sample after the other
            out["first_pass_total_ms"] = first
            return out
        finally:
            self.L.disco_host_free(p)
            if q:
                self.L.disco_host_free(q)

    def counters(self) -> dict:
        c = Counters()
        self._chk(self.L.disco_get_counters(self._h, C.byref(c)))
        return {n: int(getattr(c, n)) for n in COUNTER_NAMES}
