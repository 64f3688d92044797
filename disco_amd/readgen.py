"""numpy twin of disco_amd/csrc/readgen.h — the deterministic synthetic read generator.

Replaces bbmap/randomreads.sh (Java; absent here) for the BASELINE configs.
Bit-identical to the C/HIP version: every quantity is a pure function of (seed, index).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _u64(x):
    return np.asarray(x, dtype=np.uint64)


def mix64(x):
    """splitmix64 finaliser (disco_mix64)."""
    with np.errstate(over="ignore"):
        x = _u64(x) + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def genome_word(seed, w):
    with np.errstate(over="ignore"):
        s = np.uint64(seed) * np.uint64(0xD1342543DE82EF95)
        return mix64(s ^ (_u64(w) + np.uint64(0x632BE59BD9B4E019)))


def genome_bases(seed, g):
    """base codes (A0 C1 G2 T3) at absolute genome coordinates g (array)."""
    g = _u64(g)
    w = genome_word(seed, g >> np.uint64(5))
    sh = np.uint64(62) - np.uint64(2) * (g & np.uint64(31))
    return ((w >> sh) & np.uint64(3)).astype(np.uint8)


@dataclass
class GenSpec:
    seed: int
    n_reads: int
    contig_len: int
    n_contigs: int = 1
    len_min: int = 150
    len_max: int = 150
    skew: int = 0  # 1: metagenome-like contig abundances (product of two uniform draws, see csrc/readgen.h)
    long_len: int = 0  # a tail of long reads: their length ...
    long_share: int = 0  # ... and their share of the reads in 1 / 65536 (which reads: a pure function of seed and read)

    @property
    def skew_word(self) -> int:
        """the ABI's `skew` field (csrc/readgen.h: DISCO_GEN_SKEW_WORD)"""
        return (1 if self.skew else 0) | (int(self.long_len) << 1) | (int(self.long_share) << 16)

    @property
    def longest(self) -> int:
        return max(self.len_max, self.long_len if self.long_share else 0)

    @staticmethod
    def coverage(seed: int, n_reads: int, read_len: int = 150, cov: float = 30.0, n_contigs: int = 1,
                 len_max: int | None = None, skew: int = 0, long_len: int = 0, long_share: int = 0) -> "GenSpec":
        """uniform-random genome sized for the given (mean) coverage (skew=1: config 5)."""
        len_max = read_len if len_max is None else len_max
        f = long_share / 65536.0
        mean = (1 - f) * (read_len + len_max) / 2.0 + f * long_len
        longest = max(len_max, long_len if long_share else 0)
        total = max(int(n_reads * mean / cov), n_contigs * (longest + 1))
        return GenSpec(seed, n_reads, max(total // n_contigs, longest + 1), n_contigs, read_len, len_max, skew, long_len if long_share else 0, long_share)


def read_locations(spec: GenSpec, r0: int = 0, r1: int | None = None, ids=None):
    """(genome position, length, strand) of the reads r0..r1, or of the given read ids"""
    r1 = spec.n_reads if r1 is None else r1
    with np.errstate(over="ignore"):
        r = np.arange(r0, r1, dtype=np.uint64) if ids is None else np.asarray(ids, dtype=np.uint64)
        sr = np.uint64(spec.seed) ^ np.uint64(0xA5A5A5A55A5A5A5A)
        h0 = mix64(sr + np.uint64(4) * r)
        h1 = mix64(sr + np.uint64(4) * r + np.uint64(1))
        h2 = mix64(sr + np.uint64(4) * r + np.uint64(2))
        length = (np.uint64(spec.len_min) + h2 % np.uint64(spec.len_max - spec.len_min + 1)).astype(np.uint64)
        if getattr(spec, "long_share", 0):
            is_long = (mix64(h2 ^ np.uint64(0x6C6F6E6772656164)) & np.uint64(0xFFFF)) < np.uint64(spec.long_share)
            length = np.where(is_long, np.uint64(spec.long_len), length).astype(np.uint64)
        contig = (h0 & np.uint64(0x7FFFFFFFFFFFFFFF)) % np.uint64(spec.n_contigs)
        if spec.skew:
            h3 = mix64(sr + np.uint64(4) * r + np.uint64(3))
            contig = (contig * (h3 % np.uint64(spec.n_contigs))) // np.uint64(spec.n_contigs)
        pos = h1 % (np.uint64(spec.contig_len) - length + np.uint64(1))
        gpos = contig * np.uint64(spec.contig_len) + pos
        strand = (h0 >> np.uint64(63)).astype(np.uint8)
    return gpos, length.astype(np.uint32), strand


def generate_codes(spec: GenSpec, r0: int = 0, r1: int | None = None, ids=None):
    """returns (codes uint8 [sum len], off uint64 [n+1]) — base codes of reads r0..r1, or of the given read ids."""
    gpos, length, strand = read_locations(spec, r0, r1, ids)
    n = len(gpos)
    off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(length, out=off[1:])
    total = int(off[-1])
    rid = np.repeat(np.arange(n, dtype=np.int64), length.astype(np.int64))
    i = np.arange(total, dtype=np.int64) - off[:-1].astype(np.int64)[rid]
    L = length.astype(np.int64)[rid]
    rev = strand[rid].astype(bool)
    g = gpos.astype(np.int64)[rid] + np.where(rev, L - 1 - i, i)
    b = genome_bases(spec.seed, g.astype(np.uint64))
    b = np.where(rev, 3 - b, b).astype(np.uint8)
    return b, off


def substitute(codes, off, seed: int, rate_ppm: int, r0: int = 0):
    """numpy twin of disco_substituted_base (csrc/readgen.h): base p of read r0 + i is replaced with probability rate_ppm / 10^6"""
    n = len(off) - 1
    length = (off[1:] - off[:-1]).astype(np.int64)
    rid = np.repeat(np.arange(n, dtype=np.uint64) + np.uint64(r0), length)
    p = (np.arange(int(off[-1]), dtype=np.int64) - off[:-1].astype(np.int64)[np.repeat(np.arange(n), length)]).astype(np.uint64)
    with np.errstate(over="ignore"):
        h = mix64((np.uint64(seed) * np.uint64(0x9FB21C651E98DF25)) ^ ((rid << np.uint64(15)) | p))
    hit = (h % np.uint64(1000000)) < np.uint64(rate_ppm)
    out = np.asarray(codes, dtype=np.uint8).copy()
    out[hit] = ((out[hit].astype(np.uint64) + np.uint64(1) + (h[hit] >> np.uint64(40)) % np.uint64(3)) & np.uint64(3)).astype(np.uint8)
    return out


_ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)


def codes_to_reads(codes, off):
    s = _ASCII[codes].tobytes()
    return [s[int(off[i]):int(off[i + 1])].decode() for i in range(len(off) - 1)]


def generate_reads(spec: GenSpec, r0: int = 0, r1: int | None = None):
    return codes_to_reads(*generate_codes(spec, r0, r1))


def write_fasta(path: str, reads, line_width: int = 0):
    with open(path, "w") as f:
        for i, s in enumerate(reads):
            f.write(f">r{i + 1}\n")
            if line_width and line_width > 0:
                for p in range(0, len(s), line_width):
                    f.write(s[p:p + line_width] + "\n")
            else:
                f.write(s + "\n")


def pack_reads(codes, off, stride_words: int | None = None):
    """2-bit pack (MSB-first, A0 C1 G2 T3; BG/HashTable.cpp:456-477) into a fixed-stride [n][stride] u64 array."""
    n = len(off) - 1
    length = (off[1:] - off[:-1]).astype(np.int64)
    maxw = int((length.max() + 31) // 32) if n else 1
    stride = maxw if stride_words is None else stride_words
    assert stride >= maxw
    packed = np.zeros((n, stride), dtype=np.uint64)
    rid = np.repeat(np.arange(n, dtype=np.int64), length)
    i = np.arange(int(off[-1]), dtype=np.int64) - off[:-1].astype(np.int64)[rid]
    sh = (62 - 2 * (i & 31)).astype(np.uint64)
    vals = codes.astype(np.uint64) << sh
    np.bitwise_or.at(packed, (rid, i >> 5), vals)
    return packed, length.astype(np.uint16)


def generate_pairs(seed: int, n_pairs: int, genome_len: int, len_min: int, len_max: int, ins_min: int = 600, ins_max: int = 900):
    """interleaved paired-end reads of one uniform-random genome (BASELINE config 1's stand-in for test/Ecoli_250_500_test.fna,
    which the reference does not ship): read 2i from the forward strand at the fragment's start, read 2i+1 the
    reverse complement of the fragment's end. numpy's PCG64 with the given seed: bit-reproducible."""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, genome_len, dtype=np.uint8)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = []
    for _ in range(n_pairs):
        ins = int(rng.integers(ins_min, ins_max + 1))
        p = int(rng.integers(0, genome_len - ins))
        l1, l2 = int(rng.integers(len_min, len_max + 1)), int(rng.integers(len_min, len_max + 1))
        l1, l2 = min(l1, ins), min(l2, ins)
        r1 = genome[p:p + l1]
        r2 = comp[genome[p + ins - l2:p + ins]][::-1]
        reads.append(letters[r1].tobytes().decode())
        reads.append(letters[r2].tobytes().decode())
    return reads
