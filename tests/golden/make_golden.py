#!/usr/bin/env python3
"""Generate tests/golden/* with the REAL reference buildG (oracle/_ref/buildG_ref).

Runs only in the build container (needs /root/reference to have been compiled by `make -C oracle ref`).
Every case = inputs (regenerated from a seed by disco_amd.readgen, or a small hand-made FASTA/FASTQ committed under
tests/golden/inputs/) + the canonical outputs of the reference at -t 1:
    <case>.edges.txt      canonical edge lines (flag column dropped), sorted
    <case>.contained.txt  canonical contained rows, sorted
or, for the larger cases, only their SHA-256 in cases.json.
tests/golden/reference_data/ holds the reference's own data files (two FASTAs + its one golden edge list).
"""
from __future__ import annotations

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from disco_amd import readgen  # noqa: E402
from oracle import pyoracle, refrun  # noqa: E402

COMP = str.maketrans("ACGT", "TGCA")


def rc(s):
    return s.translate(COMP)[::-1]


def repeat_reads(seed, n_reads, copies, rep_len, flank, lmin, lmax):
    rng = np.random.default_rng(seed)
    rep = "".join(rng.choice(list("ACGT"), rep_len))
    genome = "".join("".join(rng.choice(list("ACGT"), flank)) + rep for _ in range(copies))
    reads = []
    for _ in range(n_reads):
        L = int(rng.integers(lmin, lmax + 1))
        p = int(rng.integers(0, len(genome) - L))
        s = genome[p:p + L]
        reads.append(rc(s) if rng.random() < 0.5 else s)
    return reads


# generator-seeded cases: (name, GenSpec kwargs, min_overlap, store_full_text)
GEN_CASES = [
    ("u150_5k", dict(seed=42, n_reads=5000, read_len=150, cov=30.0), 40, True),
    ("mixed_4k", dict(seed=7, n_reads=4000, read_len=100, cov=30.0, len_max=250), 40, True),
    ("k30_6k", dict(seed=11, n_reads=6000, read_len=60, cov=20.0, len_max=90), 31, True),
    ("k64_3k", dict(seed=13, n_reads=3000, read_len=80, cov=25.0, len_max=120), 65, True),
    ("long_2k", dict(seed=5, n_reads=2000, read_len=300, cov=20.0, len_max=600), 40, True),
    ("contigs_20k", dict(seed=9, n_reads=20000, read_len=150, cov=30.0, n_contigs=4), 40, False),
    ("u150_100k", dict(seed=43, n_reads=100000, read_len=150, cov=30.0), 40, False),
    # round 4: k above 64 (the reference's k-mers are strings: no limit there, BG/HashTable.cpp:396-416)
    ("k79_4k", dict(seed=21, n_reads=4000, read_len=250, cov=30.0, len_max=500), 80, True),
    ("k94_4k", dict(seed=22, n_reads=4000, read_len=250, cov=30.0, len_max=500), 95, True),
    # round 4: 150 bp reads with a tail of 600 bp reads (1 %): the set the two classes of rows are for (include/disco_hip.h, disco_long_rows)
    ("tail_20k", dict(seed=23, n_reads=20000, read_len=150, cov=30.0, long_len=600, long_share=650), 40, False),
]


def write_inputs():
    """small hand-made inputs that exercise the parser and the read filter (BG/Dataset.cpp:255-305,403-452)"""
    d = os.path.join(HERE, "inputs")
    os.makedirs(d, exist_ok=True)
    spec = readgen.GenSpec.coverage(seed=77, n_reads=600, read_len=90, cov=15.0, len_max=140)
    good = readgen.generate_reads(spec)
    # FASTA with multi-line records, lower case, N, too-short reads, low-complexity reads, micro-repeat ends, CRLF-free
    recs = []
    for i, s in enumerate(good[:300]):
        if i % 7 == 0:
            s = s.lower()
        if i % 31 == 0:
            s = s[:20] + "N" + s[21:]
        recs.append(s)
    recs.insert(5, "ACGT" * 8)                       # 32 bp <= min overlap -> rejected
    recs.insert(9, "A" * 80 + good[0][:20])           # > 70 % one base
    recs.insert(12, "ACACACACACACACACACACACACACACA" + good[1][:70])  # micro-repeat prefix
    recs.insert(20, good[2][:70] + "TTCTTCTTCTTCTTCTTCTTCTTCTTCTT")  # micro-repeat suffix
    recs.insert(25, "AT" * 45)                        # dimer over > 50 %
    recs.insert(30, "")                               # empty record
    with open(os.path.join(d, "filter_multi.fasta"), "w") as f:
        for i, s in enumerate(recs):
            f.write(f">rec{i} some description\n")
            for p in range(0, len(s), 60):
                f.write(s[p:p + 60] + "\n")
    # FASTQ
    with open(os.path.join(d, "reads.fastq"), "w") as f:
        for i, s in enumerate(good[300:450]):
            f.write(f"@q{i}\n{s}\n+\n{'I' * len(s)}\n")
    # second FASTA, single-line, no trailing newline at EOF
    with open(os.path.join(d, "plain.fasta"), "w") as f:
        f.write("\n".join(f">p{i}\n{s}" for i, s in enumerate(good[450:])))
    return [os.path.join(d, x) for x in ("filter_multi.fasta", "reads.fastq", "plain.fasta")]


def canon_to_files(name, edges, cont, full):
    et, ct = pyoracle.edges_text(edges), pyoracle.contained_text(cont)
    if full:
        open(os.path.join(HERE, name + ".edges.txt"), "w").write(et)
        open(os.path.join(HERE, name + ".contained.txt"), "w").write(ct)
    return dict(edges_sha256=pyoracle.digest(et), contained_sha256=pyoracle.digest(ct), n_edges=int(len(edges)),
                n_contained=int(len(cont)), full_text=bool(full))


def main():
    assert refrun.available(), "build the reference first: make -C oracle ref"
    cases = {}
    import tempfile

    only = None  # make_golden.py --only a,b : (re)generate these generated cases and merge them into cases.json
    if len(sys.argv) > 2 and sys.argv[1] == "--only":
        only = set(sys.argv[2].split(","))
        cases = json.load(open(os.path.join(HERE, "cases.json")))
    for name, kw, minovl, full in GEN_CASES:
        if only is not None and name not in only:
            continue
        spec = readgen.GenSpec.coverage(**kw)
        reads = readgen.generate_reads(spec)
        d = tempfile.mkdtemp(prefix="golden_")
        fa = os.path.join(d, "r.fasta")
        readgen.write_fasta(fa, reads)
        r = refrun.run_reference([fa], minovl, threads=1, workdir=d)
        info = canon_to_files(name, r["edges"], r["contained"], full)
        info.update(kind="generated", spec=kw, min_overlap=minovl)
        cases[name] = info
        print(name, info["n_edges"], info["n_contained"], f"{r['wall']:.1f}s", flush=True)

    if only is not None:
        json.dump(cases, open(os.path.join(HERE, "cases.json"), "w"), indent=1, sort_keys=True)
        return
    # order-dependent regime (cap of 4 edges per k-mer binds / asymmetric pairs): the reference differs from ITSELF between
    # thread counts here (SURVEY.md preamble item 6); recorded to document the parity domain, not asserted bit-exact
    reads = repeat_reads(99, 8000, 30, 500, 300, 100, 200)
    d = tempfile.mkdtemp(prefix="golden_")
    fa = os.path.join(d, "r.fasta")
    readgen.write_fasta(fa, reads)
    r = refrun.run_reference([fa], 40, threads=1, workdir=d)
    info = canon_to_files("repeats_8k", r["edges"], r["contained"], False)
    info.update(kind="repeats", args=[99, 8000, 30, 500, 300, 100, 200], min_overlap=40)
    cases["repeats_8k"] = info
    print("repeats_8k", info["n_edges"], info["n_contained"], flush=True)

    # the reference's own FASTAs (min overlap 30 = disco.cfg default)
    for fa in ("10reads_containedReads.fasta", "10reads_forward.fasta"):
        r = refrun.run_reference([os.path.join(HERE, "reference_data", fa)], 30, threads=1)
        name = "ref_" + fa.split(".")[0]
        info = canon_to_files(name, r["edges"], r["contained"], True)
        info.update(kind="file", files=["reference_data/" + fa], se=True, min_overlap=30)
        cases[name] = info

    # parser / filter / multi-file indexing: -pe a,b -se c
    a, b, c = write_inputs()
    d = tempfile.mkdtemp(prefix="golden_")
    cfg = os.path.join(d, "disco.cfg")
    refrun.write_cfg(cfg, 35)
    import glob
    import subprocess

    prefix = os.path.join(d, "g")
    p = subprocess.run([refrun.REF_BIN, "-pe", f"{a},{b}", "-se", c, "-f", prefix, "-p", cfg, "-t", "1", "-m", "8"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    edges = refrun.parse_pargraph(sorted(glob.glob(prefix + "_*_parGraph.txt")))
    cont = refrun.parse_contained(sorted(glob.glob(prefix + "_*_containedReads.txt")))
    info = canon_to_files("multifile", edges, cont, True)
    idmap = open(prefix + "_ReadIDMap.txt").read().replace(os.path.dirname(a) + "/", "")
    good = [int(x) for x in __import__("re").findall(r"\s+(\d+) good reads in current dataset", p.stdout)]
    info.update(kind="files", pe=["inputs/filter_multi.fasta", "inputs/reads.fastq"], se=["inputs/plain.fasta"], min_overlap=35,
                read_id_map=idmap, good_reads_per_file=good)
    cases["multifile"] = info
    print("multifile", info["n_edges"], info["n_contained"], good)

    json.dump(cases, open(os.path.join(HERE, "cases.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
