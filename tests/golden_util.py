"""load the committed golden cases (tests/golden/cases.json, made by tests/golden/make_golden.py with the REAL reference)"""
from __future__ import annotations

import json
import os

import numpy as np

from disco_amd import readgen
from oracle import pyoracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(GOLD, "cases.json")))
COMP = str.maketrans("ACGT", "TGCA")


def repeat_reads(seed, n_reads, copies, rep_len, flank, lmin, lmax):
    """same construction as tests/golden/make_golden.py:repeat_reads"""
    rng = np.random.default_rng(seed)
    rep = "".join(rng.choice(list("ACGT"), rep_len))
    genome = "".join("".join(rng.choice(list("ACGT"), flank)) + rep for _ in range(copies))
    reads = []
    for _ in range(n_reads):
        L = int(rng.integers(lmin, lmax + 1))
        p = int(rng.integers(0, len(genome) - L))
        s = genome[p:p + L]
        reads.append(s.translate(COMP)[::-1] if rng.random() < 0.5 else s)
    return reads


def case_inputs(name):
    """returns (good reads, file index array (1-based over all records), min_overlap)"""
    c = CASES[name]
    mo = c["min_overlap"]
    if c["kind"] == "generated":
        reads = readgen.generate_reads(readgen.GenSpec.coverage(**c["spec"]))
        return reads, np.arange(1, len(reads) + 1, dtype=np.uint64), mo
    if c["kind"] == "repeats":
        reads = repeat_reads(*c["args"])
        return reads, np.arange(1, len(reads) + 1, dtype=np.uint64), mo
    if c["kind"] == "file":
        paths = [os.path.join(GOLD, f) for f in c["files"]]
    else:
        paths = [os.path.join(GOLD, f) for f in c["pe"] + c["se"]]
    reads, fidx, _total = pyoracle.load_good_reads(paths, mo)
    return reads, fidx, mo


def check_against_golden(name, edges_canon, contained_canon):
    c = CASES[name]
    et, ct = pyoracle.edges_text(edges_canon), pyoracle.contained_text(contained_canon)
    if c["full_text"]:
        assert et == open(os.path.join(GOLD, name + ".edges.txt")).read(), f"{name}: edge list differs from the reference"
        assert ct == open(os.path.join(GOLD, name + ".contained.txt")).read(), f"{name}: contained rows differ from the reference"
    assert len(edges_canon) == c["n_edges"] and len(contained_canon) == c["n_contained"], (name, len(edges_canon), c["n_edges"])
    assert pyoracle.digest(et) == c["edges_sha256"], f"{name}: edge digest differs from the reference"
    assert pyoracle.digest(ct) == c["contained_sha256"], f"{name}: contained digest differs from the reference"


BIT_EXACT_CASES = [n for n, c in CASES.items() if c["kind"] != "repeats"]
