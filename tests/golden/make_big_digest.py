#!/usr/bin/env python3
"""Pin a BASELINE-size case: canonical digests of files written by the REAL reference buildG (oracle/_ref/buildG_ref).

The reference run itself is started by hand (it takes hours at 50 M reads and ~40 GB of memory):

    disco_amd/bin/readgen r.fasta N 150 30 42          (same generator and contig layout as bench.py: 5 Mbp contigs)
    oracle/_ref/buildG_ref -se r.fasta -f out/x -p disco.cfg -t T -m 56

and this script turns out/x_<t>_parGraph.txt / _containedReads.txt into the canonical forms of SURVEY.md §8c-3 and stores
their sha256 (over the sorted int64 arrays, oracle/pyoracle.digest_array) in tests/golden/cases_big.json. Only the digests
and the generator parameters are committed; the GPU test regenerates the reads on the device from the same parameters.

usage: make_big_digest.py NAME PREFIX THREADS --reads N --contigs C [--seed 42 --len 150 --len-max 0 --coverage 30 --min-overlap 40]
"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyoracle  # noqa: E402


def read_cols(path, usecols):
    """tab/comma separated integer columns of a reference output file (the NA column is never selected)"""
    if os.path.getsize(path) == 0:
        return np.zeros((0, len(usecols)), np.int64)
    with open(path, "rb") as f:
        tr = subprocess.Popen(["tr", "\\t", ","], stdin=f, stdout=subprocess.PIPE)
        df = pd.read_csv(tr.stdout, header=None, usecols=usecols, dtype=np.int64, engine="c")
        tr.wait()
    return df[usecols].to_numpy(dtype=np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("prefix")
    ap.add_argument("threads", type=int)
    ap.add_argument("--reads", type=int, required=True)
    ap.add_argument("--contigs", type=int, required=True)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--len", type=int, default=150)
    ap.add_argument("--len-max", type=int, default=0, help="longest read (uniform lengths in [len, len-max]); 0 = fixed length")
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--min-overlap", type=int, default=40)
    ap.add_argument("--skew", type=int, default=0, help="1: metagenome-like contig abundances (readgen's skew)")
    a = ap.parse_args()
    # edge line: src dst orient,ovl,0,0,len1,start1,stop1,len2,0,stop2,NA[,flag] -> columns 0 1 2 | 7 (start1) 6 (len1) 9 (len2)
    e = np.concatenate([read_cols(f"{a.prefix}_{t}_parGraph.txt", [0, 1, 2, 7, 6, 9]) for t in range(a.threads)])
    ce = pyoracle.canonical_edges_large(e[:, 0], e[:, 1], e[:, 2], e[:, 3], e[:, 4], e[:, 5])
    del e
    # contained line: contained super orient,len2,0,0,len2,0,len2,len1,start,stop -> columns 0 1 2 3 | 9 (len1) 10 (start)
    c = np.concatenate([read_cols(f"{a.prefix}_{t}_containedReads.txt", [0, 1, 2, 3, 9, 10]) for t in range(a.threads)])
    order = np.lexsort(tuple(c[:, i] for i in range(5, -1, -1)))
    cc = c[order]
    assert len(np.unique(cc[:, 0])) == len(cc), "a contained read is listed twice"
    out = os.path.join(HERE, "cases_big.json")
    cases = json.load(open(out)) if os.path.exists(out) else {}
    cases[a.name] = dict(kind="generated", seed=a.seed, reads=a.reads, read_len=a.len, len_max=a.len_max or a.len, coverage=a.coverage, n_contigs=a.contigs, skew=a.skew,
                         min_overlap=a.min_overlap, n_edges=int(len(ce)), n_contained=int(len(cc)),
                         edges_sha256=pyoracle.digest_array(ce), contained_sha256=pyoracle.digest_array(cc),
                         reference="oracle/_ref/buildG_ref -se <generated fasta> -t %d" % a.threads)
    json.dump(cases, open(out, "w"), indent=1, sort_keys=True)
    print(a.name, cases[a.name])


if __name__ == "__main__":
    main()
