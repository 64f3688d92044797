"""Run the REAL reference buildG (oracle/_ref/buildG_ref, built by `make -C oracle ref`) and canonicalise its files.

TEST INFRASTRUCTURE ONLY.  Used (a) in the build container to pin the C restatement and to generate
tests/golden/, (b) by bench.py's cpu_baseline leg (kind "reference") on the GPU box, where the prebuilt binary
travels with the snapshot but /root/reference does not exist.
"""
from __future__ import annotations

import glob
import os
import subprocess
import tempfile
import time

import numpy as np

from . import pyoracle

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_BIN = os.path.join(_HERE, "_ref", "buildG_ref")


def available() -> bool:
    return os.path.isfile(REF_BIN) and os.access(REF_BIN, os.X_OK)


def parse_pargraph(paths):
    rows = []
    for p in paths:
        with open(p) as f:
            for line in f:
                line = line.rstrip("\n")
                if not line:
                    continue
                a, b, info = line.split("\t")
                t = info.split(",")
                # orient, ovl, subst, edits, len1, start1, stop1, len2, start2, stop2, NA, flag
                rows.append((int(a), int(b), int(t[0]), int(t[5]), int(t[4]), int(t[7])))
    if not rows:
        return pyoracle.canonical_edges([], [], [], [], [], [])
    r = np.asarray(rows, dtype=np.int64)
    return pyoracle.canonical_edges(r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4], r[:, 5])


def parse_contained(paths):
    rows = []
    for p in paths:
        with open(p) as f:
            for line in f:
                line = line.rstrip("\n")
                if not line:
                    continue
                a, b, info = line.split("\t")
                t = info.split(",")
                # orient, len2, 0, 0, len2, 0, len2, len1, start, start+len2
                rows.append((int(a), int(b), int(t[0]), int(t[1]), int(t[7]), int(t[8])))
    if not rows:
        return pyoracle.canonical_contained([], [], [], [], [], [])
    r = np.asarray(rows, dtype=np.int64)
    return pyoracle.canonical_contained(*[r[:, i] for i in range(6)])


def write_cfg(path: str, min_overlap: int):
    with open(path, "w") as f:
        f.write(f"MinOverlap4BuildGraph = {min_overlap}\n")


def run_reference(fasta_paths, min_overlap: int, threads: int = 1, mem_gb: int = 8, se: bool = True, workdir=None,
                  binary: str = REF_BIN):
    """returns dict(edges=canonical, contained=canonical, wall=seconds, log=str, files=dir)"""
    own = workdir is None
    workdir = workdir or tempfile.mkdtemp(prefix="disco_ref_")
    cfg = os.path.join(workdir, "disco.cfg")
    write_cfg(cfg, min_overlap)
    prefix = os.path.join(workdir, "g")
    cmd = [binary, "-se" if se else "-pe", ",".join(fasta_paths), "-f", prefix, "-p", cfg, "-t", str(threads), "-m",
           str(mem_gb)]
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    t0 = time.perf_counter()
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, text=True)
    wall = time.perf_counter() - t0
    edges = parse_pargraph(sorted(glob.glob(prefix + "_*_parGraph.txt")))
    cont = parse_contained(sorted(glob.glob(prefix + "_*_containedReads.txt")))
    return dict(edges=edges, contained=cont, wall=wall, log=p.stdout, returncode=p.returncode, workdir=workdir, own=own,
                prefix=prefix)
