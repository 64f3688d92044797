#!/usr/bin/env python3
"""scale-only defects do not show in 5 000-read parity tests (round 3 found one with a 50 M-read run: rows of up to 64 hits on the
sequential accept path). This sweep runs shapes at 10-50 M reads — lengths, coverage, substitution errors, abundance skew — through
  (a) the single-GPU pass,  (b) the same with the two-pass verify,  (c) round 2's probe (DISCO_NO_RUNS=1),
  (d) 4 ranks on one GPU,   (e) 4 ranks with the index kept partitioned
and demands the same result counters everywhere (e_pre, e_out, contained reads, cap-bound sites, one-sided pairs) plus identical
digests of the canonical edge list and contained rows between (a) and (d).   python tools/scale_consistency.py [quick]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco_amd import buildgraph, readgen  # noqa: E402
from oracle import pyoracle  # noqa: E402
from tests.dist_util import run_ranks  # noqa: E402

KEYS = ("e_pre", "e_out", "n_contained", "cap_bind_sites", "asymmetric_pairs")
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
SHAPES = [  # reads, len_min, len_max, coverage, contigs, skew, errors ppm, substitutions tolerated (f-4; 0 = exact overlaps)
    (20_000_000, 150, 150, 30.0, 20, 0, 0, 0),
    (20_000_000, 150, 150, 30.0, 20, 0, 1000, 0),
    (20_000_000, 100, 250, 30.0, 40, 1, 0, 0),
    (10_000_000, 100, 250, 30.0, 20, 0, 3000, 0),
    (10_000_000, 150, 150, 100.0, 5, 0, 0, 0),
    (10_000_000, 60, 120, 25.0, 10, 0, 500, 0),
    (5_000_000, 250, 500, 25.0, 10, 0, 0, 0),
    (20_000_000, 150, 150, 30.0, 20, 0, 3000, 3),   # inexact overlaps: (b) does not apply, the ranks take regime 1
    (10_000_000, 100, 250, 30.0, 20, 0, 6000, 2),
    (10_000_000, 150, 150, 30.0, 20, 0, 0, 3),      # no errors: no hidden finds, the few-extras merge, flags cleared in place
]
if quick:
    SHAPES = [(s[0] // 10,) + s[1:] for s in SHAPES]


def digests(e, r):
    one = np.int64(1)
    ce = pyoracle.canonical_edges_large(e["src"].astype(np.int64) + one, e["dst"].astype(np.int64) + one, e["orient"], e["offset"], e["len_src"], e["len_dst"])
    cc = np.stack([r["contained"].astype(np.int64) + one, r["super"].astype(np.int64) + one] + [np.asarray(r[k], dtype=np.int64) for k in ("orient", "len2", "len1", "start")], axis=1)
    cc = cc[np.lexsort(tuple(cc[:, i] for i in range(5, -1, -1)))]
    return pyoracle.digest_array(ce), pyoracle.digest_array(cc)


fails = 0
for n, lmin, lmax, cov, nc, skew, ppm, tsub in SHAPES:
    spec = readgen.GenSpec.coverage(42, n, lmin, cov, n_contigs=nc, len_max=lmax, skew=skew)
    label = f"n={n} len={lmin}-{lmax} cov={cov} contigs={nc} skew={skew} errors_ppm={ppm} max_substitutions={tsub}"
    t0 = time.time()

    def single(flags=0, env=None):
        for k, v in (env or {}).items():
            os.environ[k] = v
        try:
            with buildgraph.BuildGraph(min_overlap=40, flags=flags, max_substitutions=tsub) as g:
                g.generate_reads(spec)
                if ppm:
                    g.substitute_bases(7, ppm)
                g.run_graph()
                g.run_graph()  # a second pass on kept buffers must reproduce the first
                c = g.counters()
                d = digests(g.fetch_edges(), g.fetch_contained()) if not flags and not env else None
            return {k: c[k] for k in KEYS}, d
        finally:
            for k in (env or {}):
                os.environ.pop(k, None)

    def setup(g):
        g.dist_generate_reads(spec)
        if ppm:
            g.substitute_bases(7, ppm)

    try:
        a, da = single()
        b, _ = single(flags=buildgraph.FLAG_TWO_PASS_VERIFY) if not tsub else single(env={"DISCO_FORCE_TWIN_CHECK": "1"})  # (inexact: the full twin search instead)
        c_, _ = single(env={"DISCO_NO_RUNS": "1", "DISCO_NO_DROP_LIST": "1"})  # round 2's probe; the twin search by bitmap instead of the drop list
        e4, r4, info, _ = run_ranks(4, 40, setup, max_substitutions=tsub)
        d4 = digests(e4, r4)
        _, _, infop, _ = run_ranks(4, 40, setup, partitioned_index=True, max_substitutions=tsub)
        ok = a == b == c_ == {k: info[k] for k in KEYS} == {k: infop[k] for k in KEYS} and da == d4
        print(("ok  " if ok else "FAIL"), label, a, f"regime {info['regime']}/{infop['regime']}", f"{time.time() - t0:.0f} s", flush=True)
        if not ok:
            fails += 1
            print("     two-pass", b, "\n     old probe", c_, "\n     4 ranks", {k: info[k] for k in KEYS}, "\n     4 ranks, partitioned index", {k: infop[k] for k in KEYS},
                  "\n     digests equal", da == d4, flush=True)
    except Exception as ex:  # noqa: BLE001
        fails += 1
        print("FAIL", label, repr(ex)[:300], flush=True)
print(f"{len(SHAPES) - fails}/{len(SHAPES)} shapes consistent")
sys.exit(1 if fails else 0)
