"""-m gpu : inexact overlaps — an EXTENSION: the reference compares exactly and writes 0 into the
substitutions column (BG/OverlapGraph.cpp:815-816), so there is no reference output to pin. What is checked: the HIP path with
disco_params.max_substitutions = t against the oracle's statement of the same rule (oracle_build_graph_inexact: the two compares
of checkOverlap* tolerate t differing bases, the seed k-mer stays exact, containment in its order-free form), bit-exact — edges,
contained rows, counters and the substitutions of every edge — and properties that need no second implementation."""
import numpy as np
import pytest

from disco_amd import buildgraph, readgen
from oracle import pyoracle
from tests.util import canon_hip

pytestmark = pytest.mark.gpu


def _mutated(seed, n, lmin, lmax, cov, rate):
    """error-free reads of one genome with substitutions at the given per-base rate"""
    spec = readgen.GenSpec.coverage(seed, n, lmin, cov, len_max=lmax)
    codes, off = readgen.generate_codes(spec)
    rng = np.random.default_rng(seed + 1000)
    m = rng.random(len(codes)) < rate
    codes = codes.copy()
    codes[m] = (codes[m] + rng.integers(1, 4, int(m.sum())).astype(np.uint8)) % 4
    return codes, off


def _hip(codes, off, min_overlap, t, **kw):
    reads = readgen.codes_to_reads(codes, off)
    with buildgraph.BuildGraph(min_overlap=min_overlap, max_substitutions=t, **kw) as g:
        g.upload_ascii(reads)
        g.run_graph()
        return g.fetch_edges(), g.fetch_contained(), g.counters(), g.fetch_edge_substitutions()


def _edge_key(e):
    """canonical (src < dst) identity of each edge record, as one sortable row per edge"""
    return pyoracle.canonical_edges_large(e["src"].astype(np.int64) + 1, e["dst"].astype(np.int64) + 1, e["orient"], e["offset"],
                                          e["len_src"], e["len_dst"])


def _subs_by_edge(e, subs):
    """canonical rows with the substitution count appended, sorted"""
    src, dst = e["src"].astype(np.int64), e["dst"].astype(np.int64)
    assert np.all(src < dst)  # both implementations emit an edge from its smaller endpoint
    t = np.stack([src, dst, e["orient"].astype(np.int64), e["offset"].astype(np.int64), subs.astype(np.int64)], axis=1)
    return t[np.lexsort((t[:, 3], t[:, 2], t[:, 1], t[:, 0]))]


def _assert_inexact_parity(codes, off, min_overlap, t, label):
    he, hr, hc, hs = _hip(codes, off, min_overlap, t)
    orows, oe, oc, osubs = pyoracle.build_graph_inexact(codes, off, min_overlap, t)
    ce, cc = canon_hip(he, hr)
    oce, occ = canon_hip(oe, orows)
    assert np.array_equal(cc, occ), f"{label}: contained rows differ ({len(cc)} vs {len(occ)})"
    assert np.array_equal(ce, oce), f"{label}: edge list differs ({len(ce)} vs {len(oce)})"
    for key in ("probes", "n_contained", "e_pre", "e_out", "cap_bind_sites", "asymmetric_pairs"):
        assert hc[key] == oc[key], f"{label}: counter {key}: hip {hc[key]} oracle {oc[key]}"
    assert np.array_equal(_subs_by_edge(he, hs), _subs_by_edge(oe, osubs)), f"{label}: substitutions per edge differ"
    assert hs.max(initial=0) <= t
    return hc, hs


@pytest.mark.parametrize("seed,n,lmin,lmax,cov,minovl,rate,t", [
    (101, 4000, 150, 150, 30.0, 40, 0.004, 1),   # BASELINE shape, ~0.6 errors per read
    (102, 4000, 150, 150, 30.0, 40, 0.004, 3),
    (103, 3000, 100, 250, 30.0, 40, 0.006, 2),   # mixed lengths: containment within the threshold
    (104, 3000, 60, 90, 25.0, 31, 0.01, 4),      # k = 30, single word
    (105, 2000, 260, 500, 25.0, 65, 0.004, 3),   # 16-word rows, k = 64
    (106, 1200, 520, 760, 25.0, 40, 0.004, 5),   # 24-word rows
    (107, 1200, 800, 1024, 25.0, 50, 0.003, 6),  # 32-word rows
    (108, 600, 1100, 1500, 20.0, 40, 0.002, 4),  # the generic variant (rows read from global memory)
    (109, 5000, 150, 150, 100.0, 40, 0.004, 2),  # 100x: wide rows
])
def test_inexact_against_the_oracle(seed, n, lmin, lmax, cov, minovl, rate, t):
    codes, off = _mutated(seed, n, lmin, lmax, cov, rate)
    _check_inexact(codes, off, seed, minovl, t)


@pytest.mark.parametrize("seed,n,lmin,lmax,cov,minovl,rate,t", [
    (111, 4000, 150, 150, 30.0, 40, 0.004, 3),    # a few extras per row: ranked into the old row
    (112, 3000, 100, 250, 30.0, 40, 0.006, 2),
    (113, 5000, 150, 150, 100.0, 40, 0.006, 4),   # rows beyond 64 entries (scratch row) and rows with dozens of extras (sorting network)
    (114, 1500, 150, 150, 60.0, 40, 0.012, 8),    # most end k-mers carry an error: more extras than old entries
])
def test_inexact_with_every_row_rebuilt(seed, n, lmin, lmax, cov, minovl, rate, t, monkeypatch):
    """data sets of test size have few enough one-sided pairs for the sparse merge; at scale a third of all pairs are one-sided and
    every row is rebuilt (merge_rows_kernel) — force that path"""
    monkeypatch.setenv("DISCO_MERGE_REBUILD", "1")
    codes, off = _mutated(seed, n, lmin, lmax, cov, rate)
    _check_inexact(codes, off, seed, minovl, t)


def _check_inexact(codes, off, seed, minovl, t):
    hc, hs = _assert_inexact_parity(codes, off, minovl, t, f"seed{seed}")
    assert hc["e_out"] > 0 and hs.max() > 0          # the threshold was used
    assert hc["asymmetric_pairs"] > 0                # pairs hidden from one side by an error inside an end k-mer exist


def test_threshold_zero_is_the_exact_path():
    codes, off = _mutated(111, 3000, 100, 200, 30.0, 0.004)
    e0, r0, c0, s0 = _hip(codes, off, 40, 0)
    orows, oe, oc = pyoracle.build_graph(codes, off, 40)
    ce, cc = canon_hip(e0, r0)
    oce, occ = canon_hip(oe, orows)
    assert np.array_equal(ce, oce) and np.array_equal(cc, occ)
    assert not s0.any()


def test_error_free_reads_do_not_change_with_the_threshold():
    """on reads without errors a random genome has no near-identical loci: every inexact overlap is an exact one"""
    spec = readgen.GenSpec.coverage(113, 4000, 100, 30.0, len_max=180)
    codes, off = readgen.generate_codes(spec)
    e0, r0, c0, _ = _hip(codes, off, 40, 0)
    e3, r3, c3, s3 = _hip(codes, off, 40, 3)
    assert np.array_equal(_edge_key(e0), _edge_key(e3))
    assert np.array_equal(canon_hip(e0, r0)[1], canon_hip(e3, r3)[1])
    assert not s3.any()


def test_more_tolerance_recovers_the_error_free_graph():
    """the same reads with and without errors: the tolerant graph must be closer to the error-free one than the exact graph is"""
    spec = readgen.GenSpec.coverage(115, 6000, 150, 30.0)
    clean, off = readgen.generate_codes(spec)
    codes, _ = _mutated(115, 6000, 150, 150, 30.0, 0.004)
    ref = _hip(clean, off, 40, 0)[2]
    exact = _hip(codes, off, 40, 0)[2]
    tol = _hip(codes, off, 40, 4)[2]
    assert abs(tol["e_pre"] - ref["e_pre"]) < abs(exact["e_pre"] - ref["e_pre"])
    assert abs(tol["n_contained"] - ref["n_contained"]) <= abs(exact["n_contained"] - ref["n_contained"])


def test_inexact_through_three_ranks():
    """the multi-rank flow takes its order-dependent regime (lists are not symmetric before the twin pass) and must give the
    single-context answer"""
    from tests.dist_util import run_ranks_reads

    codes, off = _mutated(117, 3000, 100, 200, 30.0, 0.005)
    reads = readgen.codes_to_reads(codes, off)
    he, hr, hc, hs = _hip(codes, off, 40, 2)
    subs = []
    edges, rows, info, _ = run_ranks_reads(reads, 40, 3, max_substitutions=2, subs_out=subs)
    assert info["regime"] == 1
    assert np.array_equal(_subs_by_edge(edges, subs[0]), _subs_by_edge(he, hs))
    assert np.array_equal(canon_hip(edges, rows)[1], canon_hip(he, hr)[1])


@pytest.mark.parametrize("how", ["cli", "cfg", "two_ranks"])
def test_buildg_writes_the_substitutions_column(tmp_path, how):
    """the drop-in's files in inexact mode: every edge line of the oracle's rule, with its count in the third number of the info
    column (where the reference writes "no substitutions", BG/OverlapGraph.cpp:815), text and binary side output alike"""
    import os
    import subprocess

    from disco_amd import build, edgefile

    build.build_host()
    codes, off = _mutated(121, 3000, 100, 220, 30.0, 0.005)
    reads = readgen.codes_to_reads(codes, off)
    fa = tmp_path / "r.fasta"
    readgen.write_fasta(str(fa), reads)
    cfg = tmp_path / "disco.cfg"
    cfg.write_text("MinOverlap4BuildGraph = 40\n" + ("MaxSubstitutions4BuildGraph = 2\n" if how == "cfg" else ""))
    prefix = str(tmp_path / "g")
    cmd = [os.path.join(os.path.dirname(build.HERE), "disco_amd", "bin", "buildG"), "-se", str(fa), "-f", prefix, "-p", str(cfg), "-t", "2", "--binary-out"]
    if how != "cfg":
        cmd += ["--max-substitutions", "2"]
    if how == "two_ranks":
        cmd += ["--gpus", "2", "--same-device"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
    assert "MaxSubstitutions4BuildGraph = 2" in p.stdout
    got = []
    for t in range(2):
        for line in open(f"{prefix}_{t}_parGraph.txt"):
            a, b, info = line.rstrip("\n").split("\t")
            f = info.split(",")
            assert f[3] == "0"  # edits
            got.append((int(a) - 1, int(b) - 1, int(f[0]), int(f[5]), int(f[2])))  # src, dst, orient, start1 = offset, substitutions
    got = np.array(sorted(got), dtype=np.int64)
    orows, oe, oc, osubs = pyoracle.build_graph_inexact(codes, off, 40, 2)
    assert np.array_equal(got, _subs_by_edge(oe, osubs))
    assert got[:, 4].max() == 2
    rec, _ = edgefile.read_edges(prefix + "_edges.bin")
    binr = np.stack([rec["src"].astype(np.int64) - 1, rec["dst"].astype(np.int64) - 1, rec["orient"].astype(np.int64), rec["offset"].astype(np.int64),
                     rec["substitutions"].astype(np.int64)], axis=1)
    assert np.array_equal(binr[np.lexsort((binr[:, 3], binr[:, 2], binr[:, 1], binr[:, 0]))], got)
    texts = edgefile.text_files(prefix)
    for t in range(2):
        path = f"{prefix}_{t}_parGraph.txt"
        assert sorted(open(path).read().splitlines()) == sorted(texts[path].splitlines())
    n_rows = sum(1 for t in range(2) for _ in open(f"{prefix}_{t}_containedReads.txt"))
    assert n_rows == oc["n_contained"]


def test_device_error_model_equals_its_numpy_twin():
    """disco_substitute_bases (the error model of the bench / scale probes) against readgen.substitute, and the graph of such reads
    against the oracle"""
    spec = readgen.GenSpec.coverage(131, 3000, 100, 30.0, len_max=170)
    codes, off = readgen.generate_codes(spec)
    mut = readgen.substitute(codes, off, 9, 5000)
    assert 0.003 < (mut != codes).mean() < 0.007
    packed, lens = readgen.pack_reads(mut, off)
    with buildgraph.BuildGraph(min_overlap=40, max_substitutions=2) as g:
        g.generate_reads(spec)
        g.substitute_bases(9, 5000)
        dp, dl = g.download_reads()
        assert np.array_equal(dl, lens) and np.array_equal(dp[:, :packed.shape[1]], packed) and not dp[:, packed.shape[1]:].any()
        g.run_graph()
        he, hs, hc = g.fetch_edges(), g.fetch_edge_substitutions(), g.counters()
    orows, oe, oc, osubs = pyoracle.build_graph_inexact(mut, off, 40, 2)
    assert np.array_equal(_subs_by_edge(he, hs), _subs_by_edge(oe, osubs)) and hc["n_contained"] == oc["n_contained"]
