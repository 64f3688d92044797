"""-m gpu : the multi-GPU flow (range-partitioned reads, hash-partitioned index build, rows on request, survivor push) with
G = 1..4 ranks on ONE GPU (tests/dist_util.py). The canonical output must be identical for every G and equal to the REAL
reference's (SURVEY.md §8e 'parity across GPU counts')."""
import numpy as np
import pytest

from disco_amd import readgen
from tests import golden_util as gu
from tests.dist_util import run_ranks, run_ranks_reads
from tests.util import canon_hip, run_hip_reads, run_oracle_reads

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,G", [(n, g) for n in ("u150_5k", "mixed_4k", "k30_6k", "long_2k") for g in (1, 2, 3, 4)] +
                         [("u150_5k", 8), ("mixed_4k", 8), ("contigs_20k", 8), ("k64_3k", 5), ("k79_4k", 3), ("k94_4k", 4)])
def test_ranks_equal_reference(name, G):
    """regular regime: neighbour rows fetched on request"""
    reads, fidx, mo = gu.case_inputs(name)
    edges, rows, info, infos = run_ranks_reads(reads, mo, G)
    ce, cc = canon_hip(edges, rows, fidx)
    gu.check_against_golden(name, ce, cc)
    assert info["regime"] == 0 and info["asymmetric_pairs"] == 0 and info["world"] == G
    # round 5: ranks own loci — every read is processed by exactly one rank, whichever id range it arrived in
    assert all(i["placement"] == 1 for i in infos) and sum(i["own_reads"] for i in infos) == info["n_reads"]
    if G > 1:
        assert sum(i["bytes_sent"]["index_records"] for i in infos) > 0
        assert sum(i["bytes_sent"]["row_data"] for i in infos) > 0
        assert sum(i["bytes_sent"]["keys"] for i in infos) > 0


@pytest.mark.parametrize("name,G,extra", [("u150_5k", 2, {}), ("mixed_4k", 3, {}), ("contigs_20k", 8, {}), ("k30_6k", 4, {}), ("contigs_20k", 4, {"DISCO_DIST_ONE_COMM": "1"}),
                                          ("contigs_20k", 3, {"DISCO_DIST_KEYS_ON_PATH": "1"}), ("mixed_4k", 4, {"DISCO_DIST_ID_RANGES": "1"}),
                                          ("repeats_8k", 3, {}), ("u150_5k", 4, {"DISCO_DIST_PARTITIONED_INDEX": "1"}),
                                          # the grouping of the own reads: its counting pass inside own_select_kernel (default; every rank's share
                                          # must reach DISCO_ORDER_MIN_READS) and as a pass of its own
                                          ("contigs_20k", 2, {"DISCO_DIST_ORDER_TWO_PASSES": "1"}), ("u150_5k", 3, {"DISCO_ORDER_MIN_READS": "1"}),
                                          ("u150_5k", 3, {"DISCO_ORDER_MIN_READS": "1", "DISCO_DIST_ORDER_TWO_PASSES": "1"})])
def test_ranks_without_host_waits_in_the_transport(name, G, extra, monkeypatch):
    """DISCO_LOOP_ASYNC=1 (round 6): the in-process transport enqueues its exchanges on the ranks' streams with events between them and
    never waits for a device — RCCL's behaviour. Rounds 4-5 took host waits out of the flow (no sync behind an all-to-all, counts derived
    instead of exchanged, two communicators at once) and the blocking transport could not see a missing stream dependency; this one
    does: the same fixtures, two passes, the reference's digests"""
    monkeypatch.setenv("DISCO_LOOP_ASYNC", "1")
    for k, v in extra.items():
        monkeypatch.setenv(k, v)
    reads, fidx, mo = gu.case_inputs(name)
    edges, rows, info, infos = run_ranks_reads(reads, mo, G, passes=2)
    ce, cc = canon_hip(edges, rows, fidx)
    if name == "repeats_8k":  # outside the reference's parity domain (the cap binds): the oracle is the checker
        oe, orows, _ = run_oracle_reads(reads, mo, count_hits=False)
        oce, occ = canon_hip(oe, orows, fidx)
        assert np.array_equal(ce, oce) and np.array_equal(cc, occ)
    else:
        gu.check_against_golden(name, ce, cc)
    assert info["world"] == G


@pytest.mark.parametrize("name,G,extra", [("contigs_20k", 4, {}), ("mixed_4k", 3, {"DISCO_DIST_ID_RANGES": "1"}), ("u150_5k", 2, {"DISCO_LOOP_ASYNC": "1"})])
def test_survivor_push_when_the_first_pass_does_not_fit(name, G, extra, monkeypatch):
    """round 6: the survivors whose smaller endpoint is another rank's are written in ONE pass into the room the exchange buffer has;
    DISCO_TEST_TIGHT_PUSH=1 gives that pass room for one item — it counts what it lost, the buffer grows, the second pass delivers"""
    monkeypatch.setenv("DISCO_TEST_TIGHT_PUSH", "1")
    for k, v in extra.items():
        monkeypatch.setenv(k, v)
    reads, fidx, mo = gu.case_inputs(name)
    edges, rows, info, infos = run_ranks_reads(reads, mo, G, passes=2)
    ce, cc = canon_hip(edges, rows, fidx)
    gu.check_against_golden(name, ce, cc)
    assert sum(i["bytes_sent"]["push"] for i in infos) > 0


@pytest.mark.parametrize("name,G,extra", [("contigs_20k", 4, {}), ("mixed_4k", 3, {"DISCO_DIST_ID_RANGES": "1"}), ("repeats_8k", 3, {}), ("u150_5k", 2, {"DISCO_LOOP_ASYNC": "1"})])
def test_fetched_rows_when_the_adjacency_array_has_to_move(name, G, extra, monkeypatch):
    """round 6: the rows fetched from other ranks go behind the rank's own rows in the same array (one reference word per node, one kind
    of entry for the marking kernel). The hit buffer has the room by construction (64 slots per read allotted); DISCO_TEST_TIGHT_TAIL=1
    makes every placement MOVE the array first — the path a buffer sized exactly (the partitioned probe, a merged adjacency) takes — in
    both request rounds, both ownerships, and the regime whose rows were rebuilt by the merge of the twins"""
    monkeypatch.setenv("DISCO_TEST_TIGHT_TAIL", "1")
    for k, v in extra.items():
        monkeypatch.setenv(k, v)
    reads, fidx, mo = gu.case_inputs(name)
    edges, rows, info, infos = run_ranks_reads(reads, mo, G, passes=2)
    ce, cc = canon_hip(edges, rows, fidx)
    if name == "repeats_8k":  # outside the reference's parity domain (the cap binds): the oracle is the checker
        oe, orows, _ = run_oracle_reads(reads, mo, count_hits=False)
        oce, occ = canon_hip(oe, orows, fidx)
        assert np.array_equal(ce, oce) and np.array_equal(cc, occ)
    else:
        gu.check_against_golden(name, ce, cc)
    assert sum(i["bytes_sent"]["row_data"] for i in infos) > 0 and sum(i["device_allocs"] for i in infos) >= 0


@pytest.mark.parametrize("name,G", [("u150_5k", 3), ("mixed_4k", 4), ("contigs_20k", 8), ("long_2k", 2)])
def test_ranks_over_id_ranges_equal_reference(name, G, monkeypatch):
    """DISCO_DIST_ID_RANGES=1: the ownership of rounds 1-4 (rank r owns the ids [r per, (r + 1) per)) — what a pass falls back to when it
    ends up gathering the whole adjacency, and the baseline the work-inflation figures of DESIGN.md section 6 are compared with"""
    monkeypatch.setenv("DISCO_DIST_ID_RANGES", "1")
    reads, fidx, mo = gu.case_inputs(name)
    edges, rows, info, infos = run_ranks_reads(reads, mo, G)
    ce, cc = canon_hip(edges, rows, fidx)
    gu.check_against_golden(name, ce, cc)
    assert info["regime"] == 0 and all(i["placement"] == 0 for i in infos)
    assert all(i["own_reads"] == i["own_hi"] - i["own_lo"] for i in infos)


def test_dealt_rows_or_the_gathered_table_same_result(monkeypatch):
    """DISCO_DIST_NO_DEAL_ROWS=1: the index pass of a rank's loci waits for the all-gather of all reads instead of getting its own reads'
    rows by an all-to-all of their own — same result"""
    monkeypatch.setenv("DISCO_DIST_NO_DEAL_ROWS", "1")
    reads, fidx, mo = gu.case_inputs("contigs_20k")
    edges, rows, info, infos = run_ranks_reads(reads, mo, 4)
    ce, cc = canon_hip(edges, rows, fidx)
    gu.check_against_golden("contigs_20k", ce, cc)
    assert all(i["placement"] == 1 and i["bytes_sent"]["reads_dealt"] == 0 for i in infos)


@pytest.mark.parametrize("G", [2, 3])
def test_ranks_forced_adjacency_gather_equals_reference(G, monkeypatch):
    """the order-dependent regime's path (whole adjacency gathered, every rank finishes on its own), forced on regular data"""
    monkeypatch.setenv("DISCO_DIST_FORCE_GATHER", "1")
    reads, fidx, mo = gu.case_inputs("mixed_4k")
    edges, rows, info, _ = run_ranks_reads(reads, mo, G)
    ce, cc = canon_hip(edges, rows, fidx)
    gu.check_against_golden("mixed_4k", ce, cc)
    assert info["regime"] == 1


@pytest.mark.parametrize("G,how", [(2, "twins"), (3, "twins"), (5, "twins"), (2, "gather"), (3, "gather"), (3, "rebuild")])
def test_ranks_order_dependent_regime_equals_single_gpu(G, how, monkeypatch):
    """repeats: the cap binds, pairs are found from one side only. The lists of the reads that dropped a hit are completed across
    the ranks and the pass carries on in the regular regime (2); without that exchange — or when a rank's rows cannot grow in
    place — the adjacency is gathered (1). Result = the single-GPU pass = the oracle, either way"""
    from oracle import pyoracle

    if how == "gather":
        monkeypatch.setenv("DISCO_DIST_NO_TWIN_PUSH", "1")
    if how == "rebuild":
        monkeypatch.setenv("DISCO_MERGE_REBUILD", "1")
    reads, fidx, mo = gu.case_inputs("repeats_8k")
    edges, rows, info, _ = run_ranks_reads(reads, mo, G)
    ce, cc = canon_hip(edges, rows, fidx)
    oce, occ, ocnt = pyoracle.oracle_canonical(reads, fidx, mo)
    assert info["regime"] == (2 if how == "twins" else 1) and info["dropped_hits"] > 0
    assert info["asymmetric_pairs"] == ocnt["asymmetric_pairs"] > 0 and info["e_pre"] == ocnt["e_pre"]
    assert np.array_equal(cc, occ) and np.array_equal(ce, oce)
    if how == "twins":
        assert info["bytes_sent"]["twins"] > 0 and info["bytes_sent"]["adjacency"] == 0 and info["bytes_sent"]["row_data"] > 0
        assert info["placement"] == 1  # the lists are completed across ranks that own loci
    else:
        assert info["placement"] == 0  # the gather of the whole adjacency walks id ranges: the pass was redone over them


@pytest.mark.parametrize("G", [2, 4])
def test_ranks_generated_reads_match_single_gpu(G):
    """reads generated on the device, range by range; counters of the whole job equal the single-GPU pass"""
    from disco_amd import buildgraph

    spec = readgen.GenSpec.coverage(seed=7, n_reads=200_000, read_len=150, cov=30.0)
    with buildgraph.BuildGraph(min_overlap=40) as g:
        g.generate_reads(spec)
        g.run_graph()
        e1, r1, c1 = g.fetch_edges(), g.fetch_contained(), g.counters()
    edges, rows, info, infos = run_ranks(G, 40, lambda g: g.dist_generate_reads(spec), passes=2)
    for key in ("e_pre", "e_out", "n_contained", "probes", "kmer_hits", "cap_bind_sites"):
        assert info[key] == c1[key], (key, info[key], c1[key])
    ce1, cc1 = canon_hip(e1, r1)
    ce2, cc2 = canon_hip(edges, rows)
    assert np.array_equal(ce1, ce2) and np.array_equal(cc1, cc2)
    assert info["tr_rounds"] >= 1


def test_ranks_with_empty_ranges():
    """fewer reads than ranks x 64: some ranks own nothing and still take part in every collective"""
    reads, fidx, mo = gu.case_inputs("ref_10reads_containedReads")
    edges, rows, info, _ = run_ranks_reads(reads, mo, 3)
    ce, cc = canon_hip(edges, rows, fidx)
    gu.check_against_golden("ref_10reads_containedReads", ce, cc)


def test_third_sweeps_take_the_second_round():
    """reads of very different lengths over a repeat-free genome with both strands: some nodes sweep a third neighbour whose
    row round 1 did not fetch; they are redone after the request-all round and the result does not change"""
    spec = readgen.GenSpec.coverage(seed=11, n_reads=60_000, read_len=100, cov=40.0, len_max=250)
    from disco_amd import buildgraph

    with buildgraph.BuildGraph(min_overlap=40) as g:
        g.generate_reads(spec)
        g.run_graph()
        e1, r1, c1 = g.fetch_edges(), g.fetch_contained(), g.counters()
    edges, rows, info, _ = run_ranks(3, 40, lambda g: g.dist_generate_reads(spec))
    ce1, cc1 = canon_hip(e1, r1)
    ce2, cc2 = canon_hip(edges, rows)
    assert np.array_equal(ce1, ce2) and np.array_equal(cc1, cc2)
    assert info["e_pre"] == c1["e_pre"]


def test_ownership_arithmetic_matches_the_library():
    from disco_amd import buildgraph, launch

    gs = [buildgraph.BuildGraph(min_overlap=40) for _ in range(3)]
    try:
        buildgraph.BuildGraph.comm_init_local(gs)
        for n in (0, 1, 64, 65, 1000, 12345677):
            for r, g in enumerate(gs):
                assert g.dist_range(n) == launch.owner_range(n, r, 3)
                assert (g.rank, g.world) == (r, 3)
    finally:
        for g in gs:
            g.close()


def test_partition_of_host_edges_keeps_components_together():
    """disco_partition_edges (buildG --gpus N): the edges of all ranks, held by the host, dealt to files by component"""
    from disco_amd import buildgraph

    rng = np.random.default_rng(5)
    n, ncomp = 20000, 400
    comp = rng.integers(0, ncomp, n)
    order = np.argsort(comp, kind="stable")
    e = np.zeros(0, dtype=buildgraph.EDGE_DTYPE)
    src, dst = [], []
    for c in range(ncomp):  # a path through every component
        ids = order[comp[order] == c]
        src += list(ids[:-1])
        dst += list(ids[1:])
    e = np.zeros(len(src), dtype=buildgraph.EDGE_DTYPE)
    e["src"], e["dst"] = np.minimum(src, dst), np.maximum(src, dst)
    with buildgraph.BuildGraph(min_overlap=40) as g:
        f = g.partition_edges(e, n, 7)
    assert f.max() < 7 and len(np.unique(f)) == 7
    file_of_comp = {}
    for s, fi in zip(e["src"], f):
        assert file_of_comp.setdefault(int(comp[s]), int(fi)) == int(fi)


@pytest.mark.parametrize("name,G", [("mixed_4k", 2), ("mixed_4k", 3), ("k30_6k", 4), ("contigs_20k", 2)])
def test_ranks_two_pass_verify_equal_reference(name, G):
    """DISCO_FLAG_TWO_PASS_VERIFY inside a multi-rank pass: the containment exchange moves between the two verify passes"""
    from disco_amd import buildgraph

    reads, fidx, mo = gu.case_inputs(name)
    edges, rows, info, _ = run_ranks_reads(reads, mo, G, flags=buildgraph.FLAG_TWO_PASS_VERIFY)
    ce, cc = canon_hip(edges, rows, fidx)
    gu.check_against_golden(name, ce, cc)
    assert info["regime"] == 0


def test_config4_reads_through_8_ranks_hash_like_the_reference():
    """BASELINE config 4's data — the 50 M x 150 bp reads of config 3 — through the 8-rank flow (here: 8 contexts on ONE GPU over the
    in-process transport; RCCL refuses two ranks per device): the union of the ranks' edges and contained rows must hash to the
    digests of the files the REAL reference wrote for these reads (tests/golden/cases_big.json: u150_50m)."""
    import json
    import os

    from oracle import pyoracle

    c = json.load(open(os.path.join(gu.GOLD, "cases_big.json")))["u150_50m"]
    spec = readgen.GenSpec.coverage(c["seed"], c["reads"], c["read_len"], c["coverage"], n_contigs=c["n_contigs"])
    e, r, info, infos = run_ranks(8, c["min_overlap"], lambda g: g.dist_generate_reads(spec))
    assert info["regime"] == 0 and info["asymmetric_pairs"] == 0 and info["cap_bind_sites"] == 0
    assert (info["e_out"], info["n_contained"]) == (c["n_edges"], c["n_contained"])
    one = np.int64(1)
    ce = pyoracle.canonical_edges_large(e["src"].astype(np.int64) + one, e["dst"].astype(np.int64) + one, e["orient"], e["offset"], e["len_src"], e["len_dst"])
    del e
    assert pyoracle.digest_array(ce) == c["edges_sha256"]
    cc = np.stack([r["contained"].astype(np.int64) + one, r["super"].astype(np.int64) + one] + [np.asarray(r[k], dtype=np.int64) for k in ("orient", "len2", "len1", "start")], axis=1)
    cc = cc[np.lexsort(tuple(cc[:, i] for i in range(5, -1, -1)))]
    assert pyoracle.digest_array(cc) == c["contained_sha256"]
    # every exchange of the north-star design moved bytes, none of them the order-dependent regime's
    sent = {k: sum(i["bytes_sent"][k] for i in infos) for k in infos[0]["bytes_sent"]}
    assert all(sent[k] > 0 for k in ("reads", "index_records", "index_shards", "contain", "row_requests", "row_data", "push")) and sent["adjacency"] == 0


def test_config5_shape_through_8_ranks_equals_reference():
    """BASELINE config 5's generator settings (metagenome-like, 100-250 bp, 79 % of the reads contained) at the 10 M reads the REAL
    reference was run on, through 8 ranks with the two-pass verify (the containment exchange between its passes): same digests"""
    import json
    import os

    from disco_amd import buildgraph
    from oracle import pyoracle

    c = json.load(open(os.path.join(gu.GOLD, "cases_big.json")))["s100_250_10m"]
    spec = readgen.GenSpec.coverage(c["seed"], c["reads"], c["read_len"], c["coverage"], n_contigs=c["n_contigs"], len_max=c["len_max"], skew=c["skew"])
    e, r, info, _ = run_ranks(8, c["min_overlap"], lambda g: g.dist_generate_reads(spec), flags=buildgraph.FLAG_TWO_PASS_VERIFY)
    assert info["regime"] == 0 and (info["e_out"], info["n_contained"]) == (c["n_edges"], c["n_contained"])
    one = np.int64(1)
    ce = pyoracle.canonical_edges_large(e["src"].astype(np.int64) + one, e["dst"].astype(np.int64) + one, e["orient"], e["offset"], e["len_src"], e["len_dst"])
    assert pyoracle.digest_array(ce) == c["edges_sha256"]
    cc = np.stack([r["contained"].astype(np.int64) + one, r["super"].astype(np.int64) + one] + [np.asarray(r[k], dtype=np.int64) for k in ("orient", "len2", "len1", "start")], axis=1)
    cc = cc[np.lexsort(tuple(cc[:, i] for i in range(5, -1, -1)))]
    assert pyoracle.digest_array(cc) == c["contained_sha256"]


def test_ranks_one_communicator_equals_reference(monkeypatch):
    """DISCO_DIST_ONE_COMM=1: the all-gather of the reads on the pass's own communicator and stream, in front of the index exchanges
    (no second communicator in flight) — same result"""
    monkeypatch.setenv("DISCO_DIST_ONE_COMM", "1")
    reads, fidx, mo = gu.case_inputs("mixed_4k")
    edges, rows, info, _ = run_ranks_reads(reads, mo, 3)
    ce, cc = canon_hip(edges, rows, fidx)
    gu.check_against_golden("mixed_4k", ce, cc)
    assert info["regime"] == 0 and info["bytes_sent"]["reads"] > 0


@pytest.mark.parametrize("name,G", [("u150_5k", 2), ("u150_5k", 3), ("mixed_4k", 4), ("contigs_20k", 8), ("u150_5k", 1)])
def test_ranks_with_the_index_kept_partitioned_equal_reference(name, G):
    """DISCO_DIST_KEEP_INDEX_PARTITIONED (the split hashData of buildG-MPIRMA): no rank ever holds more than its
    slice of the bucket table and of the records; every lookup travels to the bucket's owner, the matching records travel back"""
    reads, fidx, mo = gu.case_inputs(name)
    edges, rows, info, infos = run_ranks_reads(reads, mo, G, partitioned_index=True)
    ce, cc = canon_hip(edges, rows, fidx)
    gu.check_against_golden(name, ce, cc)
    assert info["regime"] == 0 and info["asymmetric_pairs"] == 0
    if G > 1:
        assert sum(i["bytes_sent"]["queries"] for i in infos) > 0 and sum(i["bytes_sent"]["hits"] for i in infos) > 0
        assert sum(i["bytes_sent"]["index_shards"] for i in infos) == 0  # nothing of the built index is ever replicated


@pytest.mark.parametrize("G", [2, 3])
def test_ranks_partitioned_index_on_repeats_equals_single_gpu(G):
    """repeats: windows whose smallest m-mer hash ties (reads without a usable run list: their lookups are made the long way, one
    per range of windows with the same occurrence and strand), the cap binds, pairs are found from one side only"""
    from oracle import pyoracle

    reads, fidx, mo = gu.case_inputs("repeats_8k")
    edges, rows, info, _ = run_ranks_reads(reads, mo, G, partitioned_index=True)
    ce, cc = canon_hip(edges, rows, fidx)
    oce, occ, ocnt = pyoracle.oracle_canonical(reads, fidx, mo, count_hits=True)
    assert info["asymmetric_pairs"] == ocnt["asymmetric_pairs"] > 0 and info["e_pre"] == ocnt["e_pre"] and info["kmer_hits"] == ocnt["kmer_hits"]
    assert np.array_equal(cc, occ) and np.array_equal(ce, oce)


def test_ranks_partitioned_index_generated_reads_match_single_gpu():
    """200 k generated reads, 4 ranks: every counter of the job equals the single-GPU pass, k-mer hits included"""
    from disco_amd import buildgraph

    spec = readgen.GenSpec.coverage(seed=7, n_reads=200_000, read_len=150, cov=30.0)
    with buildgraph.BuildGraph(min_overlap=40) as g:
        g.generate_reads(spec)
        g.run_graph()
        e1, r1, c1 = g.fetch_edges(), g.fetch_contained(), g.counters()
    edges, rows, info, infos = run_ranks(4, 40, lambda g: g.dist_generate_reads(spec), passes=2, partitioned_index=True)
    for key in ("e_pre", "e_out", "n_contained", "probes", "kmer_hits", "cap_bind_sites"):
        assert info[key] == c1[key], (key, info[key], c1[key])
    ce1, cc1 = canon_hip(e1, r1)
    ce2, cc2 = canon_hip(edges, rows)
    assert np.array_equal(ce1, ce2) and np.array_equal(cc1, cc2)


def test_config2_reads_through_4_ranks_with_partitioned_index_hash_like_the_reference():
    """BASELINE config 2's reads (1 M x 150 bp) through 4 ranks that keep the index partitioned: the digests of the REAL reference's
    files (tests/golden/cases_big.json: u150_1m)"""
    import json
    import os

    from oracle import pyoracle

    c = json.load(open(os.path.join(gu.GOLD, "cases_big.json")))["u150_1m"]
    spec = readgen.GenSpec.coverage(c["seed"], c["reads"], c["read_len"], c["coverage"], n_contigs=c.get("n_contigs", 1))
    e, r, info, infos = run_ranks(4, c["min_overlap"], lambda g: g.dist_generate_reads(spec), partitioned_index=True)
    assert info["regime"] == 0 and (info["e_out"], info["n_contained"]) == (c["n_edges"], c["n_contained"])
    one = np.int64(1)
    ce = pyoracle.canonical_edges_large(e["src"].astype(np.int64) + one, e["dst"].astype(np.int64) + one, e["orient"], e["offset"], e["len_src"], e["len_dst"])
    assert pyoracle.digest_array(ce) == c["edges_sha256"]
    cc = np.stack([r["contained"].astype(np.int64) + one, r["super"].astype(np.int64) + one] + [np.asarray(r[k], dtype=np.int64) for k in ("orient", "len2", "len1", "start")], axis=1)
    cc = cc[np.lexsort(tuple(cc[:, i] for i in range(5, -1, -1)))]
    assert pyoracle.digest_array(cc) == c["contained_sha256"]
    sent = {k: sum(i["bytes_sent"][k] for i in infos) for k in infos[0]["bytes_sent"]}
    assert sent["queries"] > 0 and sent["hits"] > 0 and sent["index_shards"] == 0


@pytest.mark.parametrize("name,G", [("k30_6k", 3), ("long_2k", 2), ("k64_3k", 4), ("k94_4k", 3)])
def test_ranks_partitioned_index_on_shapes_without_runs(name, G):
    """windows other than 17 m-mers / reads beyond 256 bases: the index pass leaves no minimizer runs, every read's lookups are made
    the long way (pq_slow_kernel) — same result"""
    reads, fidx, mo = gu.case_inputs(name)
    edges, rows, info, infos = run_ranks_reads(reads, mo, G, partitioned_index=True)
    ce, cc = canon_hip(edges, rows, fidx)
    gu.check_against_golden(name, ce, cc)
    assert sum(i["bytes_sent"]["queries"] for i in infos) > 0 and sum(i["bytes_sent"]["index_shards"] for i in infos) == 0


def test_a_pass_allocates_nothing_and_reports_what_it_did():
    """round 4: the arena of a multi-GPU context (one device allocation before the first collective) serves every buffer of a pass; the
    pass reports its operations on the communicator, its blocking host waits, the allocations that reached the runtime and its
    kernel time (disco_dist_info)"""
    spec = readgen.GenSpec.coverage(seed=5, n_reads=60_000, read_len=150, cov=30.0, n_contigs=3)
    edges, rows, info, infos = run_ranks(3, 40, lambda g: g.dist_generate_reads(spec), passes=3)
    for i in infos:   # the last of three passes on warm contexts
        assert i["device_allocs"] == 0 and i["device_frees"] == 0, i
        assert i["arena_bytes"] > 0 and 0 < i["arena_peak"] <= i["arena_bytes"] and i["hbm_peak"] >= i["arena_peak"]
        assert 10 <= i["comm_ops"] <= 24 and 10 <= i["host_syncs"] <= 45 and i["kernel_ms"] > 0, i  # (round 4: 23 / 52 at this shape)
    assert info["e_out"] == len(edges) > 0


@pytest.mark.parametrize("G", [2, 3, 4])
def test_ranks_with_a_tail_of_long_reads_keep_64_byte_rows(G):
    """round 6: two classes of rows under a communicator. tail_20k — 20 000 reads, 1 % of them 600 bp — used to put every
    rank on a 24-word stride (the generic probe, the wide-row verify); now every rank converts its replica once the gather is through and
    runs the 64-byte kernels, the long reads of its range through the long class. Two passes (the second finds the table converted);
    the REAL reference's files"""
    reads, fidx, mo = gu.case_inputs("tail_20k")
    seen = {}
    edges, rows, info, infos = run_ranks_reads(reads, mo, G, passes=2, inspect=seen)
    ce, cc = canon_hip(edges, rows, fidx)
    gu.check_against_golden("tail_20k", ce, cc)
    n_long = sum(len(r) > 256 for r in reads)
    assert all(seen[r]["long_rows"] == n_long > 100 and seen[r]["probe_run_words"] > 0 for r in range(G)), seen
    assert all(i["placement"] == 0 for i in infos)  # over id ranges: the own reads' index pass needs the converted table


def test_ranks_with_long_reads_one_stride_on_request(monkeypatch):
    """DISCO_DIST_NO_TWO_CLASS=1: the table of rounds 1-5 (one stride for everybody) — same files"""
    monkeypatch.setenv("DISCO_DIST_NO_TWO_CLASS", "1")
    reads, fidx, mo = gu.case_inputs("tail_20k")
    seen = {}
    edges, rows, info, infos = run_ranks_reads(reads, mo, 3, inspect=seen)
    ce, cc = canon_hip(edges, rows, fidx)
    gu.check_against_golden("tail_20k", ce, cc)
    assert all(seen[r]["long_rows"] == 0 for r in range(3))


def test_ranks_two_classes_without_host_waits_and_with_three_word_kmers(monkeypatch):
    """the same flow over the transport without host waits, and at min-overlap 80 (k = 79) on a generated set: the single-GPU pass's result"""
    from tests.test_gpu_two_class import mixed_reads

    monkeypatch.setenv("DISCO_LOOP_ASYNC", "1")
    reads = mixed_reads(21, 4000, 150, 150, 30.0, 0.02, 300, 700)
    for mo in (40, 80):
        e1, r1, c1 = run_hip_reads(reads, mo)
        seen = {}
        edges, rows, info, _ = run_ranks_reads(reads, mo, 4, inspect=seen)
        a, b = canon_hip(edges, rows), canon_hip(e1, r1)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), mo
        assert all(seen[r]["long_rows"] > 0 for r in range(4)) and info["e_out"] == c1["e_out"]
