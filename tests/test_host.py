"""the C++ host of the buildG drop-in: input stage on CPU (vs the oracle's restatement of Dataset), whole CLI on the GPU."""
import glob
import os
import subprocess

import numpy as np
import pytest

from disco_amd import build
from oracle import pyoracle, refrun
from tests import golden_util as gu

BIN = os.path.join(os.path.dirname(build.HERE), "disco_amd", "bin")


def _dump(min_overlap, pe=(), se=()):
    build.build_host()
    cmd = [os.path.join(BIN, "fastx_dump"), str(min_overlap)]
    if pe:
        cmd += ["-pe", ",".join(pe)]
    if se:
        cmd += ["-se", ",".join(se)]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, check=True, text=True).stdout.strip().split("\n")
    tail = out[-1].split()
    reads = [l.split("\t") for l in out[:-1] if l]
    return [r[1] for r in reads], np.array([int(r[0]) for r in reads], dtype=np.uint64), int(tail[1]), int(tail[3])


def test_input_stage_matches_oracle_on_the_multifile_case():
    c = gu.CASES["multifile"]
    pe = [os.path.join(gu.GOLD, f) for f in c["pe"]]
    se = [os.path.join(gu.GOLD, f) for f in c["se"]]
    reads, fidx, total, stride = _dump(c["min_overlap"], pe, se)
    oreads, ofidx, ototal = pyoracle.load_good_reads(pe + se, c["min_overlap"])
    assert total == ototal and reads == oreads and np.array_equal(fidx, ofidx)
    assert stride == (max(map(len, reads)) + 31) // 32


def test_input_stage_reference_fastas_and_gz(tmp_path):
    import gzip

    fa = os.path.join(gu.GOLD, "reference_data", "10reads_containedReads.fasta")
    reads, fidx, total, _ = _dump(30, se=[fa])
    oreads, ofidx, ototal = pyoracle.load_good_reads([fa], 30)
    assert (reads, total) == (oreads, ototal) and np.array_equal(fidx, ofidx)
    gz = str(tmp_path / "r.fasta.gz")
    with gzip.open(gz, "wb") as f:
        f.write(open(fa, "rb").read())
    reads2, fidx2, total2, _ = _dump(30, se=[gz])
    assert reads2 == reads and total2 == total


def test_input_stage_errors(tmp_path):
    build.build_host()
    bad = tmp_path / "bad.txt"
    bad.write_text("ACGT\n")
    p = subprocess.run([os.path.join(BIN, "fastx_dump"), "30", "-se", str(bad)], stderr=subprocess.PIPE, text=True)
    assert p.returncode != 0 and "Unknown input file format" in p.stderr
    p = subprocess.run([os.path.join(BIN, "fastx_dump"), "30", "-se", str(tmp_path / "missing.fa")], stderr=subprocess.PIPE, text=True)
    assert p.returncode != 0 and "Unable to open file" in p.stderr


def test_cli_usage_and_unknown_flag():
    build.build_host()
    exe = os.path.join(BIN, "buildG")
    p = subprocess.run([exe], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    assert p.returncode == 0 and "Usage: buildG" in p.stderr          # BG/main.cpp:93-102
    p = subprocess.run([exe, "-x"], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    assert p.returncode == 1 and "Unknown option: -x" in p.stderr     # BG/main.cpp:133-148
    p = subprocess.run([exe, "-h"], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    assert p.returncode == 0
    p = subprocess.run([exe, "-se", "x.fa", "-f", "/tmp/none", "-p", "/nonexistent.cfg"], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    assert p.returncode == 1 and "Unable to open parameter file" in p.stderr


def test_cli_rejects_malformed_numbers(tmp_path):
    """a missing or non-numeric value prints the usage and exits 1 (the reference's stoull would abort the process)"""
    build.build_host()
    exe = os.path.join(BIN, "buildG")
    cfg = tmp_path / "disco.cfg"
    cfg.write_text("MinOverlap4BuildGraph = 40\n")
    for argv in (["-se", "x.fa", "-f", "g", "-p", str(cfg), "-t"], ["-se", "x.fa", "-f", "g", "-p", str(cfg), "-t", "four"],
                 ["-se", "x.fa", "-f", "g", "-p", str(cfg), "-t", "70000"], ["-se", "x.fa", "-f", "g", "-p", str(cfg), "-m", "-3"],
                 ["-se", "x.fa", "-f", "g", "-p", str(cfg), "--gpus", "0"]):
        p = subprocess.run([exe] + argv, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
        assert p.returncode == 1 and "Usage: buildG" in p.stderr and "needs a number" in p.stderr, (argv, p.stderr)
    bad = tmp_path / "bad.cfg"
    bad.write_text("MinOverlap4BuildGraph = abc\n")
    p = subprocess.run([exe, "-se", "x.fa", "-f", "g", "-p", str(bad)], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    assert p.returncode == 1 and "is not a number" in p.stderr


def _file_tags(threads, gpus, mpi_names):
    """(edge-file tags, contained-file tags) the run must produce"""
    if not mpi_names:
        t = [str(i) for i in range(threads)]
        return t, t
    first = 1 if threads > 1 else 0  # runDisco-MPI.sh:165-186: edge files of threads 1..t-1, contained files of threads 0..t-1
    return ([f"{r}_{i}" for r in range(gpus) for i in range(first, threads)], [f"{r}_{i}" for r in range(gpus) for i in range(threads)])


@pytest.mark.gpu
def test_buildg_multi_rank_watchdog_ends_a_stalled_stage(tmp_path):
    """round 5: a rank that never enters the pass (DISCO_TEST_STALL_RANK) leaves the others inside their first collective; after
    DISCO_WATCHDOG_S seconds without progress on any rank buildG says where every rank stands and exits non-zero — no hang, no
    re-exec, no GC=Complete line (the reference's MPI binaries hang for ever in MPI_Recv, MPI/OverlapGraph.cpp:218-246)"""
    import time

    build.build_host()
    c = gu.CASES["multifile"]
    cfg = tmp_path / "disco.cfg"
    cfg.write_text(f"MinOverlap4BuildGraph = {c['min_overlap']}\n")
    prefix = str(tmp_path / "g")
    se = ",".join(os.path.join(gu.GOLD, f) for f in c["se"])
    env = dict(os.environ, DISCO_TEST_STALL_RANK="1", DISCO_WATCHDOG_S="2")
    t0 = time.time()
    p = subprocess.run([os.path.join(BIN, "buildG"), "-se", se, "-f", prefix, "-p", str(cfg), "-t", "2", "--gpus", "2", "--same-device"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, env=env, timeout=120)
    assert p.returncode == 3, p.stdout
    # the stalled rank stands IN FRONT of the pass (it never entered it: ADVICE r5), the other one inside its first collective
    assert "no rank has made progress" in p.stdout and "rank 1 in front of disco_dist_run_graph" in p.stdout and "rank 0 in disco_dist_run_graph" in p.stdout, p.stdout
    # round 6: the launcher — a process that never touched the GPU — started the stage ONCE more with one communicator before giving up
    assert p.stdout.count("no rank has made progress") == 2 and "starting it ONCE more with one communicator" in p.stdout, p.stdout
    assert time.time() - t0 < 60
    assert not os.path.exists(prefix + "_CheckpointInfo.txt") or "GC=Complete" not in open(prefix + "_CheckpointInfo.txt").read()


@pytest.mark.gpu
def test_buildg_launcher_retries_a_stalled_first_try_and_delivers(tmp_path):
    """the first child stalls (DISCO_TEST_STALL_FIRST_TRY), its watchdog ends it, the launcher's second child — DISCO_DIST_ONE_COMM=1 — builds
    the graph: exit 0, the reference's files"""
    build.build_host()
    c = gu.CASES["multifile"]
    cfg = tmp_path / "disco.cfg"
    cfg.write_text(f"MinOverlap4BuildGraph = {c['min_overlap']}\n")
    prefix = str(tmp_path / "g")
    pe = ",".join(os.path.join(gu.GOLD, f) for f in c["pe"])
    se = ",".join(os.path.join(gu.GOLD, f) for f in c["se"])
    env = dict(os.environ, DISCO_TEST_STALL_RANK="0", DISCO_TEST_STALL_FIRST_TRY="1", DISCO_WATCHDOG_S="2", DISCO_VERBOSE="1")
    p = subprocess.run([os.path.join(BIN, "buildG"), "-pe", pe, "-se", se, "-f", prefix, "-p", str(cfg), "-t", "2", "--gpus", "2", "--same-device"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, env=env, timeout=180)
    assert p.returncode == 0, p.stdout
    assert p.stdout.count("no rank has made progress") == 1 and "starting it ONCE more with one communicator" in p.stdout, p.stdout
    edges = refrun.parse_pargraph(sorted(glob.glob(prefix + "_*_parGraph.txt")))
    cont = refrun.parse_contained(sorted(glob.glob(prefix + "_*_containedReads.txt")))
    gu.check_against_golden("multifile", edges, cont)
    assert "GC=Complete" in open(prefix + "_CheckpointInfo.txt").read()


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [2, 3])
def test_buildg_multi_rank_on_a_set_with_a_few_long_reads(tmp_path, gpus):
    """round 6: `buildG --gpus N` on a set with 1 % reads of 600 bp (tail_20k, the REAL reference's files as the fixture): every rank runs the
    64-byte-row kernels (two classes of rows under a communicator) — the log says so — and the files are the reference's"""
    from disco_amd import readgen

    build.build_host()
    reads, fidx, mo = gu.case_inputs("tail_20k")
    fa = tmp_path / "tail.fasta"
    readgen.write_fasta(str(fa), reads)
    cfg = tmp_path / "disco.cfg"
    cfg.write_text(f"MinOverlap4BuildGraph = {mo}\n")
    prefix = str(tmp_path / "g")
    p = subprocess.run([os.path.join(BIN, "buildG"), "-se", str(fa), "-f", prefix, "-p", str(cfg), "-t", "2", "--gpus", str(gpus), "--same-device"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, DISCO_VERBOSE="1"), timeout=300)
    assert p.returncode == 0, p.stdout
    assert p.stdout.count("two classes of rows") == gpus, p.stdout  # one line per rank's context
    edges = refrun.parse_pargraph(sorted(glob.glob(prefix + "_*_parGraph.txt")))
    cont = refrun.parse_contained(sorted(glob.glob(prefix + "_*_containedReads.txt")))
    gu.check_against_golden("tail_20k", edges, cont)


@pytest.mark.gpu
@pytest.mark.parametrize("threads,gpus,mpi_names,part", [(1, 1, False, False), (4, 1, False, False), (3, 2, False, False), (3, 2, True, False), (2, 3, True, False),
                                                         (1, 2, True, False), (3, 3, False, True)])
def test_buildg_cli_multifile_matches_reference(tmp_path, threads, gpus, mpi_names, part):
    """whole drop-in: argv in, files out; canonical content identical to the real reference's files — on one GPU and with
    --gpus N ranks (here all on one device: --same-device), plain and buildG-MPI file names"""
    build.build_host()
    c = gu.CASES["multifile"]
    cfg = tmp_path / "disco.cfg"
    cfg.write_text(f"# test\nMinOverlap4BuildGraph = {c['min_overlap']}\nPrintContigs = false\n")
    prefix = str(tmp_path / "g")
    pe = ",".join(os.path.join(gu.GOLD, f) for f in c["pe"])
    se = ",".join(os.path.join(gu.GOLD, f) for f in c["se"])
    cmd = [os.path.join(BIN, "buildG"), "-pe", pe, "-se", se, "-f", prefix, "-p", str(cfg), "-t", str(threads), "-m", "8"]
    if gpus > 1:
        cmd += ["--gpus", str(gpus), "--same-device"]
    if mpi_names:
        cmd += ["--mpi-names"]
    if part:  # the index stays hash-partitioned over the ranks (buildG-MPIRMA's split hashData)
        cmd += ["--partitioned-index"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
    etags, ctags = _file_tags(threads, gpus, mpi_names)
    for t in etags:  # every file must exist, empty or not (SG/DataSet.cpp:294-295)
        assert os.path.exists(f"{prefix}_{t}_parGraph.txt"), t
    for t in ctags:
        assert os.path.exists(f"{prefix}_{t}_containedReads.txt"), t
    assert len(glob.glob(prefix + "_*_parGraph.txt")) == len(etags) and len(glob.glob(prefix + "_*_containedReads.txt")) == len(ctags)
    edges = refrun.parse_pargraph(sorted(glob.glob(prefix + "_*_parGraph.txt")))
    cont = refrun.parse_contained(sorted(glob.glob(prefix + "_*_containedReads.txt")))
    gu.check_against_golden("multifile", edges, cont)
    idmap = open(prefix + "_ReadIDMap.txt").read().replace(gu.GOLD + "/inputs/", "")
    assert idmap == c["read_id_map"]
    assert open(prefix + "_CheckpointInfo.txt").read() == "CCR=Complete\nGC=Complete\n"
    # flag rule: a node flagged as marked in file t has ALL its edges in file t
    all_edges = {}
    per_file = []
    for t in etags:
        rows = [l.rstrip("\n").split("\t") for l in open(f"{prefix}_{t}_parGraph.txt")]
        per_file.append(rows)
        for a, b, info in rows:
            all_edges.setdefault(int(a), set()).add((int(a), int(b)))
            all_edges.setdefault(int(b), set()).add((int(a), int(b)))
    for rows in per_file:
        here = {}
        marked = set()
        for a, b, info in rows:
            a, b, flag = int(a), int(b), int(info.rsplit(",", 1)[1])
            here.setdefault(a, set()).add((a, b))
            here.setdefault(b, set()).add((a, b))
            if flag in (0, 2):
                marked.add(a)
            if flag in (1, 2):
                marked.add(b)
        for v in marked:
            assert here[v] == all_edges[v]
    # the files are cut along connected components: every edge is written once, with both ends marked
    n_lines = sum(len(r) for r in per_file)
    assert n_lines == len({e for es in all_edges.values() for e in es})
    assert all(info.rsplit(",", 1)[1] == "2" for rows in per_file for _, _, info in rows)
    # contained rows of one containing read are contiguous (SG/DataSet.cpp:316-335)
    for t in ctags:
        supers = [l.split("\t")[1] for l in open(f"{prefix}_{t}_containedReads.txt")]
        seen, prev = set(), None
        for s in supers:
            if s != prev:
                assert s not in seen
                seen.add(s)
                prev = s
    # re-running the same command is a no-op (BG/main.cpp:48-52)
    p2 = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p2.returncode == 0 and "Graph already exists" in p2.stdout


@pytest.mark.gpu
def test_binary_side_output_stands_for_the_text_files(tmp_path):
    """--binary-out writes fixed-size records beside the text files; converted back (disco_amd/edgefile.py) they ARE
    the text files, line for line per file (lines of a file sorted: the text writer groups by file in chunks, the binary file
    keeps the edges in fetch order)"""
    from disco_amd import edgefile

    build.build_host()
    c = gu.CASES["multifile"]
    cfg = tmp_path / "disco.cfg"
    cfg.write_text(f"MinOverlap4BuildGraph = {c['min_overlap']}\n")
    prefix = str(tmp_path / "g")
    pe = ",".join(os.path.join(gu.GOLD, f) for f in c["pe"])
    se = ",".join(os.path.join(gu.GOLD, f) for f in c["se"])
    p = subprocess.run([os.path.join(BIN, "buildG"), "-pe", pe, "-se", se, "-f", prefix, "-p", str(cfg), "-t", "3", "--binary-out"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
    texts = edgefile.text_files(prefix)
    assert len(texts) == 6
    for path, text in texts.items():
        assert sorted(open(path).read().splitlines()) == sorted(text.splitlines()), path
    assert "".join(texts[f"{prefix}_{t}_containedReads.txt"] for t in range(3)) == "".join(open(f"{prefix}_{t}_containedReads.txt").read() for t in range(3))
    # --no-text: the file lists still exist, empty; the binary pair carries everything
    prefix2 = str(tmp_path / "h")
    p = subprocess.run([os.path.join(BIN, "buildG"), "-pe", pe, "-se", se, "-f", prefix2, "-p", str(cfg), "-t", "3", "--no-text"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
    assert all(os.path.getsize(f"{prefix2}_{t}_parGraph.txt") == 0 and os.path.getsize(f"{prefix2}_{t}_containedReads.txt") == 0 for t in range(3))
    e2, _ = edgefile.read_edges(prefix2 + "_edges.bin")
    e1, _ = edgefile.read_edges(prefix + "_edges.bin")
    key = lambda r: np.sort(r[["src", "dst", "orient", "offset", "len_src", "len_dst"]], order=["src", "dst"])  # noqa: E731
    assert np.array_equal(key(e1), key(e2))
    edges = refrun.parse_pargraph(sorted(glob.glob(prefix + "_*_parGraph.txt")))
    assert len(e1) == len(edges) == c["n_edges"]


def _parsimplify(edge_file, out_file, min_ovl):
    exe = os.path.join(os.path.dirname(refrun.REF_BIN), "parsimplify_ref")
    p = subprocess.run([exe, edge_file, out_file, str(min_ovl), "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout[-2000:]
    comps = []
    for line in open(out_file):
        f = line.rstrip("\n").split("\t")
        if len(f) < 3:
            continue
        ids = {int(f[0]), int(f[1])}
        if len(f) > 3:
            ids |= {int(t.split(",")[0]) for t in f[3].strip("()").split(")(") if t}
        comps.append(frozenset(ids))
    return sorted(comps, key=lambda s: (len(s), min(s)))


@pytest.mark.gpu
@pytest.mark.skipif(not (refrun.available() and os.path.exists(os.path.join(os.path.dirname(refrun.REF_BIN), "parsimplify_ref"))),
                    reason="prebuilt reference binaries (make -C oracle ref ref_parsimplify) not present")
def test_files_load_in_the_reference_parsimplify(tmp_path):
    """loader compatibility with the immediate consumer: the reference's own parsimplify contracts the
    drop-in's edge file into the same composite edges as it does the reference buildG's edge file"""
    from disco_amd import readgen

    build.build_host()
    # two contigs, mixed lengths -> several composite edges, contained reads, both strands
    spec = readgen.GenSpec.coverage(seed=77, n_reads=6000, read_len=120, cov=18.0, n_contigs=3, len_max=180)
    fa = str(tmp_path / "r.fasta")
    readgen.write_fasta(fa, readgen.generate_reads(spec))
    cfg = tmp_path / "disco.cfg"
    cfg.write_text("MinOverlap4BuildGraph = 40\n")
    ours = str(tmp_path / "ours")
    p = subprocess.run([os.path.join(BIN, "buildG"), "-se", fa, "-f", ours, "-p", str(cfg), "-t", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
    ref = refrun.run_reference([fa], 40, threads=1, workdir=str(tmp_path / "ref") if os.makedirs(tmp_path / "ref", exist_ok=True) is None else None)
    a = _parsimplify(ours + "_0_parGraph.txt", str(tmp_path / "ours_simple.txt"), 40)
    b = _parsimplify(ref["prefix"] + "_0_parGraph.txt", str(tmp_path / "ref_simple.txt"), 40)
    assert len(a) > 1 and a == b
    # and with several partial-graph files every file loads
    ours3 = str(tmp_path / "ours3")
    p = subprocess.run([os.path.join(BIN, "buildG"), "-se", fa, "-f", ours3, "-p", str(cfg), "-t", "3"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
    parts = []
    for t in range(3):
        parts += _parsimplify(f"{ours3}_{t}_parGraph.txt", str(tmp_path / f"ours3_{t}.txt"), 40)
    # the files are cut along connected components, so the per-file pre-simplification is complete: together the three files
    # contract into exactly the composite edges of the single file
    assert sorted(parts, key=lambda s: (len(s), min(s))) == a


def test_input_stage_falls_back_on_awkward_fasta(tmp_path):
    """'>' inside a header line and in the middle of a sequence line: the parallel splitter must hand over to the literal
    sequential one (the reference treats EVERY '>' after the header line as a record delimiter, BG/Dataset.cpp:273)"""
    good = "ACGTTGCAAGCTTAGCCGATCGGATTACAGCTAGCTAGGATCCGATTAGCATGCAAGT"
    fa = tmp_path / "awkward.fasta"
    fa.write_text(f">r1 desc>more\n{good}\n>r2\n{good[:30]}>{good[5:]}\n>r3\n{good[::-1]}\n>")
    reads, fidx, total, _ = _dump(30, se=[str(fa)])
    oreads, ofidx, ototal = pyoracle.load_good_reads([str(fa)], 30)
    assert (reads, total) == (oreads, ototal) and np.array_equal(fidx, ofidx)
    # r1 | r2 up to the stray '>' | a record whose "header" is the rest of that line and whose sequence is empty | r3
    assert total == 4 and len(reads) == 2  # the 30-base half of r2 is not longer than the minimum overlap


def test_read_filter_matches_oracle_on_adversarial_reads(tmp_path):
    """low-complexity, motif-rich and micro-repeat reads around every threshold of Dataset::testRead (BG/Dataset.cpp:403-452):
    the host filter (which skips motif scans that cannot reach the threshold) must keep exactly what the restatement keeps"""
    rng = np.random.default_rng(123)
    motifs = ["AC", "AG", "AT", "CG", "CT", "GT", "AAT", "ATA", "TAA", "AAC", "ACA", "CAA", "AAG", "AGA", "GAA", "GGGGCC"]
    reads = []
    for i in range(6000):
        L = int(rng.integers(31, 200))
        kind = i % 6
        if kind == 0:      # random
            s = "".join(rng.choice(list("ACGT"), L))
        elif kind == 1:    # one base near the 70 % threshold
            b = "ACGT"[i % 4]
            frac = rng.uniform(0.62, 0.78)
            s = "".join(b if rng.random() < frac else rng.choice(list("ACGT")) for _ in range(L))
        elif kind == 2:    # motif covering about half of the read, scattered
            m = motifs[int(rng.integers(0, len(motifs)))]
            reps = int(L * rng.uniform(0.40, 0.60) / len(m))
            parts = [m] * reps + list(rng.choice(list("ACGT"), max(L - reps * len(m), 0)))
            rng.shuffle(parts)
            s = "".join(parts)[:L]
        elif kind == 3:    # motif run + random tail
            m = motifs[int(rng.integers(0, len(motifs)))]
            run = int(L * rng.uniform(0.45, 0.55))
            s = (m * (run // len(m) + 1))[:run] + "".join(rng.choice(list("ACGT"), L - run))
        elif kind == 4:    # micro-repeat prefix / suffix
            unit = ["AC", "AAG", "AAAT", "AATT", "TACA", "GTTT", "AGGG"][i % 7]
            rep = (unit * 10)[:29]
            body = "".join(rng.choice(list("ACGT"), L))
            s = rep + body if i % 2 else body + rep
        else:              # lower case, N, short
            s = "".join(rng.choice(list("ACGT"), L))
            if i % 3 == 0:
                s = s.lower()
            elif i % 3 == 1:
                s = s[:10] + "N" + s[11:]
        reads.append(s)
    fa = tmp_path / "adv.fasta"
    fa.write_text("".join(f">a{i}\n{s}\n" for i, s in enumerate(reads)))
    got, fidx, total, _ = _dump(30, se=[str(fa)])
    want, wfidx, wtotal = pyoracle.load_good_reads([str(fa)], 30)
    assert total == wtotal == len(reads)
    assert np.array_equal(fidx, wfidx) and got == want
    assert 1000 < len(got) < 5000  # both outcomes are well represented


REF_PS = os.path.join(os.path.dirname(refrun.REF_BIN), "parsimplify_ref")
REF_PS_INITLEN = REF_PS + "_initlen"  # the reference with EdgeSimple::copyEdge copying the two read lengths too (oracle/Makefile)


def _par_lines(path, drop_length=False):
    """sorted lines of a ParSimpleEdges file; drop_length: without the edge-length column (offset + length of the destination),
    which the STOCK reference fills from uninitialised members on some composite edges (SG/EdgeSimple.cpp:50-71); fullsimplify never
    reads that column (SG/OverlapGraph.cpp:2028-2094 takes fields 0, 1 and 5)"""
    out = []
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        info = f[2].split(",")
        out.append("\t".join(f[:2] + [",".join(info[:2] + info[3:]) if drop_length else f[2]] + f[3:]))
    return sorted(out)


def _run_ps(exe, edge_file, out, min_ovl, threads):
    p = subprocess.run([exe, str(edge_file), out, str(min_ovl), str(threads)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout[-2000:]


@pytest.mark.skipif(not os.path.exists(REF_PS_INITLEN), reason="prebuilt reference parsimplify (make -C oracle ref_parsimplify) not present")
@pytest.mark.parametrize("name", ["u150_5k", "mixed_4k", "k30_6k", "k64_3k", "long_2k", "multifile"])
@pytest.mark.parametrize("split", [False, True])
def test_partial_simplification_equals_the_reference_parsimplify(tmp_path, name, split):
    """disco_amd/bin/parsimplify (= the code behind buildG --par-simple) against the REAL parsimplify on the same
    edge file: the same lines — every (composite) edge, inner read lists and lengths included. split: a file that owns only the
    lower half of the nodes, edges to the other half carrying flags 0 / 1 — nodes that are not marked must neither be absorbed
    nor removed. The comparator is the reference with its uninitialised-length defect repaired (deterministic); the stock binary
    must agree too up to the lines that defect decides."""
    build.build_host()
    src = os.path.join(gu.GOLD, name + ".edges.txt")
    lines = [l.rstrip("\n") for l in open(src) if l.strip()]
    ids = sorted({int(x) for l in lines for x in l.split("\t")[:2]})
    cut = ids[len(ids) // 2]
    edge_file = tmp_path / "in_parGraph.txt"
    with open(edge_file, "w") as f:
        for l in lines:
            a, b = (int(x) for x in l.split("\t")[:2])
            if not split:
                f.write(l + ",2\n")
            elif a < cut or b < cut:
                f.write(l + ("," + ("2" if (a < cut and b < cut) else "0" if a < cut else "1")) + "\n")
    ours, ref, stock = str(tmp_path / "ours.txt"), str(tmp_path / "ref.txt"), str(tmp_path / "stock.txt")
    _run_ps(os.path.join(BIN, "parsimplify"), edge_file, ours, 30, 4)
    _run_ps(REF_PS_INITLEN, edge_file, ref, 30, 2)
    a, b = _par_lines(ours), _par_lines(ref)
    assert a == b
    assert len(a) < len(lines)  # something was contracted
    _run_ps(REF_PS, edge_file, stock, 30, 1)
    s1, a1 = _par_lines(stock, True), _par_lines(ours, True)
    assert len(set(s1) ^ set(a1)) <= max(4, len(a1) // 50)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_PS_INITLEN), reason="prebuilt reference parsimplify (make -C oracle ref_parsimplify) not present")
@pytest.mark.parametrize("gpus", [1, 2])
def test_buildg_par_simple_equals_parsimplify_on_its_edge_files(tmp_path, gpus):
    """buildG --par-simple: the files fullsimplify would otherwise make by running parsimplify on every edge file — written from
    the edges while they are in memory. Each must hold the same (composite) edges as the REAL parsimplify run on the edge file of
    the same run."""
    from disco_amd import readgen

    build.build_host()
    spec = readgen.GenSpec.coverage(seed=5, n_reads=30000, read_len=110, cov=20.0, n_contigs=6, len_max=190)
    fa = str(tmp_path / "r.fasta")
    readgen.write_fasta(fa, readgen.generate_reads(spec))
    cfg = tmp_path / "disco.cfg"
    cfg.write_text("MinOverlap4BuildGraph = 40\nMinOverlap4SimplifyGraph = 40\n")
    os.makedirs(tmp_path / "graph")
    prefix = str(tmp_path / "graph" / "asm")
    cmd = [os.path.join(BIN, "buildG"), "-se", fa, "-f", prefix, "-p", str(cfg), "-t", "3"] + (["--gpus", str(gpus), "--same-device"] if gpus > 1 else [])
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, DISCO_PAR_SIMPLE="1"))
    assert p.returncode == 0 and "Partial simplification" in p.stdout, p.stdout
    total = 0
    for t in range(3):
        mine = str(tmp_path / "assembly" / f"asm_{t}_ParSimpleEdges.txt")  # runDisco.sh's layout: <out>/graph/<name> -> <out>/assembly/<name>
        assert os.path.exists(mine)
        ref = str(tmp_path / f"ref_{t}.txt")
        _run_ps(REF_PS_INITLEN, f"{prefix}_{t}_parGraph.txt", ref, 40, 2)
        a, b = _par_lines(mine), _par_lines(ref)
        assert a == b, (t, len(a), len(b))
        total += len(a)
    assert 0 < total < 3000  # 6 contigs, mixed lengths: a few hundred composite edges, not 25 000 simple ones


@pytest.mark.gpu
def test_buildg_par_simple_at_scale_equals_the_reference(tmp_path):
    """10 M metagenome-like reads (BASELINE config 5's generator settings): the partial simplification buildG leaves beside its
    graph must hash like the output of the REAL parsimplify (length defect repaired, oracle/Makefile) on the edge file the REAL
    buildG wrote for the same reads — both halves of that reference run were done once in the build container
    (tests/golden/cases_big.json: s100_250_10m)."""
    import hashlib
    import json

    build.build_host()
    c = json.load(open(os.path.join(gu.GOLD, "cases_big.json")))["s100_250_10m"]
    fa = str(tmp_path / "r.fasta")
    subprocess.run([os.path.join(BIN, "readgen"), fa, str(c["reads"]), str(c["read_len"]), repr(float(c["coverage"])), str(c["seed"]), str(c["len_max"]), "583333", "1"],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    cfg = tmp_path / "disco.cfg"
    cfg.write_text("MinOverlap4BuildGraph = 40\nMinOverlap4SimplifyGraph = 30\n")
    os.makedirs(tmp_path / "graph")
    p = subprocess.run([os.path.join(BIN, "buildG"), "-se", fa, "-f", str(tmp_path / "graph" / "x"), "-p", str(cfg), "-t", "1", "--no-text"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, DISCO_PAR_SIMPLE="1"))
    assert p.returncode == 0, p.stdout[-2000:]
    lines = sorted(open(tmp_path / "assembly" / "x_0_ParSimpleEdges.txt").read().splitlines())
    assert len(lines) == c["par_simple_lines"]
    assert hashlib.sha256(("\n".join(lines) + "\n").encode()).hexdigest() == c["par_simple_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("gpus,shape", [(1, "uniform"), (1, "metagenome"), (2, "metagenome"), (1, "errors")])
def test_chains_contracted_on_the_gpu_equal_the_host_walk(tmp_path, gpus, shape):
    """--par-simple starts from the chains the GPU has contracted by RANKING them (disco_contract_chains, pointer jumping; with
    several ranks disco_contract_chains_of on the gathered edges); DISCO_PAR_SIMPLE_HOST=1 walks them on the host as the
    reference does. Same files, line for line — on one long chain per contig, on the ragged graph of a metagenome and on reads
    with errors (tips, bubbles, dead ends: several rounds)"""
    from disco_amd import readgen

    build.build_host()
    if shape == "uniform":
        spec = readgen.GenSpec.coverage(seed=15, n_reads=400_000, read_len=150, cov=30.0, n_contigs=4)
    else:
        spec = readgen.GenSpec.coverage(seed=16, n_reads=400_000, read_len=100, cov=25.0, n_contigs=40, len_max=250, skew=1 if shape == "metagenome" else 0)
    codes, off = readgen.generate_codes(spec)
    if shape == "errors":
        codes = readgen.substitute(codes, off, 3, 2000)
    fa = str(tmp_path / "r.fasta")
    readgen.write_fasta(fa, readgen.codes_to_reads(codes, off))
    cfg = tmp_path / "disco.cfg"
    cfg.write_text("MinOverlap4BuildGraph = 40\nMinOverlap4SimplifyGraph = 45\n")  # a filter that bites: some edges are not loaded
    out = {}
    for how in ("gpu", "host"):
        prefix = str(tmp_path / f"g_{how}")
        cmd = [os.path.join(BIN, "buildG"), "-se", fa, "-f", prefix, "-p", str(cfg), "-t", "4", "--par-simple", str(tmp_path / f"s_{how}"), "--no-text"]
        if gpus > 1:
            cmd += ["--gpus", str(gpus), "--same-device"]
        env = dict(os.environ, DISCO_VERBOSE="1")
        if how == "host":
            env["DISCO_PAR_SIMPLE_HOST"] = "1"
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
        assert p.returncode == 0, p.stdout
        assert ("contract chains on the GPU" in p.stdout) == (how == "gpu")
        out[how] = [sorted(open(str(tmp_path / f"s_{how}_{t}_ParSimpleEdges.txt")).read().splitlines()) for t in range(4)]
        stats = [l for l in p.stdout.splitlines() if l.startswith("Partial simplification")][0]
        out[how + "_stats"] = stats.split("files")[0].replace(" 1 rounds", " R rounds").replace(" 2 rounds", " R rounds")
    assert out["gpu"] == out["host"]
    assert sum(len(x) for x in out["gpu"]) > 0
    # the same edges in, out and absorbed (the round count differs by the round the GPU did)
    import re
    assert re.sub(r"\d+ rounds", "R rounds", out["gpu_stats"]) == re.sub(r"\d+ rounds", "R rounds", out["host_stats"])


@pytest.mark.gpu
@pytest.mark.parametrize("case,threads", [("multifile", 1), ("multifile", 4), ("generated", 16), ("generated", 3)])
def test_edge_lines_formatted_on_the_gpu_are_the_host_writers_bytes(tmp_path, case, threads):
    """the edge files are formatted where the edges are (disco_format_edges: one thread per line, a scan per file) — line for line, byte for byte,
    what the host writer produces (DISCO_HOST_TEXT=1), with filtered records (file index != read id + 1) and without"""
    from disco_amd import readgen

    build.build_host()
    if case == "multifile":
        c = gu.CASES["multifile"]
        inputs = ["-pe", ",".join(os.path.join(gu.GOLD, f) for f in c["pe"]), "-se", ",".join(os.path.join(gu.GOLD, f) for f in c["se"])]
        mo = c["min_overlap"]
    else:
        fa = str(tmp_path / "r.fasta")
        readgen.write_fasta(fa, readgen.generate_reads(readgen.GenSpec.coverage(seed=31, n_reads=200_000, read_len=100, cov=30.0, n_contigs=7, len_max=260)))
        inputs, mo = ["-se", fa], 40
    cfg = tmp_path / "disco.cfg"
    cfg.write_text(f"MinOverlap4BuildGraph = {mo}\n")
    files = {}
    for how in ("gpu", "host"):
        prefix = str(tmp_path / how)
        env = dict(os.environ, DISCO_VERBOSE="1")
        if how == "host":
            env["DISCO_HOST_TEXT"] = "1"
        p = subprocess.run([os.path.join(BIN, "buildG")] + inputs + ["-f", prefix, "-p", str(cfg), "-t", str(threads)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, env=env)
        assert p.returncode == 0, p.stdout
        assert ("format edge lines on the GPU" in p.stdout) == (how == "gpu")
        # (the order of the edges inside a file is the order of the emission, which differs from run to run: lines compared sorted)
        files[how] = [sorted(open(f"{prefix}_{t}_parGraph.txt", "rb").read().split(b"\n")) for t in range(threads)] + \
                     [open(f"{prefix}_{t}_startRead.txt", "rb").read() for t in range(threads)]
    assert files["gpu"] == files["host"]
    assert sum(len(x) for x in files["gpu"][:threads]) > 200  # lines


REF_PS_BIN = REF_PS + "_bin"  # the reference parsimplify with the binary loader spliced in (oracle/Makefile: ref_binary_loaders)


@pytest.mark.skipif(not (os.path.exists(REF_PS_INITLEN) and os.path.exists(REF_PS_BIN)),
                    reason="prebuilt reference parsimplify with / without the binary loader (make -C oracle ref_parsimplify ref_binary_loaders) not present")
@pytest.mark.parametrize("name", ["u150_5k", "mixed_4k", "k30_6k", "multifile"])
def test_reference_loader_patch_reads_the_binary_edge_file(tmp_path, name):
    """SURVEY.md 8 f-3, the consumer's half: the REAL parsimplify with oracle/patches spliced into its loader, given an EMPTY
    <prefix>_<t>_parGraph.txt next to <prefix>_edges.bin (the state buildG --no-text leaves), writes the lines it writes for the text
    file — two files, flags 0 / 1 on the edges that cross them, so that the file filter and the marks of the binary path are exercised."""
    from disco_amd import edgefile

    lines = [l.rstrip("\n") for l in open(os.path.join(gu.GOLD, name + ".edges.txt")) if l.strip()]
    ids = sorted({int(x) for l in lines for x in l.split("\t")[:2]})
    cut = ids[len(ids) // 2]
    text = str(tmp_path / "t")
    files = [open(f"{text}_{t}_parGraph.txt", "w") for t in range(2)]
    for l in lines:
        a, b = (int(x) for x in l.split("\t")[:2])
        lo, hi = a < cut, b < cut
        if lo or hi:
            files[0].write(l + (",2\n" if lo and hi else (",0\n" if lo else ",1\n")))
        if not lo or not hi:
            files[1].write(l + (",2\n" if not lo and not hi else (",0\n" if not lo else ",1\n")))
    for f in files:
        f.close()
    for t in range(2):
        open(f"{text}_{t}_containedReads.txt", "w").close()
    binp = str(tmp_path / "b")
    ne, _ = edgefile.from_text(text, 2, out_prefix=binp, empty_text=True)
    assert ne >= len(lines) and os.path.getsize(f"{binp}_0_parGraph.txt") == 0
    for t in range(2):
        want, got = str(tmp_path / f"want_{t}.txt"), str(tmp_path / f"got_{t}.txt")
        _run_ps(REF_PS_INITLEN, f"{text}_{t}_parGraph.txt", want, 30, 2)
        p = subprocess.run([REF_PS_BIN, f"{binp}_{t}_parGraph.txt", got, "30", "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert p.returncode == 0 and "edges loaded to memory from" in p.stdout, p.stdout[-2000:]
        a, b = _par_lines(want), _par_lines(got)
        assert len(a) > 0 and a == b, (t, len(a), len(b))
        # the patched binary on the TEXT file takes the reference's own path
        again = str(tmp_path / f"again_{t}.txt")
        _run_ps(REF_PS_BIN, f"{text}_{t}_parGraph.txt", again, 30, 2)
        assert _par_lines(again) == a


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(REF_PS_INITLEN) and os.path.exists(REF_PS_BIN)), reason="prebuilt reference parsimplify binaries not present")
def test_buildg_no_text_feeds_the_patched_reference_loader(tmp_path):
    """buildG --no-text (binary side output only, text files left empty) -> the REAL parsimplify with the loader patch, file by file:
    the lines it writes for the text files of an ordinary buildG run on the same reads"""
    from disco_amd import readgen

    build.build_host()
    spec = readgen.GenSpec.coverage(seed=9, n_reads=40000, read_len=110, cov=25.0, n_contigs=5, len_max=200)
    fa = str(tmp_path / "r.fasta")
    readgen.write_fasta(fa, readgen.generate_reads(spec))
    cfg = tmp_path / "disco.cfg"
    cfg.write_text("MinOverlap4BuildGraph = 40\n")
    out = {}
    for how, extra in (("text", []), ("bin", ["--no-text"])):
        prefix = str(tmp_path / how)
        p = subprocess.run([os.path.join(BIN, "buildG"), "-se", fa, "-f", prefix, "-p", str(cfg), "-t", "3"] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert p.returncode == 0, p.stdout
        out[how] = []
        for t in range(3):
            res = str(tmp_path / f"{how}_{t}.out")
            _run_ps(REF_PS_INITLEN if how == "text" else REF_PS_BIN, f"{prefix}_{t}_parGraph.txt", res, 40, 2)
            out[how].append(_par_lines(res))
    assert os.path.getsize(str(tmp_path / "bin_0_parGraph.txt")) == 0 and os.path.getsize(str(tmp_path / "bin_edges.bin")) > 1000
    assert out["text"] == out["bin"] and sum(len(x) for x in out["text"]) > 0
