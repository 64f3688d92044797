"""-m gpu : the input stage on the GPU (disco_ingest_fasta: records, Dataset::testRead, ids and 2-bit rows as kernels) against the
CPU restatement of the reference's parser and filter (oracle/pyoracle.load_good_reads, pinned in tests/test_oracle_golden.py /
tests/test_host.py) and against the host stage of the drop-in: the same reads kept, the same file indices, the same packed rows —
and the files it must DECLINE (so that the host stage, which follows the reference's getline calls literally, takes them)."""
import glob
import gzip
import os
import subprocess

import numpy as np
import pytest

from disco_amd import build, buildgraph
from oracle import pyoracle
from tests import golden_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "disco_amd", "bin")
MOTIFS = ["AC", "AG", "AT", "CG", "CT", "GT", "AAT", "ATA", "TAA", "AAC", "ACA", "CAA", "AAG", "AGA", "GAA", "GGGGCC"]


def _decode(packed, lens):
    out = []
    for row, L in zip(packed, lens):
        s = []
        for t in range(int(L)):
            s.append("ACGT"[(int(row[t >> 5]) >> (62 - 2 * (t & 31))) & 3])
        out.append("".join(s))
    return out


def _adversarial(rng, n):
    """reads around every threshold of Dataset::testRead (BG/Dataset.cpp:403-452), lower case, N, CR, short and very long ones"""
    reads = []
    for i in range(n):
        L = int(rng.integers(25, 260))
        kind = i % 8
        if kind == 0:
            s = "".join(rng.choice(list("ACGT"), L))
        elif kind == 1:  # one base near the 70 % threshold
            b = "ACGT"[i % 4]
            frac = rng.uniform(0.62, 0.78)
            s = "".join(b if rng.random() < frac else rng.choice(list("ACGT")) for _ in range(L))
        elif kind == 2:  # a motif covering about half of the read, scattered
            m = MOTIFS[int(rng.integers(0, len(MOTIFS)))]
            reps = int(L * rng.uniform(0.40, 0.60) / len(m))
            parts = [m] * reps + list(rng.choice(list("ACGT"), max(L - reps * len(m), 0)))
            rng.shuffle(parts)
            s = "".join(parts)[:L]
        elif kind == 3:  # motif run + random tail
            m = MOTIFS[int(rng.integers(0, len(MOTIFS)))]
            run = int(L * rng.uniform(0.45, 0.55))
            s = (m * (run // len(m) + 1))[:run] + "".join(rng.choice(list("ACGT"), L - run))
        elif kind == 4:  # micro-repeat prefix / suffix
            unit = ["AC", "AAG", "AAAT", "AATT", "TACA", "GTTT", "AGGG"][i % 7]
            rep = (unit * 10)[:29]
            body = "".join(rng.choice(list("ACGT"), L))
            s = rep + body if i % 2 else body + rep
        elif kind == 5:  # lower case, N, mixed case
            s = "".join(rng.choice(list("ACGT"), L))
            s = s.lower() if i % 3 == 0 else (s[:10] + "N" + s[11:] if i % 3 == 1 else s[:L // 2].lower() + s[L // 2:])
        elif kind == 6:  # lengths around the minimum overlap and MIN_READ_SIZE
            s = "".join(rng.choice(list("ACGT"), int(rng.integers(20, 45))))
        else:            # a few very long reads (beyond the 15-bit length field: dropped and counted)
            s = "".join(rng.choice(list("ACGT"), 33000 if i % 64 == 7 else L))
        reads.append(s)
    return reads


def _ingest(paths, min_overlap):
    with buildgraph.BuildGraph(min_overlap=min_overlap) as g:
        res = g.ingest_fasta(paths, threads=8)
        if res is None:
            return None
        info, files = res
        ln, fi = g.ingest_fetch()
        packed, lens = g.download_reads()
        assert np.array_equal(ln, lens) and info["n_reads"] == len(ln) == g.num_reads
        return info, files, _decode(packed, lens), fi


@pytest.mark.parametrize("min_overlap,eol,final_newline", [(30, "\n", True), (40, "\n", False), (31, "\r\n", True)])
def test_ingest_keeps_what_the_reference_parser_keeps(tmp_path, min_overlap, eol, final_newline):
    rng = np.random.default_rng(7 + min_overlap)
    reads = _adversarial(rng, 5000)
    text = "".join(f">r{i} some description{eol}{s}{eol}" for i, s in enumerate(reads))
    if not final_newline:
        text = text[:-len(eol)]
    fa = tmp_path / "adv.fasta"
    fa.write_bytes(text.encode())
    want, wfidx, wtotal = pyoracle.load_good_reads([str(fa)], min_overlap)
    # reads beyond the 15-bit length field of the index records (BG/HashTable.cpp:531) are dropped and counted, by the host stage too
    keep = [i for i, s in enumerate(want) if len(s) <= 32767]
    n_long = len(want) - len(keep)
    want, wfidx = [want[i] for i in keep], np.asarray(wfidx)[keep]
    if eol == "\r\n":  # a CR is a character that is not ACGT (the reference strips the newline only): every read is rejected, and a job
        assert want == [] and _ingest([str(fa)], min_overlap) is None  # without a good read goes to the host stage for its error message
        return
    info, files, got, fidx = _ingest([str(fa)], min_overlap)
    assert info["total_records"] == wtotal == len(reads) and info["too_long"] == n_long
    assert got == want and np.array_equal(fidx.astype(np.int64), np.asarray(wfidx, dtype=np.int64))
    assert files[0]["good"] == len(want) and files[0]["good"] + files[0]["bad"] == len(reads)
    assert 800 < len(got) < 4500 and n_long > 0  # both outcomes well represented; the long reads were counted


def test_ingest_of_several_files_numbers_the_records_through(tmp_path):
    rng = np.random.default_rng(3)
    paths, all_reads = [], []
    for f in range(3):
        reads = _adversarial(rng, 700 + 100 * f)
        p = tmp_path / f"f{f}.fa"
        p.write_text("".join(f">x{i}\n{s}\n" for i, s in enumerate(reads)))
        paths.append(str(p))
        all_reads += reads
    want, wfidx, wtotal = pyoracle.load_good_reads(paths, 35)
    keep = [i for i, s in enumerate(want) if len(s) <= 32767]
    want, wfidx = [want[i] for i in keep], np.asarray(wfidx)[keep]
    info, files, got, fidx = _ingest(paths, 35)
    assert info["total_records"] == wtotal == len(all_reads) and got == want
    assert np.array_equal(fidx.astype(np.int64), np.asarray(wfidx, dtype=np.int64))
    assert [f["first_index"] for f in files] == [1, 701, 1501] and files[2]["last_index"] == len(all_reads)


@pytest.mark.parametrize("tail", ["full", "no_final_newline", "header_only", "three_lines"])
def test_ingest_reads_fastq_by_counting_lines(tmp_path, tail):
    """FASTQ: records of four lines, found by the line count (the reference's four getline calls, BG/Dataset.cpp:255-293) — quality lines
    that begin with '@' or '>' must not start a record; a truncated last record still counts as a record"""
    rng = np.random.default_rng(17)
    reads = _adversarial(rng, 3000)
    reads = [r for r in reads if len(r) <= 32767]
    recs = []
    for i, s in enumerate(reads):
        q = "".join(rng.choice(list("@>IF#+"), len(s)))
        if i % 5 == 0:
            q = "@" + q[1:]
        if i % 7 == 0:
            q = ">" + q[1:]
        recs.append(f"@r{i} x\n{s}\n+\n{q}\n")
    text = "".join(recs)
    if tail == "no_final_newline":
        text = text[:-1]
    elif tail == "header_only":
        text += "@last"
    elif tail == "three_lines":
        text += "@last\n" + "ACGT" * 20 + "\n+"
    fq = tmp_path / "r.fastq"
    fq.write_text(text)
    want, wfidx, wtotal = pyoracle.load_good_reads([str(fq)], 33)
    info, files, got, fidx = _ingest([str(fq)], 33)
    assert info["total_records"] == wtotal == len(reads) + (tail in ("header_only", "three_lines"))
    assert got == want and np.array_equal(fidx.astype(np.int64), np.asarray(wfidx, dtype=np.int64)) and len(got) > 500
    # FASTA and FASTQ files in one job
    fa = tmp_path / "r.fasta"
    fa.write_text("".join(f">x{i}\n{s}\n" for i, s in enumerate(reads[:500])))
    want2, wfidx2, wtotal2 = pyoracle.load_good_reads([str(fa), str(fq)], 33)
    info2, files2, got2, fidx2 = _ingest([str(fa), str(fq)], 33)
    assert got2 == want2 and np.array_equal(fidx2.astype(np.int64), np.asarray(wfidx2, dtype=np.int64)) and info2["total_records"] == wtotal2


def _wrapped(rng, s, how):
    """the sequence s over several lines: 'w60' = wrapped at 60 (the last line shorter or full), 'irregular' = lines of random widths,
    'blank' = a blank line inside, 'trail' = blank lines behind the last one"""
    if how.startswith("w"):
        w = int(how[1:])
        return "\n".join(s[i:i + w] for i in range(0, len(s), w)) if s else ""
    if how == "irregular":
        out, i = [], 0
        while i < len(s):
            w = int(rng.integers(1, 90))
            out.append(s[i:i + w])
            i += w
        return "\n".join(out)
    if how == "blank":
        h = len(s) // 2
        return s[:h] + "\n\n" + s[h:]
    if how == "trail":
        return s + "\n\n"
    return s


@pytest.mark.parametrize("final_newline", [True, False])
def test_ingest_reads_wrapped_fasta(tmp_path, final_newline):
    """round 4: FASTA records over several lines — a record is its header line and everything up to the next '>' with the newlines
    taken out (BG/Dataset.cpp:270-281): fixed widths (arithmetic addressing), irregular widths, blank lines inside and behind (walking),
    next to one-line records, against the CPU restatement of the reference's parser"""
    rng = np.random.default_rng(23)
    reads = [r for r in _adversarial(rng, 4000) if len(r) <= 32767]
    hows = ["w60", "w70", "w7", "w1", "irregular", "blank", "trail", "one", "w150", "w29"]
    text = "".join(f">r{i} d\n{_wrapped(rng, s, hows[i % len(hows)])}\n" for i, s in enumerate(reads))
    if not final_newline:
        text = text.rstrip("\n")
    fa = tmp_path / "wrapped.fasta"
    fa.write_text(text)
    want, wfidx, wtotal = pyoracle.load_good_reads([str(fa)], 33)
    info, files, got, fidx = _ingest([str(fa)], 33)
    assert info["total_records"] == wtotal == len(reads)
    assert got == want and np.array_equal(fidx.astype(np.int64), np.asarray(wfidx, dtype=np.int64)) and len(got) > 600
    # a long record wrapped at one width is addressed by arithmetic (kept, or dropped for its length like a one-line one); an irregular
    # one beyond the walking limit sends the file to the host stage
    long_seq = "".join(rng.choice(list("ACGT"), 20000))
    ok = tmp_path / "long_regular.fasta"
    ok.write_text(text + "\n>long\n" + _wrapped(rng, long_seq, "w80") + "\n")
    want2, wfidx2, wtotal2 = pyoracle.load_good_reads([str(ok)], 33)
    info2, files2, got2, fidx2 = _ingest([str(ok)], 33)
    assert got2 == want2 and got2[-1] == long_seq and info2["total_records"] == wtotal2
    bad = tmp_path / "long_irregular.fasta"
    bad.write_text(text + "\n>long\n" + _wrapped(rng, long_seq, "irregular") + "\n")
    assert _ingest([str(bad)], 33) is None


def test_ingest_declines_what_only_the_literal_parser_handles(tmp_path):
    good = ">a\nACGTTGCAAGCTAGCTAGGATCGATCGTAGCTAGCTAGCATCGATGCTAGCTAGTCGATCGAT\n"
    # (round 4: sequences over several lines and blank lines between records are read on the device: test_ingest_reads_wrapped_fasta)
    for name, text in {"multiline.fa": ">a\nACGTTGCAAGCTAGCTAGGATCGATCG\nTAGCTAGCTAGCATCGATGCTAGCTAGTCGATCGAT\n", "blank_line.fa": good + "\n" + good}.items():
        p = tmp_path / name
        p.write_text(text)
        want, wfidx, wtotal = pyoracle.load_good_reads([str(p)], 30)
        info, files, got, fidx = _ingest([str(p)], 30)
        assert got == want and len(got) == wtotal and list(fidx) == list(wfidx), name
    cases = {
        "gt_inside.fa": ">a>b\nACGT\n" + good.replace(">a", ">c d>e"),
        "empty.fa": "",
        "no_header.fa": "ACGT\n",
    }
    for name, text in cases.items():
        p = tmp_path / name
        p.write_text(text)
        assert _ingest([str(p)], 30) is None, name
    gz = tmp_path / "r.fa.gz"
    with gzip.open(gz, "wt") as f:
        f.write(good)
    assert _ingest([str(gz)], 30) is None
    assert _ingest([str(tmp_path / "does_not_exist.fa")], 30) is None
    ok = tmp_path / "ok.fa"
    ok.write_text(good * 3 + ">")  # a '>' that is the very last byte starts no record (disco_amd/host/fastx.cpp)
    info, files, got, fidx = _ingest([str(ok)], 30)
    assert info["total_records"] == 3 and len(got) == 3 and list(fidx) == [1, 2, 3]


@pytest.mark.parametrize("threads", [1, 5])
def test_buildg_with_the_device_input_stage_writes_the_host_stages_files(tmp_path, threads):
    """the whole drop-in on a FASTA the device stage accepts, against DISCO_HOST_INPUT=1: every output file line for line; and on the
    `multifile` fixture (FASTQ, multi-line FASTA, a .gz among the files: declined) the run still equals the reference's files"""
    from disco_amd import readgen

    build.build_host()
    rng = np.random.default_rng(11)
    spec = readgen.GenSpec.coverage(seed=21, n_reads=20000, read_len=100, cov=20.0, n_contigs=3, len_max=180)
    reads = list(readgen.generate_reads(spec)) + _adversarial(rng, 2000)
    order = rng.permutation(len(reads))
    fa = tmp_path / "r.fasta"
    fa.write_text("".join(f">q{i}\n{reads[j]}\n" for i, j in enumerate(order)))
    cfg = tmp_path / "disco.cfg"
    cfg.write_text("MinOverlap4BuildGraph = 40\n")
    out = {}
    for how in ("device", "host"):
        prefix = str(tmp_path / how)
        env = dict(os.environ, DISCO_VERBOSE="1", **({"DISCO_HOST_INPUT": "1"} if how == "host" else {}))
        p = subprocess.run([os.path.join(BIN, "buildG"), "-se", str(fa), "-f", prefix, "-p", str(cfg), "-t", str(threads)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
        assert p.returncode == 0, p.stdout
        assert ("input stage on the GPU" in p.stdout) == (how == "device"), p.stdout[-1500:]
        out[how] = {os.path.basename(f)[len(how):]: sorted(open(f, "rb").read().splitlines()) for f in sorted(glob.glob(prefix + "_*"))}
        out[how + "_log"] = [l for l in p.stdout.splitlines() if "reads in current dataset" in l or "read length in all datasets" in l]
    assert out["device"].keys() == out["host"].keys() and len(out["device"]) >= 3 * threads + 2
    for k in out["device"]:  # (the order of the edge lines inside a file follows the emission's atomics: compared as sets of lines)
        assert out["device"][k] == out["host"][k], k
    assert out["device_log"] == out["host_log"] and len(out["device"]["_ReadIDMap.txt"]) > 0
    # the `multifile` fixture — wrapped FASTA with filter cases, FASTQ, plain FASTA — through the device stage (round 4), and with one
    # of its files gzipped: declined, the host stage takes over inside the same run; the reference's files either way
    import shutil

    from oracle import refrun

    c = gu.CASES["multifile"]
    cfg.write_text(f"MinOverlap4BuildGraph = {c['min_overlap']}\n")
    gz = tmp_path / "plain.fasta.gz"
    with open(os.path.join(gu.GOLD, c["se"][0]), "rb") as fi, gzip.open(gz, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    for how, se in (("dev", ",".join(os.path.join(gu.GOLD, f) for f in c["se"])), ("declined", str(gz))):
        prefix = str(tmp_path / ("m_" + how))
        cmd = [os.path.join(BIN, "buildG"), "-pe", ",".join(os.path.join(gu.GOLD, f) for f in c["pe"]), "-se", se, "-f", prefix, "-p", str(cfg), "-t", "2"]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, DISCO_VERBOSE="1"))
        assert p.returncode == 0, p.stdout[-1500:]
        assert ("the host input stage takes this job" in p.stdout) == (how == "declined") and ("input stage on the GPU" in p.stdout) == (how == "dev"), p.stdout[-1500:]
        gu.check_against_golden("multifile", refrun.parse_pargraph(sorted(glob.glob(prefix + "_*_parGraph.txt"))), refrun.parse_contained(sorted(glob.glob(prefix + "_*_containedReads.txt"))))


@pytest.mark.parametrize("wrapped", [False, True])
def test_ingest_packs_a_tail_of_long_reads_per_class(tmp_path, wrapped):
    """a few long reads among short ones (up to the format's 32767 bases): the device stage packs 64-byte rows for everybody and
    full / tail rows for the long ones straight from the text (two classes of rows: the one-stride table — n rows as wide as the
    longest read — is never made); what comes back is the table as the parser defines it, and the graph is the oracle's"""
    from tests.test_gpu_two_class import mixed_reads
    from tests.util import canon_hip, run_oracle_reads

    rng = np.random.default_rng(5 + wrapped)
    reads = mixed_reads(31 + wrapped, 6000, 100, 250, 30.0, 0.01, 257, 3000) + _adversarial(rng, 600)
    reads += ["".join(rng.choice(list("ACGT"), 32000))]  # one outlier: one stride would cost 8 KB for each of the 6601 rows
    order = rng.permutation(len(reads))
    paths = []
    for f, part in enumerate((order[:4000], order[4000:])):
        p = tmp_path / f"t{f}.fa"
        with open(p, "w") as fh:
            for i, j in enumerate(part):
                s = reads[j]
                fh.write(f">t{i}\n" + ("\n".join(s[q:q + 70] for q in range(0, len(s), 70)) if wrapped and s else s) + "\n")
        paths.append(str(p))
    want, wfidx, wtotal = pyoracle.load_good_reads(paths, 40)
    keep = [i for i, s in enumerate(want) if len(s) <= 32767]
    want, wfidx = [want[i] for i in keep], np.asarray(wfidx)[keep]
    with buildgraph.BuildGraph(min_overlap=40) as g:
        info, files = g.ingest_fasta(paths, threads=4)
        n_long = sum(len(s) > 256 for s in want)
        assert g.long_rows == n_long > 30 and g.stride_words == 1000
        ln, fi = g.ingest_fetch()
        packed, lens = g.download_reads()
        assert _decode(packed, lens) == want and np.array_equal(fi.astype(np.int64), np.asarray(wfidx, dtype=np.int64))
        g.run_graph()
        he, hr, hc = g.fetch_edges(), g.fetch_contained(), g.counters()
    oe, orows, oc = run_oracle_reads(want, 40)
    a, b = canon_hip(he, hr), canon_hip(oe, orows)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and hc["kmer_hits"] == oc["kmer_hits"]
