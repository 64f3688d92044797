"""differential fuzzing of the input stage (disco_amd/bin/fastx_dump = the parser / filter / id assignment of the drop-in buildG)
against the oracle's restatement of Dataset::readDataset + testRead, on randomly malformed FASTA / FASTQ text.
   python tools/fuzz_fastx.py [ITERATIONS=200] [SEED=1] [gpu]      (CPU only; with "gpu": the input stage on the GPU, disco_ingest_fasta, on
   the same files as well — whatever it ACCEPTS must come out exactly as the oracle's parser has it; declining is always allowed)"""
import gzip, os, subprocess, sys, tempfile
sys.path.insert(0, '.')
import numpy as np
from oracle import pyoracle
from disco_amd import build

build.build_host()
BIN = os.path.join("disco_amd", "bin", "fastx_dump")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
GPU = len(sys.argv) > 3 and sys.argv[3] == "gpu"
gpu_accepted = 0


def ingest(mo, paths):
    """(reads, file indices, total records) through disco_ingest_fasta, or None when it declines"""
    from disco_amd import buildgraph

    with buildgraph.BuildGraph(min_overlap=mo) as g:
        res = g.ingest_fasta(paths, threads=4)
        if res is None:
            return None
        info, _ = res
        ln, fi = g.ingest_fetch()
        packed, lens = g.download_reads()
    reads = []
    for row, L in zip(packed, lens):
        reads.append("".join("ACGT"[(int(row[t >> 5]) >> (62 - 2 * (t & 31))) & 3] for t in range(int(L))))
    return reads, [int(x) for x in fi], int(info["total_records"])


def dump(mo, paths):
    p = subprocess.run([BIN, str(mo), "-se", ",".join(paths)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0:
        return None
    out = p.stdout.strip().split("\n")
    tail = out[-1].split()
    reads = [l.split("\t") for l in out[:-1] if l]
    return [r[1] for r in reads], [int(r[0]) for r in reads], int(tail[1])


def rand_seq(L):
    kind = rng.integers(0, 10)
    s = "".join(rng.choice(list("ACGT"), L))
    if kind == 0:
        s = s.lower()
    elif kind == 1 and L > 3:
        i = int(rng.integers(0, L))
        s = s[:i] + rng.choice(list("NnRY-*")) + s[i + 1:]
    elif kind == 2:
        s = (rng.choice(["AC", "AAT", "GGGGCC", "A", "TAA"]) * L)[:L]
    return s


def fasta(nrec):
    eol = "\r\n" if rng.random() < 0.15 else "\n"
    t = ""
    for i in range(nrec):
        L = int(rng.integers(0, 260))
        s = rand_seq(L)
        hdr = ">r%d" % i + (" x>y" if rng.random() < 0.05 else "")
        w = int(rng.choice([0, 0, 60, 70, 7, -1]))
        if w < 0 and s:  # lines of random widths (round 4: the device stage walks irregularly wrapped records)
            cuts = sorted(set(int(x) for x in rng.integers(1, max(L, 2), size=int(rng.integers(1, 6)))))
            body = eol.join(s[a:b] for a, b in zip([0] + cuts, cuts + [L]))
        else:
            body = s if not w or not s else eol.join(s[j:j + w] for j in range(0, L, w))
        if rng.random() < 0.03 and L > 10:
            body = body[:5] + ">" + body[5:]
        t += hdr + eol + body + eol
        if rng.random() < 0.05:
            t += eol
    if rng.random() < 0.3:
        t = t.rstrip("\r\n")
    if rng.random() < 0.05:
        t += ">"
    return t


def fastq(nrec):
    t = ""
    for i in range(nrec):
        L = int(rng.integers(0, 260))
        s = rand_seq(L)
        q = "".join(rng.choice(list("IIII@>+#!5"), L))
        t += "@q%d\n%s\n+%s\n%s\n" % (i, s, "" if rng.random() < 0.7 else "q%d" % i, q)
    if rng.random() < 0.3:
        t = t.rstrip("\n")
    return t


fails = 0
with tempfile.TemporaryDirectory() as d:
    for it in range(iters):
        mo = int(rng.choice([31, 40, 65]))
        paths, plain = [], []
        for f in range(int(rng.integers(1, 4))):
            text = fasta(int(rng.integers(0, 60))) if rng.random() < 0.6 else fastq(int(rng.integers(0, 60)))
            p = os.path.join(d, "f%d_%d.%s" % (it, f, "fa"))
            open(p, "w", newline="").write(text)
            plain.append(p)
            if rng.random() < 0.2:  # the drop-in reads the gzip, the oracle the same bytes uncompressed
                with gzip.open(p + ".gz", "wb") as fh:
                    fh.write(text.encode())
                p += ".gz"
            paths.append(p)
        # the reference exits when a file contributes no record (BG/Dataset.cpp:113-114,124-125)
        if any(len(pyoracle.parse_records(open(p, "rb").read())) == 0 for p in plain):
            oreads = None
        else:
            want = pyoracle.load_good_reads(plain, mo)
            oreads, ofidx, ototal = want[0], [int(x) for x in want[1]], want[2]
        got = dump(mo, paths)
        ok = (got is None and oreads is None) or (got is not None and oreads is not None and got[0] == oreads and got[1] == ofidx and got[2] == ototal)
        if GPU:
            gi = ingest(mo, paths)
            if gi is not None:
                gpu_accepted += 1
                keep_idx = None if oreads is None else [i for i, r in enumerate(oreads) if len(r) <= 32767]
                if oreads is None or gi[0] != [oreads[i] for i in keep_idx] or gi[1] != [ofidx[i] for i in keep_idx] or gi[2] != ototal:
                    ok = False
                    print("     (device input stage differs: %s)" % (None if gi is None else (len(gi[0]), gi[2]),), flush=True)
        if not ok:
            fails += 1
            keep = os.path.join(tempfile.gettempdir(), "fuzz_fastx_fail_%d" % it)
            os.makedirs(keep, exist_ok=True)
            for p in paths:
                subprocess.run(["cp", p, keep])
            print("FAIL it%d mo=%d files=%s kept in %s : got %s want %s" % (it, mo, [os.path.basename(p) for p in paths], keep,
                  None if got is None else (len(got[0]), got[2]), None if oreads is None else (len(oreads), ototal)), flush=True)
        for p in set(paths + plain):
            os.remove(p)
print("%d/%d ok" % (iters - fails, iters) + (" (device input stage accepted %d of the jobs)" % gpu_accepted if GPU else ""))
sys.exit(1 if fails else 0)
