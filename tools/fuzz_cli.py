"""end-to-end differential fuzzing of the drop-in executable: disco_amd/bin/buildG vs the REAL reference (oracle/_ref/buildG_ref,
-t 1) on random FASTA / FASTQ inputs given as -pe / -se lists; canonical content of the files they write must be identical
(only where the reference itself is order independent: cases with asymmetric pairs / cap-bound sites are compared on the
contained rows and counted).   python tools/fuzz_cli.py [ITERATIONS=30] [SEED=1]      (needs a GPU and the prebuilt reference)"""
import glob, os, subprocess, sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np
from disco_amd import build, readgen
from oracle import refrun

build.build_host()
MINE = os.path.join("disco_amd", "bin", "buildG")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
assert refrun.available(), "oracle/_ref/buildG_ref missing (make -C oracle ref in the build container)"
fails = skipped = 0
t0 = time.time()
for it in range(iters):
    with tempfile.TemporaryDirectory() as d:
        mo = int(rng.choice([31, 40, 50, 80]))
        lmin = int(rng.choice([70, 100, 150, 250]))
        if mo >= lmin:  # (no read would pass the length filter: both binaries stop with "No reads found", with different exit codes by design)
            mo = 50
        lmax = lmin if rng.random() < 0.5 else lmin + int(rng.integers(1, lmin))
        n = int(rng.integers(200, 3000))
        cov = float(rng.choice([10, 30, 80]))
        tailed = lmax <= 256 and rng.random() < 0.35  # a tail of long reads: the library re-lays the table in two classes of rows
        spec = readgen.GenSpec.coverage(int(rng.integers(1, 1 << 30)), n, lmin, cov, n_contigs=int(rng.integers(1, 4)), len_max=lmax,
                                        long_len=int(rng.choice([300, 600, 1000])) if tailed else 0, long_share=int(rng.choice([300, 2000])) if tailed else 0)
        reads = list(readgen.generate_reads(spec))
        # sprinkle reads the filter must drop (ids still advance) and lower case
        for i in rng.integers(0, n, n // 20):
            reads[i] = rng.choice(["ACGT" * 40, reads[i][:10] + "N" + reads[i][11:], reads[i].lower(), "A" * lmin, reads[i][:25]])
        nfiles = int(rng.integers(1, 4))
        cuts = sorted(rng.integers(1, n, nfiles - 1).tolist()) if nfiles > 1 else []
        parts = [reads[a:b] for a, b in zip([0] + cuts, cuts + [n])]
        files, kinds, gz = [], [], set()
        for fi, part in enumerate(parts):
            if not part:
                part = [reads[0]]
            fq = rng.random() < 0.3
            p = os.path.join(d, "in%d.%s" % (fi, "fastq" if fq else "fasta"))
            with open(p, "w") as f:
                for i, s in enumerate(part):
                    if fq:
                        f.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
                    else:
                        w = int(rng.choice([0, 0, 60]))
                        f.write(">r%d\n%s\n" % (i, s if not w else "\n".join(s[j:j + w] for j in range(0, len(s), w))))
            if rng.random() < 0.25:  # gzipped for the drop-in: its device input stage declines, the host stage (reads back to back) takes the
                import gzip, shutil  # whole job; the reference build here has no zlib (READGZ): it gets the file as it was
                with open(p, "rb") as fi, gzip.open(p + ".gz", "wb") as fo:
                    shutil.copyfileobj(fi, fo)
                gz.add(p)
            files.append(p)
            kinds.append("-pe" if rng.random() < 0.4 else "-se")
        pe = [f for f, k in zip(files, kinds) if k == "-pe"]
        se = [f for f, k in zip(files, kinds) if k == "-se"]
        cfg = os.path.join(d, "disco.cfg")
        open(cfg, "w").write("MinOverlap4BuildGraph = %d\n" % mo)
        threads = int(rng.choice([1, 2, 5]))

        # the drop-in also as N ranks on the one GPU (in-process exchanges), at random; with and without buildG-MPI's file names
        extra = []
        if rng.random() < 0.5:
            extra = ["--gpus", str(int(rng.integers(2, 5))), "--same-device"] + (["--mpi-names"] if rng.random() < 0.3 else [])

        def run(exe, prefix, t):
            nm = (lambda f: f + ".gz" if f in gz else f) if exe == MINE else (lambda f: f)
            cmd = [exe] + (["-pe", ",".join(map(nm, pe))] if pe else []) + (["-se", ",".join(map(nm, se))] if se else []) + ["-f", prefix, "-p", cfg, "-t", str(t), "-m", "8"]
            if exe == MINE:
                cmd += extra
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            return p.returncode, p.stdout

        os.makedirs(os.path.join(d, "ref")); os.makedirs(os.path.join(d, "mine"))
        rc1, log1 = run(refrun.REF_BIN, os.path.join(d, "ref", "g"), 1)
        rc2, log2 = run(MINE, os.path.join(d, "mine", "g"), threads)
        label = "it%d n=%d len=%d-%d%s mo=%d cov=%g files=%s t=%d %s" % (it, n, lmin, lmax, " +tail" if tailed else "", mo, cov, [k + ":" + os.path.basename(f) + (".gz" if f in gz else "") for f, k in zip(files, kinds)],
                                                                          threads, " ".join(extra))
        try:
            assert rc2 == 0, log2[-500:]
            e1 = refrun.parse_pargraph(sorted(glob.glob(os.path.join(d, "ref", "g_*_parGraph.txt"))))
            c1 = refrun.parse_contained(sorted(glob.glob(os.path.join(d, "ref", "g_*_containedReads.txt"))))
            e2 = refrun.parse_pargraph(sorted(glob.glob(os.path.join(d, "mine", "g_*_parGraph.txt"))))
            c2 = refrun.parse_contained(sorted(glob.glob(os.path.join(d, "mine", "g_*_containedReads.txt"))))
            m1 = open(os.path.join(d, "ref", "g_ReadIDMap.txt")).read().replace(d, "")
            m2 = open(os.path.join(d, "mine", "g_ReadIDMap.txt")).read().replace(d, "").replace(".gz", "")
            assert m1 == m2, "ReadIDMap differs"
            assert np.array_equal(c1, c2), "contained rows differ (%d vs %d)" % (len(c1), len(c2))
            flat = log2.replace(" ", "")
            order_dep = ("asymmetric_pairs:0" not in flat) or ("cap_bind_sites:0" not in flat)
            if not np.array_equal(e1, e2):
                if order_dep:
                    skipped += 1
                    print("skip", label, "(order-dependent regime reported by the drop-in)", flush=True)
                    continue
                raise AssertionError("edges differ (%d vs %d)" % (len(e1), len(e2)))
            print("ok  ", label, "edges", len(e2), "contained", len(c2), flush=True)
        except Exception as e:
            fails += 1
            print("FAIL", label, repr(e)[:400], flush=True)
print("%d/%d ok, %d skipped, in %.0f s" % (iters - fails - skipped, iters, skipped, time.time() - t0))
sys.exit(1 if fails else 0)
