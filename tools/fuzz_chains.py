"""differential fuzzing of the chain contraction: buildG --par-simple with the chains ranked on the GPU (disco_contract_chains /
disco_contract_chains_of) against the same command with DISCO_PAR_SIMPLE_HOST=1 (chains walked on the host, the form that is pinned
against the real parsimplify in tests/test_host.py) — random read sets incl. circular genomes (rings of absorbable nodes), repeats,
sequencing errors, several ranks.   python tools/fuzz_chains.py [ITERATIONS=40] [SEED=1]"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np
from disco_amd import build, readgen

build.build_host()
BIN = os.path.join("disco_amd", "bin", "buildG")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
comp = str.maketrans("ACGT", "TGCA")
fails = 0
t0 = time.time()
for it in range(iters):
    n = int(rng.integers(300, 30000))
    lmin = int(rng.choice([60, 100, 150, 250]))
    lmax = lmin if rng.random() < 0.5 else lmin + int(rng.integers(1, lmin))
    cov = float(rng.choice([6, 12, 25, 40]))
    circular = rng.random() < 0.4
    err = float(rng.choice([0, 0, 0.001, 0.004]))
    gpus = int(rng.choice([1, 1, 2, 3]))
    t = int(rng.choice([1, 2, 5]))
    mo_s = int(rng.choice([0, 40, 50, 70]))
    label = f"it{it} n={n} len={lmin}-{lmax} cov={cov} circular={circular} err={err} gpus={gpus} t={t} minOvlSimplify={mo_s}"
    try:
        reads = []
        nc = int(rng.integers(1, 8))
        glen = max(int(n * (lmin + lmax) / 2 / cov / nc), lmax + 50)
        genomes = ["".join(rng.choice(list("ACGT"), glen)) for _ in range(nc)]
        if rng.random() < 0.3:  # a repeat shared by the genomes: branch nodes
            rep = "".join(rng.choice(list("ACGT"), int(rng.integers(60, 300))))
            genomes = [g[:len(g) // 2] + rep + g[len(g) // 2:] for g in genomes]
        for _ in range(n):
            g = genomes[int(rng.integers(0, nc))]
            L = int(rng.integers(lmin, lmax + 1))
            if circular:
                p = int(rng.integers(0, len(g)))
                s = (g + g)[p:p + L] if L < len(g) else g[:L]
            else:
                p = int(rng.integers(0, len(g) - L + 1))
                s = g[p:p + L]
            if err:
                b = np.frombuffer(s.encode(), dtype=np.uint8).copy()
                hit = rng.random(len(b)) < err
                b[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(hit.sum()))]
                s = b.tobytes().decode()
            reads.append(s.translate(comp)[::-1] if rng.random() < 0.5 else s)
        with tempfile.TemporaryDirectory() as d:
            fa = os.path.join(d, "r.fasta")
            readgen.write_fasta(fa, reads)
            cfg = os.path.join(d, "disco.cfg")
            open(cfg, "w").write(f"MinOverlap4BuildGraph = 40\nMinOverlap4SimplifyGraph = {mo_s}\n")
            out = {}
            for how in ("gpu", "host"):
                cmd = [BIN, "-se", fa, "-f", os.path.join(d, "g_" + how), "-p", cfg, "-t", str(t), "--par-simple", os.path.join(d, "s_" + how), "--no-text"]
                if gpus > 1:
                    cmd += ["--gpus", str(gpus), "--same-device"]
                env = dict(os.environ)
                if how == "host":
                    env["DISCO_PAR_SIMPLE_HOST"] = "1"
                p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
                if p.returncode != 0:
                    raise RuntimeError(how + ": " + p.stdout[-250:])
                out[how] = [sorted(open(os.path.join(d, f"s_{how}_{i}_ParSimpleEdges.txt")).read().splitlines()) for i in range(t)]
            if out["gpu"] != out["host"]:
                a = set(l for f in out["gpu"] for l in f)
                b = set(l for f in out["host"] for l in f)
                raise AssertionError(f"files differ: {len(a - b)} lines only with the GPU chains, {len(b - a)} only with the host walk; e.g. "
                                     + " | ".join(x[:150] for x in sorted(a - b)[:2]) + " <> " + " | ".join(x[:150] for x in sorted(b - a)[:2]))
        print("ok  ", label, "lines", sum(len(x) for x in out["gpu"]), flush=True)
    except Exception as e:
        fails += 1
        print("FAIL", label, repr(e)[:1200], flush=True)
print(f"{iters - fails}/{iters} ok in {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
