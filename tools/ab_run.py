#!/usr/bin/env python3
"""time several builds of libdisco_hip.so against each other in ONE gpurun call (boxes differ by a few per cent):
   python tools/ab_run.py [--reads N] [--rounds R] [--steps K] NAME=path/lib.so ...   ->  one line per build and round: pass and phase times, counters
   (bench.py with DISCO_LIB set, no CPU baseline / stage / host-to-host leg). Builds come from tools/ab_build.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
reads, rounds, steps, extra = 50_000_000, 2, 5, []
while args and args[0].startswith("--"):
    k = args.pop(0)
    if k == "--reads":
        reads = int(args.pop(0))
    elif k == "--rounds":
        rounds = int(args.pop(0))
    elif k == "--steps":
        steps = int(args.pop(0))
    else:
        extra += [k, args.pop(0)]  # handed to bench.py (e.g. --min-overlap 35)
libs = [a.split("=", 1) for a in args]
for rnd in range(rounds):
    for name, path in libs:
        env = dict(os.environ)
        envs = path.split(",")  # lib.so[,VAR=value...]
        if envs[0]:
            env["DISCO_LIB"] = os.path.join(ROOT, envs[0])
        for kv in envs[1:]:
            a, b = kv.split("=", 1)
            env[a] = b
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", str(reads), "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline", "--no-stage",
                            "--no-host-to-host"] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = next((l for l in reversed(p.stdout.splitlines()) if l.startswith("{")), None)
        if not line:
            print(f"{name:14s} FAILED rc={p.returncode} {p.stderr[-400:]}", flush=True)
            continue
        d = json.loads(line)
        ph = d["config"]["phase_ms_rank0"]
        c = d["config"]
        print(f"{name:14s} pass {d['ms_per_step']:7.2f}  " + " ".join(f"{k} {v:6.2f}" for k, v in ph.items() if v > 0.05) +
              f"  | e_pre {c['e_pre']} e_out {c['e_out']} contained {c['n_contained']} kmer_hits {c['kmer_hits']}", flush=True)
