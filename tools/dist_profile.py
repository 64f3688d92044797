#!/usr/bin/env python3
"""Multi-GPU pass measured on ONE GPU (no 8-GPU node in this pool): G ranks = G host threads over the in-process transport, the
compute segments of the ranks serialised on the device (DISCO_LOOP_SERIALIZE=1: a rank's phase timers then show its own kernels).
Writes what the time model of DESIGN.md section 6 is built from:
  per rank   bytes sent per exchange, operations on the communicator, blocking host waits, device allocations inside the pass,
             the arena and its high water, kernel milliseconds per phase;
  whole job  work inflation = sum over the ranks of the kernel milliseconds / the single-GPU pass on the same reads and GPU.
usage: dist_profile.py [G=8] [N_READS=50000000] [OUT=gpurun_out/r04_dist8.json]   (PARTITIONED_INDEX=1: the index stays partitioned)"""
import json
import os
import sys
import threading
import time

os.environ.setdefault("DISCO_LOOP_SERIALIZE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco_amd import buildgraph, readgen  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
out_path = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/r05_dist8.json"
passes = 3
genome = int(n * 150 / 30.0)
spec = readgen.GenSpec.coverage(42, n, 150, 30.0, n_contigs=max(1, genome // 5_000_000))
part = bool(os.environ.get("PARTITIONED_INDEX"))
MODE = os.environ.get("MODE", "both")  # single | dist | both (under rocprofv3 --kernel-trace --stats: one flow per process)
if MODE == "dist":
    passes = 1

# single GPU, same reads
single_wall, single_ph, single_cnt = 0.0, {}, None
if MODE != "dist":
    with buildgraph.BuildGraph(min_overlap=40, device=0) as g:
        g.generate_reads(spec)
        for _ in range(0 if MODE != "both" else 2):
            g.run_graph()
        g.synchronize()
        t0 = time.perf_counter()
        g.run_graph()
        g.synchronize()
        single_wall = (time.perf_counter() - t0) * 1e3
        single_ph = g.phase_ms()
        single_cnt = g.counters()
single_kernel = sum(single_ph.values()) or 1.0
if MODE == "single":
    print(json.dumps({"single_wall_ms": single_wall, "phase_ms": single_ph}))
    raise SystemExit(0)

gs = [buildgraph.BuildGraph(min_overlap=40, device=0) for _ in range(G)]
buildgraph.BuildGraph.comm_init_local(gs)
res, errors = [None] * G, []


def work(r):
    try:
        g = gs[r]
        g.dist_generate_reads(spec)
        walls = []
        for _ in range(passes):
            t0 = time.perf_counter()
            g.dist_run_graph(True, part)
            walls.append((time.perf_counter() - t0) * 1e3)
        res[r] = (g.dist_info(), g.phase_ms(), walls)
    except Exception as e:  # pragma: no cover
        errors.append((r, repr(e)))


th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
for t in th:
    t.start()
for t in th:
    t.join()
for g in gs:
    g.close()
if errors:
    raise SystemExit(str(errors))
ranks = []
for info, ph, walls in res:
    ranks.append({"rank": info["rank"], "kernel_ms": round(info["kernel_ms"], 3), "phase_ms": {k: round(v, 3) for k, v in ph.items()},
                  "bytes_sent": info["bytes_sent"], "exchange_host_ms": {k: round(v, 3) for k, v in info["ms"].items()},
                  "comm_ops": info["comm_ops"], "host_syncs": info["host_syncs"], "device_allocs_in_pass": info["device_allocs"],
                  "device_frees_in_pass": info["device_frees"], "arena_bytes": info["arena_bytes"], "arena_peak": info["arena_peak"],
                  "hbm_peak": info["hbm_peak"], "pass_wall_ms_serialised": [round(w, 2) for w in walls],
                  "own_reads": info["own_reads"], "placement": "loci" if info["placement"] else "id ranges"})
i0 = res[0][0]
kernel_table_path = os.path.join(os.path.dirname(out_path) or ".", os.path.basename(out_path).replace(".json", "_kernels.json"))
kernel_table = json.load(open(kernel_table_path)) if os.path.exists(kernel_table_path) else {}
assert single_cnt is None or (i0["e_pre"] == single_cnt["e_pre"] and i0["e_out"] == single_cnt["e_out"]), "the multi-rank pass must reproduce the single-GPU counters"
sum_kernel = sum(r["kernel_ms"] for r in ranks)
phases = sorted(single_ph) if single_ph else sorted(ranks[0]["phase_ms"])
per_phase = {k: {"single_ms": round(single_ph.get(k, 0.0), 3), "sum_over_ranks_ms": round(sum(r["phase_ms"].get(k, 0.0) for r in ranks), 3)} for k in phases}
for v in per_phase.values():
    v["inflation"] = round(v["sum_over_ranks_ms"] / v["single_ms"], 3) if v["single_ms"] > 0.05 else None
out = {"what": f"{G} ranks on ONE MI355X over the in-process transport, compute segments serialised (DISCO_LOOP_SERIALIZE=1); {n} x 150 bp, 30x, min-overlap 40 "
               f"(BASELINE config 4's data); last of {passes} passes; index {'kept partitioned' if part else 'replicated after the partitioned build'}",
       "ranks": G, "reads": n, "regime": i0["regime"], "e_pre": i0["e_pre"], "e_out": i0["e_out"], "n_contained": i0["n_contained"],
       "single_gpu": {"pass_wall_ms": round(single_wall, 2), "kernel_ms": round(single_kernel, 3), "phase_ms": {k: round(v, 3) for k, v in single_ph.items()}},
       # a phase timer brackets the phase on the rank's stream: the phases with an exchange inside (index: the routed records and the
       # replicated table; contain: the keys; csr: the degrees) also hold the wait for the other ranks' serialised turns. The job's work
       # inflation is therefore taken from the kernel table of the same pass (profiles/prof_dist.sh -> tools/dist_kernel_table.py).
       "phases_with_exchange_waits": ["index", "contain", "csr"], "sum_phase_timers_ms_over_ranks": round(sum_kernel, 3),
       "work_inflation": kernel_table.get("work_inflation"), "work_inflation_source": kernel_table_path if kernel_table else None,
       "per_phase": per_phase,
       "per_rank": ranks,
       "bytes_sent_per_rank_mean": {k: int(sum(r["bytes_sent"][k] for r in ranks) / G) for k in ranks[0]["bytes_sent"]},
       "placement": ranks[0]["placement"], "own_reads_max_over_mean": round(max(r["own_reads"] for r in ranks) * G / max(1, sum(r["own_reads"] for r in ranks)), 4),
       "comm_ops_per_pass": max(r["comm_ops"] for r in ranks), "host_syncs_per_pass": max(r["host_syncs"] for r in ranks),
       "device_allocs_in_pass": max(r["device_allocs_in_pass"] for r in ranks), "device_frees_in_pass": max(r["device_frees_in_pass"] for r in ranks)}
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("placement", "own_reads_max_over_mean", "work_inflation", "sum_phase_timers_ms_over_ranks", "comm_ops_per_pass", "host_syncs_per_pass", "device_allocs_in_pass",
                                       "device_frees_in_pass", "bytes_sent_per_rank_mean")}, indent=1))
print(json.dumps(per_phase, indent=1))
print("single", single_wall, "arena", ranks[0]["arena_bytes"], ranks[0]["arena_peak"], "hbm_peak", ranks[0]["hbm_peak"])
