#!/usr/bin/env python3
"""bench.py — overlaps/sec of the BuildGraph stage on synthetic 150 bp reads (BASELINE.json metric).

A step = one pass of the hot path (index build -> fused probe+verify -> containment -> edge selection -> twin check ->
transitive reduction -> emission) over reads that are already resident in HBM, results left in HBM.
  N = 1 : the whole pass on one MI355X.
  N > 1 : strong scaling — the same job, the reads range-partitioned in HBM when a step starts (one rank per GPU), dealt to the ranks by
          their read-level minimizer inside the pass (ranks own loci), every exchange an RCCL collective inside libdisco_hip.so
          (disco_dist_run_graph).  value = E_pre of the whole job / max-over-ranks time.
          `python bench.py --gpus N` without a launcher starts its own N ranks (a child torch.distributed.run).
Prints ONE JSON line on rank 0.

Which timer `value` is (three walls are in the line; tests/test_gpu_bench.py asserts this mapping). The build contract of this repository
(task statement, section "Measurement") defines it in these words: "`value` is whole-job throughput with inputs already resident in HBM
when the timed region starts (if the boundary hands over host buffers, note the PCIe-inclusive rate in DESIGN.md — it is never `value`)".
So:  value                  = E_pre / (HBM-resident pass), "timer": "hbm_resident"  — the contract's definition, quoted above;
     value_host_to_host     = SURVEY.md §8(d)'s *graph* wall t_graph: pinned host buffers in -> host structs out (PCIe inclusive);
     stage_drop_in.wall_s   = SURVEY.md §8(d)'s *stage* wall: buildG's argv to files closed.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import subprocess
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the pool's host driver supports dmabuf IPC only: without this RCCL's buffer sharing between the rank processes fails with
# "hipIpcGetMemHandle: invalid argument". Exported on the boxes already; set before the HIP runtime loads in case a launcher drops it.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 GB/s is attainable
# vector-issue roof (MI355X_MICROARCH.md): 256 CUs x 4 SIMDs, a wavefront instruction occupies its SIMD's issue slot for 4 cycles (64 lanes on
# 16-lane SIMDs), 2.4 GHz maximum clock (the chip holds less under load: the roof is the nominal one, like the 8 TB/s). The scalar unit of a
# CU serves its four SIMDs in turn — one scalar instruction per SIMD every 4 cycles — so the same figure bounds SQ_INSTS_SALU.
N_SIMDS, CYCLES_PER_WAVE_INST, CLOCK_GHZ = 1024, 4, 2.4
ISSUE_PEAK_GINST_S = N_SIMDS / CYCLES_PER_WAVE_INST * CLOCK_GHZ  # 614.4 G wavefront-instructions / s


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("DISCO_BENCH_READS", 50_000_000)))
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--min-overlap", type=int, default=40)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--cpu-sample-reads", type=int, default=1_000_000)  # = BASELINE config 2
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-to-host", action="store_true", help="skip the host-buffers-in / host-structs-out pass")
    ap.add_argument("--no-stage", action="store_true", help="skip the stage wall (FASTA -> files through disco_amd/bin/buildG) at the benched config")
    ap.add_argument("--errors-ppm", type=int, default=0, help="per-base substitution rate in 10^-6 (SURVEY.md 8d: the optional Illumina-like variant, 1000; "
                    "not a BASELINE config: implies --no-cpu-baseline --no-stage --no-host-to-host)")
    ap.add_argument("--partitioned-index", action="store_true", help="multi-GPU passes keep the index hash-partitioned: lookups travel to the buckets' owners, "
                    "matching records back (slower than replicating the built slices wherever the index fits every GPU; DESIGN.md section 5)")
    ap.add_argument("--force-distributed", action="store_true", help="run the multi-GPU code path (RCCL communicator, every exchange) even with one rank")
    return ap.parse_args()


# phase of disco_phase_ms -> the kernel(s) behind it
PHASE_KERNELS = {"index": "index_runs_kernel + scan + index_fill_ordered_kernel", "probe_kernel": "probe_runs_kernel", "verify": "verify_flat_kernel",
                 "contain": "contain_flags_kernel", "select": "edge_select_flat_kernel", "trmark": "transitive_mark_kernel", "emit": "emit_half_kernel"}


def algorithmic_bytes(cnt, n_reads, words_mean):
    """SURVEY.md §8(d) contract: B = N*R (pack/read once for indexing) + 2N*16 (index entries) + N*R (read again to generate
    queries) + Q*8 (one 8-byte index word per probe) + H*R (fetch candidate sequence to verify) + E_pre*16 (write overlap
    records) + 2*E_pre*16 (read both directions for reduction) + E_out*16 + C*16 (write results).
    Returns (whole path, {phase: its share}) for one pass; every term of the contract belongs to exactly one phase:
      index   = N*R + 2N*16        (reads packed once + the two index entries of a read)
      probe   = N*R + Q*8          (query reads + the index words of all probes)
      verify  = H*R + E_pre*16     (candidate rows + overlap records)
      select  = E_pre*16           (the overlap records read once: 2 E_pre directed hits of 8 bytes)
      trmark  = E_pre*16           (... and the adjacency they became, read for the reduction)
      emit    = E_out*16 ; contain = C*16   (results)"""
    R = 8.0 * words_mean
    N, Q, H = n_reads, cnt["probes"], cnt["kmer_hits"]
    E_pre, E_out, C = cnt["e_pre"], cnt["e_out"], cnt["n_contained"]
    per_phase = {"index": N * R + 2 * N * 16, "probe_kernel": N * R + Q * 8, "verify": H * R + E_pre * 16, "select": E_pre * 16, "trmark": E_pre * 16,
                 "emit": E_out * 16, "contain": C * 16}
    return sum(per_phase.values()), per_phase


def issue_phase(t, ms, reads):
    """one phase against the VECTOR-ISSUE roof — the roof that binds these kernels: wavefront instructions per launch from the
    stamped counter profile (SQ_INSTS_VALU / SQ_INSTS_SALU of the phase's kernels: t = its entry in profiles/probe_traffic.json; same
    source-fingerprint rule as `traffic`) over the issue rate of the chip, against the phase's duration measured live (ms).
    frac = the share of the phase's time its vector instructions alone need at full issue."""
    if not t.get("valu") or ms <= 0:
        return None
    floor = {u: (t.get(u) or 0.0) / (ISSUE_PEAK_GINST_S * 1e9) * 1e3 for u in ("valu", "salu")}
    lds = t.get("lds") or 0.0
    return {"bound": "valu_issue", "valu_insts": t["valu"], "salu_insts": t.get("salu"), "achieved": t["valu"] / (ms * 1e-3) / 1e9, "peak": ISSUE_PEAK_GINST_S,
            "unit": "G wavefront-instructions/s", "frac": floor["valu"] / ms, "frac_salu": floor["salu"] / ms, "floor_ms": floor["valu"], "floor_ms_salu": floor["salu"],
            "valu_insts_per_read": t["valu"] / reads, "lds_insts": lds,
            # vector AND scalar instructions together per second (round 6: the marking kernel issued 1.04e12 of them per second, half of
            # them scalar, and its time fell with either kind — a phase near 1000 G/s is bound by instruction issue whatever `frac` says)
            "achieved_valu_plus_salu": (t["valu"] + (t.get("salu") or 0.0)) / (ms * 1e-3) / 1e9,
            "lds_bank_conflict_cycles_per_lds_inst": ((t.get("lds_bank_conflict") or 0.0) / lds) if lds else None}


def issue_pass(ph_issue, ms_per_step, reads, note):
    """the whole pass against the vector-issue roof: the instructions of every phase with a counter profile, at 4 cycles per wavefront
    instruction on 1024 SIMDs at 2.4 GHz (0.63 of it against 0.235 of the HBM roof in round 5)"""
    if not all(ph_issue.get(ph) for ph in ("index", "probe_kernel", "verify", "select", "trmark")):
        return {"bound": "valu_issue", "frac": None, "source": note}
    v_tot = sum(x["valu_insts"] for x in ph_issue.values() if x)
    s_tot = sum((x["salu_insts"] or 0.0) for x in ph_issue.values() if x)
    floor = v_tot / (ISSUE_PEAK_GINST_S * 1e9) * 1e3
    return {"bound": "valu_issue", "valu_insts_per_step": v_tot, "salu_insts_per_step": s_tot, "cycles_per_wave_inst": CYCLES_PER_WAVE_INST, "simds": N_SIMDS,
            "clock_ghz": CLOCK_GHZ, "peak": ISSUE_PEAK_GINST_S, "unit": "G wavefront-instructions/s", "achieved": v_tot / (ms_per_step * 1e-3) / 1e9,
            "floor_ms": floor, "frac": floor / ms_per_step, "valu_insts_per_read": v_tot / reads,
            "achieved_valu_plus_salu": (v_tot + s_tot) / (ms_per_step * 1e-3) / 1e9,
            "per_phase": {ph: {k: x[k] for k in ("valu_insts", "salu_insts", "floor_ms", "frac", "frac_salu", "valu_insts_per_read", "achieved_valu_plus_salu", "lds_bank_conflict_cycles_per_lds_inst")}
                          for ph, x in ph_issue.items() if x},
            "source": note.replace("FETCH_SIZE + WRITE_SIZE", "SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS / SQ_LDS_BANK_CONFLICT")}


KERNEL_SOURCES = tuple(sorted("disco_amd/csrc/" + f for f in os.listdir(os.path.join(ROOT, "disco_amd", "csrc")) if f.endswith((".h", ".hip"))))


def kernels_sha16():
    """fingerprint of the kernel sources: profiles/probe_traffic.json carries the one it was measured with, and its counter
    figures are only quoted for exactly these kernels"""
    import hashlib

    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def reference_at_config3():
    """the REAL reference's run on BASELINE config 3 (50 M x 150 bp), recorded once (it takes an hour; the GPU box's baseline leg of every bench
    run samples config 2 instead): on the host cores of a GPU box of this pool (round 6: tests/golden/u150_50m.reference_run_gpubox.txt, 16
    usable cores — the same-box figure) and, before that, in the build container (tests/golden/u150_50m.reference_run.txt, 7 threads)"""
    out = None
    for fn, threads, where in (("u150_50m.reference_run_gpubox.txt", 16, "host cores of an MI355X box of this pool (16 usable of 256: the job's cgroup)"),
                               ("u150_50m.reference_run.txt", 7, "build container, 8 vCPU")):
        path = os.path.join(ROOT, "tests", "golden", fn)
        try:
            txt = open(path).read()
            t = sum(float(re.search(r"Function %s\(\) finished in ([0-9.eE+-]+) Seconds" % f, txt).group(1)) for f in ("insertDataset", "buildOverlapGraphFromHashTable"))
            wall = float(re.search(r"Function main\(\) finished in ([0-9.eE+-]+) Seconds", txt).group(1))
            cases = json.load(open(os.path.join(ROOT, "tests", "golden", "cases_big.json")))
            e_pre = 903_537_181  # = the HIP path's count on the same (digest-checked) reads; the reference does not print it
            rec = {"overlaps_per_s": e_pre / t, "graph_s": t, "whole_process_s": wall, "threads": threads, "where": where,
                   "edges": cases["u150_50m"]["n_edges"], "source": "tests/golden/" + fn}
            if out is None:
                out = rec
            else:
                out["other_run"] = rec
        except Exception:
            continue
    return out


def usable_cores():
    """host cores this process may actually use: the cgroup CPU quota when there is one (os.cpu_count() reports the
    machine's cores even inside a container limited to a fraction of them)"""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return n


def cpu_baseline(args, spec_full):
    """reference buildG (oracle/_ref/buildG_ref, kind 'reference') or the C restatement (kind 'port') on a bounded
    sample of the same workload: same read length / coverage / min-overlap, smaller genome."""
    import numpy as np

    from disco_amd import buildgraph, readgen
    from oracle import pyoracle, refrun

    n = min(args.cpu_sample_reads, args.reads)
    spec = readgen.GenSpec.coverage(args.seed + 1, n, args.read_len, args.coverage)
    # the unit count of the sample comes from the (parity-checked) HIP path
    with buildgraph.BuildGraph(min_overlap=args.min_overlap, device=0) as g:
        g.generate_reads(spec)
        g.run_graph()
        e_pre = g.counters()["e_pre"]
    cores = usable_cores()
    sample = f"{n} x {args.read_len} bp reads, {args.coverage:g}x of a {spec.contig_len * spec.n_contigs} bp random genome, min-overlap {args.min_overlap}"
    if refrun.available():
        d = tempfile.mkdtemp(prefix="disco_cpu_")
        fa = os.path.join(d, "sample.fasta")
        gen = os.path.join(ROOT, "disco_amd", "bin", "readgen")  # same generator, C++ (the numpy twin needs ~1 s per 100 k reads)
        if os.path.exists(gen) and spec.n_contigs == 1:
            subprocess.run([gen, fa, str(n), str(args.read_len), repr(float(args.coverage)), str(args.seed + 1), str(args.read_len),
                            str(spec.contig_len)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        else:
            readgen.write_fasta(fa, readgen.generate_reads(spec))
        r = refrun.run_reference([fa], args.min_overlap, threads=cores, mem_gb=max(8, 2 * cores), workdir=d)
        # graph timer of the reference = HashTable::insertDataset + OverlapGraph (containment, edges, reduction, write)
        t = 0.0
        for fn in ("insertDataset", "buildOverlapGraphFromHashTable"):
            m = re.search(r"Function %s\(\) finished in ([0-9.eE+-]+) Seconds" % fn, r["log"])
            t += float(m.group(1)) if m else 0.0
        if t <= 0:
            t = r["wall"]
        out = dict(value=e_pre / t, unit="overlaps/s", cores=cores, kind="reference",
                   sample=sample + f"; reference buildG -t {cores}: graph {t:.2f} s, whole process {r['wall']:.2f} s")
        # the same FASTA through the drop-in executable: the *stage* wall of SURVEY.md 8(d) (argv to files closed: parse + filter +
        # pack on the host cores, upload, graph on the GPU, fetch, text output), next to the reference's whole-process time
        mine = os.path.join(ROOT, "disco_amd", "bin", "buildG")
        if os.path.exists(mine):
            try:
                os.makedirs(os.path.join(d, "mine"), exist_ok=True)
                t0 = time.perf_counter()
                p = subprocess.run([mine, "-se", fa, "-f", os.path.join(d, "mine", "g"), "-p", os.path.join(d, "disco.cfg"), "-t", str(cores)],
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                wall = time.perf_counter() - t0
                if p.returncode == 0:
                    out["stage_drop_in"] = {"wall_s": wall, "overlaps_per_s": e_pre / wall, "reference_wall_s": r["wall"],
                                            "what": "disco_amd/bin/buildG on the same FASTA, process start to files closed"}
            except Exception:
                pass
        return out
    codes, off = readgen.generate_codes(spec)
    t0 = time.perf_counter()
    pyoracle.build_graph(codes, off, args.min_overlap)
    t = time.perf_counter() - t0
    return dict(value=e_pre / t, unit="overlaps/s", cores=1, kind="port", sample=sample + f"; C restatement {t:.2f} s")


def stage_wall(args, spec, e_pre):
    """SURVEY.md §8(d) 'stage' wall at the BENCHED configuration: the FASTA of the same reads (disco_amd/bin/readgen) through the
    drop-in executable, process start to files closed — parse + filter + pack on the host cores, upload, graph on the GPU,
    fetch, text output. What the reference's 'Function main() finished in' line measures."""
    import shutil

    gen = os.path.join(ROOT, "disco_amd", "bin", "readgen")
    exe = os.path.join(ROOT, "disco_amd", "bin", "buildG")
    if not (os.path.exists(gen) and os.path.exists(exe)):
        return {"skipped": "disco_amd/bin/readgen or buildG not built"}
    need = args.reads * (args.read_len + 12) * 1.5 + (1 << 30)
    d = tempfile.mkdtemp(prefix="disco_stage_")
    try:
        if shutil.disk_usage(d).free < need:
            return {"skipped": "not enough free disk for the FASTA and the edge files (%.1f GB needed)" % (need / 1e9)}
        fa = os.path.join(d, "reads.fasta")
        t0 = time.perf_counter()
        subprocess.run([gen, fa, str(args.reads), str(args.read_len), repr(float(args.coverage)), str(args.seed), str(args.read_len), "5000000"],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t_gen = time.perf_counter() - t0
        with open(os.path.join(d, "disco.cfg"), "w") as f:
            f.write(f"MinOverlap4BuildGraph = {args.min_overlap}\n")
        cores = usable_cores()
        # three runs, the median reported (all three listed): on the shared boxes of this pool one run in three meets a burst of
        # interference (other tenants' jobs on the node; the runtime's allocator lock behind the 28 GB hit buffer of a process that
        # starts while the driver still scrubs what the previous one freed) that doubles single phases — profiles/r05_stage_runs.txt.
        # The input is in the page cache in every run (the generator wrote it seconds ago: "input_page_cache": "warm").
        # round 6: a process that starts while the driver is still reclaiming the device memory of one that has just exited finds ONE of its
        # first large allocations blocked for 1-3.7 s (tools/micro/alloc_probe.cpp: after a process that held 10-45 GB, one hipMalloc in a
        # series — any size, the 18th of 48 x 1 GB in one trial — waits; a fresh device never does): three stage runs back to back measure
        # that, twice. The three timed runs therefore start SETTLE_S seconds after the process before them has gone — a sample processed
        # on its own, which is how the stage runs — and one more run immediately behind them is reported as wall_s_back_to_back.
        SETTLE_S = 4.0
        runs = []
        for rep in range(4):
            for f in os.listdir(d):
                if f.startswith("g_"):
                    os.unlink(os.path.join(d, f))
            if rep < 3:
                time.sleep(SETTLE_S)
            t0 = time.perf_counter()
            p = subprocess.run([exe, "-se", fa, "-f", os.path.join(d, "g"), "-p", os.path.join(d, "disco.cfg"), "-t", str(cores)],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, DISCO_VERBOSE="1"))
            runs.append((time.perf_counter() - t0, p))
            if p.returncode != 0:
                return {"failed": p.stdout[-500:]}
        back_to_back = runs.pop()[0]
        wall, p = sorted(runs, key=lambda x: x[0])[1]

        def laps(text):
            out = {}
            for key, pat in (("parse_filter_pack_s", r"Function readDataset\(\) finished in ([0-9.eE+-]+)"), ("graph_s", r"\[GPU\] finished in ([0-9.eE+-]+)"),
                             ("upload_s", r"host->device ([0-9.eE+-]+)"), ("fetch_and_write_s", r"Function saveParGraphToFile\(\) finished in ([0-9.eE+-]+)"),
                             ("process_start_to_context_s", r"process start to context ready\s+([0-9.eE+-]+)"), ("files_into_hbm_s", r"files into HBM ([0-9.eE+-]+) s"),
                             # round 6: parse_filter_pack_s taken apart (file -> pinned ring -> HBM is files_into_hbm_s; then the kernels), and what follows the graph
                             ("records_filter_ids_rows_kernels_s", r"records \+ filter \+ ids \+ rows ([0-9.eE+-]+) s"),
                             ("reader_waited_for_copies_ms", r"waited [0-9.]+ ms for pieces, ([0-9.eE+-]+) ms for copies"),
                             ("lengths_to_host_s", r"lengths \+ file indices to the host\s+([0-9.eE+-]+)"), ("fetch_contained_rows_s", r"fetch contained rows\s+([0-9.eE+-]+)"),
                             ("partition_edges_into_files_s", r"partition edges into files\s+([0-9.eE+-]+)"), ("format_edge_lines_gpu_s", r"format edge lines on the GPU\s+([0-9.eE+-]+)"),
                             ("edge_lines_into_files_s", r"edge lines into the files\s+([0-9.eE+-]+)"), ("main_s", r"Function main\(\) finished in ([0-9.eE+-]+)")):
                m = re.search(pat, text)
                if m:
                    out[key] = float(m.group(1))
            return out

        parts = laps(p.stdout)
        # once with the input NOT in the page cache (the three runs above read it from memory: the generator wrote it seconds ago): the
        # file's pages are dropped (fsync + POSIX_FADV_DONTNEED: no privilege needed) and the stage reads it from the disk
        cold = None
        try:
            fd = os.open(fa, os.O_RDONLY)
            os.fsync(fd)
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
            os.close(fd)
            for f in os.listdir(d):
                if f.startswith("g_"):
                    os.unlink(os.path.join(d, f))
            time.sleep(SETTLE_S)
            t0 = time.perf_counter()
            pc = subprocess.run([exe, "-se", fa, "-f", os.path.join(d, "g"), "-p", os.path.join(d, "disco.cfg"), "-t", str(cores)],
                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, DISCO_VERBOSE="1"))
            if pc.returncode == 0:
                cl = laps(pc.stdout)
                cold = {"wall_s": round(time.perf_counter() - t0, 3), "files_into_hbm_s": cl.get("files_into_hbm_s"), "main_s": cl.get("main_s"),
                        "what": "one more run after the FASTA's pages were dropped from the page cache: the 8 GB come from the disk"}
        except Exception as e:
            cold = {"skipped": str(e)}
        m = re.search(r"overlaps \(pre-reduction\) : (\d+)", p.stdout)
        same = (int(m.group(1)) == e_pre) if m else None
        out_bytes = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f.startswith("g_"))
        return {"wall_s": round(wall, 3), "wall_s_runs": [round(w, 3) for w, _ in runs],
                "wall_s_is": "median of three runs, each started %.0f s after the process before it had exited" % SETTLE_S,
                "wall_s_back_to_back": round(back_to_back, 3), "input_page_cache": "warm",
                "overlaps_per_s": e_pre / wall, "host_threads": cores, "fasta_bytes": os.path.getsize(fa), "output_bytes": out_bytes,
                "same_overlap_count_as_the_bench_pass": same, "fasta_generation_s": round(t_gen, 2), **parts, "cold_page_cache": cold,
                "what": "disco_amd/bin/buildG on the FASTA of the benched reads, process start to files closed (the reference's main() timer)"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a CHILD torch.distributed.run (this
    process has not touched the GPU and never will — no exec of a GPU-initialised process), relay its output and exit code.
    Replaces mpirun of the reference's multi-process binaries (MPI/main.cpp:29-37)."""
    import socket

    from disco_amd import launch

    have = launch.gpu_count()  # from the KFD topology in sysfs: this parent never opens the HIP runtime
    if have < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this machine has {have} GPU(s); one rank per GPU is required "
                         f"(refusing to print a {args.gpus}-GPU line measured on fewer)")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse_args()
    if args.errors_ppm:
        args.no_cpu_baseline = args.no_stage = args.no_host_to_host = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    if world > 1:
        # a multi-rank run that stops making progress (a rank lost, a collective that never completes) must end by itself with a
        # message, not sit in a collective until somebody else's limit kills the box: stacks to stderr, then exit
        import faulthandler
        import threading

        limit = float(os.environ.get("DISCO_BENCH_WATCHDOG_S", "900"))

        def _expired():
            sys.stderr.write(f"bench.py: rank {rank} of {world} still running after {limit:.0f} s — giving up (DISCO_BENCH_WATCHDOG_S)\n")
            faulthandler.dump_traceback(file=sys.stderr)
            sys.stderr.flush()
            os._exit(124)

        wd_state = {"t": None}

        def kick():
            """progress: re-arm the watchdog (every step and every phase of the run calls it)"""
            if wd_state["t"] is not None:
                wd_state["t"].cancel()
            wd_state["t"] = threading.Timer(limit, _expired)
            wd_state["t"].daemon = True
            wd_state["t"].start()

        def unwatch():
            if wd_state["t"] is not None:
                wd_state["t"].cancel()
                wd_state["t"] = None

        kick()
    else:
        def kick():
            pass

        def unwatch():
            pass
    import torch

    from disco_amd import buildgraph, launch, readgen

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    torch.zeros(1, device=device)  # torch's HIP runtime opens the device before libdisco_hip.so does (tests/conftest.py)
    sharded = world > 1 or args.force_distributed
    genome = int(args.reads * args.read_len / args.coverage)
    n_contigs = max(1, genome // 5_000_000)  # 5 Mbp contigs, reads never span contigs
    spec = readgen.GenSpec.coverage(args.seed, args.reads, args.read_len, args.coverage, n_contigs=n_contigs)
    g = buildgraph.BuildGraph(min_overlap=args.min_overlap, device=local_rank)
    cp = None
    if sharded:
        # control plane: gloo (rendezvous, barrier, max over ranks). Data plane: RCCL inside libdisco_hip.so — the unique id of
        # its communicator travels over the control plane, as MPI_Bcast would carry it (disco_comm_init)
        cp = launch.ControlPlane("gloo")
        g.comm_init(cp.broadcast_unique_id(buildgraph.BuildGraph.comm_unique_id), world, rank)
        g.dist_generate_reads(spec)  # every rank generates ITS range of the reads: the inputs are range-partitioned in HBM
    else:
        g.generate_reads(spec)  # inputs resident in HBM before the timed region
    if args.errors_ppm:
        g.substitute_bases(args.seed + 1, args.errors_ppm)  # every rank: its own range of the table

    def step():
        if not sharded:
            g.run_graph()
        else:
            g.dist_run_graph(gather_reads=True, partitioned_index=args.partitioned_index)  # every pass starts from the range-partitioned reads: the all-gather is timed

    def fence():
        g.synchronize()
        torch.cuda.synchronize(device)
        if sharded:
            cp.barrier()
        g.synchronize()
        torch.cuda.synchronize(device)

    first_pass_info = None
    for w in range(args.warmup):
        step()
        if sharded and w == 0:  # the first pass of a multi-rank context waits behind every exchange (first contact): its table shows the exchanges themselves
            first_pass_info = g.dist_info()
        kick()
    kern_ms = {k: [] for k in PHASE_KERNELS}
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        ph = g.phase_ms()  # HIP events around each kernel launch on the stream it was launched on
        for k in kern_ms:
            kern_ms[k].append(ph.get(k, 0.0))
        kick()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = cp.max_over_ranks(elapsed)
    ms_per_step = elapsed / max(args.steps, 1) * 1e3

    cnt = g.counters()
    phases = g.phase_ms()
    info = g.dist_info() if sharded else None
    e_pre = info["e_pre"] if sharded else cnt["e_pre"]
    e_out = info["e_out"] if sharded else cnt["e_out"]
    words_mean = float((args.read_len + 31) // 32)  # W of SURVEY.md §8: packed words per read (device rows are padded to 64 B)
    if sharded:  # whole-job counters from the pass itself
        cnt_all = dict(cnt, probes=info["probes"], kmer_hits=info["kmer_hits"], e_pre=e_pre, e_out=e_out, n_contained=info["n_contained"],
                       cap_bind_sites=info["cap_bind_sites"], asymmetric_pairs=info["asymmetric_pairs"])
    else:
        cnt_all = cnt
    total_b, kern_b = algorithmic_bytes(cnt_all, args.reads, words_mean)
    avg_ms = {k: sum(v) / max(len(v), 1) for k, v in kern_ms.items()}
    dominant = max(avg_ms, key=lambda k: avg_ms[k])  # the longest kernel of the pass
    tj, traffic_note = {}, "no counter profile for this workload"
    tfile = os.path.join(ROOT, "profiles", "probe_traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            if tj.get("reads") != args.reads or tj.get("gpus", 1) != world:
                tj, traffic_note = {}, "profiles/probe_traffic.json was taken on another workload"
            elif tj.get("kernels_sha16") != kernels_sha16():
                tj, traffic_note = {}, ("profiles/probe_traffic.json was taken with other kernel sources (%s, now %s): re-run profiles/prof_pmc.sh"
                                        % (tj.get("kernels_sha16"), kernels_sha16()))
            else:
                traffic_note = "FETCH_SIZE + WRITE_SIZE of profiles/probe_traffic.json (rocprofv3 --pmc, same kernel sources: %s)" % tj["kernels_sha16"]
        except Exception:
            tj = {}

    try:  # attainable HBM bandwidth on this box (streaming copy kernel), reported beside the nominal peak
        hbm_measured = g.measure_hbm(2 << 30, 5)
    except Exception:
        hbm_measured = None
    try:  # ... and of the pattern the dominant kernels actually issue: random 64-byte rows out of the packed-read table
        gather_measured = g.measure_gather(max(args.reads * 64, 1 << 20), 3)
    except Exception:
        gather_measured = None

    def roof(ph):
        b_launch = kern_b[ph] / world  # one launch processes one shard
        ms = avg_ms[ph]
        ach = b_launch / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        # counter traffic: FETCH_SIZE reads 1.014 x the known bytes on a pure 64-byte row gather and 0.500 x on a wide coalesced stream
        # (calibration kernels in the same profile; MI355X_MICROARCH.md, HBM section); the hot kernels mix both, so the true figure lies
        # between traffic_lo (fetch x 1 + write) and traffic_hi (fetch x 2 + write). `traffic` = traffic_lo.
        t = tj.get("phases", {}).get(ph) or {}
        return {"bound": "hbm", "kernel": PHASE_KERNELS[ph], "phase": ph, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "peak_measured_copy": hbm_measured, "peak_measured_row_gather": gather_measured,
                "traffic": t.get("lo"), "traffic_lo": t.get("lo"), "traffic_hi": t.get("hi"), "traffic_source": traffic_note,
                "algorithmic_bytes_per_launch": b_launch, "avg_launch_ms": ms, "issue": issue(ph)}

    def issue(ph):
        return issue_phase(tj.get("phases", {}).get(ph) or {}, avg_ms[ph], args.reads)

    out = {
        "metric": "overlaps/sec (BuildGraph stage), 150 bp reads",
        "value": e_pre / (ms_per_step * 1e-3),
        "timer": "hbm_resident",  # reads resident in HBM when the timed region starts, results left in HBM; SURVEY.md 8(d)'s t_graph
                                  # (host buffers in, host structs out) is value_host_to_host, its stage timer stage_drop_in
        "unit": "overlaps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": f"{args.reads} x {args.read_len} bp synthetic reads, {args.coverage:g}x of {n_contigs} x {spec.contig_len} bp "
                        f"uniform-random contigs, both strands, {'error-free' if not args.errors_ppm else f'{args.errors_ppm} substitutions per 10^6 bases (NOT a BASELINE config)'}, min-overlap {args.min_overlap} (k={args.min_overlap - 1}), "
                        f"transitive reduction on, seed {args.seed}",
            "reads": args.reads, "read_len": args.read_len, "min_overlap": args.min_overlap,
            "parallelism": "1 GPU" if not sharded else (
                f"{world} GPU(s), one rank each: reads and graph nodes range-partitioned, index built hash-partitioned (all-to-all of "
                f"records, all-gather of shards), neighbour rows on request, survivor push; RCCL inside libdisco_hip.so"),
            "e_pre": e_pre, "e_out": e_out, "n_contained": cnt_all["n_contained"],
            "cap_bind_sites": cnt_all["cap_bind_sites"], "asymmetric_pairs": cnt_all["asymmetric_pairs"],
            "probes": cnt_all["probes"], "kmer_hits": cnt_all["kmer_hits"],
            "reads_per_s": args.reads / (ms_per_step * 1e-3),
            "phase_ms_rank0": {k: round(v, 3) for k, v in phases.items()},
            "algorithmic_bytes_per_step": total_b,
            "path_gbs": total_b / (ms_per_step * 1e-3) / 1e9,
        },
        "roofline": roof(dominant),
        "roofline_other": [roof(k) for k in sorted(avg_ms, key=lambda k: -avg_ms[k]) if k != dominant],
        # share of ms_per_step that the kernels with a roofline entry account for (the rest: the grouping's scan / scatter, small
        # bookkeeping kernels, launch gaps)
        "roofline_coverage_of_step": sum(avg_ms.values()) / ms_per_step if ms_per_step > 0 else None,
    }
    # round 6: the pass against the VECTOR-ISSUE roof — the one that binds it
    out["roofline_issue"] = issue_pass({ph: issue(ph) for ph in PHASE_KERNELS}, ms_per_step, args.reads, traffic_note)
    if not sharded and not args.no_host_to_host:
        # SURVEY.md §8(d) 'graph' wall: host buffers in, host structs out (the HBM-resident `value` never includes the copies)
        try:
            h2h = g.host_to_host_pass()
            h2h["overlaps_per_s"] = e_pre / (h2h["total_ms"] * 1e-3)
            h2h["what"] = ("pinned packed reads in host memory (rows at the words they use) -> disco_upload_reads (chunked copy, rows spread and the index's count "
                           "pass behind it) -> whole pass (contained rows leave on a side stream during it) -> disco_fetch_contained + disco_fetch_edges into "
                           "host structs (12 bytes per row / edge over the link); second of two such passes: the context keeps its buffers and the caller "
                           "its result arrays")
            out["graph_host_to_host"] = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in h2h.items()}
            # SURVEY.md 8(d)'s t_graph (host buffers in, host structs out) next to the HBM-resident `value`
            out["value_host_to_host"] = h2h["overlaps_per_s"]  # steady state (second pass of a warm context)
            if h2h.get("first_pass_total_ms"):
                out["value_host_to_host_first_pass"] = e_pre / (h2h["first_pass_total_ms"] * 1e-3)  # what a cold caller sees
        except Exception as e:
            out["graph_host_to_host"] = {"failed": str(e)}
    if sharded:  # rank 0's view of the exchanges of the last pass
        out["config"]["exchanges_rank0"] = {"regime": {0: "regular", 2: "regular after twin completion across ranks"}.get(info["regime"], "order-dependent (adjacency gathered)"),
                                            "transport": g.transport, "tr_rounds": info["tr_rounds"], "tr_deferred": info["tr_deferred"],
                                            "bytes_sent": info["bytes_sent"], "ms": {k: round(v, 3) for k, v in info["ms"].items()},
                                            "ms_pass": round(info["ms_total"], 3),
                                            "ms_is": "host time to ISSUE each exchange in the last timed pass (stream ordered: nothing waits behind them)"}
        if first_pass_info is not None:
            out["config"]["exchanges_rank0"]["first_pass_ms"] = {k: round(v, 3) for k, v in first_pass_info["ms"].items()}
            out["config"]["exchanges_rank0"]["first_pass_ms_is"] = ("the untimed first (warm-up) pass waits behind every all-to-all on multi-rank RCCL: the exchanges "
                                                                    "themselves, link time included (DISCO_DIST_NO_FIRST_CONTACT=1 takes the waits away)")
    g.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            kick()
            out["cpu_baseline"] = cpu_baseline(args, spec)
        except Exception as e:  # the baseline is a reported extra; never lose the bench line over it
            out["cpu_baseline"] = {"value": None, "unit": "overlaps/s", "cores": os.cpu_count(), "kind": "reference", "sample": f"failed: {e}"}
        if args.reads == 50_000_000 and args.read_len == 150 and args.min_overlap == 40 and args.seed == 42:
            ref3 = reference_at_config3()
            if ref3:
                out["cpu_baseline"]["reference_at_benched_config"] = ref3
    if rank == 0 and world == 1 and not args.no_stage:
        try:
            out["stage_drop_in"] = stage_wall(args, spec, e_pre)
        except Exception as e:
            out["stage_drop_in"] = {"failed": str(e)}
    if sharded:
        cp.close()
    unwatch()
    if rank == 0:
        try:  # RCCL writes its version banner through C stdio, which is flushed at exit: push it out before the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
