"""-m gpu : bench.py prints ONE JSON line, last on stdout, with the fields the driver reads (small workload so that it takes
seconds); also through the sharded code path over RCCL with one rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "300000", "--steps", "2", "--warmup", "1", "--cpu-sample-reads", "20000", "--no-stage"] + extra
    env = dict(os.environ, MASTER_PORT="29641")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.strip().split("\n") if l.strip()]
    return json.loads(lines[-1])  # the JSON line is the LAST line of stdout (RCCL's banner must not follow it)


def test_bench_line_has_the_contract_fields():
    d = _run([])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "overlaps/s" and d["higher_is_better"] is True
    assert d["value"] > 1e8 and d["ms_per_step"] > 0 and "workload" in d["config"] and d["vs_baseline"] is None
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["value"] > 0
    assert d["config"]["cap_bind_sites"] == 0 and d["config"]["asymmetric_pairs"] == 0
    h = d["graph_host_to_host"]  # host buffers in, host structs out
    assert h["e_out"] == d["config"]["e_out"] and h["total_ms"] > 0 and h["upload_bytes"] > 0
    assert "traffic_source" in r
    # round 4: the timer is named, every term of the SURVEY.md 8(d) byte contract belongs to a phase with a roofline entry, the
    # counter traffic comes as a bracket (FETCH_SIZE x 1 .. x 2)
    assert d["timer"] == "hbm_resident" and d["value_host_to_host"] > 0 and d["value_host_to_host_first_pass"] > 0
    entries = [r] + d["roofline_other"]
    assert {e["phase"] for e in entries} == {"index", "probe_kernel", "verify", "contain", "select", "trmark", "emit"}
    for e in entries:
        for k in ("kernel", "achieved", "frac", "traffic", "traffic_lo", "traffic_hi", "algorithmic_bytes_per_launch", "avg_launch_ms"):
            assert k in e, (e["phase"], k)
    assert abs(sum(e["algorithmic_bytes_per_launch"] for e in entries) - d["config"]["algorithmic_bytes_per_step"]) < 1.0
    assert 0.2 < d["roofline_coverage_of_step"] <= 1.05
    # round 6: the vector-issue roof beside the HBM one, per phase and for the pass. The instruction counts come from the stamped counter
    # profile of the BENCHED workload (50 M reads): at this test's size the entry says so instead of quoting them
    ri = d["roofline_issue"]
    assert ri["bound"] == "valu_issue" and "source" in ri
    assert ri["frac"] is None and "another workload" in ri["source"]
    for e in entries:
        assert "issue" in e and e["issue"] is None, e["phase"]


def test_bench_stage_wall_through_the_drop_in_executable():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "200000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-host-to-host"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.strip().split("\n") if l.strip()][-1])
    s = d["stage_drop_in"]
    assert s.get("same_overlap_count_as_the_bench_pass") is True and s["wall_s"] > 0 and s["output_bytes"] > 0, s


def test_bench_refuses_more_ranks_than_gpus():
    """`--gpus 2` on a 1-GPU box must fail loudly instead of printing a line measured on one GPU"""
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has 2 GPUs")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "300000", "--steps", "1", "--warmup", "0"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and "--gpus 2" in p.stderr and not p.stdout.strip()


def test_bench_sharded_code_path_with_one_rank():
    d = _run(["--force-distributed", "--no-cpu-baseline"])
    assert d["n_gpus"] == 1 and d["value"] > 1e8 and d["config"]["e_out"] > 0
    # the RCCL transport itself, not the in-process stand-in (a silent fall-back to LoopComm would keep everything else green)
    assert d["config"]["exchanges_rank0"]["transport"] == "rccl", d["config"]["exchanges_rank0"]


def test_bench_illumina_like_variant():
    """the optional variant with substitution errors (not a BASELINE config) — fewer overlaps survive the exact
    compare, and the line says what it was measured on"""
    clean = _run(["--no-cpu-baseline", "--no-host-to-host"])
    d = _run(["--errors-ppm", "1000"])
    assert "1000 substitutions per 10^6 bases" in d["config"]["workload"] and "cpu_baseline" not in d
    assert 0.5 * clean["config"]["e_pre"] < d["config"]["e_pre"] < clean["config"]["e_pre"]
    assert d["config"]["e_out"] > clean["config"]["e_out"]  # broken transitivity leaves more edges
