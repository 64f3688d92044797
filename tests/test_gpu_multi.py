"""-m gpu : the multi-GPU flow over RCCL on MORE THAN ONE PHYSICAL GPU — `buildG --gpus N` without --same-device and
`bench.py --gpus N` (child ranks under torch.distributed.run). Everything else in the suite drives the multi-rank code with ranks
that share one device over the in-process transport (tests/dist_util.py), because RCCL refuses two ranks per device; these tests
are the ones that put RcclComm's grouped send / recv, reduce-scatter(MIN) and the two communicators on two streams on real links.
On a 1-GPU box every test here is SKIPPED with the reason (tests/conftest.py adds -rs: the reason is in the run's summary).

The parent process never touches the GPU before the children are started (launch.gpu_count reads the KFD topology in sysfs)."""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from disco_amd import build, launch, readgen
from oracle import pyoracle, refrun
from tests import golden_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "disco_amd", "bin")
N_GPUS = launch.gpu_count()
need2 = pytest.mark.skipif(N_GPUS < 2, reason=f"needs >= 2 physical GPUs for one RCCL rank per device; this box has {N_GPUS}")


def _big_case(name):
    c = json.load(open(os.path.join(gu.GOLD, "cases_big.json")))[name]
    spec = readgen.GenSpec.coverage(c["seed"], c["reads"], c["read_len"], c["coverage"], n_contigs=c.get("n_contigs", 1),
                                    **({"len_max": c["len_max"]} if c.get("len_max") else {}), **({"skew": c["skew"]} if c.get("skew") else {}))
    return c, spec


def _canonical_files(prefix):
    edges = refrun.parse_pargraph(sorted(glob.glob(prefix + "_*_parGraph.txt")))
    cont = refrun.parse_contained(sorted(glob.glob(prefix + "_*_containedReads.txt")))
    return np.asarray(edges, dtype=np.int64), np.asarray(cont, dtype=np.int64)


@need2
@pytest.mark.parametrize("gpus,regime_env", [(2, {}), (2, {"DISCO_DIST_FORCE_GATHER": "1"}), (min(max(N_GPUS, 2), 8), {})])
def test_buildg_on_physical_gpus_writes_the_single_gpu_files(tmp_path, gpus, regime_env):
    """BASELINE config 2's reads (u150_1m: 1 M x 150 bp; the REAL reference's counts in tests/golden/cases_big.json, its digests
    reproduced by the single-GPU pass in tests/test_gpu_big.py) through `buildG --gpus N`, one RCCL rank per physical GPU, in the
    regular regime (rows on request) and the gathered-adjacency regime: canonical edge list and contained rows identical to the
    files of the single-GPU run on the same FASTA"""
    build.build_host()
    c, spec = _big_case("u150_1m")
    fa = str(tmp_path / "r.fasta")
    readgen.write_fasta(fa, readgen.generate_reads(spec))
    cfg = tmp_path / "disco.cfg"
    cfg.write_text(f"MinOverlap4BuildGraph = {c['min_overlap']}\n")
    out = {}
    for tag, extra, env in (("one", [], {}), ("many", ["--gpus", str(gpus)], regime_env)):
        prefix = str(tmp_path / f"g_{tag}")
        cmd = [os.path.join(BIN, "buildG"), "-se", fa, "-f", prefix, "-p", str(cfg), "-t", "4"] + extra
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, DISCO_VERBOSE="1", **env), timeout=900)
        assert p.returncode == 0, p.stdout[-4000:]
        if tag == "many":
            assert "transport rccl" in p.stdout, p.stdout[-2000:]  # the transport the ranks used (DISCO_VERBOSE)
        out[tag] = _canonical_files(prefix)
    assert (len(out["one"][0]), len(out["one"][1])) == (c["n_edges"], c["n_contained"])
    assert np.array_equal(out["one"][0], out["many"][0]) and np.array_equal(out["one"][1], out["many"][1])


@need2
@pytest.mark.parametrize("gpus", sorted({2, min(max(N_GPUS, 2), 8)}))
def test_bench_on_physical_gpus_counts_what_one_gpu_counts(gpus):
    """bench.py --gpus N (its own child ranks over RCCL) at BASELINE config 2's size: the job-wide counters of the line equal the
    single-GPU line's, and the line says N GPUs"""

    def run(extra):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "1000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-stage",
               "--no-host-to-host"] + extra
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        return json.loads([l for l in p.stdout.strip().split("\n") if l.strip()][-1])

    one = run([])
    many = run(["--gpus", str(gpus)])
    assert many["n_gpus"] == gpus and many["scaling"] == "strong"
    for k in ("e_pre", "e_out", "n_contained", "cap_bind_sites", "asymmetric_pairs"):
        assert many["config"][k] == one["config"][k], k
    ex = many["config"]["exchanges_rank0"]
    assert ex["transport"] == "rccl" and ex["regime"] == "regular"


def test_gpu_count_reads_sysfs_without_hip():
    """the launcher's device count must not come from the HIP runtime (a parent that only starts child ranks stays off the GPU)"""
    src = open(os.path.join(ROOT, "disco_amd", "launch.py")).read()
    body = src[src.index("def gpu_count"):src.index("def owner_range")]
    code = "\n".join(l for l in body.split("\n") if not l.strip().startswith(("\"\"\"", "SIMDs", "hipGetDeviceCount", "the way")))
    assert "import torch" not in code and "/sys/class/kfd" in code
    assert N_GPUS >= 1  # this is a GPU test: the box has one
