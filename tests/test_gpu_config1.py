"""-m gpu : BASELINE config 1, the graph half. The command runDisco.sh issues for `-inP reads -n 4` (runDisco.sh:200) through the
drop-in executable must reproduce the canonical graph of the run done once in the build container with the REAL reference
(tools/run_config1.py -> tests/golden/config1.json: runDisco.sh + real buildG / fullsimplify / parsimplify to scaffolds, and the
same pipeline with the drop-in's files, which ended in the same scaffold)."""
import glob
import hashlib
import json
import os
import subprocess

import pytest

from disco_amd import build, readgen
from oracle import pyoracle, refrun

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = json.load(open(os.path.join(ROOT, "tests", "golden", "config1.json")))


def test_config1_graph_matches_the_reference_run(tmp_path):
    build.build_host()
    reads = readgen.generate_pairs(**FX["spec"])
    fa = tmp_path / "reads.fasta"
    fa.write_text("".join(f">p{i // 2 + 1}/{i % 2 + 1}\n{r}\n" for i, r in enumerate(reads)))
    assert hashlib.sha256(fa.read_bytes()).hexdigest() == FX["reads_sha256"]
    cfg = tmp_path / "disco.cfg"
    cfg.write_text(f"MinOverlap4BuildGraph = {FX['min_overlap']}\n")
    os.makedirs(tmp_path / "graph")
    prefix = str(tmp_path / "graph" / "disco")
    p = subprocess.run([os.path.join(ROOT, "disco_amd", "bin", "buildG"), "-pe", str(fa), "-f", prefix, "-p", str(cfg), "-t", str(FX["threads"]), "-m", "8"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
    for t in range(FX["threads"]):  # the file lists runDisco.sh builds (runDisco.sh:155-170)
        assert os.path.exists(f"{prefix}_{t}_parGraph.txt") and os.path.exists(f"{prefix}_{t}_containedReads.txt")
    e = refrun.parse_pargraph(sorted(glob.glob(prefix + "_*_parGraph.txt")))
    c = refrun.parse_contained(sorted(glob.glob(prefix + "_*_containedReads.txt")))
    want = FX["graph_reference_t1"]
    assert (len(e), len(c)) == (want["n_edges"], want["n_contained"])
    assert pyoracle.digest(pyoracle.edges_text(e)) == want["edges_sha256"] == FX["graph_reference"]["edges_sha256"]
    assert pyoracle.digest(pyoracle.contained_text(c)) == want["contained_sha256"]
    assert FX["scaffolds_drop_in"] == FX["scaffolds_reference"] and FX["scaffolds_reference"]["sequences"] >= 1
