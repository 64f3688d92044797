#!/usr/bin/env python3
"""BASELINE config 1 — the plumbing run through the reference's own shell driver (runDisco.sh:194-245), end to end.

The reference does not ship test/Ecoli_250_500_test.fna; the stand-in is 20 000 interleaved pairs of 250-500 bp
reads from one 0.6 Mbp uniform-random genome (25x; at the 3x of a 4.6 Mbp genome the reference pipeline prints no scaffold at all) (disco_amd.readgen.generate_pairs, seed 42), MinOverlap4BuildGraph = 30 (disco.cfg:9).

  gpu  DIR   (MI355X box)       : the graph half, exactly the command runDisco.sh issues for -inP (runDisco.sh:200):
                                  disco_amd/bin/buildG -pe reads.fasta -f DIR/out/graph/disco -p disco.cfg -t 4 -m 8
  ref  DIR   (build container)  : runDisco.sh + disco*.cfg copied from /root/reference next to the REAL buildG / fullsimplify /
                                  parsimplify (oracle/_ref, built by oracle/Makefile), whole pipeline to scaffolds
  ours DIR GRAPHDIR (container) : the same directory layout with the drop-in's graph files put where buildG writes them, then
                                  runDisco.sh -osg: the reference's fullsimplify / parsimplify consume the drop-in's files
  ours-simple DIR GRAPHDIR      : as `ours`, with the drop-in's <out>/assembly/disco_<i>_ParSimpleEdges.txt in place too (buildG under
                                  DISCO_PAR_SIMPLE=1): fullsimplify loads them and never starts parsimplify
  ours-binary DIR GRAPHDIR      : as `ours`, with the graph as the BINARY side output only (<prefix>_edges.bin / _contained.bin,
                                  the text files emptied — the state `buildG --no-text` leaves) and the reference's fullsimplify /
                                  parsimplify with the loader patch of oracle/patches spliced in (oracle/Makefile: ref_binary_loaders)
  fixture REFDIR OURSDIR [OURSSIMPLEDIR [OURSBINARYDIR]] : canonical digests of both graphs + scaffold statistics of both runs -> tests/golden/config1.json
"""
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SPEC = dict(seed=42, n_pairs=20000, genome_len=600_000, len_min=250, len_max=500)
MIN_OVERLAP = 30
THREADS = 4


def write_reads(path):
    from disco_amd import readgen

    reads = readgen.generate_pairs(**SPEC)
    with open(path, "w") as f:
        for i, r in enumerate(reads):
            f.write(f">p{i // 2 + 1}/{i % 2 + 1}\n{r}\n")
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def seq_stats(fasta):
    lens, cur = [], 0
    if os.path.exists(fasta):
        for line in open(fasta):
            if line.startswith(">"):
                if cur:
                    lens.append(cur)
                cur = 0
            else:
                cur += len(line.strip())
        if cur:
            lens.append(cur)
    lens.sort(reverse=True)
    tot, acc, n50 = sum(lens), 0, 0
    for x in lens:
        acc += x
        if acc * 2 >= tot:
            n50 = x
            break
    return {"sequences": len(lens), "total_bp": tot, "n50": n50, "longest": lens[0] if lens else 0}


def graph_digests(prefix):
    from oracle import pyoracle, refrun

    e = refrun.parse_pargraph(sorted(glob.glob(prefix + "_*_parGraph.txt")))
    c = refrun.parse_contained(sorted(glob.glob(prefix + "_*_containedReads.txt")))
    return {"n_edges": int(len(e)), "n_contained": int(len(c)), "edges_sha256": pyoracle.digest(pyoracle.edges_text(e)),
            "contained_sha256": pyoracle.digest(pyoracle.contained_text(c))}


def stage_dir(d, build_g, suffix=""):
    """runDisco.sh looks for buildG / fullsimplify / parsimplify and disco*.cfg beside itself (runDisco.sh:3-7,16-18,142-150)"""
    b = os.path.join(d, "bin")
    os.makedirs(b, exist_ok=True)
    for f in ("runDisco.sh", "disco.cfg", "disco_2.cfg", "disco_3.cfg"):
        shutil.copy(os.path.join("/root/reference", f), b)  # at run time, in the build container only; nothing is committed
    os.chmod(os.path.join(b, "runDisco.sh"), 0o755)
    ref = os.path.join(ROOT, "oracle", "_ref")
    shutil.copy(build_g, os.path.join(b, "buildG"))
    shutil.copy(os.path.join(ref, "fullsimplify_ref" + suffix), os.path.join(b, "fullsimplify"))
    shutil.copy(os.path.join(ref, "parsimplify_ref" + suffix), os.path.join(b, "parsimplify"))
    return b


def main():
    cmd = sys.argv[1]
    if cmd == "gpu":
        d = os.path.abspath(sys.argv[2])
        os.makedirs(os.path.join(d, "out", "graph"), exist_ok=True)
        sha = write_reads(os.path.join(d, "reads.fasta"))
        open(os.path.join(d, "disco.cfg"), "w").write(f"MinOverlap4BuildGraph = {MIN_OVERLAP}\n")
        log = os.path.join(d, "out", "disco.log")
        # DISCO_PAR_SIMPLE=1: buildG also leaves <out>/assembly/disco_<i>_ParSimpleEdges.txt — fullsimplify then skips its parsimplify
        # step (SG/OverlapGraph.cpp:1027-1049); the `ours-simple` run below checks that the pipeline ends in the same scaffold
        rc = subprocess.call([os.path.join(ROOT, "disco_amd", "bin", "buildG"), "-pe", os.path.join(d, "reads.fasta"), "-f", os.path.join(d, "out", "graph", "disco"),
                              "-p", os.path.join(d, "disco.cfg"), "-t", str(THREADS), "-m", "8"], stdout=open(log, "w"), stderr=subprocess.STDOUT,
                             env=dict(os.environ, DISCO_PAR_SIMPLE="1"))
        print("buildG rc", rc, "reads sha256", sha, graph_digests(os.path.join(d, "out", "graph", "disco")))
        os.remove(os.path.join(d, "reads.fasta"))  # regenerated where it is needed
        sys.exit(rc)
    if cmd == "ref":
        d = os.path.abspath(sys.argv[2])
        os.makedirs(d, exist_ok=True)
        write_reads(os.path.join(d, "reads.fasta"))
        b = stage_dir(d, os.path.join(ROOT, "oracle", "_ref", "buildG_ref"))
        rc = subprocess.call([os.path.join(b, "runDisco.sh"), "-inP", os.path.join(d, "reads.fasta"), "-d", os.path.join(d, "out"), "-n", str(THREADS), "-m", "8"], cwd=d)
        print("runDisco.sh rc", rc, seq_stats(os.path.join(d, "out", "disco_scaffoldsFinalCombined.fasta")))
        sys.exit(rc)
    if cmd in ("ours", "ours-simple", "ours-binary"):
        d, graph = os.path.abspath(sys.argv[2]), os.path.abspath(sys.argv[3])
        os.makedirs(os.path.join(d, "out", "graph"), exist_ok=True)
        write_reads(os.path.join(d, "reads.fasta"))
        for f in glob.glob(os.path.join(graph, "disco_*")):
            shutil.copy(f, os.path.join(d, "out", "graph"))
        if cmd == "ours-simple":  # the partial simplification the drop-in did on the resident graph, where fullsimplify looks for it
            os.makedirs(os.path.join(d, "out", "assembly"), exist_ok=True)
            n = 0
            for f in glob.glob(os.path.join(os.path.dirname(graph), "assembly", "disco_*_ParSimpleEdges.txt")):
                shutil.copy(f, os.path.join(d, "out", "assembly"))
                n += 1
            assert n == THREADS, n
        if cmd == "ours-binary":  # binary records instead of text: the loaders must take their binary path for every file
            from disco_amd import edgefile

            ne, nc = edgefile.from_text(os.path.join(d, "out", "graph", "disco"), THREADS, empty_text=True)
            print("binary side output:", ne, "edge records,", nc, "contained rows; text files emptied")
        b = stage_dir(d, os.path.join(ROOT, "disco_amd", "bin", "buildG"), "_bin" if cmd == "ours-binary" else "")  # buildG: present beside the script, not run (-osg)
        rc = subprocess.call([os.path.join(b, "runDisco.sh"), "-inP", os.path.join(d, "reads.fasta"), "-d", os.path.join(d, "out"), "-n", str(THREADS), "-m", "8", "-osg"], cwd=d)
        log = open(os.path.join(d, "out", "disco.log")).read()
        print("runDisco.sh -osg rc", rc, seq_stats(os.path.join(d, "out", "disco_scaffoldsFinalCombined.fasta")),
              "| parsimplify step skipped:", "Partial graphs already exist" in log,
              "| binary loader used:", log.count("edges loaded to memory from"), "edge files,", log.count("_contained.bin"), "contained files")
        sys.exit(rc)
    if cmd == "fixture":
        refd, oursd = os.path.abspath(sys.argv[2]), os.path.abspath(sys.argv[3])
        g_ref = graph_digests(os.path.join(refd, "out", "graph", "disco"))
        g_ours = graph_digests(os.path.join(oursd, "out", "graph", "disco"))
        # the reference's containment pass races between its threads on reads of mixed length: with -n 4 a few
        # contained reads name another of their containing reads; -t 1 is its deterministic form
        t1 = os.path.join(refd, "t1")
        os.makedirs(os.path.join(t1, "g"), exist_ok=True)
        open(os.path.join(t1, "disco.cfg"), "w").write(f"MinOverlap4BuildGraph = {MIN_OVERLAP}\n")
        subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "buildG_ref"), "-pe", os.path.join(refd, "reads.fasta"), "-f", os.path.join(t1, "g", "disco"),
                               "-p", os.path.join(t1, "disco.cfg"), "-t", "1", "-m", "8"], stdout=open(os.path.join(t1, "log"), "w"), stderr=subprocess.STDOUT)
        g_ref1 = graph_digests(os.path.join(t1, "g", "disco"))
        fx = {"spec": SPEC, "min_overlap": MIN_OVERLAP, "threads": THREADS, "graph_reference_t1": g_ref1,
              "edges_identical": g_ref["edges_sha256"] == g_ours["edges_sha256"] == g_ref1["edges_sha256"],
              "contained_identical_to_reference_t1": g_ref1 == g_ours,
              "reads_sha256": hashlib.sha256(open(os.path.join(refd, "reads.fasta"), "rb").read()).hexdigest(),
              "graph_reference": g_ref, "graph_drop_in": g_ours,
              "scaffolds_reference": seq_stats(os.path.join(refd, "out", "disco_scaffoldsFinalCombined.fasta")),
              "scaffolds_drop_in": seq_stats(os.path.join(oursd, "out", "disco_scaffoldsFinalCombined.fasta")),
              **({"scaffolds_drop_in_with_its_own_partial_simplification": seq_stats(os.path.join(os.path.abspath(sys.argv[4]), "out", "disco_scaffoldsFinalCombined.fasta")),
                  "scaffold_sequences_identical": open(os.path.join(os.path.abspath(sys.argv[4]), "out", "disco_scaffoldsFinalCombined.fasta")).read().split("\n", 1)[1:] ==
                  open(os.path.join(refd, "out", "disco_scaffoldsFinalCombined.fasta")).read().split("\n", 1)[1:]} if len(sys.argv) > 4 else {}),
              **({"scaffolds_from_binary_side_output_through_patched_reference_loaders": seq_stats(os.path.join(os.path.abspath(sys.argv[5]), "out", "disco_scaffoldsFinalCombined.fasta")),
                  "scaffold_sequences_identical_from_binary_side_output":
                  open(os.path.join(os.path.abspath(sys.argv[5]), "out", "disco_scaffoldsFinalCombined.fasta")).read().split("\n", 1)[1:] ==
                  open(os.path.join(refd, "out", "disco_scaffoldsFinalCombined.fasta")).read().split("\n", 1)[1:]} if len(sys.argv) > 5 else {}),
              "how": "tools/run_config1.py: ref = runDisco.sh with the real buildG / fullsimplify / parsimplify; drop-in = disco_amd/bin/buildG on the MI355X "
                     "(the command runDisco.sh:200 issues) + runDisco.sh -osg with the real fullsimplify / parsimplify on its files"}
        json.dump(fx, open(os.path.join(ROOT, "tests", "golden", "config1.json"), "w"), indent=1, sort_keys=True)
        print(json.dumps(fx, indent=1))
        return
    raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
