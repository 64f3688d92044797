"""CPU: the C restatement (oracle/) against the golden vectors — the reference's own and those generated with the real
reference binary (tests/golden/make_golden.py). This is what pins the oracle."""
import os

import numpy as np
import pytest

from oracle import pyoracle, refrun
from tests import golden_util as gu


def test_reference_golden_vector():
    """src/BuildGraph/bench_test_0_parGraph.txt, the only golden file in the reference tree"""
    gold = open(os.path.join(gu.GOLD, "reference_data", "bench_test_0_parGraph.txt")).read()
    reads, fidx, total = pyoracle.load_good_reads([os.path.join(gu.GOLD, "reference_data", "10reads_containedReads.fasta")], 30)
    assert total == 17 and len(reads) == 16  # one 8 bp read is rejected
    ce, cc, cnt = pyoracle.oracle_canonical(reads, fidx, 30)
    want = "\n".join(line.rsplit(",", 1)[0] for line in gold.strip().split("\n")) + "\n"  # drop the batch flag
    assert pyoracle.edges_text(ce) == want
    assert pyoracle.contained_text(cc) == "11\t12\t3,35,0,0,35,0,35,42,0,35\n13\t4\t3,35,0,0,35,0,35,35,0,35\n14\t4\t3,35,0,0,35,0,35,35,0,35\n"
    assert cnt["e_pre"] == 23 and cnt["asymmetric_pairs"] == 0 and cnt["cap_bind_sites"] == 0


@pytest.mark.parametrize("name", [n for n in gu.BIT_EXACT_CASES if gu.CASES[n].get("spec", {}).get("n_reads", 0) <= 20000])
def test_oracle_matches_reference_fixture(name):
    reads, fidx, mo = gu.case_inputs(name)
    ce, cc, cnt = pyoracle.oracle_canonical(reads, fidx, mo)
    gu.check_against_golden(name, ce, cc)
    assert cnt["asymmetric_pairs"] == 0 and cnt["cap_bind_sites"] == 0  # inside the order-independent parity domain


def test_oracle_matches_reference_100k_digest():
    reads, fidx, mo = gu.case_inputs("u150_100k")
    ce, cc, _ = pyoracle.oracle_canonical(reads, fidx, mo)
    gu.check_against_golden("u150_100k", ce, cc)


def test_multifile_read_id_map_and_filter():
    c = gu.CASES["multifile"]
    counts = []
    for f in c["pe"] + c["se"]:
        reads, fidx, total = pyoracle.load_good_reads([os.path.join(gu.GOLD, f)], c["min_overlap"])
        counts.append(len(reads))
    assert counts == c["good_reads_per_file"]


def test_order_dependent_regime_is_reported():
    """40-copy-repeat style input: the reference is order dependent here (differs from itself between -t 1 and -t 8);
    the bulk form must flag it through the two counters instead of claiming parity."""
    reads, fidx, mo = gu.case_inputs("repeats_8k")
    ce, cc, cnt = pyoracle.oracle_canonical(reads, fidx, mo)
    c = gu.CASES["repeats_8k"]
    assert pyoracle.digest(pyoracle.contained_text(cc)) == c["contained_sha256"]  # containment is order independent
    assert cnt["asymmetric_pairs"] > 0 or cnt["cap_bind_sites"] > 0
    assert abs(len(ce) - c["n_edges"]) <= 0.05 * c["n_edges"]


def test_filter_rules():
    t = pyoracle.test_read
    good = "ACGTTGCAAGCTTAGCCGATCGGATTACAGCTAGCTAGGATCCGATTAGC"
    assert t(good)
    assert not t(good[:29])                                  # MIN_READ_SIZE 30
    assert not t(good.replace("G", "N", 1))                  # non-ACGT
    assert not t("A" * 36 + good[:14])                       # >= 70 % one base
    assert not t("ACACACACACACACACACACACACACACA" + good)      # micro-repeat prefix
    assert not t(good + "TTCTTCTTCTTCTTCTTCTTCTTCTTCTT")      # micro-repeat suffix
    assert not t("AT" * 30)                                  # dimer over >= 50 %
    assert not t("GGGGCC" * 5 + good[:20])                   # 6-mer motif over >= 50 %


def test_parse_records_fasta_fastq():
    fa = b">a desc\nACGT\nAC\n>b\n\n>c\nGGG"
    assert pyoracle.parse_records(fa) == [b"ACGTAC", b"", b"GGG"]
    fq = b"@q1\nACGT\n+\nIIII\n@q2\nGG\n+\nII\n"
    assert pyoracle.parse_records(fq) == [b"ACGT", b"GG"]
    with pytest.raises(ValueError):
        pyoracle.parse_records(b"ACGT\n")


@pytest.mark.ref
@pytest.mark.skipif(not refrun.available(), reason="reference binary not built (make -C oracle ref)")
def test_oracle_vs_live_reference_small(tmp_path):
    """live check against the real reference where it is available (build container / GPU box with the prebuilt binary)"""
    from disco_amd import readgen

    reads = readgen.generate_reads(readgen.GenSpec.coverage(seed=2024, n_reads=1500, read_len=90, cov=20.0, len_max=150))
    fa = str(tmp_path / "r.fasta")
    readgen.write_fasta(fa, reads)
    ref = refrun.run_reference([fa], 40, threads=1, workdir=str(tmp_path))
    ce, cc, _ = pyoracle.oracle_canonical(reads, np.arange(1, len(reads) + 1), 40)
    assert np.array_equal(ce, ref["edges"]) and np.array_equal(cc, ref["contained"])
