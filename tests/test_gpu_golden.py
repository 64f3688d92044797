"""-m gpu : the HIP path through the C-ABI against the committed outputs of the REAL reference (tests/golden)."""
import numpy as np
import pytest

from tests import golden_util as gu
from tests.util import canon_hip, run_hip_reads

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", gu.BIT_EXACT_CASES)
def test_hip_matches_reference_fixture(name):
    reads, fidx, mo = gu.case_inputs(name)
    edges, rows, cnt = run_hip_reads(reads, mo)
    ce, cc = canon_hip(edges, rows, fidx)
    gu.check_against_golden(name, ce, cc)
    assert cnt["asymmetric_pairs"] == 0 and cnt["cap_bind_sites"] == 0


def test_hip_order_dependent_regime_matches_oracle_and_reports():
    from oracle import pyoracle

    reads, fidx, mo = gu.case_inputs("repeats_8k")
    edges, rows, cnt = run_hip_reads(reads, mo)
    ce, cc = canon_hip(edges, rows, fidx)
    oce, occ, ocnt = pyoracle.oracle_canonical(reads, fidx, mo)
    assert np.array_equal(ce, oce) and np.array_equal(cc, occ)
    assert cnt["asymmetric_pairs"] == ocnt["asymmetric_pairs"] > 0 or cnt["cap_bind_sites"] == ocnt["cap_bind_sites"] > 0
