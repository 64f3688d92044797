"""CPU: the issue-roofline arithmetic of bench.py (round 6: the roof that binds the pass is vector issue, not HBM) on the committed
counter profile, and the stamping rule of profiles/probe_traffic.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_issue_peak_is_1024_simds_at_4_cycles_and_2400_mhz():
    assert bench.ISSUE_PEAK_GINST_S == 1024 / 4 * 2.4


def test_issue_phase_and_pass_on_known_counts():
    t = {"valu": 8.16e9, "salu": 3.69e9, "lds": 1.36e9, "lds_bank_conflict": 2.78e9}
    e = bench.issue_phase(t, 21.5, 50_000_000)
    assert abs(e["floor_ms"] - 8.16e9 / 614.4e9 * 1e3) < 1e-9 and abs(e["frac"] - e["floor_ms"] / 21.5) < 1e-12
    assert abs(e["lds_bank_conflict_cycles_per_lds_inst"] - 2.78 / 1.36) < 1e-9 and abs(e["valu_insts_per_read"] - 163.2) < 1e-9
    assert bench.issue_phase({}, 21.5, 50_000_000) is None and bench.issue_phase(t, 0.0, 50_000_000) is None
    phases = {ph: bench.issue_phase(t, 20.0, 50_000_000) for ph in ("index", "probe_kernel", "verify", "select", "trmark")}
    p = bench.issue_pass(phases, 100.0, 50_000_000, "FETCH_SIZE + WRITE_SIZE of x")
    assert abs(p["valu_insts_per_step"] - 5 * 8.16e9) < 1 and abs(p["frac"] - 5 * 8.16e9 / 614.4e9 * 1e3 / 100.0) < 1e-12
    assert set(p["per_phase"]) == set(phases) and p["source"].startswith("SQ_INSTS_VALU")
    phases["verify"] = None  # a phase without counters: no pass figure
    assert bench.issue_pass(phases, 100.0, 50_000_000, "n")["frac"] is None


def test_committed_counter_profile_carries_the_instruction_counts():
    """profiles/probe_traffic.json (profiles/make_traffic.py) is what bench.py quotes: every long phase has its instruction counts"""
    tj = json.load(open(os.path.join(ROOT, "profiles", "probe_traffic.json")))
    assert len(tj["kernels_sha16"]) == 16 and tj["reads"] == 50_000_000
    for ph in ("index", "probe_kernel", "verify", "select", "trmark"):
        t = tj["phases"][ph]
        assert t["valu"] > 1e9 and t["salu"] > 1e8 and t["lo"] > 0, ph
        e = bench.issue_phase(t, 20.0, tj["reads"])
        assert 0 < e["frac"] < 1.5
