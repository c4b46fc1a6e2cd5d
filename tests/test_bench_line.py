"""bench.py's derived blocks (no GPU): the per-stage rooflines put together from a step's phase timers, its unit counts and the stamped stage profile."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITS = {"bases": 4907148927, "seed_positions": 39274360, "markers": 4895178, "candidate_pairs": 9500, "enumerated_positions": 371913175, "anchors": 117380842,
         "listed_query_positions": 371913175, "chunks": 2354774, "candidate_intervals": 2944813, "accepted_intervals": 2944813}
TM = {"seed_ms": 33.6, "sketch_build_ms": 15.3, "screen_ms": 2.8, "chain_ms": 67.5}       # ten steps


def test_stage_rooflines_without_a_profile():
    rows = bench.stage_rooflines(TM, 10, UNITS, 125, 1000, 9500, None)
    assert [r["stage"] for r in rows] == ["seeding", "tables", "screen", "join", "chunking + DP", "selection + estimate"]
    seeding = rows[0]
    assert abs(seeding["algorithmic_bytes"] - 0.354 * UNITS["bases"]) < 1e6 and abs(seeding["ms_live"] - 3.36) < 1e-9
    assert abs(seeding["achieved"] - seeding["algorithmic_bytes"] / 3.36e-3 / 1e9) < 1e-6 and abs(seeding["frac"] - seeding["achieved"] / 8000.0) < 1e-12
    assert rows[2]["algorithmic_bytes"] == 8.0 * UNITS["markers"]                         # 8 B per marker incidence
    assert all(r["ms_live"] is None and r["achieved"] is None and r["counter_bytes"] is None for r in rows[3:])   # the chaining's parts have no timer of their own
    assert bench.stage_rooflines(TM, 10, None, 125, 1000, 9500, None) is None and bench.stage_rooflines(TM, 10, {"error": "x"}, 125, 1000, 9500, None) is None


def test_stage_rooflines_with_the_committed_profile():
    sp = json.load(open(os.path.join(ROOT, "profiles", "stage_profile.json")))
    assert len(sp["kernel_sources_sha256_16"]) == 16 and any("join_count_kernel" in k["kernel"] for k in sp["kernels"])
    rows = bench.stage_rooflines(TM, 10, UNITS, 125, 1000, 9500, sp)
    by = {r["stage"]: r for r in rows}
    parts = [by[s] for s in ("join", "chunking + DP", "selection + estimate")]
    assert all(p["ms_live_apportioned"] and p["ms_live"] > 0 for p in parts) and abs(sum(p["ms_live"] for p in parts) - 6.75) < 1e-6   # the one chaining timer, dealt out
    for r in rows:
        assert r["ms_profile"] > 0 and r["counter_bytes"] > 0 and 0 < r["frac"] < 1 and r["kernels"] and r["kernels"][0]["ms_per_step"] >= r["kernels"][-1]["ms_per_step"]
    assert by["join"]["kernels"][0]["kernel"].startswith("join_count_kernel") and 3 < by["join"]["kernels"][0]["waves_per_simd"] < 6


def test_the_profile_is_refused_for_other_sources_or_workloads(monkeypatch):
    sp, note = bench.load_stage_profile(False)
    assert sp is None and "default workload" in note
    monkeypatch.setattr(bench, "kernel_sources_sha", lambda: "0" * 16)
    sp, note = bench.load_stage_profile(True)
    assert sp is None and "other kernel sources" in note
