"""CPU tests of bench.py's bookkeeping: the byte budget of the design from a run's counts, the spread of step times,
the hash that guards the committed counter profiles against edited kernels, and the roofline arithmetic on
synthetic counter files (no GPU: nothing here renders)."""
import json
import os
import types

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_bytes_follow_the_counts():
    frame = dict(paths=100.0, bounces=250.0, shadow_rays=40.0, nee_samples=60.0, stored_paths=30.0)
    fused = dict(per_frame=frame, carries_colour=False, recompute_primaries=True, fused=True, inline_light_samples=True)
    b = bench.design_bytes(fused)
    assert b["path_state"] == 30 * 2 * 48 and b["results"] == 100 * 24
    assert b["fresh_samples"] == b["hit_records"] == b["light_samples"] == 0
    assert b["total"] == 30 * 96 + 2400
    wide = dict(per_frame=frame, carries_colour=True, recompute_primaries=False, fused=False, inline_light_samples=False)
    w = bench.design_bytes(wide)
    assert w["path_state"] == 30 * 2 * 64 and w["fresh_samples"] == 100 * 64
    assert w["hit_records"] == 250 * 40 and w["light_samples"] == 40 * 128
    assert w["total"] == sum(v for k, v in w.items() if k != "total")


def test_spread_of_step_times():
    s = bench.spread([5.0, 7.0, 6.0, 100.0])
    assert s == {"min": 5.0, "median": 6.5, "mean": 29.5, "max": 100.0, "n": 4}
    assert bench.spread([]) is None


def test_profiles_are_stale_when_the_kernel_sources_change(tmp_path, monkeypatch):
    h = bench.csrc_sha256()
    assert len(h) == 64 and h == bench.csrc_sha256()
    manifest = json.load(open(os.path.join(ROOT, "profiles", "MANIFEST.json")))
    assert set(manifest) >= {"sample2_glass", "sample3_mis_cornell", "mesh_1m", "sample1_diffuse"}
    for w, e in manifest.items():
        for f in e["files"]:
            assert os.path.exists(os.path.join(ROOT, "profiles", f)), (w, f)
    # a workload without an entry, or with another hash, is stale
    assert bench.profiles_stale("no_such_workload")
    monkeypatch.setattr(bench, "csrc_sha256", lambda: "0" * 64)
    assert bench.profiles_stale("sample2_glass")


def test_roofline_names_the_binding_roof_and_no_fraction_exceeds_one(monkeypatch):
    """valu bound: frac = issue fraction x lanes / 64; hbm bound: traffic / duration / peak."""
    info = types.SimpleNamespace(bvh_width=2)
    base = dict(fused=True, info=info)
    # 1 ms launches; 1.2288e9 VALU instructions x 2 cycles = the chip's issue slots in 1 ms -> issue fraction 1.0
    monkeypatch.setattr(bench, "pmc_traffic", lambda w, p: (800_000_000, "x.json"))
    monkeypatch.setattr(bench, "sq_counters", lambda w, p: ({"SQ_INSTS_VALU": 0.5 * 1.2288e9, "SQ_THREAD_CYCLES_VALU": 32.0,
                                                            "SQ_ACTIVE_INST_VALU": 1.0}, "y.json"))
    monkeypatch.setattr(bench, "rocprof_average_ms", lambda w, p: (1.0, "z.csv"))
    monkeypatch.setattr(bench, "profiles_stale", lambda w: False)
    m = dict(base, stage={"intersect": (0.0, 0, 0), "shade": (10.0, 10, 1e7), "shadow": (0.0, 0, 0), "generate": (0.0, 0, 0)})
    r = bench.build_roofline("w", m)
    assert r["bound"] == "valu" and r["kernel"] == "k_trace_shade"
    assert r["valu"]["issue_frac"] == pytest.approx(0.5) and r["valu"]["lanes_active"] == pytest.approx(32.0)
    assert r["frac"] == pytest.approx(0.25) and r["unit"] == "Tlane-instr/s"
    assert r["hbm"]["frac"] == pytest.approx(0.1) and r["frac"] <= 1 and r["hbm"]["frac"] <= 1
    # the same counters with ten times the traffic: HBM binds
    monkeypatch.setattr(bench, "pmc_traffic", lambda w, p: (6_400_000_000, "x.json"))
    r = bench.build_roofline("w", m)
    assert r["bound"] == "hbm" and r["frac"] == pytest.approx(0.8) and r["unit"] == "GB/s"
    # no counter files: unmeasured, nothing invented
    monkeypatch.setattr(bench, "pmc_traffic", lambda w, p: (None, None))
    monkeypatch.setattr(bench, "sq_counters", lambda w, p: (None, None))
    r = bench.build_roofline("w", m)
    assert r["bound"] == "unmeasured" and r["frac"] is None and r["achieved"] is None
