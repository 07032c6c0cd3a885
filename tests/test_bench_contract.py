"""bench.py's host-side pieces and the committed records, without a GPU:
  * every `profiles/r5/bench_*.json` (the unabridged records of the round's measurements) honours the line's contract: the fields the
    driver reads, a roofline whose numbers follow from each other (achieved = algorithmic bytes / the kernel's launch time, frac =
    achieved / peak <= 1), a CPU baseline that says what it is, parity from the same run;
  * the compact stdout form keeps every contract field and number and fits a log tail;
  * `latency_model`: the floor is the sum of its three terms, priced as documented;
  * `match_bytes`: the tallies' price list;
  * the terrestrial stand-in generator: deterministic, the density really spans orders of magnitude, H_true is rigid and maps the
    movable scan onto the fixed one's surfaces;
  * `profiles/latest_pmc.json` names the kernel sources it was measured on.
"""
import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
RECORDS = sorted((ROOT / "profiles" / "r6").glob("bench_*.json"))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


@pytest.mark.parametrize("path", RECORDS, ids=[p.stem for p in RECORDS])
def test_committed_record_honours_the_contract(path):
    d = json.loads(path.read_text())
    for k in CONTRACT:
        assert k in d, k
    assert d["unit"] == "iterations/s" and d["higher_is_better"] is True and d["scaling"] in ("weak", "strong")
    assert d["vs_baseline"] is None                       # BASELINE.md publishes no number for this metric on this hardware
    assert d["dtype"] == "f64" and isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 2e-5          # iterations/s = 1 / (seconds per step); the compact lines (bench_under_rocprof*) carry 6 digits
    assert d["n_gpus"] >= 1 and d["steps"] >= 1 and d["warmup"] >= 0
    for name in ("roofline", "roofline_match"):
        r = d.get(name)
        if r is None:
            continue
        assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
        if r["frac"] is None:                         # (the kernel-trace run's by-product line: nothing to price the pruned search on)
            assert "under_rocprof" in path.stem and r["achieved"] is None and r.get("bytes_alg_source", "none").startswith("none")
            continue
        assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-5 * r["frac"] and 0 < r["frac"] <= 1
        # achieved = algorithmic bytes per launch / the kernel's average launch time (GB/s = bytes / ms / 1e6)
        assert abs(r["achieved"] - r["bytes_alg_per_launch"] / r["avg_ms"] / 1e6) <= 1e-4 * r["achieved"]          # (the byte count is stored rounded)
        assert r["traffic"] is None or r["traffic"] > 0
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "iterations/s" and c["sample"]
    if "parity" in d:
        assert d["parity"]["ok"] is True
    for k, t in d.items():
        if k.startswith("throughput_point"):
            assert t["parity"]["ok"] is True and 0 < t["roofline"]["frac"] <= 1
            assert abs(t["correspondences_per_s"] - t["correspondences"] * t["iterations_per_s"]) <= 1e-6 * t["correspondences_per_s"]


def test_final_headline_record_quotes_counter_traffic_of_its_own_tree():
    d = json.loads((ROOT / "profiles" / "r6" / "bench_C4.json").read_text())
    pmc = json.loads((ROOT / "profiles" / "latest_pmc.json").read_text())
    assert d["roofline"]["kernel"] == "k_icp_tail" and d["roofline"]["traffic"] == pytest.approx(pmc["k_icp_tail"], rel=1e-3)
    assert d["roofline_match"]["traffic"] == pytest.approx(pmc["k_grid_nn"], rel=1e-3)
    # traffic at or above the algorithmic bytes of the dominant kernel, and not wildly so (re-reads would show here first)
    assert 1.0 <= d["roofline"]["traffic"] / d["roofline"]["bytes_alg_per_launch"] < 1.5
    assert len(pmc["_csrc_hash"]) == 16 and "FETCH_SIZE" in pmc["_note"]


def test_compact_line_keeps_the_contract_and_fits_a_log_tail():
    import bench
    out = json.loads((ROOT / "profiles" / "r6" / "bench_C4.json").read_text())
    line = bench.compact_line(out)
    text = json.dumps(line)
    assert len(text) < 8192 and "\n" not in text
    for k in CONTRACT + ("cpu_baseline", "latency_model", "parity", "steady_us_per_step", "scaling_relevant"):
        assert k in line, k
    assert line["value"] == pytest.approx(out["value"], rel=1e-5) and line["ms_per_step"] == pytest.approx(out["ms_per_step"], rel=1e-5)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"]
        if isinstance(out["roofline"][k], float):
            assert line["roofline"][k] == pytest.approx(out["roofline"][k], rel=1e-5)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"]
    for k in ("throughput_point", "throughput_point_q1000000"):
        assert line[k]["parity"]["ok"] is True and line[k]["roofline"]["frac"] == pytest.approx(out[k]["roofline"]["frac"], rel=1e-5)


def test_latency_model_is_the_sum_of_its_terms():
    import bench
    cyc = {"load": 4700.0, "select": 9600.0, "keep": 1200.0, "lm": 7800.0, "final": 2400.0}
    m = bench.latency_model(21.0, 25.5, cyc)
    t = m["terms_us"]
    assert t["kernel_boundaries"] == pytest.approx(2 * 1.45) and t["global_round_trips"] == pytest.approx(5 * 0.375)
    assert t["tail_on_chip"] == pytest.approx((9600 + 1200 + 7800 + 2400) / 2400.0)            # the load phase IS one of the round trips
    assert m["floor_us"] == pytest.approx(sum(t.values())) and m["frac_of_floor"] == pytest.approx(m["floor_us"] / 21.0)
    assert m["floor_us"] < 21.0 and m["from_cold_us_per_step"] == 25.5
    assert bench.latency_model(0.0, 1.0, cyc)["frac_of_floor"] is None


def test_match_bytes_price_list():
    import bench
    per = {"candidates": 1000, "rows": 100}
    assert bench.match_bytes("k_grid_nn", per, 10) == 1000 * 32 + 100 * 8 + 10 * 96
    assert bench.match_bytes("k_grid_nn16", per, 10) == bench.match_bytes("k_grid_nn", per, 10)
    assert bench.match_bytes("k_grid_nn16f", per, 10) == 1000 * 16 + 100 * 8 + 10 * 168
    assert set(bench.GRID_KERNELS) == {"k_grid_nn", "k_grid_nn16", "k_grid_nn16f"}


def test_terrestrial_stand_in_generator():
    import bench
    from scipy.spatial import cKDTree
    n = 60_000
    Xf, Xm, H = bench.terrestrial_pair(n)
    Xf2, Xm2, H2 = bench.terrestrial_pair(n)
    assert np.array_equal(Xf, Xf2) and np.array_equal(Xm, Xm2) and np.array_equal(H, H2)           # pinned seeds
    assert Xf.shape == (n, 3) and Xm.shape == (n, 3) and Xf.flags.c_contiguous
    r = np.linalg.norm(Xf, axis=1)
    assert 2.0 - 0.05 <= r.min() and r.max() <= 80.0 + 0.05                                         # the scanner's range gate
    # uniform angular sampling: the density on the surfaces falls by orders of magnitude between the scanner's feet and the far walls
    tree = cKDTree(Xf)
    near = Xf[np.argsort(r)[:200]]
    far = Xf[np.argsort(r)[-200:]]
    c_near = np.mean([len(x) for x in tree.query_ball_point(near, 0.5)])
    c_far = np.mean([len(x) for x in tree.query_ball_point(far, 0.5)])
    assert c_near > 100 * c_far
    # H_true is rigid and puts the movable scan onto the fixed one's surfaces: the same ground plane, the same walls
    R, t = H[:3, :3], H[:3, 3]
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and np.linalg.det(R) == pytest.approx(1.0) and np.array_equal(H[3], [0, 0, 0, 1])
    moved = Xm @ R.T + t
    d, _ = tree.query(moved[::20])
    assert np.median(d) < 0.25                    # (two samplings of one scene: sampling distance, not a residual offset)
    d0, _ = tree.query(Xm[::20])
    assert np.median(d0) > np.median(d)           # ... and without H_true they are visibly apart
