"""tests/operator_flow.py::reference_loop on FRESH data: the committed fixture generator (oracle/make_golden.py: the unmodified
reference with recording wrappers around its CorrPts / SimpleICPOptimization methods) is pointed at seeded random cloud pairs,
and the reference's loop is then replayed with the mirror classes (on the oracle-backed stand-in context) against those
per-stage dumps -- picks, bit-equal distances, rows after each rejection, estimate -- exactly as the GPU tests do against the
committed fixtures.  CPU only; skipped where /root/reference does not exist."""
import sys
from pathlib import Path

import numpy as np
import pytest

import operator_flow as flow
import oracle_backend
from conftest import ROOT
from live_data import cloud_pair as _pair

pytestmark = pytest.mark.skipif(not Path("/root/reference/python/simpleicp").exists(),
                                reason="the reference package is not on this machine")


@pytest.mark.parametrize("seed,kwargs", [
    (0, {}),
    (1, {"max_overlap_distance": 0.6, "correspondences": 400}),
    (3, {"rbp_observed_values": (0.0, 0.0, 1.0, 0.05, 0.0, 0.0), "rbp_observation_weights": (np.inf, 0.0, 20.0, 100.0, 0.0, 0.0)}),
    (5, {"correspondences": 2500, "neighbors": 8, "distance_weights": None, "max_iterations": 4}),
])
def test_reference_loop_replayed_on_fresh_data(seed, kwargs, monkeypatch, tmp_path):
    if str(ROOT / "oracle") not in sys.path:
        sys.path.insert(0, str(ROOT / "oracle"))
    import make_golden as mg                                  # (imports the unmodified reference, lmfit stand-in on the path)
    P, M = _pair(seed)
    data = {"fix.xyz": P, "mov.xyz": M}
    monkeypatch.setattr(mg, "GOLD", tmp_path)
    monkeypatch.setattr(mg, "store_cloud", lambda name: data[name])
    mg.run_case("live", "fix.xyz", "mov.xyz", kwargs)
    g = np.load(tmp_path / "live.npz", allow_pickle=False)
    assert int(g["iterations"]) >= 2
    monkeypatch.setattr(flow, "load_golden", lambda name: (g, ["fix.xyz", "mov.xyz"], dict(kwargs)))
    oracle_backend.install(monkeypatch)
    flow.reference_loop("live", lambda f: data[Path(f).name])
