"""BASELINE.json's third configuration by shape (1.34 M points per cloud, 10 000 correspondences, k = 10; the airborne files themselves
are missing upstream, `bench.py --config C3` is SURVEY 8(d)'s surface at that size): the mid-Q regime -- four queries per wave in
the match (k_grid_nn16), the one-workgroup rejection (k_reject), the many-workgroup minimisation (k_lm_all) -- held against the
oracle on EVERY correspondence (1.3e10 pairs per brute-force pass: seconds on the GPU box's host cores):
  * estimate_normals: neighbour lists bit for bit, normals / planarity to a float32 ulp;
  * three iterations: matched indices, distances, keep mask, median, MAD bit-exact, the estimate to 1e-9;
  * the chained loop lands on the same estimate.
"""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


@pytest.fixture(scope="module")
def c3():
    import bench
    from simpleicp_amd import _lib
    cfg = bench.CONFIGS["C3"]
    Xf, Xm, H_true = bench.synthetic_pair(cfg["n"])
    sel = np.unique(np.round(np.linspace(0, cfg["n"] - 1, cfg["Q"])).astype(np.int64))
    c = _lib.Context(0)
    c.upload(_lib.FIX, Xf)
    c.upload(_lib.MOV, Xm)
    yield c, Xf, Xm, sel, cfg["k"]
    c.close()


def test_normals_at_c3_shape_equal_oracle(c3):
    from simpleicp_amd import _lib
    from oracle import orc
    c, Xf, _, sel, k = c3
    nv, pl, nn = c.estimate_normals(_lib.FIX, sel, k, want_nn=True)
    onn, _ = orc.knn(Xf, Xf[sel], k=k)
    assert np.array_equal(nn, onn)
    onv, opl = orc.normals(Xf, onn)
    assert np.abs(nv - onv).max() <= 2e-7 and np.abs(pl - opl).max() <= 2e-6


def test_iterations_at_c3_shape_equal_oracle(c3):
    from simpleicp_amd import _lib
    from test_gpu_fullsize import check_large_q_iteration
    c, Xf, Xm, sel, k = c3
    nv, pl = c.estimate_normals(_lib.FIX, sel, k)
    z = np.zeros(6)
    c.icp_setup(sel, nv, pl)
    x = z.copy()
    for it in range(3):
        R = c.icp_iterate(x, z, z, 0.3, 1.0)
        assert c.last_match_kernel() == "k_grid_nn16"
        check_large_q_iteration(c, Xf, Xm, sel, nv, pl, x, R, len(sel))       # every query, not a sample
        x = np.array(R.x[:])
    c.icp_setup(sel, nv, pl)
    whole = c.icp_run(z, z, z, 0.3, 1.0, max_iterations=3, min_change=0.0)
    assert np.abs(np.array(whole[-1].x[:]) - x).max() < 1e-12
    assert whole[-1].n_kept == R.n_kept and abs(whole[-1].median - R.median) < 1e-12 and abs(whole[-1].mad - R.mad) < 1e-12
