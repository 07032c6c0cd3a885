"""BASELINE.json's largest configuration on ONE GPU: 100 M-vs-100 M points, Q = 1 M (config 5; the 8-GPU form shards these
clouds).  Generation dominates the cost (the pinned SURVEY 8(d) generator, ~1 minute on the host), so one module fixture is
shared by three checks, each against the CPU oracle on a 300-query sample:

  * estimate_normals (pointcloud.py:173-203): k = 10 neighbour lists bit for bit, normals / planarity to 1 ulp(f32);
  * two ICP iterations at Q = 1 M (corrpts.py:124-188, optimization.py:65-124): the match on the sample (brute force over all
    1e8 movable points), distances / keep mask / median / MAD / minimiser on ALL correspondences;
  * select_in_range over all 1e8 fixed points (pointcloud.py:149-171): sampled verdicts and the strict bound.
"""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.slow]
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
N, Q, SAMPLE = 100_000_000, 1_000_000, 300


@pytest.fixture(scope="module")
def c5():
    import bench
    from simpleicp_amd import _lib
    Xf, Xm, H_true = bench.synthetic_pair(N)
    sel = np.unique(np.round(np.linspace(0, N - 1, Q)).astype(np.int64))
    pick = np.unique(np.round(np.linspace(0, len(sel) - 1, SAMPLE)).astype(np.int64))
    c = _lib.Context(0)
    c.upload(_lib.FIX, Xf)
    c.upload(_lib.MOV, Xm)
    yield c, Xf, Xm, H_true, sel, pick
    c.close()


@pytest.fixture(scope="module")
def c5_normals(c5):
    from simpleicp_amd import _lib
    c, Xf, Xm, H_true, sel, pick = c5
    c.timing_enable(True, count_work=True); c.timing_reset()
    nv, pl, nn = c.estimate_normals(_lib.FIX, sel, 10, want_nn=True)
    work = c.knn_work()
    c.timing_enable(False)
    return nv, pl, nn, work


def test_normals_at_c5_size(c5, c5_normals):
    from oracle import orc
    c, Xf, Xm, H_true, sel, pick = c5
    nv, pl, nn, work = c5_normals
    onn, _ = orc.knn(Xf, Xf[sel[pick]], k=10)
    assert np.array_equal(nn[pick], onn)
    assert np.array_equal(nn[:, 0], sel)                          # every query is its own nearest neighbour
    onv, opl = orc.normals(Xf, onn)
    assert np.abs(nv[pick] - onv).max() <= 2e-7 and np.abs(pl[pick] - opl).max() <= 2e-6
    assert np.isfinite(nv).all() and np.isfinite(pl).all()
    assert work["sweeps"] >= len(sel) and work["slow_queries"] == 0, work      # the one-sweep kernel, never its k-round path


def test_two_iterations_at_c5_size(c5, c5_normals):
    from test_gpu_fullsize import check_large_q_iteration
    c, Xf, Xm, H_true, sel, pick = c5
    nv, pl, _, _ = c5_normals
    z = np.zeros(6)
    c.icp_setup(sel, nv, pl)
    x = z.copy()
    for it in range(2):
        R = c.icp_iterate(x, z, z, 0.3, 1.0)
        assert c.last_match_kernel() == "k_grid_nn16f"
        check_large_q_iteration(c, Xf, Xm, sel, nv, pl, x, R, SAMPLE)
        x = np.array(R.x[:])


def test_select_in_range_at_c5_size(c5):
    from simpleicp_amd import _lib
    from oracle import orc
    c, Xf, Xm, H_true, sel, pick = c5
    rows = sel[pick]
    oidx, od2 = orc.knn(Xm, Xf[rows], k=1, H=H_true)
    dist = np.sqrt(od2[:, 0])
    bound = float(np.median(dist))
    near = c.select_in_range(_lib.FIX, _lib.MOV, None, H_true, bound)          # all 1e8 fixed points against all 1e8 movable ones
    assert near.shape == (N,) and 0 < near.sum() < N
    ridx, _ = orc.knn(Xm, Xf[rows], k=1, H=H_true, max_dist=bound)
    assert np.array_equal(near[rows], ridx[:, 0] >= 0)
    # the bound is strict (d2 < max_range^2, like cKDTree's distance_upper_bound), also at this size
    j = int(np.flatnonzero((dist * dist == od2[:, 0]) & (dist > 0))[0])
    assert not c.select_in_range(_lib.FIX, _lib.MOV, rows[j:j + 1], H_true, float(dist[j]))[0]
    assert c.select_in_range(_lib.FIX, _lib.MOV, rows[j:j + 1], H_true, float(np.nextafter(dist[j], np.inf)))[0]
