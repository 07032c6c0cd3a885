"""Runs inside a process with the AddressSanitizer runtime preloaded and SICP_LIBRARY pointing at the
sanitizer build (tests/test_asan.py).  Exercises the host translation units: the .xyz reader/writer on ordinary, ragged
and extreme inputs, the ABI's argument checks and error texts -- and, with `--gpu`, one whole ICP run (uploads, grid
build, normals, chained iterations, state download), so the host side of the device path runs under the sanitizer too."""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from simpleicp_amd import _lib  # noqa: E402
from simpleicp_amd import io as sio  # noqa: E402


def io_round_trips(tmp):
    rng = np.random.default_rng(5)
    for n, threads in ((0, 1), (1, 1), (7, 3), (10_000, 8), (200_001, 16)):
        X = np.round(rng.normal(0, 1e5, (n, 3)), 3)
        f = tmp / f"c{n}.xyz"
        sio.write_xyz(f, X, decimals=3, threads=threads)
        Y = sio.read_xyz(f, threads=threads)
        assert Y.shape == (n, 3) and np.array_equal(X, Y), (n, threads)
    # ragged / extreme text: long lines, exponents, nan/inf rows, blank lines, more than three columns, CRLF
    f = tmp / "odd.xyz"
    f.write_text("1 2 3\n\n   4.5e300\t-1e-320   7  99 100\r\n" + "nan inf -inf\n" + "0." + "1" * 700 + " 2 3\n" + "9 8 7")
    Y = sio.read_xyz(f, threads=4)
    assert Y.shape == (5, 3) and Y[1, 0] == 4.5e300 and np.isnan(Y[2, 0]) and Y[4, 2] == 7.0
    sio.write_xyz(tmp / "wide.xyz", np.array([[1e308, -1e308, 5e-324]]), decimals=60, threads=2)
    for bad in ("1 2\n", "1 2 3\n4 5\n"):
        g = tmp / "bad.xyz"
        g.write_text(bad)
        try:
            sio.read_xyz(g)
        except Exception as exc:  # noqa: BLE001
            assert "column" in str(exc) or "numeric" in str(exc), exc
        else:
            raise AssertionError("malformed file accepted")
    try:
        sio.read_xyz(tmp / "does_not_exist.xyz")
    except Exception:  # noqa: BLE001
        pass
    else:
        raise AssertionError("missing file accepted")


def abi_errors():
    L = _lib.load()
    import ctypes as C
    assert L.sicp_abi_version() == 7
    assert L.sicp_ctx_create(0, None) != 0 and b"null" in L.sicp_last_error()
    assert L.sicp_cloud_upload(None, 0, None, 0, 0) != 0
    assert L.sicp_icp_iterate(None, None, None) != 0
    assert L.sicp_ctx_destroy(None) == 0
    H = np.empty(16)
    x = np.array([0.1, -0.2, 0.3, 1.0, 2.0, 3.0])
    assert L.sicp_params_to_H(x.ctypes.data_as(C.c_void_p), H.ctypes.data_as(C.c_void_p)) == 0
    assert abs(np.linalg.det(H.reshape(4, 4)[:3, :3]) - 1) < 1e-14
    # every export with a NULL context / NULL pointers: a code and a message, never a wild access
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "helpers"))
    import null_abi_probe
    for name, (rc, msg) in null_abi_probe.probe().items():
        assert (rc == 0 and name == "sicp_ctx_destroy") or (rc < 0 and msg), (name, rc, msg)
    n = C.c_int(-1)
    L.sicp_device_count(C.byref(n))
    return n.value


def gpu_run():
    from simpleicp_amd import PointCloud, SimpleICP
    rng = np.random.default_rng(1)
    n = 60_000
    xy = rng.uniform(-30, 30, (n, 2))
    X = np.column_stack((xy, 2 * np.sin(xy[:, 0] / 5) * np.cos(xy[:, 1] / 7)))
    c, s = np.cos(0.02), np.sin(0.02)
    Xm = X @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]]).T + [0.1, -0.05, 0.02] + rng.normal(0, 0.005, X.shape)
    for kw in ({}, {"correspondences": 5000, "max_overlap_distance": 2.0}):
        icp = SimpleICP()
        icp.add_point_clouds(PointCloud(X, columns=["x", "y", "z"]), PointCloud(Xm, columns=["x", "y", "z"]))
        H, Xt, rbp, res = icp.run(**kw)
        assert np.abs(Xt - X).max() < 0.1 and len(res) > 100


def gpu_operators():
    """the operator-by-operator road (sicp_corr_* / sicp_estimate_parameters behind the mirror classes) under the sanitizers"""
    from simpleicp_amd import PointCloud
    from simpleicp_amd.corrpts import CorrPts
    from simpleicp_amd.optimization import SimpleICPOptimization
    rng = np.random.default_rng(2)
    for n, q in ((30_000, 800), (90_000, 20_000)):               # single-workgroup and multi-workgroup selection
        xy = rng.uniform(-30, 30, (n, 2))
        X = np.column_stack((xy, 2 * np.sin(xy[:, 0] / 5) * np.cos(xy[:, 1] / 7)))
        pc_fix = PointCloud(X, columns=["x", "y", "z"])
        pc_mov = PointCloud(X + [0.05, -0.03, 0.02] + rng.normal(0, 0.004, X.shape), columns=["x", "y", "z"])
        pc_fix.select_n_points(q)
        pc_fix.estimate_normals(10)
        cp = CorrPts(pc_fix, pc_mov)
        cp.match()
        cp.reject_wrt_planarity(0.3)
        cp.reject_wrt_point_to_plane_distances()
        optim = SimpleICPOptimization(cp, None, (0.,) * 6, (0.,) * 6, (0., 0., np.inf, 0., 0., 1.0))
        res = optim.estimate_parameters()
        optim.estimate_parameter_uncertainties()
        x = np.array(optim.rbp.get_parameter_attributes_as_list("estimated_value"))
        assert len(res) == cp.num_corr_pts > 100 and x[2] == 0.0 and abs(x[3] + 0.05) < 0.01, x


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as d:
        io_round_trips(Path(d))
    devices = abi_errors()
    if "--gpu" in sys.argv:
        assert devices > 0
        gpu_run()
        gpu_operators()
    print("asan exercise OK", "(with device run)" if "--gpu" in sys.argv else "", flush=True)
    os._exit(0)     # skip interpreter teardown: numpy/pandas extension destructors are not what is under test
