"""CPU-side checks: C-ABI surface, loud failure without a GPU, host mirror of the reference API."""
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

from conftest import ROOT, has_gpu


def _declared():
    text = (ROOT / "include" / "simpleicp_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sicp_[A-Za-z0-9_]+)\s*\(", text)) - {"sicp_exchange_fn"})


def test_library_exports_every_declared_symbol():
    """hipcc cross-compiles without a GPU: the .so must load here and export the whole header."""
    from simpleicp_amd import _lib
    L = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/simpleicp_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == names
    assert L.sicp_abi_version() == _lib.ABI_VERSION == 7


def test_params_to_H_matches_reference_convention():
    """mathutils.py:39-68,81-93 (host-only entry point, no device needed)."""
    from oracle import orc, ref_port
    from simpleicp_amd import _lib
    from simpleicp_amd.rbp import H_from_params
    rng = np.random.default_rng(0)
    for _ in range(20):
        x = np.concatenate((rng.uniform(-np.pi, np.pi, 3), rng.uniform(-10, 10, 3)))
        H = _lib.params_to_H(x)
        assert np.array_equal(H, orc.params_to_H(x))
        assert np.allclose(H, ref_port.params_to_H(x), rtol=0, atol=1e-15)
        assert np.allclose(H, H_from_params(x), rtol=0, atol=1e-15)
        assert np.allclose(H[:3, :3] @ H[:3, :3].T, np.eye(3), atol=1e-14)


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    """No silent CPU fallback: every compute entry point raises BackendError when no MI355X is visible."""
    from simpleicp_amd import PointCloud, SimpleICP, _lib, backend
    assert _lib.device_count() == 0
    with pytest.raises(_lib.BackendError) as e:
        _lib.Context(0)
    assert e.value.code == _lib.ERR_NO_DEVICE and "no CPU path" in str(e.value)
    backend.reset_context()
    X = np.random.default_rng(0).uniform(0, 1, (50, 3))
    pc = PointCloud(X, columns=["x", "y", "z"])
    for call in (lambda: pc.estimate_normals(5), lambda: pc.transform_by_H(np.eye(4)),
                 lambda: pc.select_in_range(X, 0.1)):
        with pytest.raises(_lib.BackendError):
            call()
    icp = SimpleICP(verbose=False)
    icp.add_point_clouds(pc, PointCloud(X.copy(), columns=["x", "y", "z"]))
    with pytest.raises(_lib.BackendError):
        icp.run()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under simpleicp_amd/ may reference it."""
    for f in (ROOT / "simpleicp_amd").rglob("*"):
        if f.suffix in (".py", ".cpp", ".hip", ".h"):
            t = f.read_text()
            assert "import oracle" not in t and "from oracle" not in t and "libsicp_oracle" not in t, f
            assert "orc_" not in t.replace("sicp_oracle.c:orc_normals", ""), f   # a comment cites the oracle twin
    code = "import sys; sys.path.insert(0, %r); import simpleicp_amd; assert not any(m.startswith('oracle') for m in sys.modules)" % str(ROOT)
    subprocess.run([sys.executable, "-c", code], check=True)


def test_pointcloud_mirror_of_reference_api():
    """pointcloud.py:15-147 behaviours that need no device."""
    from simpleicp_amd import PointCloud, PointCloudException
    X = np.arange(30, dtype=float).reshape(10, 3)
    pc = PointCloud(X, columns=["x", "y", "z"])
    assert isinstance(pc, pd.DataFrame) and pc.num_points == 10 and pc.num_selected_points == 10
    assert pc["selected"].dtype == bool
    assert np.array_equal(pc.X, X) and np.array_equal(pc.x, X[:, 0]) and np.array_equal(pc.z, X[:, 2])
    with pytest.raises(PointCloudException, match='Column "z" is missing'):
        PointCloud(X[:, :2], columns=["x", "y"])
    pc.select_n_points(4)                                   # round-half-even of linspace(0, 9, 4) = 0 3 6 9
    assert pc.idx_selected.tolist() == [0, 3, 6, 9]
    assert np.array_equal(pc.X_selected, X[[0, 3, 6, 9]]) and np.array_equal(pc.y_selected, X[[0, 3, 6, 9], 1])
    pc.select_by_indices([3, 4, 9])
    assert pc.idx_selected.tolist() == [3, 9]
    pc.select_n_points(5)                                   # fewer selected than n: untouched
    assert pc.idx_selected.tolist() == [3, 9]
    pc.unselect_all_points()
    assert pc.num_selected_points == 0
    pc.select_all_points()
    pc.idx_selected = [1, 2]
    assert pc.idx_selected.tolist() == [1, 2]
    # half-to-even rounding case: linspace(0, 4, 3) -> 0 2 4 ; linspace(0,5,3) -> 0 2.5->2 5
    pc2 = PointCloud(np.zeros((6, 3)), columns=["x", "y", "z"])
    pc2.select_n_points(3)
    assert pc2.idx_selected.tolist() == [0, 2, 5]


def test_select_n_points_matches_reference_on_golden(clouds):
    from conftest import load_golden
    from simpleicp_amd import PointCloud
    g, files, kw = load_golden("dragon")
    pc = PointCloud(clouds(files[0]), columns=["x", "y", "z"])
    pc.select_n_points(1000)
    assert np.array_equal(pc.idx_selected, g["sel_idx"])


def test_write_xyz_format(tmp_path):
    from simpleicp_amd import PointCloud
    pc = PointCloud(np.array([[1.23456, 2.0, -3.5], [0.0004, 0.0005, 10]]), columns=["x", "y", "z"])
    pc.write_xyz(tmp_path / "a.xyz")
    assert (tmp_path / "a.xyz").read_text().splitlines() == ["//X Y Z", "1.235 2.000 -3.500", "0.000 0.001 10.000"]


def test_rigid_body_parameters_mirror():
    """optimization.py:291-382."""
    from simpleicp_amd import RigidBodyParameters
    rbp = RigidBodyParameters()
    assert np.isnan(rbp.alpha1.estimated_value) and rbp.tx.scale_for_logging == 1
    rbp.set_parameter_attributes_from_list("estimated_value", [0.1, 0.2, 0.3, 1, 2, 3])
    assert rbp.get_parameter_attributes_as_list("estimated_value") == [0.1, 0.2, 0.3, 1, 2, 3]
    assert np.isclose(rbp.alpha3.estimated_value_scaled, np.degrees(0.3)) and rbp.tz.estimated_value_scaled == 3
    H = rbp.H
    assert H.shape == (4, 4) and np.array_equal(H[3], [0, 0, 0, 1]) and np.array_equal(H[:3, 3], [1, 2, 3])
    # R = Rx(a1) Ry(a2) Rz(a3) (README.md:96-100)
    def rx(a): return np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    def ry(a): return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    def rz(a): return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    assert np.allclose(H[:3, :3], rx(0.1) @ ry(0.2) @ rz(0.3), atol=1e-15)
    from simpleicp_amd.rbp import euler_from_rotation
    assert np.allclose(euler_from_rotation(H[:3, :3]), [0.1, 0.2, 0.3], atol=1e-15)


def test_argument_checks_need_no_gpu():
    """simpleicp.py:327-353: same exceptions, raised before any device work."""
    from simpleicp_amd import SimpleICP, SimpleICPException
    icp = SimpleICP(verbose=False)
    for kw, msg in [({"distance_weights": -1}, "distance_weights must be > 0."),
                    ({"rbp_observed_values": (0,) * 5}, "rbp_observed_values must have exactly 6 elements."),
                    ({"rbp_observation_weights": (0,) * 7}, "rbp_observation_weights must have exactly 6 elements."),
                    ({"rbp_observation_weights": (0, 0, -1, 0, 0, 0)}, "must be >= 0"),
                    ({"rbp_observation_weights": (np.inf,) * 6}, "At least one element")]:
        with pytest.raises(SimpleICPException, match=msg):
            icp.run(**kw)


def test_pointcloud_storage_views_need_no_copy():
    """What run() uploads: the frame's own storage (zero-copy view of the caller's (n,3) array, or the three
    column vectors once columns were assigned) -- always the same numbers as the copying ``X``."""
    from simpleicp_amd import PointCloud
    rng = np.random.default_rng(3)
    X = rng.normal(size=(1000, 3))
    pc = PointCloud(X, columns=["x", "y", "z"])
    kind, buf = pc._xyz_buffers()
    assert kind == "aos" and buf.flags.c_contiguous and np.shares_memory(buf, X) and not buf.flags.writeable
    assert np.array_equal(buf, pc.X) and not np.shares_memory(pc.X, X)
    pc["x"], pc["y"], pc["z"] = X[:, 2] * 2, X[:, 0], X[:, 1]           # like transform_by_H
    kind, buf = pc._xyz_buffers()
    assert kind == "soa" and all(b.flags.c_contiguous for b in buf)
    assert np.array_equal(np.column_stack(buf), pc.X)
    pc3 = PointCloud(pd.DataFrame({"z": X[:, 2], "x": X[:, 0].astype(np.float32), "y": X[:, 1]}))
    kind, buf = pc3._xyz_buffers()                                       # float32 column: gathered, converted copy
    assert kind == "aos" and buf.dtype == np.float64 and np.array_equal(buf, pc3.X)
    pc4 = PointCloud(np.asfortranarray(X), columns=["x", "y", "z"])
    kind, buf = pc4._xyz_buffers()
    assert np.array_equal(np.column_stack(buf) if kind == "soa" else buf, X)


def test_sparse_attribute_columns_equal_dense_construction():
    """estimate_normals' columns built in O(selected) == SparseArray(dense NaN-filled vector) (pointcloud.py:199-203),
    and reading them back at the selected rows in O(selected) == the dense read."""
    from simpleicp_amd import PointCloud
    n = 5000
    pc = PointCloud(np.zeros((n, 3)), columns=["x", "y", "z"])
    sel = np.unique(np.round(np.linspace(0, n - 1, 37)).astype(np.int64))
    rng = np.random.default_rng(0)
    vals = {c: rng.normal(size=len(sel)).astype(np.float32) for c in ("nx", "ny", "nz", "planarity")}
    vals["planarity"][5] = np.nan                                        # a degenerate neighbourhood
    for c, v in vals.items():
        pc[c] = pc._sparse_column(sel, v)
        dense = np.full(n, np.nan, np.float32)
        dense[sel] = v
        ref = pd.arrays.SparseArray(dense)
        assert pc[c].dtype == ref.dtype == pd.SparseDtype(np.float32, np.nan)
        assert np.array_equal(pc[c].array.sp_index.indices, ref.sp_index.indices)
        assert np.array_equal(pc[c].array.sp_values, ref.sp_values)
        assert np.array_equal(pc[c].to_numpy(), dense.astype(pc[c].to_numpy().dtype), equal_nan=True)
    for idx in (sel, sel[::3], np.array([0, 1, 2, n - 1]), np.empty(0, np.int64)):
        nv, pl = pc._attributes_of(idx)
        assert nv.dtype == pl.dtype == np.float32 and nv.shape == (len(idx), 3)
        for j, c in enumerate(("nx", "ny", "nz")):
            assert np.array_equal(nv[:, j], pc[c].to_numpy().astype(np.float32)[idx], equal_nan=True)
        assert np.array_equal(pl, pc["planarity"].to_numpy().astype(np.float32)[idx], equal_nan=True)
    pc["nx"] = np.arange(n, dtype=np.float64)                            # a caller-assigned dense column
    assert np.array_equal(pc._attributes_of(sel)[0][:, 0], np.arange(n, dtype=np.float32)[sel])


def _build_c_demo(tmp_path):
    import subprocess
    from simpleicp_amd import build
    build.build()
    exe = tmp_path / "c_abi_demo"
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O2", f"-I{ROOT / 'include'}",
                        str(ROOT / "examples" / "c_abi_demo.c"), f"-L{ROOT / 'simpleicp_amd'}", "-lsimpleicp_hip", "-lm",
                        f"-Wl,-rpath,{ROOT / 'simpleicp_amd'}", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c_and_a_c_host_links(tmp_path):
    """include/simpleicp_hip.h is C99 (-pedantic -Werror) and a C program links against the library; without a device
    it fails LOUDLY with SICP_ERR_NO_DEVICE (exit code 3 of the demo), it does not compute on the host."""
    import subprocess
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c",
                        str(ROOT / "include" / "simpleicp_hip.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exe = _build_c_demo(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is visible: the GPU flavour of this test runs the demo to the end")
    r = subprocess.run([str(exe), "20000", "500"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "ABI version 7" in r.stdout and "no HIP device" in r.stderr, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_c_host_runs_an_icp(tmp_path):
    """examples/c_abi_demo.c: uploads, normals, the whole iteration loop and the result through the C ABI from plain C."""
    import subprocess
    exe = _build_c_demo(tmp_path)
    r = subprocess.run([str(exe), "200000", "1000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "max |x - x_true|" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_every_export_refuses_null_arguments():
    """'Nothing throws across the ABI' includes not crashing: every entry point called with a NULL context / NULL
    pointers returns a negative code and leaves a message (sicp_ctx_destroy(NULL) is a no-op, like free)."""
    import subprocess
    from simpleicp_amd import _lib
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "helpers" / "null_abi_probe.py")], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    seen = {}
    for line in r.stdout.splitlines():
        name, rc, *msg = line.split(" ", 2)
        seen[name] = (int(rc), msg[0] if msg else "")
    expected = [n for n in _lib.EXPORTS if n not in ("sicp_abi_version", "sicp_last_error")]
    assert sorted(seen) == sorted(expected)
    for name, (rc, msg) in seen.items():
        if name == "sicp_ctx_destroy":
            assert rc == 0
        else:
            assert rc == _lib.ERR_INVALID and msg, (name, rc, msg)


def test_binding_refuses_a_library_of_another_abi_version(monkeypatch):
    """SICP_LIBRARY makes it easy to point the binding at a stale build: load() compares sicp_abi_version() with the version
    this binding was written for and refuses instead of calling entry points with the wrong arguments."""
    from simpleicp_amd import _lib
    _lib.load()                                                   # (the real library is fine)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", 99)
    with pytest.raises(_lib.BackendError, match="ABI version"):
        _lib.load()


def test_bench_withholds_stale_pmc_traffic(tmp_path, monkeypatch):
    """profiles/latest_pmc.json is quoted as `traffic` only when it was measured on THESE kernel sources (bench.csrc_hash)."""
    import json
    import bench
    pmc, src = bench.load_pmc()
    committed = json.loads((ROOT / "profiles" / "latest_pmc.json").read_text())
    if committed.get("_csrc_hash") == bench.csrc_hash():
        assert pmc.get("k_icp_tail", 0) > 0 and "NOT collected in this run" in src
    else:
        assert pmc == {} and "stale" in src
    # a file stamped with another hash is refused whatever it holds
    fake = tmp_path / "profiles"
    fake.mkdir()
    (fake / "latest_pmc.json").write_text(json.dumps({"_csrc_hash": "0" * 16, "k_icp_tail": 1.0}))
    (tmp_path / "simpleicp_amd").symlink_to(ROOT / "simpleicp_amd")
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    pmc, src = bench.load_pmc()
    assert pmc == {} and "stale" in src and bench.csrc_hash() in src
