"""The host-side mirror against the UNMODIFIED reference package imported next to it (build container only: the
GPU box has no copy; `lmfit`, absent from the image, is the 40-line stand-in of oracle/shim that make_golden.py uses too):
everything the mirror does WITHOUT the device -- the math helpers, the PointCloud selection bookkeeping, the
RigidBodyParameters schema, the argument checks and their messages, the public surface of the operator classes -- is
compared call by call on seeded inputs.  CPU only; skipped where /root/reference does not exist."""
import dataclasses
import inspect
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import ROOT

REF = Path("/root/reference/python")
pytestmark = pytest.mark.skipif(not (REF / "simpleicp").exists(), reason="the reference package is not on this machine")


@pytest.fixture(scope="module")
def ref():
    for p in (str(ROOT / "oracle" / "shim"), str(REF)):
        if p not in sys.path:
            sys.path.insert(0, p)
    import simpleicp
    from simpleicp import corrpts, mathutils, optimization, pointcloud, simpleicp as driver
    return dict(pkg=simpleicp, corrpts=corrpts, mathutils=mathutils, optimization=optimization, pointcloud=pointcloud,
                driver=driver)


def test_mathutils_call_by_call(ref):
    from simpleicp_amd import mathutils as mine
    theirs = ref["mathutils"]
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = rng.uniform(-1.5, 1.5, 3)
        R0, R1 = theirs.euler_angles_to_rotation_matrix(*a), mine.euler_angles_to_rotation_matrix(*a)
        assert np.abs(R0 - R1).max() <= 2.3e-16                             # np.cos on a scalar vs libm: an ulp at most
        assert np.allclose(mine.rotation_matrix_to_euler_angles(R1), a, atol=1e-14)
        assert np.allclose(theirs.rotation_matrix_to_euler_angles(R0), mine.rotation_matrix_to_euler_angles(R0), atol=1e-15)
        assert np.array_equal(theirs.euler_angles_to_linearized_rotation_matrix(*a), mine.euler_angles_to_linearized_rotation_matrix(*a))
        t = rng.normal(size=3)
        assert np.array_equal(theirs.create_homogeneous_transformation_matrix(R0, t), mine.create_homogeneous_transformation_matrix(R0, t))
    X = rng.normal(size=(50, 3))
    Xh0, Xh1 = theirs.euler_coord_to_homogeneous_coord(X), mine.euler_coord_to_homogeneous_coord(X)
    assert np.array_equal(Xh0, Xh1)
    Xh0[:, 3] = rng.uniform(0.5, 2.0, 50)
    assert np.array_equal(theirs.homogeneous_coord_to_euler_coord(Xh0), mine.homogeneous_coord_to_euler_coord(Xh0))


def test_pointcloud_selection_bookkeeping_call_by_call(ref):
    """pointcloud.py:51-147: the same sequence of host-side calls on both classes leaves the same state."""
    from simpleicp_amd import PointCloud as Mine
    Theirs = ref["pkg"].PointCloud
    rng = np.random.default_rng(1)
    for n in (1, 7, 1000, 20_011):
        X = rng.normal(size=(n, 3))
        a, b = Theirs(X, columns=["x", "y", "z"]), Mine(X, columns=["x", "y", "z"])

        def same():
            assert a.num_points == b.num_points and a.num_selected_points == b.num_selected_points
            assert np.array_equal(a.idx_selected, b.idx_selected)
            assert np.array_equal(a.X, b.X) and np.array_equal(a.X_selected, b.X_selected)
            for c in "xyz":
                assert np.array_equal(getattr(a, c), getattr(b, c))
                assert np.array_equal(getattr(a, c + "_selected"), getattr(b, c + "_selected"))
            assert a["selected"].dtype == b["selected"].dtype == bool
        same()
        for m in (n + 5, max(1, n // 2), max(1, n // 3), 5, 1):           # select_n_points narrows the CURRENT selection
            a.select_n_points(m), b.select_n_points(m)
            same()
        a.select_all_points(), b.select_all_points()
        pick = rng.choice(n, max(1, n // 4), replace=False)
        a.select_by_indices(pick), b.select_by_indices(pick)
        same()
        a.select_by_indices(pick[: len(pick) // 2]), b.select_by_indices(pick[: len(pick) // 2])
        same()
        a.idx_selected = np.sort(pick)
        b.idx_selected = np.sort(pick)
        same()
        a.unselect_all_points(), b.unselect_all_points()
        same()
        a.select_n_points(3), b.select_n_points(3)                         # nothing selected: nothing to narrow
        same()
    with pytest.raises(ref["pointcloud"].PointCloudException) as e0:
        Theirs(np.zeros((3, 2)), columns=["x", "y"])
    from simpleicp_amd import PointCloudException
    with pytest.raises(PointCloudException) as e1:
        Mine(np.zeros((3, 2)), columns=["x", "y"])
    assert str(e0.value) == str(e1.value)
    with pytest.raises(ref["pointcloud"].PointCloudException) as e0:         # pointcloud.py:158-159, before any search
        Theirs(np.zeros((3, 3)), columns=["x", "y", "z"]).select_in_range(np.zeros((4, 2)), 1.0)
    with pytest.raises(PointCloudException) as e1:
        Mine(np.zeros((3, 3)), columns=["x", "y", "z"]).select_in_range(np.zeros((4, 2)), 1.0)
    assert str(e0.value) == str(e1.value)


def test_rigid_body_parameters_schema_and_H(ref):
    from simpleicp_amd import Parameter as MyP, RigidBodyParameters as MyR
    TheirR, TheirP = ref["optimization"].RigidBodyParameters, ref["optimization"].Parameter
    assert [f.name for f in dataclasses.fields(TheirR)] == [f.name for f in dataclasses.fields(MyR)]
    assert [f.name for f in dataclasses.fields(TheirP)] == [f.name for f in dataclasses.fields(MyP)]
    rng = np.random.default_rng(2)
    a, b = TheirR(), MyR()
    for attr in ("initial_value", "observed_value", "observation_weight", "estimated_value", "estimated_uncertainty"):
        v = list(rng.uniform(-0.3, 0.3, 6))
        a.set_parameter_attributes_from_list(attr, v), b.set_parameter_attributes_from_list(attr, v)
        assert a.get_parameter_attributes_as_list(attr) == b.get_parameter_attributes_as_list(attr)
    assert np.abs(a.H - b.H).max() <= 2.3e-16
    for name in ("alpha1", "alpha2", "alpha3", "tx", "ty", "tz"):
        p, q = getattr(a, name), getattr(b, name)
        for prop in ("initial_value_scaled", "observed_value_scaled", "estimated_value_scaled", "estimated_uncertainty_scaled"):
            assert getattr(p, prop) == getattr(q, prop)
        assert p.scale_for_logging == q.scale_for_logging


@pytest.mark.parametrize("kwargs", [
    {"distance_weights": 0}, {"distance_weights": -1.5}, {"rbp_observed_values": (0, 0, 0)},
    {"rbp_observation_weights": (0,) * 5}, {"rbp_observation_weights": (0, 0, -1, 0, 0, 0)},
    {"rbp_observation_weights": (np.inf,) * 6},
])
def test_argument_checks_raise_the_reference_s_messages(ref, kwargs):
    """simpleicp.py:327-353: both raise before any work is done (here: before a device is asked for)."""
    from simpleicp_amd import PointCloud as MineP, SimpleICP as Mine, SimpleICPException
    X = np.random.default_rng(3).normal(size=(50, 3))
    a = ref["pkg"].SimpleICP(verbose=False)
    a.add_point_clouds(ref["pkg"].PointCloud(X, columns=["x", "y", "z"]), ref["pkg"].PointCloud(X + 0.01, columns=["x", "y", "z"]))
    b = Mine(verbose=False)
    b.add_point_clouds(MineP(X, columns=["x", "y", "z"]), MineP(X + 0.01, columns=["x", "y", "z"]))
    with pytest.raises(ref["driver"].SimpleICPException) as e0:
        a.run(**kwargs)
    with pytest.raises(SimpleICPException) as e1:
        b.run(**kwargs)
    assert str(e0.value) == str(e1.value)


def test_public_surface_of_every_mirrored_class(ref):
    """Same public methods / properties, same parameter names and defaults."""
    from simpleicp_amd import PointCloud, SimpleICP
    from simpleicp_amd.corrpts import CorrPts
    from simpleicp_amd.optimization import SimpleICPOptimization
    pairs = [(ref["pkg"].PointCloud, PointCloud), (ref["pkg"].SimpleICP, SimpleICP), (ref["corrpts"].CorrPts, CorrPts),
             (ref["optimization"].SimpleICPOptimization, SimpleICPOptimization)]
    import pandas as pd
    for theirs, mine in pairs:
        base = set(dir(pd.DataFrame)) if issubclass(theirs, pd.DataFrame) else set()
        pub = [n for n in vars(theirs) if not n.startswith("_") and n not in base]
        for n in pub:
            assert hasattr(mine, n), (theirs.__name__, n)
            t, m = inspect.getattr_static(theirs, n), inspect.getattr_static(mine, n)
            if isinstance(t, property):
                assert isinstance(m, property), (theirs.__name__, n)
            elif callable(t) or isinstance(t, staticmethod):
                st, sm = inspect.signature(getattr(theirs, n)), inspect.signature(getattr(mine, n))
                tp = [(p.name, p.default) for p in st.parameters.values() if not p.name.startswith("_")]
                mp = [(p.name, p.default) for p in sm.parameters.values() if not p.name.startswith("_")]
                assert [x[0] for x in tp] == [x[0] for x in mp], (theirs.__name__, n, tp, mp)
                for (pn, d0), (_, d1) in zip(tp, mp):
                    same = (d0 is d1) or (d0 == d1) or (isinstance(d0, float) and isinstance(d1, float) and np.isinf(d0) and np.isinf(d1))
                    assert same, (theirs.__name__, n, pn, d0, d1)
        ti, mi = inspect.signature(theirs.__init__), inspect.signature(mine.__init__)
        if not issubclass(theirs, pd.DataFrame):
            assert list(ti.parameters) == list(mi.parameters), theirs.__name__
    assert set(ref["pkg"].__dict__) >= {"SimpleICP", "PointCloud", "RigidBodyParameters"}
    import simpleicp_amd
    for n in ("SimpleICP", "PointCloud", "RigidBodyParameters"):
        assert hasattr(simpleicp_amd, n)
