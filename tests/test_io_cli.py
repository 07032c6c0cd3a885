"""Host-side .xyz I/O (native, multithreaded) and the CLI argument surface.  CPU only, except the CLI run."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_read_xyz_equals_genfromtxt(tmp_path):
    from simpleicp_amd import io
    rng = np.random.default_rng(0)
    X = np.round(rng.uniform(-1e4, 1e4, (50_003, 3)), 4)
    f = tmp_path / "a.xyz"
    np.savetxt(f, X, fmt="%.4f")
    assert np.array_equal(io.read_xyz(f), np.genfromtxt(f))
    # scientific notation, 18 significant digits, header / blank lines, extra columns, no trailing newline, CRLF
    g = tmp_path / "b.xyz"
    Y = rng.normal(size=(1000, 4)) * 10.0 ** rng.integers(-8, 8, (1000, 1))
    lines = ["//X Y Z I", "# comment", ""] + [" ".join("%.18e" % v for v in r) for r in Y[:500]] + ["", "   "] + \
            ["\t".join(repr(float(v)) for v in r) + "\r" for r in Y[500:]]
    g.write_text("\n".join(lines))
    got = io.read_xyz(g, threads=3)
    assert got.shape == (1000, 3) and np.array_equal(got, Y[:, :3])
    # empty file, missing file, short row
    (tmp_path / "e.xyz").write_text("")
    assert io.read_xyz(tmp_path / "e.xyz").shape == (0, 3)
    with pytest.raises(OSError, match="cannot open"):
        io.read_xyz(tmp_path / "nope.xyz")
    (tmp_path / "s.xyz").write_text("1 2 3\n4 5\n")
    with pytest.raises(OSError, match="fewer than 3"):
        io.read_xyz(tmp_path / "s.xyz")


def test_write_xyz_bytes_equal_reference_writers(tmp_path):
    """pointcloud.py:219-226 (pandas to_csv, '%.3f') and corrpts.py:213-237 (np.savetxt, '%.18e')."""
    from simpleicp_amd import PointCloud, io
    rng = np.random.default_rng(1)
    X = rng.normal(size=(70_001, 3)) * 100
    X[0] = [0.0005, -0.0005, 1e-9]
    io.write_xyz(tmp_path / "a.xyz", X)
    pd.DataFrame(X, columns=list("xyz")).to_csv(tmp_path / "b.xyz", sep=" ", header=["//X", "Y", "Z"], index=False,
                                                float_format="%.3f")
    assert (tmp_path / "a.xyz").read_bytes() == (tmp_path / "b.xyz").read_bytes()
    Z = rng.normal(size=(1234, 7))
    io.write_xyz(tmp_path / "c.xyz", Z, decimals=-1, header="//X1 Y1 Z1 X2 Y2 Z2 point_to_plane_distance")
    np.savetxt(tmp_path / "d.xyz", Z, delimiter=" ", header="X1 Y1 Z1 X2 Y2 Z2 point_to_plane_distance", comments="//")
    assert (tmp_path / "c.xyz").read_bytes() == (tmp_path / "d.xyz").read_bytes()
    pc = PointCloud(X[:100], columns=["x", "y", "z"])
    pc.write_xyz(tmp_path / "p.xyz")
    assert np.allclose(io.read_xyz(tmp_path / "p.xyz"), X[:100], atol=5.1e-4)


def test_xyz_extreme_values_and_non_finite_rows(tmp_path):
    """Values whose '%.3f' image is hundreds of characters long (1e100, DBL_MAX) and nan / inf are written with the
    bytes pandas writes, and rows that START with nan / inf are data rows for the reader as they are for
    np.genfromtxt (the 64-byte scratch buffer was over-read for |v| >= 1e59)."""
    from simpleicp_amd import io
    X = np.array([[1e100, -1e59, 1.0], [np.finfo(float).max, -np.finfo(float).max, 5e-324],
                  [np.nan, np.inf, -np.inf], [1.5, 2.5, 3.5], [-np.inf, 0.0, np.nan]])
    io.write_xyz(tmp_path / "a.xyz", X)
    pd.DataFrame(X, columns=list("xyz")).to_csv(tmp_path / "b.xyz", sep=" ", header=["//X", "Y", "Z"], index=False,
                                                float_format="%.3f", na_rep="nan")
    assert (tmp_path / "a.xyz").read_bytes() == (tmp_path / "b.xyz").read_bytes()
    got = io.read_xyz(tmp_path / "a.xyz")
    ref = np.genfromtxt(tmp_path / "a.xyz", comments="//")
    assert got.shape == (5, 3) and np.array_equal(got, ref, equal_nan=True)
    io.write_xyz(tmp_path / "c.xyz", X, decimals=-1, header=None)
    np.savetxt(tmp_path / "d.xyz", X)
    assert (tmp_path / "c.xyz").read_bytes() == (tmp_path / "d.xyz").read_bytes()
    with pytest.raises(OSError):
        io.write_xyz(tmp_path / "e.xyz", X, decimals=100)


def test_cli_options_mirror_reference():
    """c++/src/simpleicp-cli.cpp:12-35 / rust/src/main.rs:8-46: same short and long names and defaults."""
    from simpleicp_amd.cli import build_parser
    a = build_parser().parse_args(["-f", "a.xyz", "-m", "b.xyz"])
    assert (a.correspondences, a.neighbors, a.min_planarity, a.max_overlap_distance, a.min_change, a.max_iterations) == \
        (1000, 10, 0.3, -1.0, 1.0, 100)
    a = build_parser().parse_args(["--fixed", "a", "--movable", "b", "-c", "5", "-n", "6", "-p", "0.5", "-o", "2",
                                   "-i", "3", "-x", "7"])
    assert (a.correspondences, a.neighbors, a.min_planarity, a.max_overlap_distance, a.min_change, a.max_iterations) == \
        (5, 6, 0.5, 2.0, 3.0, 7)


@pytest.mark.gpu
def test_cli_end_to_end(tmp_path, clouds):
    from conftest import load_golden
    from simpleicp_amd import io
    g, files, kw = load_golden("bunny")
    io.write_xyz(tmp_path / "f.xyz", clouds(files[0]), decimals=4, header=None)
    io.write_xyz(tmp_path / "m.xyz", clouds(files[1]), decimals=4, header=None)
    r = subprocess.run([sys.executable, "-m", "simpleicp_amd", "-f", str(tmp_path / "f.xyz"), "-m", str(tmp_path / "m.xyz"),
                        "-o", "1", "--output", str(tmp_path / "out.xyz")], capture_output=True, text=True, cwd=ROOT,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    text = r.stdout + r.stderr
    assert "Estimated transformation matrix H:" in text and "Finished in" in text
    assert "[    0.984798    -0.173702    -0.000053     0.000676]" in text          # python/README.md:62
    assert io.read_xyz(tmp_path / "out.xyz").shape == clouds(files[1]).shape


def test_xyz_round_trip_properties(tmp_path):
    """Property test (hypothesis): any finite double written with any `decimals` has Python's own '%.<d>f' / '%.18e' bytes,
    and what the reader returns is float() of those bytes -- for every thread count, with and without a header."""
    from hypothesis import given, settings
    from hypothesis import strategies as st
    from simpleicp_amd import io

    doubles = st.floats(allow_nan=False, allow_infinity=False, width=64)
    rows = st.lists(st.tuples(doubles, doubles, doubles), min_size=0, max_size=40)

    @settings(max_examples=120, deadline=None)
    @given(rows=rows, decimals=st.integers(min_value=-1, max_value=17), threads=st.integers(min_value=1, max_value=5),
           header=st.sampled_from([None, "//X Y Z", "# anything"]))
    def check(rows, decimals, threads, header):
        X = np.array(rows, dtype=np.float64).reshape(-1, 3)
        f = tmp_path / "p.xyz"
        io.write_xyz(f, X, decimals=decimals, header=header, threads=threads)
        fmt = "%.18e" if decimals < 0 else f"%.{decimals}f"
        want = ([header] if header is not None else []) + [" ".join(fmt % v for v in r) for r in X]
        text = f.read_text()
        assert text == "".join(line + "\n" for line in want)
        back = io.read_xyz(f, threads=threads)
        ref = np.array([[float(fmt % v) for v in r] for r in X], dtype=np.float64).reshape(-1, 3)
        assert back.shape == ref.shape and np.array_equal(back, ref)
        assert np.array_equal(np.signbit(back), np.signbit(ref))          # -0.0 stays -0.0
    check()
