"""The golden fixtures are outputs of the UNMODIFIED reference: where the reference is on disk (the build container;
the GPU box has no copy) regenerate all of them with the committed script and require identity with the committed
files -- every array of every fixture, the input clouds included -- and the README known-answer the script checks on
the way (python/README.md:62-73).  CPU only; skipped where /root/reference does not exist."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

REF = Path("/root/reference/python/simpleicp")


@pytest.mark.skipif(not REF.exists(), reason="the reference package is not on this machine")
def test_committed_fixtures_are_what_the_reference_produces(tmp_path):
    r = subprocess.run([sys.executable, str(ROOT / "oracle" / "make_golden.py"), "--out", str(tmp_path)], capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "known-answer (Bunny H, rbp, uncertainties): reproduced to all printed digits" in r.stdout
    committed = sorted(p.relative_to(GOLDEN) for p in GOLDEN.rglob("*.npz"))
    fresh = sorted(p.relative_to(tmp_path) for p in tmp_path.rglob("*.npz"))
    assert committed == fresh and len(committed) == 17
    for rel in committed:
        a, b = np.load(GOLDEN / rel, allow_pickle=False), np.load(tmp_path / rel, allow_pickle=False)
        assert sorted(a.files) == sorted(b.files), rel
        for k in a.files:
            if k == "log":                                   # carries the run's wall time ("Finished in ... seconds!")
                strip = lambda t: [ln for ln in str(t).splitlines() if not ln.startswith("Finished in")]   # noqa: E731
                assert strip(a[k]) == strip(b[k]), (rel, k)
            else:
                assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), (rel, k)
