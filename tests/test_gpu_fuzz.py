"""Seeded random whole-iteration cases (tests/fuzz_flow.py) on the real library against the oracle.  GPU only."""
import pytest

import fuzz_flow
import fuzz_sequence

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from simpleicp_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


NEAR = [s for s in range(16) if s % 3]          # data around the origin: the estimate itself is compared (1e-9)
FAR = [s for s in range(16) if s % 3 == 0]      # origin kilometres away: parameters ill-determined, the motion of the data is compared


@pytest.mark.parametrize("seed", NEAR)
def test_random_iteration_case(ctx, seed):
    assert fuzz_flow.run_case(ctx, seed) == []


@pytest.mark.parametrize("seed", FAR)
def test_random_iteration_case_far_origin(ctx, seed):
    assert fuzz_flow.run_case(ctx, seed) == []


@pytest.mark.parametrize("seed", range(32))
def test_random_call_sequence(seed):
    """Random sequences of ABI calls on one context (uploads to either slot, transform, planarity column, k-NN, select_in_range,
    setups, host-driven and chained iterations, the operator road), the many-queries kernels' thresholds forced low; every result
    against the oracle on a model of what the library should hold (tests/fuzz_sequence.py).  Seeds 0, 8, 16, 24 start with the
    order round 5's stale-slot bug needed (a chained run through the filtered search, a new setup of the same size, an operator
    match, a chained run): with that fix reverted (`c->slot_cnt = -1` in sicp_icp_setup) they fail in the first iteration of that run."""
    bad, log, kernels = fuzz_sequence.run_sequence(seed)
    assert bad == [], "\n".join(log[-14:] + bad)
    if seed % 8 == 0:
        assert "k_grid_nn16f" in kernels, (kernels, log)


def test_call_sequence_fuzz_finds_the_stale_slot_bug_when_its_fix_is_reverted():
    """The scripted seeds of the call-sequence fuzz against a build WITHOUT round 5's last fix (sicp_icp_setup forgetting the filtered
    search's slots: -DSICP_TEST_REVERT_SLOT_FIX): the fuzz must report it."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    from simpleicp_amd import build
    lib = build.build_variant("revert_slot_fix", ["-DSICP_TEST_REVERT_SLOT_FIX"], units=("sicp_icp",))
    root = Path(__file__).resolve().parent.parent
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import fuzz_sequence\n"
            "bad, log, kernels = fuzz_sequence.run_sequence(0)\n"
            "print('FOUND' if bad else 'CLEAN', bad[:2])\n") % (str(root), str(root / "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SICP_LIBRARY=str(lib)), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FOUND" in r.stdout and "icp_run it 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
