"""Seeded random whole-iteration cases (tests/fuzz_flow.py) on the real library against the oracle.  GPU only."""
import pytest

import fuzz_flow

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
