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


@pytest.mark.parametrize("seed", range(16))
def test_random_iteration_case(ctx, seed):
    assert fuzz_flow.run_case(ctx, seed) == []
