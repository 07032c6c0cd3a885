"""The iteration operator by operator on the real library: the mirror classes simpleicp_amd.corrpts.CorrPts /
simpleicp_amd.optimization.SimpleICPOptimization and the C-ABI entry points below them (sicp_corr_match,
sicp_corr_reject_planarity, sicp_corr_reject_distances, sicp_estimate_parameters) against the fixtures of the unmodified
reference and the CPU oracle.  The flows live in tests/operator_flow.py.  GPU only."""
import pytest

import operator_flow as flow
from conftest import GOLDEN_CASES, GOLDEN_CHAIN

pytestmark = pytest.mark.gpu


def _ctx():
    from simpleicp_amd import _lib
    return _lib.Context(0)


@pytest.mark.parametrize("name", GOLDEN_CASES + GOLDEN_CHAIN)
def test_reference_loop_operator_by_operator(name, clouds):
    flow.reference_loop(name, clouds)


@pytest.mark.parametrize("Q", [700, 3000, 20_000])
def test_operators_vs_oracle_and_fused_iteration(Q):
    flow.abi_operators_vs_oracle(_ctx, Q)


def test_rejections_commute_like_row_filters():
    flow.rejections_commute(_ctx)


def test_corrpts_object_semantics(clouds, tmp_path):
    flow.corrpts_object_semantics(clouds, tmp_path)
