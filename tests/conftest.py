import ast
import os
import re
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"
GOLDEN_CASES = ["dragon", "bunny", "multisensor", "webots", "dragon_q5000", "bunny_obs", "dragon_kw"]
# the movable cloud carries a `planarity` column (and, dragon_chain, a partial `selected` mask): it was the fixed
# cloud of an earlier reference run (oracle/make_golden.py CHAIN_CASES; corrpts.py:131-135,158-163)
GOLDEN_CHAIN = ["dragon_chain", "bunny_chain"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the full-size legs that take 15-60 s each (100 M-point clouds, 10 M x 10 M select_in_range, "
                                       "1 M-query oracle samples): run with SICP_TEST_SLOW=1 -- scripts/final_measure.sh does, and "
                                       "commits the log under profiles/ -- so that a plain `-m gpu` stays well inside ten minutes")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if os.environ.get("SICP_TEST_SLOW") != "1":
        skip_slow = pytest.mark.skip(reason="full-size leg: set SICP_TEST_SLOW=1 (scripts/final_measure.sh runs them; log under profiles/)")
        for item in items:
            if "slow" in item.keywords:
                item.add_marker(skip_slow)
    # `-m gpu` on a box without a GPU must fail loudly, not skip silently;
    # plain runs without a GPU skip the gpu-marked tests.
    if has_gpu():
        return
    markexpr = config.getoption("-m") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    g = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    files = [str(f) for f in g["files"]]
    # repr() of a plain dict written by make_golden.py; `inf` is the only non-literal in it
    kwargs = ast.literal_eval(re.sub(r"\binf\b", "1e999", str(g["kwargs"])))
    return g, files, kwargs


def movable_columns(g, n_mov):
    """(selected rows or None, dense float32 planarity or None) of the movable cloud of a fixture."""
    if "mov_sel_idx" not in g.files:
        return None, None
    sel = g["mov_sel_idx"]
    pl = np.full(n_mov, np.nan, np.float32)
    pl[g["mov_planarity_rows"]] = g["mov_planarity_vals"]
    return (None if len(sel) == n_mov else sel), pl


def load_cloud(stem):
    q = np.load(GOLDEN / "data" / f"{Path(stem).stem}.npz")["q"]
    return q.astype(np.float64) / 1e4


@pytest.fixture(scope="session")
def clouds():
    cache = {}

    def get(stem):
        stem = Path(stem).stem
        if stem not in cache:
            cache[stem] = load_cloud(stem)
        return cache[stem]
    return get
