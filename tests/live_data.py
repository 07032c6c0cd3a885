"""Seeded random cloud pairs for the tests that run the unmodified reference live (tests/test_live_*.py)."""
import numpy as np

from oracle import orc


def cloud_pair(seed):
    rng = np.random.default_rng(500 + seed)
    n = int(rng.integers(4000, 9000))
    xy = rng.uniform(-8, 8, (n, 2))
    z = 1.5 * np.sin(xy[:, 0] / 2.0) * np.cos(xy[:, 1] / 3.0) + 0.4 * np.sin(xy[:, 0] * 1.3 + 1) + rng.normal(0, 0.005, n)
    P = np.column_stack((xy, z))
    x_true = np.concatenate((rng.uniform(-0.01, 0.01, 2), [np.deg2rad(1.0) + rng.uniform(-0.004, 0.004)], rng.uniform(-0.06, 0.06, 3)))
    M = orc.transform(np.linalg.inv(orc.params_to_H(x_true)), P[rng.permutation(n)[: n - 500]] + rng.normal(0, 0.005, (n - 500, 3)))
    if seed == 1:
        M = M[M[:, 0] > -3.0]                                  # partial overlap
    return np.ascontiguousarray(P), np.ascontiguousarray(M)
