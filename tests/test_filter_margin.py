"""The float32 filter of the many-queries search (simpleicp_amd/csrc/sicp_gridf.hip) is exact only if its margin really bounds the
difference between a candidate's float32 value and its contract distance.  This restates the kernel's arithmetic in numpy -- float32
coordinates relative to the centre of the cloud's box, float32 differences, one multiplication and two fused multiply-adds -- and the
margin's formula (filter_margin and the per-pass eps there; change one, change the other), and checks |v - d2| <= E on the inputs the
GPU tests use: a surface, UTM offsets, coordinates of 1e-6, queries far outside the cloud, quantised lattices, a line, identical
points; small and large rigid transforms.  CPU only: no device code runs here."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import orc  # noqa: E402


def margin_E(A, eps, slack):
    e_c = eps + 6.0e-8 * A
    return e_c * (3.5 * A + 3.0 * e_c) + 2.4e-7 * A * A + 2.02 * A * slack + slack * slack


def worst_ratio(P, Qp, H):
    Hinv = np.linalg.inv(H)
    lo, hi = P.min(0), P.max(0)
    c0 = 0.5 * (lo + hi)
    eps_p = 6.0e-8 * max((hi - c0).max(), (c0 - lo).max())
    rmax = np.sqrt((P ** 2).sum(1).max())
    pf = (P - c0).astype(np.float32)
    X = P @ H[:3, :3].T + H[:3, 3]
    worst = 0.0
    for q in Qp:
        qp = Hinv[:3, :3] @ q + Hinv[:3, 3]
        rq = qp - c0
        qf = rq.astype(np.float32)
        qinf = np.abs(rq).max()
        slack = 1e-12 * (rmax + np.abs(qp).sum() + 1.0)
        d = pf - qf                                                   # float32 subtraction
        t = d[:, 0] * d[:, 0]                                         # float32 product
        t = (d[:, 1].astype(np.float64) * d[:, 1].astype(np.float64) + t.astype(np.float64)).astype(np.float32)      # fma
        v = (d[:, 2].astype(np.float64) * d[:, 2].astype(np.float64) + t.astype(np.float64)).astype(np.float32)      # fma
        d2 = ((X - q) ** 2).sum(1)
        A = np.sqrt(d2)                                               # the tightest radius each candidate can be asked about
        eps = 6.0e-8 * qinf + np.minimum(eps_p, 6.0e-8 * (qinf + A))
        worst = max(worst, float((np.abs(v.astype(np.float64) - d2) / margin_E(A, eps, slack)).max()))
    return worst


def _cases():
    rng = np.random.default_rng(3)
    H1 = orc.params_to_H(np.array([0.01, -0.006, 0.014, 0.3, -0.2, 0.1]))
    H2 = orc.params_to_H(np.array([1.1, -0.7, 2.3, 30.0, -20.0, 10.0]))
    n = 60_000
    L = np.sqrt(n / 10.0)
    x, y = rng.uniform(0, L, n), rng.uniform(0, L, n)
    S = np.column_stack((x, y, 20 * np.sin(2 * np.pi * x / 200) * np.cos(2 * np.pi * y / 300)))
    utm = np.array([4.5e5, 5.2e6, 300.0])
    line = np.zeros((20_000, 3)); line[:, 0] = np.round(rng.uniform(0, 100, 20_000), 2)
    yield "surface", S - S.mean(0), (S[::3000] - S.mean(0)) + rng.normal(0, 0.3, (20, 3)), H1
    yield "surface_large_H", S - S.mean(0), (S[::3000] - S.mean(0)) + rng.normal(0, 0.3, (20, 3)), H2
    yield "utm_offset", S + utm, S[::3000] + utm + rng.normal(0, 0.05, (20, 3)), H1
    yield "tiny_coords", rng.uniform(-1e-6, 1e-6, (30_000, 3)), rng.uniform(-1e-6, 1e-6, (20, 3)), H1
    yield "far_queries", rng.uniform(-1, 1, (30_000, 3)), np.concatenate((rng.uniform(500, 600, (12, 3)), [[1e6, -1e6, 3.0]])), np.eye(4)
    yield "quantised", np.round(rng.uniform(-20, 20, (30_000, 3)), 1), np.round(rng.uniform(-20, 20, (20, 3)), 1), H2
    yield "line", line, np.column_stack((rng.uniform(-10, 110, 20), rng.normal(0, 1, 20), rng.normal(0, 1, 20))), H1
    yield "identical_points", np.tile([[1.5, -2.5, 3.25]], (3000, 1)), rng.uniform(-5, 5, (20, 3)), H1


@pytest.mark.parametrize("name", [c[0] for c in _cases()])
def test_float32_value_is_within_the_margin_of_the_contract_distance(name):
    P, Qp, H = next(c[1:] for c in _cases() if c[0] == name)
    w = worst_ratio(P, Qp, H)
    assert w <= 1.0, (name, w)
    assert w > 0.01, (name, w)           # (a margin a hundred times too wide would send every query to the exact kernel)


def test_margin_holds_on_a_seeded_sweep_of_scales_offsets_and_transforms():
    """... and on 150 random configurations spanning twelve decades of scale, offsets up to 1e4 x the extent, flat and cubic boxes,
    quantised coordinates, small and arbitrary rotations, query distances from 1e-9 to 10 x the scale (the worst ratio seen over 400 such
    configurations was 0.50: the margin has a factor of two in hand, as its derivation's constants suggest)."""
    rng = np.random.default_rng(2025)
    worst = 0.0
    for _ in range(150):
        s = 10.0 ** rng.uniform(-6, 6)
        ext = s * 10.0 ** rng.uniform(-3, 0, 3)
        off = rng.uniform(-1, 1, 3) * s * 10.0 ** rng.uniform(0, 4) if rng.random() < 0.6 else np.zeros(3)
        P = rng.uniform(-1, 1, (1500, 3)) * ext + off
        if rng.random() < 0.3:
            P = np.round(P / (s * 1e-3)) * (s * 1e-3)
        ang = rng.uniform(-np.pi, np.pi, 3) * (1.0 if rng.random() < 0.5 else 1e-3)
        t = rng.uniform(-1, 1, 3) * s * 10.0 ** rng.uniform(-3, 1)
        H = orc.params_to_H(np.concatenate((ang, t)))
        base = P[rng.choice(len(P), 24)] @ H[:3, :3].T + H[:3, 3]
        disp = rng.normal(0, 1, (24, 3))
        disp /= np.linalg.norm(disp, axis=1)[:, None]
        Qp = base + disp * (s * 10.0 ** rng.uniform(-9, 1, (24, 1)))
        worst = max(worst, worst_ratio(P, Qp, H))
    assert worst <= 1.0, worst
