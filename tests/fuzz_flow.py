"""Seeded random whole-iteration cases for tests/test_gpu_fuzz.py: cloud size (scan and grid paths), number of
correspondences (every tail flavour), neighbours, planarity threshold, distance weight (fixed / automatic), observed and
fixed parameters, an optional planarity column on the movable cloud, a far-away origin -- two chained iterations each
through sicp_icp_iterate, held against the oracle with the bounds of the hand-picked tests (indices / distances / masks /
median / MAD bit for bit, estimate to 1e-9 relative to its size, uncertainties to 1e-8)."""
import numpy as np

from oracle import orc


def make_case(seed):
    rng = np.random.default_rng(10_000 + seed)
    n = int(rng.choice([3000, 20_000, 70_001, 150_000]))
    Q = int(rng.choice([200, 1000, 1500, 3000, 9000]))
    Q = min(Q, n // 2)
    L = np.sqrt(n / 10.0)
    a1, a2 = rng.uniform(0.5, 3.0), rng.uniform(0.2, 1.0)
    l1, l2 = rng.uniform(0.15, 0.4) * L, rng.uniform(0.05, 0.12) * L
    xy = rng.uniform(0, L, (n, 2))
    z = a1 * np.sin(2 * np.pi * xy[:, 0] / l1) * np.cos(2 * np.pi * xy[:, 1] / l1) + a2 * np.sin(2 * np.pi * xy[:, 0] / l2 + 1) * \
        np.sin(2 * np.pi * xy[:, 1] / l2) + rng.normal(0, 0.01, n)
    P = np.column_stack((xy, z))
    P -= P.mean(axis=0)
    far = seed % 3 == 0
    if far:
        P += rng.uniform(-1, 1, 3) * 1e4                                  # an origin far from the data (UTM-like coordinates)
    x_true = np.concatenate((rng.uniform(-0.004, 0.004, 3), rng.uniform(-0.08, 0.08, 3)))
    if far:
        # H rotates about the ORIGIN: kilometres away an angle of 1e-3 rad moves the cloud by metres and the two clouds no longer
        # overlap at the start (x = 0) -- a registration nobody poses (and a singular one: the oracle itself does not settle).
        # Georeferenced scans are misaligned by centimetres: keep the rotation's lever-arm effect at that size.
        x_true[:3] *= 1e-4
    Xm = orc.transform(np.linalg.inv(orc.params_to_H(x_true)), P + rng.normal(0, 0.01, P.shape))
    sel = np.sort(rng.choice(n, Q, replace=False))
    kind = rng.choice(3, 6, p=[0.6, 0.2, 0.2])                             # free / fixed / observed
    if np.count_nonzero(kind != 1) < 3:
        kind[:3] = 0
    ow = np.where(kind == 1, np.inf, np.where(kind == 2, rng.uniform(5, 200, 6), 0.0))
    obs = np.where(kind == 1, x_true, np.where(kind == 2, x_true + rng.normal(0, 1e-3, 6), 0.0))
    return dict(P=P, Xm=Xm, sel=sel, k=int(rng.integers(4, 16)), min_planarity=float(rng.uniform(0.1, 0.5)),
                w=[None, 1.0, 7.5][int(rng.integers(0, 3))], obs=obs, ow=ow,
                pl2=(rng.uniform(0, 1, n).astype(np.float32) if seed % 4 == 1 else None), x_true=x_true, far=far)


def run_case(ctx, seed):
    """Problems found (empty list = parity)."""
    from simpleicp_amd import _lib
    c = make_case(seed)
    P, Xm, sel, obs, ow = c["P"], c["Xm"], c["sel"], c["obs"], c["ow"]
    bad = []
    ctx.upload(_lib.FIX, P)
    ctx.upload(_lib.MOV, Xm)
    if c["pl2"] is not None:
        ctx.set_planarity(_lib.MOV, c["pl2"])
    nv, pl, nn = ctx.estimate_normals(_lib.FIX, sel, c["k"], want_nn=True)
    rnn, _ = orc.knn(P, P[sel], k=c["k"])
    if not np.array_equal(nn, rnn):
        bad.append("normals k-NN indices")
    rnv, rpl = orc.normals(P, rnn)
    if np.abs(nv - rnv).max() > 2e-7 or np.nanmax(np.abs(pl - rpl)) > 2e-6:
        bad.append(f"normals {np.abs(nv - rnv).max():.1e} / planarity {np.nanmax(np.abs(pl - rpl)):.1e}")
    ctx.icp_setup(sel, nv, pl)
    x = obs.copy()                                                        # run() starts from the observed values (simpleicp.py:150-156,222-224)
    w = c["w"]
    for it in range(2):
        o = orc.icp_iteration(Xm, P[sel], nv, pl, x, x, w, obs, ow, c["min_planarity"], planarity_mov=c["pl2"])
        if o["n"] < 6:
            bad.append("degenerate case (fewer than 6 correspondences): regenerate")
            break
        R = ctx.icp_iterate(x, obs, ow, c["min_planarity"], w)
        idx, dist, keep, resid = ctx.icp_state()
        for name, a, b in (("indices", idx, o["nn"]), ("distances", dist, o["dist"]), ("keep mask", keep, o["keep"])):
            if not np.array_equal(a, b):
                bad.append(f"it {it}: {name} differ in {int(np.count_nonzero(a != b))} rows")
        if R.median != o["median"] or R.mad != o["mad"] or R.n_kept != o["n"]:
            bad.append(f"it {it}: median / MAD / n_kept")
        xg = np.array(R.x[:])
        if c["far"]:
            # rotation about an origin kilometres away and translation are nearly the same motion of the data: the
            # parameters are determined to ~1e-8 only (the oracle itself, restarted 1e-7 away, returns to 6e-9), the MOTION
            # is what is determined -- compare the kept movable points under both estimates
            pts = Xm[idx][keep]
            move = np.abs(orc.transform(orc.params_to_H(xg), pts) - orc.transform(orc.params_to_H(o["x"]), pts)).max()
            if move > 1e-7:
                bad.append(f"it {it}: the two estimates move the data {move:.2e} apart")
        else:
            tol = 1e-9 * (1.0 + np.abs(o["x"]).max())
            if np.abs(xg - o["x"]).max() > tol:
                bad.append(f"it {it}: |x - oracle| = {np.abs(xg - o['x']).max():.2e} > {tol:.1e}")
        if not np.array_equal(xg[~np.isfinite(ow)], x[~np.isfinite(ow)]):
            bad.append(f"it {it}: a fixed parameter moved")
        w = R.weight_used if w is None else w
        x = xg
    if not bad:
        s = ctx.icp_uncertainties()
        so = orc.uncertainties(x, w, obs, ow, P[sel], nv, Xm[idx], keep)
        free = np.isfinite(ow)
        if not (np.allclose(s[free], so[free], rtol=1e-4 if c["far"] else 1e-8) and np.all(np.isnan(s[~free]))):
            bad.append("uncertainties")
    ctx.set_planarity(_lib.MOV, None)
    return bad
