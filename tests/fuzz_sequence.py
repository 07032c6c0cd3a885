"""Seeded random CALL SEQUENCES on one context for tests/test_gpu_fuzz.py: the C ABI keeps state between calls (grids per slot, the
last match as the next search's bound, the filtered search's slot-ordered copies of queries and bounds, windows of the rejections,
barrier counters) and a bug there only shows when calls arrive in an order nobody wrote down -- round 5's last fix (40f67a5: a new
setup followed by an operator-route match left the filtered search with another setup's slots) was found by hand, minutes before the
deadline.  Here a model of the library's state is kept next to it (the clouds as they should be, the setup, the estimate) and after
every call the result is held against the CPU oracle on that model:

    upload / upload_columns (either slot) - transform - set_planarity - knn (k = 1 with H and a distance bound, k > 1) -
    select_in_range - icp_setup - icp_iterate - icp_run (iterations chained on the device) -
    corr_match -> reject_planarity -> reject_distances -> estimate_parameters (the operator road)

The context is created with the many-queries kernels' thresholds forced low (SICP_NN16_MIN_Q, SICP_NN16F_MIN_Q, SICP_ORDER_MIN_Q), so
that clouds of 20 000 points and a few thousand queries reach the four-per-wave search (k_grid_nn16) and both flavours of the
float32-filtered one (k_grid_nn16f lean-first / full-only by seed), cell-ordered queries and slot-ordered bounds included.
Bit-level assertions as in the hand-picked tests: indices, distances, masks, median, MAD; estimates to 1e-9."""
import os

import numpy as np

from oracle import orc


def _surface(rng, n):
    L = np.sqrt(n / 10.0)
    xy = rng.uniform(0, L, (n, 2))
    l1, l2 = rng.uniform(0.2, 0.4) * L, rng.uniform(0.06, 0.12) * L
    z = 1.5 * np.sin(2 * np.pi * xy[:, 0] / l1) * np.cos(2 * np.pi * xy[:, 1] / l1) + 0.5 * np.sin(2 * np.pi * xy[:, 0] / l2 + 1) * \
        np.sin(2 * np.pi * xy[:, 1] / l2) + rng.normal(0, 0.01, n)
    P = np.column_stack((xy, z))
    return P - P.mean(axis=0)


def _small_H(rng, scale=1.0):
    return orc.params_to_H(np.concatenate((rng.uniform(-0.003, 0.003, 3), rng.uniform(-0.05, 0.05, 3))) * scale)


class Model:
    """What the library should hold, and the checks of every operation against the oracle on it."""

    def __init__(self, ctx, rng):
        from simpleicp_amd import _lib
        self.ctx, self.rng, self.L = ctx, rng, _lib
        self.F = self.M = None
        self.pl2 = None
        self.setup = None                 # (sel, normals, planarity)
        self.x = np.zeros(6)
        self.bad = []
        self.log = []
        self.kernels = set()

    # ---- operations ----
    def upload(self, slot):
        rng, L = self.rng, self.L
        n = int(rng.choice([6000, 20_000, 20_001]))
        if slot == L.FIX or self.F is None:
            P = _surface(rng, n)
        else:
            # a movable cloud that overlaps the fixed one: a resampling of it, a few centimetres and milliradians off
            k = min(n, len(self.F))
            base = self.F[rng.permutation(len(self.F))[:k]] + rng.normal(0, 0.01, (k, 3))
            P = orc.transform(np.linalg.inv(_small_H(rng)), base)
        if rng.random() < 0.5:
            self.ctx.upload(slot, P)
        else:
            self.ctx.upload_columns(slot, np.ascontiguousarray(P[:, 0]), np.ascontiguousarray(P[:, 1]), np.ascontiguousarray(P[:, 2]))
        if slot == L.FIX:
            self.F, self.setup = P, None
        else:
            self.M, self.pl2 = P, None     # (an upload clears the slot's planarity column)
        self.log.append(f"upload {'FIX' if slot == L.FIX else 'MOV'} n={len(P)}")

    def transform(self):
        H = _small_H(self.rng, 0.3)
        self.ctx.transform(self.L.MOV, H)
        self.M = orc.transform(H, self.M)
        self.log.append("transform MOV")
        if self.rng.random() < 0.5:
            got = self.ctx.download(self.L.MOV)
            if not np.array_equal(got, self.M):
                self.bad.append("transform: downloaded cloud differs from H @ X")

    def set_planarity(self):
        if self.rng.random() < 0.3:
            self.pl2 = None
            self.ctx.set_planarity(self.L.MOV, None)
        else:
            self.pl2 = self.rng.uniform(0, 1, len(self.M)).astype(np.float32)
            self.pl2[self.rng.random(len(self.M)) < 0.1] = np.nan
            self.ctx.set_planarity(self.L.MOV, self.pl2)
        self.log.append(f"set_planarity {'none' if self.pl2 is None else 'column'}")

    def knn(self):
        rng, L = self.rng, self.L
        slot = L.MOV if rng.random() < 0.7 else L.FIX
        P = self.M if slot == L.MOV else self.F
        q = int(rng.choice([7, 300, 3000]))
        Q = P[rng.integers(0, len(P), q)] + rng.normal(0, 0.05, (q, 3))
        if rng.random() < 0.6:
            H = _small_H(rng) if rng.random() < 0.7 else None
            md = float(rng.choice([np.inf, 0.08, 0.3]))
            idx, d2 = self.ctx.knn(slot, Q, k=1, H=H, max_dist=md)
            ridx, rd2 = orc.knn(P, Q, k=1, H=H, max_dist=md)
            what = f"knn k=1 q={q} H={'yes' if H is not None else 'no'} max_dist={md}"
        else:
            k = int(rng.choice([5, 12]))
            idx, d2 = self.ctx.knn(slot, Q, k=k)
            ridx, rd2 = orc.knn(P, Q, k=k)
            what = f"knn k={k} q={q}"
        self.log.append(what)
        if not (np.array_equal(idx, ridx) and np.array_equal(d2, rd2)):
            self.bad.append(f"{what}: {int(np.count_nonzero(idx != ridx))} indices differ")

    def select_in_range(self):
        rng, L = self.rng, self.L
        H = _small_H(rng) if rng.random() < 0.5 else None
        r = float(rng.choice([0.05, 0.2]))
        sel = None if rng.random() < 0.5 else np.unique(rng.integers(0, len(self.F), len(self.F) // 3))
        got = self.ctx.select_in_range(L.FIX, L.MOV, sel, H, r)
        Fq = self.F if sel is None else self.F[sel]
        want = orc.knn(self.M, Fq, k=1, H=H, max_dist=r)[0][:, 0] >= 0
        self.log.append(f"select_in_range r={r} sel={'all' if sel is None else len(sel)}")
        if not np.array_equal(got, want):
            self.bad.append(f"select_in_range: {int(np.count_nonzero(got != want))} verdicts differ")

    def icp_setup(self, Q=None):
        rng, L = self.rng, self.L
        Q = int(rng.choice([500, 3000, 3000, 9000])) if Q is None else Q
        Q = min(Q, len(self.F) // 2)
        sel = np.sort(rng.choice(len(self.F), Q, replace=False))
        nv, pl = self.ctx.estimate_normals(L.FIX, sel, int(rng.choice([8, 10])))
        self.ctx.icp_setup(sel, nv, pl)
        self.setup = (sel, nv, pl)
        self.x = np.zeros(6)
        self.log.append(f"icp_setup Q={Q}")

    def _check_iteration(self, tag, R, x_prev, w, minpl, obs, ow, state=None, exact=True):
        """exact: the iteration started from an estimate the host handed over (H(x) from libm on both sides): bit-level.  A LATER
        iteration of a chained run starts from the device's own state, whose sin / cos were carried forward by the addition
        theorem: H agrees with the oracle's to ~1e-16, so distances and statistics to rounding, not to the bit."""
        sel, nv, pl = self.setup
        o = orc.icp_iteration(self.M, self.F[sel], nv, pl, x_prev, x_prev, w, obs, ow, minpl, planarity_mov=self.pl2)
        if o["n"] < 6:
            return o
        eps = 0.0 if exact else 1e-12
        if abs(R.median - o["median"]) > eps or abs(R.mad - o["mad"]) > eps or R.n_kept != o["n"]:
            self.bad.append(f"{tag}: median / MAD / n_kept ({R.median} {R.mad} {R.n_kept} vs {o['median']} {o['mad']} {o['n']})")
        tol = 1e-9 * (1.0 + np.abs(o["x"]).max())
        if np.abs(np.array(R.x[:]) - o["x"]).max() > tol:
            self.bad.append(f"{tag}: |x - oracle| = {np.abs(np.array(R.x[:]) - o['x']).max():.2e}")
        if state is not None:
            idx, dist, keep, resid = state
            if not np.array_equal(idx, o["nn"]):
                self.bad.append(f"{tag}: indices differ in {int(np.count_nonzero(idx != o['nn']))} rows")
            elif np.abs(dist - o["dist"]).max() > eps:
                self.bad.append(f"{tag}: distances differ by {np.abs(dist - o['dist']).max():.2e}")
            if not np.array_equal(keep, o["keep"]):
                self.bad.append(f"{tag}: keep mask differs in {int(np.count_nonzero(keep != o['keep']))} rows")
        return o

    def icp_iterate(self):
        rng = self.rng
        z = np.zeros(6)
        minpl = float(rng.choice([0.2, 0.3]))
        w = [None, 1.0, 4.0][int(rng.integers(0, 3))]
        try:
            R = self.ctx.icp_iterate(self.x, z, z, minpl, w)
        except self.L.BackendError as e:
            if e.code != self.L.ERR_TOO_FEW:
                raise
            self.log.append("icp_iterate: too few"); return
        self.kernels.add(self.ctx.last_match_kernel())
        self.log.append(f"icp_iterate ({self.ctx.last_match_kernel()})")
        self._check_iteration("icp_iterate", R, self.x, w, minpl, z, z, self.ctx.icp_state())
        self.x = np.array(R.x[:])

    def icp_run(self):
        rng = self.rng
        z = np.zeros(6)
        minpl = float(rng.choice([0.2, 0.3]))
        w = [1.0, 4.0][int(rng.integers(0, 2))]
        K = int(rng.choice([2, 3, 5]))
        try:
            Rs = self.ctx.icp_run(self.x, z, z, minpl, w, max_iterations=K, min_change=0.0)
        except self.L.BackendError as e:
            if e.code != self.L.ERR_TOO_FEW:
                raise
            self.log.append("icp_run: too few"); return
        self.kernels.add(self.ctx.last_match_kernel())
        self.log.append(f"icp_run K={K} ({self.ctx.last_match_kernel()})")
        if len(Rs) != K:
            self.bad.append(f"icp_run: {len(Rs)} iterations of {K}")
        x_prev = self.x
        state = self.ctx.icp_state()
        for it, R in enumerate(Rs):
            # (each iteration from the DEVICE's previous estimate: differences do not accumulate into the comparison)
            self._check_iteration(f"icp_run it {it}", R, x_prev, w, minpl, z, z, state if it == len(Rs) - 1 else None, exact=(it == 0))
            x_prev = np.array(R.x[:])
        self.x = x_prev

    def operators(self):
        rng = self.rng
        sel, nv, pl = self.setup
        z = np.zeros(6)
        minpl = 0.3
        H = orc.params_to_H(self.x)
        idx, dist = self.ctx.corr_match(H)
        self.log.append("corr_match -> rejects -> estimate_parameters")
        nn = orc.knn(self.M, self.F[sel], k=1, H=H)[0][:, 0]
        if not np.array_equal(idx, nn):
            self.bad.append(f"corr_match: {int(np.count_nonzero(idx != nn))} indices differ")
            return
        d = orc.point_to_plane(self.F[sel], nv, self.M[nn], H)
        if not np.array_equal(dist, d):
            self.bad.append("corr_match: distances differ")
        p2col = None if self.pl2 is None else self.pl2[idx]
        n_pl = self.ctx.corr_reject_planarity(minpl, pl, p2col)
        planarity = pl if p2col is None else np.where(p2col >= np.float32(minpl), pl, np.float32(np.nan))
        keep, n, med, mad = orc.reject(d, planarity, minpl)
        if n_pl != int(np.count_nonzero(planarity >= np.float32(minpl))):
            self.bad.append("corr_reject_planarity: count")
        if n_pl < 1:
            return
        gmed, gmad, gn = self.ctx.corr_reject_distances()
        if (gmed, gmad, gn) != (med, mad, n):
            self.bad.append(f"corr_reject_distances: {(gmed, gmad, gn)} vs {(med, mad, n)}")
        if n < 6:
            return
        R = self.ctx.estimate_parameters(self.x, z, z, 1.0)
        xo, _ = orc.solve(self.x, 1.0, z, z, self.F[sel], nv, self.M[nn], keep)
        if np.abs(np.array(R.x[:]) - xo).max() > 1e-9 * (1.0 + np.abs(xo).max()):
            self.bad.append(f"estimate_parameters: |x - oracle| = {np.abs(np.array(R.x[:]) - xo).max():.2e}")
        if rng.random() < 0.5:
            self.x = np.array(R.x[:])


# the order that round 5's stale-slot bug needed (40f67a5): a chained run through the filtered search, a NEW setup of the same size,
# an operator-route match (which makes "there is a previous match" true again), and a chained run on the new setup
SCRIPTED = ["icp_setup", "icp_run", "icp_setup", "operators", "icp_run", "upload_mov", "operators", "icp_run"]


def run_sequence(seed, steps=12):
    """(problems, log, match kernels seen).  Empty problems = every call agreed with the oracle."""
    from simpleicp_amd import _lib
    rng = np.random.default_rng(77_000 + seed)
    env = {"SICP_NN16_MIN_Q": "256", "SICP_NN16F_MIN_Q": "2048", "SICP_ORDER_MIN_Q": "1024", "SICP_NN16": "far" if seed % 2 else "near"}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx = _lib.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    m = Model(ctx, rng)
    try:
        m.upload(_lib.FIX)
        m.upload(_lib.MOV)
        ops = list(SCRIPTED) if seed % 8 == 0 else []
        while len(ops) < steps:
            ops.append(str(rng.choice(["upload_fix", "upload_mov", "transform", "set_planarity", "knn", "select_in_range", "icp_setup",
                                       "icp_iterate", "icp_run", "operators"], p=[0.04, 0.08, 0.08, 0.08, 0.1, 0.06, 0.14, 0.14, 0.16, 0.12])))
        for op in ops:
            if op in ("icp_iterate", "icp_run", "operators") and m.setup is None:
                m.icp_setup()
            if op == "icp_setup" and seed % 8 == 0 and m.setup is None:
                m.icp_setup(3000)                                  # (enough queries for the filtered search)
                continue
            if op == "icp_setup" and seed % 8 == 0 and m.setup is not None:
                # (the scripted order keeps the SIZE of the setup: that is what let the stale slots pass for current ones)
                sel_old = m.setup[0]
                sel = np.sort(rng.choice(len(m.F), len(sel_old), replace=False))
                nv, pl = ctx.estimate_normals(_lib.FIX, sel, 10)
                ctx.icp_setup(sel, nv, pl)
                m.setup, m.x = (sel, nv, pl), np.zeros(6)
                m.log.append(f"icp_setup Q={len(sel)} (same size, other points)")
                continue
            {"upload_fix": lambda: m.upload(_lib.FIX), "upload_mov": lambda: m.upload(_lib.MOV), "transform": m.transform,
             "set_planarity": m.set_planarity, "knn": m.knn, "select_in_range": m.select_in_range, "icp_setup": m.icp_setup,
             "icp_iterate": m.icp_iterate, "icp_run": m.icp_run, "operators": m.operators}[op]()
            if m.bad:
                break
    finally:
        ctx.close()
    return m.bad, m.log, m.kernels
