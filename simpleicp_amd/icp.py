"""SimpleICP -- call-compatible with /root/reference/python/simpleicp/simpleicp.py:41-379.

Same constructor, ``add_point_clouds``, ``run(**kwargs)`` signature, return tuple, exceptions,
side effects on the two PointCloud objects and log lines as the reference; the per-iteration
work (match, point-to-plane distances, planarity + MAD rejection, least-squares estimate) is
ONE C-ABI call into the HIP library with both clouds resident in HBM for the whole run.

Differences by design (documented in DESIGN.md):
  * the movable cloud is never transformed back and forth on the host (simpleicp.py:188,202):
    the transform is fused into the GPU scan, so the reference's ulp-level coordinate drift
    does not occur;
  * the NLLS problem of optimization.py:93-101 is minimised by Levenberg-Marquardt on fused
    6x6 normal-equation reductions instead of lmfit/scipy TRF with a finite-difference
    Jacobian -- same objective, same minimiser (tests pin H against the reference);
  * under torch.distributed (world_size > 1) the movable cloud is sharded across the GPUs.
"""
from __future__ import annotations

import logging
import time
from dataclasses import fields
from pathlib import Path
from typing import Optional, Tuple

import numpy as np

from . import _lib, backend, dist
from .pointcloud import _ALL, PointCloud
from .rbp import H_from_params, RigidBodyParameters

_log = logging.getLogger(__name__)
_PKG_LOG = logging.getLogger(__package__)
_ATTRS = ("nx", "ny", "nz", "planarity")


class SimpleICPException(Exception):
    """Raised when the SimpleICP class is misused (simpleicp.py:382)."""


def _enable_verbose_logging() -> None:
    """INFO to stdout, once (simpleicp.py:25-38)."""
    _PKG_LOG.setLevel(logging.INFO)
    for h in _PKG_LOG.handlers:
        if getattr(h, "_simpleicp_verbose", False):
            return
    h = logging.StreamHandler()
    h.setFormatter(logging.Formatter("%(message)s"))
    h._simpleicp_verbose = True
    _PKG_LOG.addHandler(h)


def _percent_change(new: float, old: float) -> float:
    if old == 0:
        return 0.0 if new == 0 else np.inf
    return abs((new - old) / old * 100)


class SimpleICP:
    def __init__(self, verbose: bool = True) -> None:
        self.pc1: Optional[PointCloud] = None
        self.pc2: Optional[PointCloud] = None
        self.last_run_info: dict = {}
        if verbose:
            _enable_verbose_logging()

    def add_point_clouds(self, pc_fix: PointCloud, pc_mov: PointCloud) -> None:
        self.pc1 = pc_fix      # fixed
        self.pc2 = pc_mov      # movable: gets transformed

    # --------------------------------------------------------------------------------------
    def run(
        self,
        correspondences: int = 1000,
        neighbors: int = 10,
        min_planarity: float = 0.3,
        max_overlap_distance: float = np.inf,
        min_change: float = 1.0,
        max_iterations: int = 100,
        distance_weights: Optional[float] = 1,
        rbp_observed_values: Tuple[float] = (0.0, 0.0, 0.0, 0.0, 0.0, 0.0),
        rbp_observation_weights: Tuple[float] = (0.0, 0.0, 0.0, 0.0, 0.0, 0.0),
        debug_dirpath: str = "",
    ) -> Tuple[np.ndarray, np.ndarray, RigidBodyParameters, np.ndarray]:
        """See the reference docstring (simpleicp.py:88-133): identical arguments/returns.
        Returns (H, X_mov_transformed, rbp, distance_residuals)."""
        self._check_arguments(distance_weights, rbp_observed_values, rbp_observation_weights)
        t_start = time.time()
        pc1, pc2 = self.pc1, self.pc2
        ctx = backend.get_context()
        ctx._corr_owner = None            # (an operator-level CorrPts object loses the device state to this run)
        import os
        sharded = dist.is_distributed() or (os.environ.get("SICP_FORCE_EXCHANGE") == "1" and dist.is_initialized())

        if debug_dirpath:
            _log.info(f'Write debug files to directory "{debug_dirpath}"')
            Path(debug_dirpath).mkdir(parents=True, exist_ok=True)

        obs = np.array(rbp_observed_values, dtype=float)
        obs[:3] = obs[:3] * np.pi / 180                       # degree -> rad (simpleicp.py:146-148)
        ow = np.array(rbp_observation_weights, dtype=float)
        H = H_from_params(obs)

        # both clouds go to HBM once and stay there (straight from the frames' storage: no host gather).  One process: each travels
        # on the library's helper thread (a stream of its own) while this thread does what needs no device -- the masks -- and, behind
        # the movable cloud's upload, the fixed cloud's grid and normals; the first call that names a slot waits for its upload
        pc1._upload(ctx, _lib.FIX, background=not sharded)
        # CorrPts.match searches pc2.X_selected only and maps the hits through pc2.idx_selected (corrpts.py:131-135);
        # a movable cloud with a partial `selected` mask (e.g. the fixed cloud of an earlier run) is uploaded as
        # that subset, after the overlap pre-pass, which looks at ALL its points (simpleicp.py:157: pc2.X)
        partial = not bool(pc2["selected"].to_numpy().all())
        msel = pc2.idx_selected if partial else None
        if partial and not len(msel):
            raise SimpleICPException("The movable point cloud has no selected points.")
        n_search = len(msel) if partial else pc2.num_points
        rank, world = dist.rank_world() if sharded else (0, 1)

        # what the ranks shard: index ranges of the movable cloud by default; the QUERIES (cloud
        # replicated on every rank) when the match dominates, i.e. for large correspondence counts
        # ... and only when a whole copy of the cloud (+ its grid) fits comfortably in every rank's free memory; a cloud that
        # only fits in shards stays sharded whatever the correspondence count (the verdict is the same on every rank of a
        # homogeneous node; SICP_PARTITION pins it explicitly)
        qshard = sharded and (os.environ.get("SICP_PARTITION", "") == "queries"
                              or (os.environ.get("SICP_PARTITION", "") != "cloud" and correspondences >= 100_000
                                  and dist.agree(dist.queries_partition_fits(ctx, n_search))))

        def upload_movable(rows=None):
            n = pc2.num_points if rows is None else len(rows)
            lo, hi = (0, n) if qshard else dist.shard_bounds(n, rank, world)
            pc2._upload(ctx, _lib.MOV, lo, hi, index_base=lo, rows=rows)

        sel0 = pc1._selection()           # carried along (_ALL or indices): every pass over the mask is a pass over N_f
        if sharded:
            upload_movable()
        else:
            pc2._upload(ctx, _lib.MOV, background=True)       # (waits for the fixed cloud's: one helper at a time)
            ctx.upload_wait(_lib.FIX)                         # ... whose verdict (a non-finite coordinate) is raised here
        self._job = {"ranks": world, "partition": None, "exchange": None}
        if sharded:
            how = dist.attach(ctx, gn_shard=(not qshard) and os.environ.get("SICP_GN_SHARD", "") != "0"
                              and (correspondences >= 262144 or os.environ.get("SICP_GN_SHARD") == "1"),
                              partition=_lib.PART_QUERIES if qshard else _lib.PART_CLOUD)
            self._job.update(partition="queries" if qshard else "cloud", exchange=how)
        else:
            dist.detach(ctx)
        try:
            return self._run_uploaded(ctx, sharded, msel, n_search, upload_movable, sel0, t_start, obs, ow, H,
                                      correspondences, neighbors, min_planarity, max_overlap_distance, min_change,
                                      max_iterations, distance_weights, debug_dirpath)
        except BaseException:
            # ANY way out of a sharded run that is not its normal end (a backend error, a host-side exception between two
            # collectives, KeyboardInterrupt, MemoryError) may leave this rank out of step with its peers: never revive the
            # communicator such a run used
            if sharded:
                dist.forget(ctx)
            raise
        finally:
            # the exchange lives on the process-wide context: a later standalone PointCloud operator must not issue a
            # collective the other ranks never join
            dist.detach(ctx)

    def _run_uploaded(self, ctx, sharded, msel, n_search, upload_movable, sel, t_start, obs, ow, H, correspondences, neighbors,
                      min_planarity, max_overlap_distance, min_change, max_iterations, distance_weights, debug_dirpath):
        pc1, pc2 = self.pc1, self.pc2
        if debug_dirpath:
            X_fix, X_mov = pc1.X, pc2.X

        if np.isfinite(max_overlap_distance):
            _log.info("Consider partial overlap of point clouds ...")
            if sel is _ALL or len(sel):
                # both clouds are resident already: only the verdicts cross the host link
                near = ctx.select_in_range(_lib.FIX, _lib.MOV, None if sel is _ALL else sel, H, float(max_overlap_distance))
                sel = pc1._keep_selected(sel, near)
            if not len(sel) > 0:
                raise SimpleICPException(
                    "Point clouds do not overlap within max_overlap_distance = "
                    f"{max_overlap_distance:.5f}! Consider increasing the value of "
                    "max_overlap_distance."
                )

        _log.info("Select points for correspondences in fixed point cloud ...")
        sel = pc1.select_n_points(correspondences, _cur=sel)
        # (simpleicp.py:174,254 save and restore pc1's selection around every iteration because the
        # reference's rejections edit it; here the masks live on the device and pc1 is never touched)

        if not set(_ATTRS).issubset(pc1.columns):
            _log.info("Estimate normals of selected points ...")
            pc1.estimate_normals(neighbors, _ctx=ctx, _uploaded=True, _sel=sel)
        normals, planarity = pc1._attributes_of(sel)
        ctx.upload_wait(_lib.MOV)                # (the movable cloud's verdict -- a non-finite coordinate -- is raised here)
        if msel is not None:
            upload_movable(msel)                 # from here on the searched cloud is pc2's selected subset
        if "planarity" in pc2.columns:
            # reject_wrt_planarity tests pc2's column as well when it exists (corrpts.py:158-163; NaN fails)
            rows, vals = pc2._planarity_pairs(msel)
            ctx.set_planarity(_lib.MOV, vals, rows=rows, n_global=n_search)
        ctx.icp_setup(sel, normals, planarity)

        x = obs.copy()
        w = distance_weights
        stats = []            # (n, mean, std) of the residuals per iteration
        R = None
        it = -1
        _log.info("Start iterations ...")
        def too_few(e):
            if e.code == _lib.ERR_TOO_FEW:
                raise SimpleICPException(str(e)) from None
            raise e

        # without debug dumps nothing on the host needs the intermediate states: the whole loop runs behind
        # ONE ABI call (sicp_icp_run, same convergence test) and the per-iteration log is replayed below
        whole, failed = None, None
        if not debug_dirpath:
            try:
                whole = ctx.icp_run(x, obs, ow, min_planarity, w, max_iterations, min_change)
            except _lib.BackendError as e:
                if e.code != _lib.ERR_TOO_FEW:
                    raise
                whole, failed = e.results[:-1], e          # log the iterations before the failing one first
        for it in range(0, max_iterations if whole is None else len(whole)):
            if debug_dirpath:
                if it == 0:
                    pc1.write_xyz(Path(debug_dirpath).joinpath(f"iteration{it:03d}_preoptim_pcfix.xyz"))
                self._write_cloud(Path(debug_dirpath).joinpath(f"iteration{it:03d}_preoptim_pcmov.xyz"), X_mov, H)
            x_start = x.copy()
            if whole is not None:
                R = whole[it]
            else:
                try:
                    R = ctx.icp_iterate(x, obs, ow, min_planarity, w)
                except _lib.BackendError as e:
                    too_few(e)
            if debug_dirpath:
                self._write_correspondences(ctx, Path(debug_dirpath).joinpath(
                    f"iteration{it:03d}_preoptim_correspondences.xyz"), X_fix, X_mov if msel is None else X_mov[msel], sel, H)
            if w is None:
                w = R.weight_used                            # frozen after iteration 0 (simpleicp.py:229-234)
            x = np.array(R.x[:])
            H = np.array(R.H[:]).reshape(4, 4)
            stats.append((int(R.n_kept), R.res_mean, R.res_std))

            if it > 0 and self._converged(stats[it], stats[it - 1], min_change):
                _log.info("Convergence criteria fulfilled -> stop iteration!")
                break

            if it == 0:
                _log.info(f"{'Iteration':>9s} | {'correspondences':>15s} | {'mean(residuals)':>15s} | "
                          f"{'std(residuals)':>15s}")
                _log.info(f"{'orig:0':>9s} | {int(R.n_kept):15d} | {R.dist_mean:15.4f} | {R.dist_std:15.4f}")
            _log.info(f"{it + 1:9d} | {stats[it][0]:15d} | {stats[it][1]:15.4f} | {stats[it][2]:15.4f}")

        if failed is not None:
            too_few(failed)

        rbp = RigidBodyParameters()
        rbp.set_parameter_attributes_from_list("observed_value", list(obs))
        rbp.set_parameter_attributes_from_list("observation_weight", list(ow))
        residuals = np.empty(0)
        if R is not None:
            rbp.set_parameter_attributes_from_list("initial_value", list(x_start))
            rbp.set_parameter_attributes_from_list("estimated_value", list(x))
            sigma = ctx.icp_uncertainties()
            for name, s, free in zip(("alpha1", "alpha2", "alpha3", "tx", "ty", "tz"), sigma, np.isfinite(ow)):
                if free:
                    getattr(rbp, name).estimated_uncertainty = float(s)
            _, _, keep, res = ctx.icp_state(pc2_idx=False, dist=False)
            residuals = res[keep]

        self._log_result(H, rbp)

        # final transformation of the caller's movable cloud (simpleicp.py:316): all of its points
        if sharded or msel is not None:
            pc2._upload(ctx, _lib.MOV)
        X_new = pc2._transform(H, ctx, _lib.MOV)
        if debug_dirpath:
            pc2.write_xyz(Path(debug_dirpath).joinpath(f"iteration{it:03d}_postoptim_pcmov.xyz"))

        self.last_run_info = {"iterations": it + 1, "stats": stats, "seconds": time.time() - t_start, **getattr(self, "_job", {})}
        if sharded:
            # how the shards' winners met: "records_allgather" / "key_allreduces" (cloud shards) / "query_slices", and how often
            xi = ctx.exchange_info()
            self.last_run_info.update(winner_exchange=xi["form"], exchanges=xi["count"])
        _log.info(f"Finished in {time.time() - t_start:.3f} seconds!")
        return H, X_new, rbp, residuals

    # --------------------------------------------------------------------------------------
    @staticmethod
    def _converged(new, old, min_change) -> bool:
        """simpleicp.py:356-379 on (n, mean, std) triples."""
        return (_percent_change(new[1], old[1]) < min_change) and (_percent_change(new[2], old[2]) < min_change)

    @staticmethod
    def _write_cloud(file, X, H):
        Xh = np.column_stack((X, np.ones(len(X))))
        Xt = (H @ Xh.T).T[:, :3]
        from . import io
        io.write_xyz(file, Xt, decimals=3, header="//X Y Z")

    @staticmethod
    def _write_correspondences(ctx, file, X_fix, X_mov, sel, H):
        """corrpts.py:213-237: kept correspondences, movable points in the pre-optimisation pose."""
        idx, d, keep, _ = ctx.icp_state(residual=False)
        p2 = X_mov[idx[keep]]
        p2 = (H @ np.column_stack((p2, np.ones(len(p2)))).T).T[:, :3]
        from . import io
        io.write_xyz(file, np.column_stack((X_fix[sel[keep]], p2, d[keep])), decimals=-1,
                     header="//X1 Y1 Z1 X2 Y2 Z2 point_to_plane_distance")

    @staticmethod
    def _log_result(H, rbp):
        _log.info("Estimated transformation matrix H:")
        for r in range(4):
            _log.info(f"[{H[r, 0]:12.6f} {H[r, 1]:12.6f} {H[r, 2]:12.6f} {H[r, 3]:12.6f}]")
        _log.info("... which corresponds to the following rigid-body transformation parameters:")
        _log.info(f"{'parameter':>9s} | {'est.value':>15s} | {'est.uncertainty':>15s} | {'obs.value':>15s} | "
                  f"{'obs.weight':>15s}")
        for f in fields(rbp):
            p = getattr(rbp, f.name)
            _log.info(f"{f.name:>9s} | {p.estimated_value_scaled:15.6f} | {p.estimated_uncertainty_scaled:15.6f} | "
                      f"{p.observed_value_scaled:15.6f} | {p.observation_weight:15.3e}")
        _log.info("(Unit of est.value, est.uncertainty, and obs.value for alpha1/2/3 is degree)")

    @staticmethod
    def _check_arguments(distance_weights, rbp_observed_values, rbp_observation_weights):
        """simpleicp.py:327-353 -- same checks, same messages."""
        if distance_weights is not None and distance_weights <= 0:
            raise SimpleICPException("distance_weights must be > 0.")
        if len(rbp_observed_values) != 6:
            raise SimpleICPException("rbp_observed_values must have exactly 6 elements.")
        if len(rbp_observation_weights) != 6:
            raise SimpleICPException("rbp_observation_weights must have exactly 6 elements.")
        if not all(w >= 0 for w in rbp_observation_weights):
            raise SimpleICPException("All elements of rbp_observation_weights must be >= 0.")
        if not any(np.isfinite(rbp_observation_weights)):
            raise SimpleICPException("At least one element in rbp_observation_weights must be finite.")
