#!/usr/bin/env python3
"""bench.py -- ICP iterations/s (+ kNN correspondences/s) on the BASELINE.json workloads.

    python bench.py                          # C4, 1 GPU, 20 steps, 3 warm-up steps
    python bench.py --config C3              # C1 | C2 | C3 | C4 | C5size | T
    python bench.py --gpus 2                 # spawns the ranks itself when not started by torchrun
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json `configs`):
  C1      Dragon (dragon1 vs dragon2), correspondences=1000                     bundled data
  C2      Bunny partial overlap, max_overlap_distance=1, correspondences=1000   bundled data
  C3      1.34 M-vs-1.34 M synthetic stand-in for the airborne pair (the files are missing upstream), Q = 10 000
  C4      synthetic 10 M-vs-10 M surface (two independent samplings + known rigid perturbation), Q = 1000
          -- the configuration the metric is quoted on, and the default
  C5size  100 M-vs-100 M, Q = 1 M on ONE GPU (the 8-GPU config's sizes; ~12 GB of the 288 GB)
  T       1.25 M-vs-1.25 M terrestrial-scan-like stand-in (density ~ 1/r^2: the reference's Terrestrial Lidar pair is missing upstream), Q = 10 000

A "step" is ONE full ICP iteration through the C ABI (exact 1-NN match of the Q selected fixed points in
the movable cloud under the current H, point-to-plane distances, planarity + raw-MAD rejection, Levenberg-
Marquardt solve of the reference's objective).  Both clouds, the selected points and their normals are resident
in HBM before the timed region.  The timed region is K consecutive iterations of a run that starts COLD at the
initial pose (`sicp_icp_setup` was just called: no search bound from an earlier match; min_change = 0 so no step
is skipped), bracketed by barrier + device synchronisation, MAX over ranks, with kernel timing events OFF.  It
is repeated `--repeats` times (same start state every time); `value` is K / median(elapsed), the spread is in
`repeat_stats`.  Kernel splits and the roofline object come from a separate, instrumented pass of the same K
steps (HIP events on the library's own stream).

N > 1: STRONG scaling -- the same movable cloud is sharded by index range over the ranks (one process per GPU),
one all-gather exchange per iteration.

One JSON line on stdout (rank 0) -- compact (about 5 KB: every field and number, no explanatory strings); the unabridged record
goes to --out FILE (profiles/r3/bench_*.json are such files).  Extra objects on the same line:
  roofline             the kernel with the largest share of the step's GPU time; `achieved` = algorithmic bytes per
                       launch / average launch duration
  roofline_match       the 1-NN kernel of the default path (pruned grid search), priced on the bytes the pruned
                       search itself needs (candidates x 32-byte records + cell offsets + queries), with `pruning_ratio`
  roofline_bruteforce  the north-star brute-force scan, measured in a short extra leg on the same inputs
  throughput_point     the same clouds with 100 000 correspondences (SURVEY 8d's throughput point) and, _q1000000, with 1 M: K steps
                       from cold, kernel split, the search's roofline on its own tallied bytes, an oracle leg (N > 1: query shards)
  comm                 (when an exchange ran) backend, rank count as RCCL counts it, partition, shard rows, exchange us per iteration
  parity               two more iterations after the timed region, checked against the CPU oracle
  setup                upload / grid build / normals, each once per run() -- outside `value`, reported
  run_end_to_end       a real run(): cold, min_change = 1, setup included
  cpu_baseline         the reference's algorithm (oracle/ref_port.py) on this box's host cores, bounded sample
  cpu_reference        the UNMODIFIED reference timed in the build container (profiles/cpu_reference.json)
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_VALU_PEAK_TFLOPS = 157.3    # vector FP32 peak (spec) = FP32 MFMA peak

CONFIGS = {
    "C1": dict(kind="dataset", fix="dragon1", mov="dragon2", Q=1000, k=10, kwargs={}),
    "C2": dict(kind="dataset", fix="bunny_part1", mov="bunny_part2", Q=1000, k=10, kwargs={"max_overlap_distance": 1.0}),
    "C3": dict(kind="synthetic", n=1_340_000, Q=10_000, k=10, kwargs={}),
    "C4": dict(kind="synthetic", n=10_000_000, Q=1000, k=10, kwargs={}),
    "C5size": dict(kind="synthetic", n=100_000_000, Q=1_000_000, k=10, kwargs={}),
    "T": dict(kind="terrestrial", n=1_250_000, Q=10_000, k=10, kwargs={}),
}


def synthetic_pair(n, seed_fix=0, seed_mov=1):
    """SURVEY.md section 8(d) generator (pinned): 10 pts/m^2 surface, independent samplings,
    centroid removed, movable = H_true^-1 applied."""
    L = np.sqrt(n / 10.0)

    def sample(seed):
        rng = np.random.default_rng(seed)
        x = rng.uniform(0, L, n)
        y = rng.uniform(0, L, n)
        z = (20 * np.sin(2 * np.pi * x / 200) * np.cos(2 * np.pi * y / 300)
             + 5 * np.sin(2 * np.pi * x / 37 + 1) * np.sin(2 * np.pi * y / 53) + rng.normal(0, 0.02, n))
        return np.column_stack((x, y, z))

    Xf, Xm = sample(seed_fix), sample(seed_mov)
    c = Xf.mean(axis=0)
    Xf -= c
    Xm -= c
    from simpleicp_amd.rbp import H_from_params
    x_true = np.array([np.deg2rad(0.5), np.deg2rad(-0.3), np.deg2rad(0.8), 0.30, -0.20, 0.10])
    H_true = H_from_params(x_true)
    Hinv = np.linalg.inv(H_true)
    Xm = Xm @ Hinv[:3, :3].T + Hinv[:3, 3]
    return np.ascontiguousarray(Xf), np.ascontiguousarray(Xm), H_true


def terrestrial_pair(n, seed_fix=10, seed_mov=11):
    """A terrestrial-laser-scan-like pair: a STAND-IN for the reference's Terrestrial Lidar set (README.md:174, 1250 k points each;
    tests/test_simpleicp.py:54-63), whose files are missing upstream (.MISSING_LARGE_BLOBS).  One scene -- ground plane, the four
    walls of a yard, a few block-shaped buildings -- scanned from two nearby stand points with uniform ANGULAR sampling, so the point
    density on a surface falls like 1 / r^2 (x the incidence angle) between 2 m and 80 m: a few thousand points per m^2 at the scanner's
    feet, a handful at the far walls.  Each cloud is given in its own scanner frame; H_true maps the movable frame to the fixed one."""
    boxes = np.array([  # xmin, xmax, ymin, ymax, zmax (from the ground up)
        [-60.0, 60.0, 44.0, 45.0, 14.0], [-60.0, 60.0, -45.0, -44.0, 14.0], [59.0, 60.0, -45.0, 45.0, 14.0], [-60.0, -59.0, -45.0, 45.0, 14.0],
        [12.0, 24.0, 8.0, 20.0, 9.0], [-30.0, -18.0, -22.0, -6.0, 6.0], [-14.0, -8.0, 14.0, 30.0, 11.0], [30.0, 36.0, -30.0, -12.0, 4.0],
        [3.0, 4.2, -6.0, -4.8, 2.4],
    ])

    def scan(origin, yaw, seed, want):
        rng = np.random.default_rng(seed)
        out = []
        have = 0
        while have < want:
            m = int((want - have) * 1.6) + 1024
            az = rng.uniform(0, 2 * np.pi, m)
            el = rng.uniform(np.deg2rad(-55.0), np.deg2rad(35.0), m)
            d = np.column_stack((np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)))
            t = np.full(m, np.inf)
            with np.errstate(divide="ignore", invalid="ignore"):
                tg = np.where(d[:, 2] < 0, -origin[2] / d[:, 2], np.inf)                      # the ground, z = 0
                t = np.minimum(t, tg)
                for b in boxes:                                                                   # slab test per block
                    lo = np.array([b[0], b[2], 0.0]); hi = np.array([b[1], b[3], b[4]])
                    t0 = (lo - origin) / d; t1 = (hi - origin) / d
                    tn = np.nanmax(np.minimum(t0, t1), axis=1); tf = np.nanmin(np.maximum(t0, t1), axis=1)
                    hit = (tn <= tf) & (tn > 0)
                    t = np.where(hit, np.minimum(t, tn), t)
            ok = (t >= 2.0) & (t <= 80.0)
            P = origin + d[ok] * (t[ok] + rng.normal(0, 0.004, ok.sum()))[:, None]             # range noise, 4 mm
            out.append(P); have += len(P)
        P = np.concatenate(out)[:want]
        c, s_ = np.cos(yaw), np.sin(yaw)
        R = np.array([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]])
        return (P - origin) @ R, R                                                               # scanner frame: R^T (P - origin)

    of, om = np.array([0.0, 0.0, 1.8]), np.array([0.45, -0.30, 1.85])
    Xf, _ = scan(of, 0.0, seed_fix, n)
    Xm, Rm = scan(om, np.deg2rad(1.0), seed_mov, n)
    H_true = np.eye(4)
    H_true[:3, :3] = Rm
    H_true[:3, 3] = om - of
    return np.ascontiguousarray(Xf), np.ascontiguousarray(Xm), H_true


def load_workload(name, points=None, correspondences=None):
    """(Xf, Xm, H_true or None, Q, k, kwargs, description)"""
    cfg = dict(CONFIGS[name])
    Q = correspondences or cfg["Q"]
    if cfg["kind"] == "dataset":
        def cloud(stem):
            return np.load(ROOT / "tests" / "golden" / "data" / f"{stem}.npz")["q"].astype(np.float64) / 1e4
        Xf, Xm = cloud(cfg["fix"]), cloud(cfg["mov"])
        desc = f"{name} {cfg['fix']} vs {cfg['mov']} (bundled data, {len(Xf)} / {len(Xm)} points)"
        return Xf, Xm, None, Q, cfg["k"], cfg["kwargs"], desc
    n = points or cfg["n"]
    if cfg["kind"] == "terrestrial":
        Xf, Xm, H_true = terrestrial_pair(n)
        desc = (f"{name} terrestrial-scan-like {n}-vs-{n}: ground + walls + blocks, uniform angular sampling from two stand points "
                "(density ~ 1/r^2 over 2-80 m) -- a STAND-IN for the reference's Terrestrial Lidar pair (missing upstream)")
        return Xf, Xm, H_true, Q, cfg["k"], cfg["kwargs"], desc
    Xf, Xm, H_true = synthetic_pair(n)
    desc = f"{name} synthetic {n}-vs-{n} surface (SURVEY 8d generator)"
    if name == "C3":
        desc += " -- a STAND-IN: the airborne_lidar1/2.xyz files the config names are missing upstream (.MISSING_LARGE_BLOBS)"
    return Xf, Xm, H_true, Q, cfg["k"], cfg["kwargs"], desc


GRID_KERNELS = ("k_grid_nn", "k_grid_nn16", "k_grid_nn16f")


def match_bytes(kernel, per, nq):
    """Bytes the pruned search itself needs per launch, from its own tallies.  Exact kernels: a 32-B record per
    candidate, two 4-B offsets per non-empty grid row, per query its coordinates, the previous match and the 48-B result.  Filtered
    kernel (k_grid_nn16f): a 16-B float record per candidate, the same offsets, per query its 32-B slot record, the 32-B bound it reads
    and the 32-B bound it leaves, the winner's 32-B record and the 40-B result."""
    if kernel == "k_grid_nn16f":
        # (an approximation from below on data with many ties: the queries the filter leaves to the exact kernel -- `deferred`, 0.1-0.5 % on
        # the bench clouds, ALL of them on lattice data -- add their 32-byte candidates to the same tallies and are priced at 16 here)
        return per["candidates"] * 16 + per["rows"] * 8 + nq * (32 + 32 + 32 + 32 + 40)
    return per["candidates"] * 32 + per["rows"] * 8 + nq * (24 + 24 + 48)


def latency_model(steady_us, from_cold_us, tail_cycles):
    """The bound a latency-bound iteration (Q <= 2048: match kernel + ONE single-workgroup tail) can be held against -- its HBM
    roofline fraction (1e-3) says nothing.  Three terms, priced from MI355X_MICROARCH.md:
      * dependent kernel boundaries: match -> tail -> next match, 2 per iteration x 1.45 us ("boundary" row of the price table);
      * dependent global-memory round trips on the critical path, counted from the code, x the HBM-miss latency (~900 cycles at
        2.4 GHz = 0.375 us): the match has 3 (query + previous match + loop state; cell offsets; records), the tail 2 (the distances and
        verdicts the match left, with its other operands in flight beside them; the record it writes to host memory, waited for at the
        kernel's end);
      * the tail's on-chip phases -- median + MAD selection, keep mask, the minimisation, residual statistics -- from its OWN clock
        (sicp_tail_cycles; its first phase, loading, is the round trip already counted) at the 2.4 GHz peak clock.
    frac_of_floor = floor / measured steady iteration: what is NOT in the floor is the match's own arithmetic, launch ramps of 250
    workgroups, clocks below peak, and host jitter."""
    clock_ghz = 2.4
    boundaries, boundary_us = 2, 1.45
    trips = {"match": 3, "tail": 2}
    trip_us = 900.0 / (clock_ghz * 1e3)
    onchip_cycles = tail_cycles["select"] + tail_cycles["keep"] + tail_cycles["lm"] + tail_cycles["final"]
    floor = boundaries * boundary_us + sum(trips.values()) * trip_us + onchip_cycles / (clock_ghz * 1e3)
    return {"floor_us": floor, "steady_us": steady_us, "frac_of_floor": floor / steady_us if steady_us else None,
            "from_cold_us_per_step": from_cold_us,
            "terms_us": {"kernel_boundaries": boundaries * boundary_us, "global_round_trips": sum(trips.values()) * trip_us,
                         "tail_on_chip": onchip_cycles / (clock_ghz * 1e3)},
            "dependent_kernel_boundaries": boundaries, "boundary_us": boundary_us, "dependent_global_round_trips": trips,
            "round_trip_us": trip_us, "tail_cycles": tail_cycles, "clock_ghz": clock_ghz,
            "note": "floor = 2 kernel boundaries + 5 dependent HBM round trips + the tail's on-chip phases (its own cycle counters); "
                    "steady = wall clock per iteration over 30 iterations after the estimate has settled"}


def csrc_hash():
    """sha256 over the kernel sources (name + bytes, sorted): what a committed PMC summary is valid for."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted((ROOT / "simpleicp_amd" / "csrc").iterdir()):
        if f.suffix in (".hip", ".cpp", ".h"):
            h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


def load_pmc():
    """profiles/latest_pmc.json (HBM bytes per launch and kernel from separate rocprofv3 --pmc passes, written by
    scripts/summarize_profile.py) -- only if it was measured on THESE kernel sources; a stale file would describe other kernels."""
    pmc_file = ROOT / "profiles" / "latest_pmc.json"
    if not pmc_file.exists():
        return {}, None
    pmc = json.loads(pmc_file.read_text())
    if pmc.get("_csrc_hash") != csrc_hash():
        return {}, (f"profiles/latest_pmc.json is stale (measured on csrc {pmc.get('_csrc_hash')}, this tree is {csrc_hash()}): "
                    "traffic withheld")
    return pmc, f"profiles/latest_pmc.json ({pmc.get('_note', 'separate rocprofv3 --pmc passes')}); NOT collected in this run"


def iterate(ctx, n_it, x, obs, ow):
    """n_it iterations of the hot path behind one ABI call (sicp_icp_run; min_change=0 never converges early)."""
    if n_it <= 0:
        return x, 0, None
    res = ctx.icp_run(x, obs, ow, 0.3, 1.0, max_iterations=n_it, min_change=0.0)
    assert len(res) == n_it
    return np.array(res[-1].x[:]), sum(r.ne_evals for r in res), res[-1]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C4")
    ap.add_argument("--points", type=int, default=None, help="override the synthetic cloud size")
    ap.add_argument("--correspondences", type=int, default=None)
    ap.add_argument("--repeats", type=int, default=50, help="repetitions of the K-step timed run (median reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bruteforce-leg", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-work-pass", action="store_true",
                    help="skip the pass in which the grid search tallies its own candidates / rows (in-kernel atomics: a kernel "
                         "trace taken with it averages slower launches in) -- the match's roofline then falls back to PMC traffic")
    ap.add_argument("--cpu-iterations", type=int, default=2)
    ap.add_argument("--partition", choices=["auto", "cloud", "queries"], default="auto",
                    help="N > 1: shard the movable cloud (north-star scheme, default below 1e5 correspondences) or the "
                         "queries (cloud replicated, default from 1e5 correspondences)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="register the multi-GPU exchange even with one rank (measures its overhead on one GPU)")
    ap.add_argument("--throughput-q", type=str, default="100000,1000000",
                    help="correspondence counts of the `throughput_point` legs (synthetic configs; same clouds, same K steps "
                         "from cold; N > 1: under query shards); 0 = none")
    ap.add_argument("--throughput-repeats", type=int, default=7)
    ap.add_argument("--out", type=str, default=None, help="also write the UNABRIDGED record to this file")
    ap.add_argument("--verbose-line", action="store_true", help="print the unabridged record on stdout as well")
    return ap.parse_args(argv)


def _spawn_entry(local_rank, world, port, argv):
    os.environ.update({"RANK": str(local_rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    run(parse_args(argv))


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: start one process per GPU ourselves (same rendezvous the driver's launcher uses)
        import socket
        import torch.multiprocessing as mp
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(_spawn_entry, args=(args.gpus, port, sys.argv[1:]), nprocs=args.gpus, join=True)
        return
    run(args)


def run(args):
    import torch
    import torch.distributed as td

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU path)", file=sys.stderr)
        sys.exit(2)
    # SICP_BENCH_SHARE_GPU=1 (tests only): every rank on cuda:0, a gloo group with host-staged collectives -- the multi-rank
    # flow of this file on a one-GPU box (RCCL refuses two ranks on one device); never a measurement
    share_gpu = os.environ.get("SICP_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} has no GPU (only {torch.cuda.device_count()} visible)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    exchange = world > 1 or args.force_exchange
    if exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share_gpu:
            td.init_process_group("gloo", rank=rank, world_size=world)
        else:
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from simpleicp_amd import _lib, dist

    Xf, Xm, H_true, Q, k, kw, desc = load_workload(args.config, args.points, args.correspondences)
    Nf, Nm = len(Xf), len(Xm)
    qshard = exchange and (args.partition == "queries" or (args.partition == "auto" and Q >= 100_000))
    ctx = _lib.Context(local_rank)
    setup = {}
    t0 = time.perf_counter()
    ctx.upload(_lib.FIX, Xf)
    lo, hi = (0, Nm) if qshard else dist.shard_bounds(Nm, rank, world)
    ctx.upload(_lib.MOV, Xm[lo:hi], index_base=lo)
    setup["upload_ms"] = (time.perf_counter() - t0) * 1e3
    transport = None
    if exchange:
        transport = dist.attach(ctx, gn_shard=(not qshard) and Q >= 262144,
                                partition=_lib.PART_QUERIES if qshard else _lib.PART_CLOUD)

    # selection: partial-overlap pre-pass (simpleicp.py:155-170) + select_n_points (pointcloud.py:132-147)
    sel = np.arange(Nf)
    if np.isfinite(kw.get("max_overlap_distance", np.inf)):
        t0 = time.perf_counter()
        sel = sel[ctx.select_in_range(_lib.FIX, _lib.MOV, None, np.eye(4), kw["max_overlap_distance"])]
        setup["overlap_prepass_ms"] = (time.perf_counter() - t0) * 1e3
    if len(sel) > Q:
        sel = np.unique(sel[np.round(np.linspace(0, len(sel) - 1, Q)).astype(np.int64)])
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    normals, planarity = ctx.estimate_normals(_lib.FIX, sel, k)
    setup["normals_ms"] = (time.perf_counter() - t0) * 1e3          # includes the fixed cloud's grid build
    knnk = ctx.timing()["knnk_scan"]
    ctx.timing_enable(False)
    obs, ow = np.zeros(6), np.zeros(6)

    def cold():
        """state a run() starts from: selection declared, no search bound from an earlier match"""
        ctx.icp_setup(sel, normals, planarity)

    # first contact builds the movable cloud's grid: time it separately (one iteration from cold, then again)
    cold()
    t0 = time.perf_counter()
    iterate(ctx, 1, obs.copy(), obs, ow)
    first_ms = (time.perf_counter() - t0) * 1e3
    cold()
    t0 = time.perf_counter()
    iterate(ctx, 1, obs.copy(), obs, ow)
    setup["grid_build_ms"] = max(0.0, first_ms - (time.perf_counter() - t0) * 1e3)
    iterate(ctx, args.warmup, obs.copy(), obs, ow)                  # untimed warm-up steps

    def timed_run():
        cold()
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, ne_evals, last = iterate(ctx, args.steps, obs.copy(), obs, ow)
        torch.cuda.synchronize()
        if world > 1:
            td.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
            td.all_reduce(t, op=td.ReduceOp.MAX)
            el = float(t.item())
        return el, x, ne_evals, last

    times = []
    for _ in range(max(1, args.repeats)):
        el, x, ne_evals, last = timed_run()
        times.append(el)
    times = np.array(times)
    elapsed = float(np.median(times))

    # steady state (a run's iterations from about its tenth on: the estimate has stopped moving, one LM step per iteration): 30 more
    # iterations from where the K timed ones ended, events off, wall clock -- what the latency model below is held against
    steady_us = None
    if world == 1:
        xs, _, _ = iterate(ctx, 12, x.copy(), obs, ow)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iterate(ctx, 30, xs, obs, ow)
        torch.cuda.synchronize()
        steady_us = (time.perf_counter() - t0) / 30 * 1e6
    tail_cycles = ctx.tail_cycles() if len(sel) <= 2048 else None

    # instrumented pass: the same K steps with HIP events around every kernel class (perturbs the step, so it is
    # not the timed run)
    ctx.timing_enable(True)
    ctx.timing_reset()
    cold()
    iterate(ctx, args.steps, obs.copy(), obs, ow)
    timing = ctx.timing()
    match_kernel = ctx.last_match_kernel()
    # ... and one more with the grid search tallying the candidates / rows it touches (the bytes its roofline is priced on)
    work = None
    if match_kernel in GRID_KERNELS and not args.no_work_pass:
        ctx.timing_enable(True, count_work=True)
        ctx.timing_reset()
        cold()
        iterate(ctx, args.steps, obs.copy(), obs, ow)
        work = ctx.match_work()
    ctx.timing_enable(False)

    # parity leg, device side (every rank takes part in the exchange; only rank 0 consults the oracle)
    parity_rec = None if args.no_parity else parity_device(ctx, sel, normals, planarity, obs, ow)
    comm = comm_record(ctx, exchange, transport, qshard, world, hi - lo, len(sel), timing, args.steps)

    # throughput legs ("additionally Q = 100 k for a throughput point"): same clouds, Q large enough that the
    # machine is full; N > 1: query shards (the whole cloud on every rank, Q / N queries each)
    tps = []
    tq = [int(v) for v in args.throughput_q.split(",") if v.strip() and int(v) > 0] if H_true is not None else []
    for Qt in tq:
        if Qt == len(sel) or Qt > Nf:
            continue
        tps.append(throughput_point(args, ctx, Xf, Xm, Qt, k, rank, world, exchange, share_gpu))

    if rank != 0:
        if exchange:
            ctx.close()
            td.destroy_process_group()
        return

    H = _lib.params_to_H(x)
    n_local, nq = hi - lo, len(sel)
    if qshard:
        nq_local = (nq + world - 1) // world                     # queries this rank's match launch handles
    else:
        nq_local = nq
    avg = {kname: v["ms"] / max(1, v["launches"]) for kname, v in timing.items()}
    match_ms, solve_ms, select_ms = avg["match"], avg["solve"], avg["reject_select"]
    fused = nq <= 2048
    evals_per_it = ne_evals / args.steps
    bytes_bruteforce = n_local * 24 + nq * (24 + 16)        # read the searched cloud once + queries + (idx, d2)
    # 72 B per kept correspondence and evaluation + the median / MAD passes (n * 8 B x 3) when they are in the launch
    bytes_solve = int(last.n_kept) * 72 * evals_per_it + (nq * 8 * 3 if fused else 0)
    pmc, pmc_src = load_pmc()

    def roof(kernel, ms, bytes_alg, note, extra=None):
        if bytes_alg is None:                    # nothing to price this launch on in this run (no tallies, no counters): say so
            d = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                 "traffic_source": None, "kernel": kernel, "avg_ms": ms, "bytes_alg_per_launch": None, "note": note}
            d.update(extra or {})
            return d
        ach = bytes_alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        # counters are keyed by kernel for the default line and "<kernel>@<config>" for the other configurations' own passes
        traffic = pmc.get(f"{kernel}@{args.config}") if args.config != "C4" else pmc.get(kernel)
        d = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
             "traffic": traffic, "traffic_source": pmc_src if (traffic is not None or not pmc) else None,
             "kernel": kernel, "avg_ms": ms, "bytes_alg_per_launch": int(bytes_alg), "note": note}
        if traffic is not None and ms > 0:
            d["frac_on_pmc_traffic"] = traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS        # the counters' bytes over the same time
        if extra:
            d.update(extra)
        return d

    if match_kernel in GRID_KERNELS:
        # the pruned search's OWN bytes: every candidate it evaluates is one packed 32-B record (x, y, z, index),
        # every non-empty grid row two 4-B offsets, every query its coordinates, the previous match (bound) and the
        # 48-B result -- tallied by the kernel itself in a separate pass
        if work and work["launches"]:
            per = {kk: v / max(1, work["launches"]) for kk, v in work.items() if kk != "launches"}
            bytes_match = match_bytes(match_kernel, per, nq)
            extra = {"candidates_per_query": per["candidates"] / nq, "grid_rows_per_query": per["rows"] / nq}
        else:
            # --no-work-pass (the kernel-trace run of scripts/gpu_profile.sh): the counters' bytes if this tree has them -- never the
            # brute-force figure, which a pruned search does not read (it would price the launch above the HBM peak)
            bytes_match = pmc.get(match_kernel)
            extra = {"bytes_alg_source": "PMC traffic (no in-kernel tallies in this run)" if bytes_match else
                     "none: no in-kernel tallies in this run (--no-work-pass) and no counter summary of this tree"}
        extra["pruning_ratio"] = bytes_bruteforce / max(1.0, bytes_match) if bytes_match else None
        extra["bytes_bruteforce_per_launch"] = int(bytes_bruteforce)
        r_match = roof(match_kernel, match_ms, bytes_match,
                       ("exact 1-NN on a static uniform grid, one wave per query: reads only the cells the bound ball touches "
                        "(pruning_ratio = brute-force algorithmic bytes / these); ~4 dependent memory round trips per "
                        "query, i.e. latency-bound, not bandwidth-bound") if match_kernel == "k_grid_nn" else
                       ("exact 1-NN on a static uniform grid, four cell-ordered queries per wave: reads only the cells the bound "
                        "ball touches (pruning_ratio = brute-force algorithmic bytes / these); VALU-issue- and latency-bound"), extra)
    else:
        r_match = roof(match_kernel, match_ms, bytes_bruteforce,
                       "brute-force Q x N scan: VALU-bound by construction (SURVEY 8d), cloud read once")
    tail_kernel = "k_icp_tail" if fused else ("k_lm_eval" if exchange else "k_lm_all")
    r_solve = roof(tail_kernel, solve_ms, bytes_solve,
                   "everything after the match in ONE single-workgroup launch (distances, MAD rejection, LM with "
                   "device-side 6x6 solves): ~1000 correspondences = latency-bound on one CU by design, not bandwidth"
                   if fused else "the iteration's solver launches together (enqueued evaluations + finish): residual + Jacobian rows + 8x8 "
                   "Gram on the FP64 matrix pipe, 72 B/correspondence/evaluation; the last block of an evaluation advances the solver")
    solve_total = solve_ms
    dominant = r_solve if solve_total >= match_ms else r_match

    out = {
        "metric": "ICP iterations/sec (kNN correspondences/sec in `correspondences_per_s`), 10M-vs-10M pts",
        "value": args.steps / elapsed,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic" if H_true is not None else "bundled reference data sets (tests/golden/data)",
        "config": {"workload": f"{desc}, correspondences={Q}, neighbors={k}, exact 1-NN (bit-identical to brute force)",
                   "name": args.config, "n_fixed": Nf, "n_movable": Nm, "correspondences": nq, "neighbors": k,
                   "parallelism": (f"query shards x{world}, movable cloud replicated" if qshard else
                                   f"movable-cloud index shards x{world}, queries replicated")
                                  + (f"; collectives: {transport}" if transport else "")},
        "correspondences_per_s": nq * args.steps / elapsed,
        "repeat_stats": {"repeats": len(times), "ms_per_step_median": elapsed / args.steps * 1e3,
                         "ms_per_step_p10": float(np.percentile(times, 10)) / args.steps * 1e3,
                         "ms_per_step_p90": float(np.percentile(times, 90)) / args.steps * 1e3,
                         "ms_per_step_min": float(times.min()) / args.steps * 1e3,
                         "ms_per_step_first": float(times[0]) / args.steps * 1e3,
                         "timed_region": "K steps from the cold state (icp_setup just called), min_change=0, events off"},
        "roofline": dominant,
        "roofline_match": r_match,
        "kernels_instrumented": {name: {"avg_ms": avg[name], "launches": timing[name]["launches"]} for name in timing},
        "gpu_ms_per_step_instrumented": match_ms + solve_total + select_ms,
        "setup": {**setup, "normals_knn_kernel_ms": knnk["ms"],
                  # the normals' k-NN as Q * k neighbours per second of its search kernel
                  "normals_knn_neighbours_per_s": (nq * k / (knnk["ms"] * 1e-3)) if knnk["ms"] > 0 else None,
                  "note": "once per run(), outside `value`: host->HBM upload of both clouds (pageable memory), "
                          "estimate_normals for the Q selected points (with the fixed cloud's grid), the movable cloud's grid"},
        "solver": {"normal_eq_evaluations_per_iteration": evals_per_it, "final_n_kept": int(last.n_kept),
                   "final_res_std": last.res_std},
        "device": ctx.device_name(),
    }
    if H_true is not None:
        out["accuracy"] = {"max_abs_H_minus_H_true": float(np.abs(H - H_true).max())}
    if steady_us is not None:
        out["steady_us_per_step"] = steady_us
    if fused and tail_cycles is not None and steady_us and not exchange:
        out["latency_model"] = latency_model(steady_us, elapsed / args.steps * 1e6, tail_cycles)

    if comm is not None:
        out["comm"] = comm
    if parity_rec is not None:
        out["parity"] = parity_oracle(parity_rec, Xf, Xm, sel, normals, planarity, obs, ow)
    for i, tp in enumerate(tps):
        if "_parity_args" in tp:
            # sampled oracle leg, 1e10 pairs (the full-size legs of tests/test_gpu_fullsize.py check 3 000 queries per iteration)
            rec_t, sel_t, nv_t, pl_t, obs_t, ow_t = tp.pop("_parity_args")
            tp["parity"] = parity_oracle(rec_t, Xf, Xm, sel_t, nv_t, pl_t, obs_t, ow_t, pair_cap=1e10)
        if "_normals_parity_args" in tp:
            tp["roofline_normals"]["parity"] = normals_parity(Xf, *tp.pop("_normals_parity_args"))
        out["throughput_point" if i == 0 else f"throughput_point_q{tp['correspondences']}"] = tp
    if tps:
        # the headline is strong-scaled and latency-bound (1000 correspondences cannot use a second GPU): the number a 1 -> 8 GPU curve
        # CAN raise is the largest throughput leg's, whose queries are sharded over the ranks
        big = max(tps, key=lambda t: t["correspondences"])
        out["scaling_relevant"] = {"metric": f"kNN correspondences/s, {big['correspondences']} correspondences per iteration, "
                                             + ("query shards" if world > 1 else "1 GPU"),
                                   "value": big["correspondences_per_s"], "unit": "correspondences/s", "n_gpus": world,
                                   "ms_per_step": big["ms_per_step"]}
    if world == 1 and not args.no_end_to_end:
        out["run_end_to_end"] = end_to_end(Xf, Xm, Q, k, kw)
    if world == 1 and not args.no_bruteforce_leg and Nm * nq <= 2e11:
        out["roofline_bruteforce"] = bruteforce_leg(local_rank, Xf, Xm, sel, normals, planarity, pmc, pmc_src)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(Xf, Xm, sel, normals, planarity, args.cpu_iterations)
    ref_file = ROOT / "profiles" / "cpu_reference.json"
    if ref_file.exists():
        ref = json.loads(ref_file.read_text())
        if args.config in ref.get("configs", {}):
            out["cpu_reference"] = {**ref["configs"][args.config], "measured_on": ref.get("measured_on"),
                                    # (kept by the compact line: the two CPU numbers come from two different machines)
                                    "box": "NOT this run's host: the 8-core build container",
                                    "kind": "reference", "source": "profiles/cpu_reference.json (scripts/time_reference.py)"}
    if args.force_exchange:
        out["config"]["parallelism"] += " (exchange forced on 1 rank)"
    if exchange:
        td.destroy_process_group()
    # RCCL prints a version banner through C stdio; flush it first so the JSON line is the LAST line
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    sys.stdout.flush()
    full = json.dumps(out)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(full + "\n")
    # stdout carries ONE line of a size a log tail holds whole (the round-2 line was 6 KB): every field of the contract and every
    # number, without the explanatory strings; the unabridged record goes to --out (profiles/r3/bench_*.json are such files)
    print(json.dumps(out if args.verbose_line else compact_line(out)), flush=True)


def compact_line(out):
    """The stdout line: same objects, explanatory strings and second-order detail dropped."""
    drop = {"note", "traffic_source", "timed_region", "scope", "oracle", "source", "measured_on", "bytes_alg_source", "repeat_stats",
            "kernels_instrumented", "solver", "setup", "transport_chosen_by_attach"}

    def strip(v, depth=0):
        if isinstance(v, dict):
            return {k: strip(x, depth + 1) for k, x in v.items() if not (depth > 0 and k in drop)}
        if isinstance(v, float):
            return float(f"{v:.6g}")
        return v
    line = strip(out)
    line["repeat_stats"] = {k: out["repeat_stats"][k] for k in ("repeats", "ms_per_step_p10", "ms_per_step_p90")}
    line["kernels_instrumented"] = {k: round(v["avg_ms"], 6) for k, v in out["kernels_instrumented"].items() if v["launches"]}
    line["setup"] = {k: round(v, 3) for k, v in out["setup"].items() if isinstance(v, float)}
    line["solver"] = out["solver"]
    for k in list(line):
        if k.startswith("throughput_point"):
            t = out[k]
            line[k] = {"correspondences": t["correspondences"], "n_gpus": t["n_gpus"], "ms_per_step": float(f"{t['ms_per_step']:.6g}"),
                       "iterations_per_s": float(f"{t['iterations_per_s']:.6g}"),
                       "correspondences_per_s": float(f"{t['correspondences_per_s']:.6g}"), "parallelism": t["parallelism"],
                       "roofline": strip(t["roofline"], 1),
                       "roofline_normals": {kk: (vv if kk != "parity" else {"ok": vv.get("ok"), "queries_sampled": vv.get("queries_sampled")})
                                            for kk, vv in strip(t["roofline_normals"], 1).items()
                                            if kk in ("kernel", "avg_ms", "achieved", "frac", "bytes_alg_per_launch", "traffic", "frac_on_pmc_traffic",
                                                      "neighbours_per_s", "candidates_per_query", "parity")},
                       "kernel_ms": {n: round(v["avg_ms"], 6) for n, v in t["kernels_instrumented"].items() if v["launches"]},
                       "evaluations_per_iteration": t["roofline_solver"].get("evaluations_per_iteration")}
            if "parity" in t:
                p0 = t["parity"].get("iteration_0", {})
                line[k]["parity"] = {"ok": t["parity"]["ok"], "queries_sampled": p0.get("queries_sampled"),
                                     "max_abs_dx": p0.get("max_abs_dx")}
            if "comm" in t:
                line[k]["comm"] = strip(t["comm"], 1)
    if "cpu_baseline" in line:
        line["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"][:120]
        line["cpu_baseline"].pop("seconds_per_iteration_by_stage", None)          # (the unabridged record keeps the per-stage split)
    if "cpu_reference" in line:
        line["cpu_reference"]["sample"] = out["cpu_reference"]["sample"][:100]
    line["config"]["workload"] = out["config"]["workload"]
    line["detail"] = "unabridged record: --out FILE (committed examples: profiles/r5/bench_*.json)"
    return line


def comm_record(ctx, exchange, transport, qshard, world, shard_rows, nq, timing, steps):
    """What exchange ran, as the library itself reports it (sicp_comm_info: for RCCL the rank count comes from
    ncclCommCount, not from what this script believes), and what it cost per iteration."""
    if not exchange:
        return None
    info = ctx.comm_info()
    xi = ctx.exchange_info()
    x = timing.get("exchange", {"ms": 0.0, "launches": 0})
    return {"backend": info["backend"], "nranks": info["nranks"], "rank": info["rank"], "partition": info["partition"],
            "winner_exchange": xi["form"], "key_exchange_from_queries": xi["keys_min_q"],
            "gn_shard": info["gn_shard"], "transport_chosen_by_attach": transport,
            "shard_rows_this_rank": int(shard_rows), "queries_this_rank": int((nq + world - 1) // world if qshard else nq),
            "exchange_us_per_iteration": x["ms"] * 1e3 / max(1, x["launches"]), "exchanges_timed": x["launches"],
            "note": "pack + collective + unpack / lexicographic minimum, HIP events on the library's stream (instrumented pass)"}


def throughput_point(args, ctx, Xf, Xm, Qt, k, rank, world, exchange, share_gpu):
    """The same clouds with Qt correspondences: K steps from the cold state, repeated; kernel split, the grid search's own
    bytes, parity.  Every rank calls this; rank 0 gets the record."""
    import torch
    import torch.distributed as td
    from simpleicp_amd import _lib, dist
    Nf, Nm = len(Xf), len(Xm)
    sel = np.unique(np.round(np.linspace(0, Nf - 1, Qt)).astype(np.int64))
    nq = len(sel)
    transport = None
    # N > 1: query shards (every rank the whole movable cloud, its slice of the queries) unless the caller pinned cloud shards
    # (--partition cloud: index shards of the movable cloud, the winners merged per iteration -- from 32 768 queries by three
    # reductions on 8-byte keys, the 6x6 reduction sharded from 262 144)
    tp_qshard = args.partition != "cloud"
    shard_rows = Nm
    if exchange:
        dist.detach(ctx)
        if tp_qshard:
            ctx.upload(_lib.MOV, Xm)
            transport = dist.attach(ctx, gn_shard=False, partition=_lib.PART_QUERIES)
        else:
            lo, hi = dist.shard_bounds(Nm, rank, world)
            shard_rows = hi - lo
            ctx.upload(_lib.MOV, Xm[lo:hi], index_base=lo)
            transport = dist.attach(ctx, gn_shard=nq >= 262144, partition=_lib.PART_CLOUD)
    ctx.timing_enable(False)
    t0 = time.perf_counter()
    normals, planarity = ctx.estimate_normals(_lib.FIX, sel, k)
    normals_ms = (time.perf_counter() - t0) * 1e3
    obs, ow = np.zeros(6), np.zeros(6)

    def cold():
        ctx.icp_setup(sel, normals, planarity)

    cold()
    iterate(ctx, 2, obs.copy(), obs, ow)                          # grid build (after a re-upload), allocations
    times = []
    for _ in range(max(1, args.throughput_repeats)):
        cold()
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, ne_evals, last = iterate(ctx, args.steps, obs.copy(), obs, ow)
        torch.cuda.synchronize()
        if world > 1:
            td.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
            td.all_reduce(t, op=td.ReduceOp.MAX)
            el = float(t.item())
        times.append(el)
    times = np.array(times)
    elapsed = float(np.median(times))
    ctx.timing_enable(True)
    ctx.timing_reset()
    cold()
    iterate(ctx, args.steps, obs.copy(), obs, ow)
    timing = ctx.timing()
    kern = ctx.last_match_kernel()
    ctx.timing_enable(True, count_work=True)
    ctx.timing_reset()
    cold()
    iterate(ctx, args.steps, obs.copy(), obs, ow)
    work = ctx.match_work()
    ctx.timing_enable(False)
    rec = None if args.no_parity else parity_device(ctx, sel, normals, planarity, obs, ow, iterations=1)
    comm = comm_record(ctx, exchange, transport, tp_qshard, world, shard_rows, nq, timing, args.steps)
    # the one-off before the loop at this Q: estimate_normals again (grid resident), its kernels under HIP events, then once more
    # with the sweep tallying its candidates
    ctx.timing_enable(True)
    nrm_ms = []
    for _ in range(3):
        ctx.timing_reset()
        ctx.estimate_normals(_lib.FIX, sel, k)
        nrm_ms.append(ctx.timing()["knnk_scan"]["ms"])
    ctx.timing_enable(True, count_work=True)
    ctx.timing_reset()
    ctx.estimate_normals(_lib.FIX, sel, k)
    knn_work = ctx.knn_work()
    ctx.timing_enable(False)
    if rank != 0:
        return None
    avg = {name: v["ms"] / max(1, v["launches"]) for name, v in timing.items()}
    nq_local = (nq + world - 1) // world if (exchange and tp_qshard) else nq
    per = {kk: v / max(1, work["launches"]) for kk, v in work.items() if kk != "launches"}
    bytes_match = match_bytes(kern, per, nq_local)
    pmc, pmc_src = load_pmc()
    tag = f"@Q{Qt}"
    if args.config != "C4" or args.points:
        pmc = {}                  # (the per-leg counters were collected on the default line's 10 M-point clouds only)
        pmc_src = "no counter pass for this leg on these clouds"

    def roof(kernel, ms, bytes_alg, note, extra=None):
        if bytes_alg is None:                    # nothing to price this launch on in this run (no tallies, no counters): say so
            d = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                 "traffic_source": None, "kernel": kernel, "avg_ms": ms, "bytes_alg_per_launch": None, "note": note}
            d.update(extra or {})
            return d
        ach = bytes_alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        d = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
             "traffic": pmc.get(kernel + tag), "traffic_source": pmc_src if (pmc.get(kernel + tag) is not None or not pmc) else None,
             "kernel": kernel, "avg_ms": ms, "bytes_alg_per_launch": int(bytes_alg), "note": note}
        if d["traffic"] is not None and ms > 0:
            d["frac_on_pmc_traffic"] = d["traffic"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS    # the counters' bytes over the same time
        d.update(extra or {})
        return d

    evals = ne_evals / args.steps
    out = {"correspondences": nq, "n_fixed": Nf, "n_movable": Nm, "n_gpus": world, "steps": args.steps,
           "ms_per_step": elapsed / args.steps * 1e3, "iterations_per_s": args.steps / elapsed,
           "correspondences_per_s": nq * args.steps / elapsed,
           "repeat_stats": {"repeats": len(times), "ms_per_step_p10": float(np.percentile(times, 10)) / args.steps * 1e3,
                            "ms_per_step_p90": float(np.percentile(times, 90)) / args.steps * 1e3,
                            "timed_region": "K steps from the cold state (icp_setup just called), min_change=0, events off"},
           "parallelism": (f"query shards x{world}, movable cloud replicated" if tp_qshard else
                           f"cloud shards x{world}, queries replicated") if exchange else "1 GPU",
           "roofline": roof(kern, avg["match"], bytes_match,
                            ("exact 1-NN on the static grid, four or eight cell-ordered queries per wave, candidates through a float32 filter in the "
                             "cloud's frame (16-B records), the winner re-evaluated exactly, ties left to the exact kernel; bytes = the candidates and "
                             "grid rows the search itself TALLIED + 168 B per query" if kern == "k_grid_nn16f" else
                             "exact 1-NN on the static grid, one (k_grid_nn) or four (k_grid_nn16) cell-ordered queries per wave, exact arithmetic on "
                             "every candidate (32-B records); bytes = the candidates and grid rows the search itself TALLIED + 96 B per query") +
                            " (they include what neighbouring queries share in L2: `frac_on_pmc_traffic` prices the same time on the HBM "
                            "counters' bytes); issue- and latency-bound", {"candidates_per_query": per["candidates"] / max(1, nq_local),
                                           "grid_rows_per_query": per["rows"] / max(1, nq_local),
                                           "left_to_exact_kernel_per_launch": per.get("deferred"),
                                           "pruning_ratio": (shard_rows * 24 + nq_local * 40) / max(1.0, bytes_match)}),
           "roofline_solver": roof("k_lm_eval" if (comm and comm["gn_shard"]) else "k_lm_all", avg["solve"], int(last.n_kept) * 72 * evals,
                                   "the iteration's whole minimisation (one launch: evaluations as phases between grid barriers, then the "
                                   "finish): 72 B per kept correspondence and evaluation, 8x8 Gram on the FP64 matrix pipe",
                                   {"evaluations_per_iteration": evals}),
           "roofline_rejection": roof("k_hsel_all" if nq > 16384 else "k_reject", avg["reject_select"], nq * 9 * (7 if nq > 16384 else 1),
                                      "median / MAD by digit selection + keep mask + statistics in one launch: 9 B per correspondence and "
                                      "sweep (two digit passes and one collecting sweep each for median and MAD, one keep / statistics sweep)"),
           "kernels_instrumented": {name: {"avg_ms": avg[name], "launches": timing[name]["launches"]} for name in timing},
           "setup": {"normals_ms": normals_ms},
           "solver": {"final_n_kept": int(last.n_kept), "final_res_std": last.res_std}}
    # SURVEY 8(d), normals (one-off): bytes_alg = N_f * 3 * 8 + Q * (3 * 8 + k * 8 + 16); reported as Q * k neighbours / s as well
    nms = float(np.median(nrm_ms))
    nkern = "k_grid_knn_sweep4" if (k <= 32 and nq >= 32768) else "k_grid_knn_sweep"
    rn = roof(nkern, nms, Nf * 24 + nq * (24 + k * 8 + 16),
              "estimate_normals' kernels (k_grid_knn_sweep4: one sweep over the ball's cells per query, four queries per wave, survivors "
              "ranked in LDS, mean + covariance from the winners' coordinates; k_grid_knn_sweep: the same one query per wave, for what "
              "the first leaves (~0.2 % of the queries) and for small query sets; k_cov_normals: Jacobi eigen-solver, one lane per "
              "query) on the resident grid; bytes_alg is SURVEY 8(d)'s brute-force figure (the whole cloud once + per-query terms) -- "
              "the pruned search reads `bytes_read_tallied` instead; bound by vector-instruction issue and latency (~270 instructions per query)",
              {"neighbours_per_s": nq * k / (nms * 1e-3) if nms > 0 else None,
               "candidates_per_query": knn_work["candidates"] / nq, "sweeps_per_query": knn_work["sweeps"] / nq,
               "queries_on_k_round_path": knn_work["slow_queries"],
               "bytes_read_tallied": int(knn_work["candidates"] * 32 + nq * (24 + 48 + 16)), "kernel_ms_all": nrm_ms})
    for other in ("k_cov_normals", "k_grid_knn_sweep"):            # every kernel of the call
        if rn["traffic"] is not None and other != nkern and pmc.get(other + tag) is not None:
            rn["traffic"] += pmc[other + tag]
    if rn["traffic"] is not None and nms > 0:
        rn["frac_on_pmc_traffic"] = rn["traffic"] / (nms * 1e-3) / 1e9 / HBM_PEAK_GBS
    out["roofline_normals"] = rn
    if rec is not None:
        out["_normals_parity_args"] = (sel, normals, planarity, k)
    if comm is not None:
        out["comm"] = comm
    if rec is not None:
        # the oracle leg runs on the host after every GPU leg is over (the other ranks must not wait in a collective for it)
        out["_parity_args"] = (rec, sel, normals, planarity, obs, ow)
    return out


def normals_parity(Xf, sel, normals, planarity, k, sample=300):
    """The device's normals on a sample of the queries against the oracle's brute-force k-NN + covariance + eigen step
    (pointcloud.py:185-203): 1 ulp(f32) of a unit vector's component."""
    from oracle import orc
    pick = np.unique(np.round(np.linspace(0, len(sel) - 1, sample)).astype(np.int64))
    onn, _ = orc.knn(Xf, Xf[sel[pick]], k=k)
    onv, opl = orc.normals(Xf, onn)
    dn, dp = float(np.abs(normals[pick] - onv).max()), float(np.abs(planarity[pick] - opl).max())
    return {"ok": bool(dn <= 2e-7 and dp <= 2e-6), "queries_sampled": int(len(pick)), "max_abs_dnormal": dn, "max_abs_dplanarity": dp}


def parity_device(ctx, sel, normals, planarity, obs, ow, iterations=2):
    """Outside the timed region: two iterations from the cold state through the product path; what they produced."""
    rec = []
    x = obs.copy()
    ctx.icp_setup(sel, normals, planarity)
    for it in range(iterations):
        R = ctx.icp_iterate(x, obs, ow, 0.3, 1.0)
        idx, dist, keep, _ = ctx.icp_state()
        rec.append((x.copy(), R, idx, dist, keep))
        x = np.array(R.x[:])
    return rec


def parity_oracle(rec, Xf, Xm, sel, normals, planarity, obs, ow, pair_cap=3e10):
    """... checked against the CPU oracle (brute-force match, distances, rejection, solve).  At Q x N_m <= 3e10 pairs
    the oracle runs the whole problem; above that a bounded sample of the queries is checked (the match is per
    query, so a sample pins it just as well)."""
    from oracle import orc
    nq = len(sel)
    cap = int(pair_cap // len(Xm))
    out = {"oracle": "oracle/sicp_oracle.c (brute-force CPU restatement, pinned against the unmodified reference)",
           "iterations_checked": len(rec)}
    ok_all = True
    for it, (x, R, idx, dist, keep) in enumerate(rec):
        if nq <= cap:
            o = orc.icp_iteration(Xm, Xf[sel], normals, planarity, x, x, 1.0, obs, ow, 0.3)
            res = {"indices_equal": bool(np.array_equal(idx, o["nn"])), "distances_equal": bool(np.array_equal(dist, o["dist"])),
                   "keep_mask_equal": bool(np.array_equal(keep, o["keep"])),
                   "median_mad_equal": bool(R.median == o["median"] and R.mad == o["mad"]),
                   "max_abs_dx": float(np.abs(np.array(R.x[:]) - o["x"]).max())}
            res["x_within_1e-9"] = res["max_abs_dx"] < 1e-9
        else:
            # the match (Q x N_m pairs) on a bounded sample of the queries, brute force over the WHOLE movable cloud; everything
            # downstream of it -- distances, planarity / MAD rejection, the minimiser -- on ALL correspondences, recomputed by the
            # oracle from the device's matched indices (per-correspondence work and reductions: cheap on the host at any Q)
            pick = np.unique(np.round(np.linspace(0, nq - 1, max(1, cap))).astype(np.int64))
            Hx = orc.params_to_H(x)
            nn, _ = orc.knn(Xm, Xf[sel[pick]], k=1, H=Hx)
            p1, p2 = Xf[sel], Xm[idx]
            d = orc.point_to_plane(p1, normals, p2, Hx)
            okeep, on, omed, omad = orc.reject(d, planarity, 0.3)
            ox, _ = orc.solve(x, 1.0, obs, ow, p1, normals, p2, okeep)
            res = {"indices_equal": bool(np.array_equal(idx[pick], nn[:, 0])), "queries_sampled": int(len(pick)),
                   "distances_equal": bool(np.array_equal(dist, d)), "keep_mask_equal": bool(np.array_equal(keep, okeep)),
                   "median_mad_equal": bool(R.median == omed and R.mad == omad and R.n_kept == on),
                   "max_abs_dx": float(np.abs(np.array(R.x[:]) - ox).max()),
                   "scope": "indices on the sample; distances, keep mask, median / MAD, minimiser on all correspondences"}
            res["x_within_1e-9"] = res["max_abs_dx"] < 1e-9
        ok_all = ok_all and all(v for kk, v in res.items() if isinstance(v, bool))
        out[f"iteration_{it}"] = res
    out["ok"] = bool(ok_all)
    return out


def end_to_end(Xf, Xm, Q, k, kw):
    """A real SimpleICP.run(): DataFrames in, cold, min_change = 1 (the reference's default), setup included.  The clock
    brackets run() alone: its results are held until the clock has stopped (dropping the 240 MB X_t it returns is the
    caller's munmap, not run()'s)."""
    from simpleicp_amd import PointCloud, SimpleICP

    def one(own):
        pc_fix = PointCloud(Xf, columns=["x", "y", "z"])
        pc_mov = PointCloud(Xm.copy() if own else Xm, columns=["x", "y", "z"])
        icp = SimpleICP(verbose=False)
        icp.add_point_clouds(pc_fix, pc_mov)
        t0 = time.perf_counter()
        held = icp.run(correspondences=Q, neighbors=k, **kw)
        dt = time.perf_counter() - t0
        return dt, icp.last_run_info["iterations"], held

    one(False)                                           # first pass: context, allocator, page cache
    # the frames wrap arrays the caller still holds; three passes, the middle one counts (a run() of 5-20 ms sits next to one-off
    # costs of the same size: the runtime pinning an address range it has not seen, the allocator returning a block to the OS)
    passes = []
    for _ in range(3):
        dt, iters, held = one(False)
        passes.append(dt)
        del held
    dt = sorted(passes)[1]
    dt_own, _, _ = one(True)                             # the movable frame is the only owner of its (n,3) array
    return {"seconds": dt, "iterations": iters, "iterations_per_s": iters / dt, "seconds_passes": passes,
            "seconds_frame_owns_its_array": dt_own,
            "note": "SimpleICP.run() on DataFrames: upload, overlap pre-pass, normals, grid build, iterations to the "
                    "reference's convergence test (min_change=1), final transform + download; warm process.  "
                    "seconds_frame_owns_its_array: the same call when nothing else references the movable frame's "
                    "coordinate block, so assigning the transformed columns (as the reference does, pointcloud.py:215-217) "
                    "frees it inside run() -- host munmap time, no device work"}


def bruteforce_leg(device, Xf, Xm, sel, normals, planarity, pmc, pmc_src):
    """The north-star kernel: brute-force scan (FP32 conservative filter + exact FP64 verification) of the
    same Q x N problem, 6 iterations; also checks that it lands on the same estimate as the default path."""
    from simpleicp_amd import _lib
    os.environ["SICP_KNN1"] = "filter"
    try:
        ctx = _lib.Context(device)
    finally:
        del os.environ["SICP_KNN1"]
    ctx.upload(_lib.FIX, Xf)
    ctx.upload(_lib.MOV, Xm)
    ctx.icp_setup(sel, normals, planarity)
    obs, ow = np.zeros(6), np.zeros(6)
    iterate(ctx, 2, obs.copy(), obs, ow)
    ctx.timing_enable(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    x, _, _ = iterate(ctx, 6, obs.copy(), obs, ow)
    dt = time.perf_counter() - t0
    tm = ctx.timing()["match"]
    ms = tm["ms"] / max(1, tm["launches"])
    n, q = len(Xm), len(sel)
    bytes_alg = n * 24 + q * 40
    pairs = n * q
    ach = bytes_alg / (ms * 1e-3) / 1e9
    kern = ctx.last_match_kernel()
    ctx.close()
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": pmc.get(kern), "traffic_source": pmc_src if pmc.get(kern) is not None else None,
            "frac_on_pmc_traffic": (pmc[kern] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if pmc.get(kern) is not None else None,
            "kernel": kern, "avg_ms": ms, "bytes_alg_per_launch": bytes_alg,
            "pair_evals_per_s": pairs / (ms * 1e-3),
            # SURVEY 8(d) prices a pair at 8 flop (3 sub + 1 mul + 2 fma in the plain distance form); the filter this kernel runs
            # spends 6 on it (|p|^2 - 2 q.p: 3 fma): both fractions of the 157.3 TF FP32 vector peak, named for what they count
            "valu_frac_8flop_per_pair": pairs * 8 / (ms * 1e-3) / (FP32_VALU_PEAK_TFLOPS * 1e12),
            "valu_frac_6flop_executed": pairs * 6 / (ms * 1e-3) / (FP32_VALU_PEAK_TFLOPS * 1e12),
            "iterations_per_s": 6 / dt,
            "note": "brute-force Q x N scan is compute-bound by construction (SURVEY 8d: 8 flop per pair; the FP32 filter executes "
                    "6, |p|^2 - 2 q.p as three fma): fractions are against the 157.3 TF FP32 vector = FP32 matrix peak; the few "
                    "passing (query, group) pairs are recorded and re-evaluated exactly in FP64 by k_knn1_fixup; the cloud is "
                    "read from HBM once per query block"}


def cpu_baseline(Xf, Xm, sel, normals, planarity, iterations):
    """The reference's algorithm on the host: oracle/ref_port.py (checker-side code, used here only
    as the timed CPU baseline).  Bounded sample: `iterations` ICP iterations of the same workload,
    normals injected (the reference's own estimate_normals needs ~10 min at 10M points)."""
    from oracle import ref_port
    cores = os.cpu_count() or 1
    if len(Xm) > 20_000_000:
        iterations = 1
    t0 = time.perf_counter()
    res = ref_port.run(Xf, Xm, correspondences=len(sel), max_iterations=iterations, min_change=0.0,
                       normals=normals, planarity=planarity, sel_idx=sel)
    dt = time.perf_counter() - t0
    import platform
    per = res.per_iter
    split = {k: float(np.mean([p[k] for p in per])) for k in per[0] if k.endswith("_s")} if per else {}
    return {"value": res.iterations / dt, "unit": "iterations/s", "cores": cores, "kind": "port",
            "box": f"this run's host ({platform.processor() or platform.machine()}, {cores} logical cores)",
            "sample": f"{res.iterations} ICP iterations of the same {len(Xm)}-vs-{len(Xf)} workload "
                      f"(cKDTree rebuild + query workers=-1 + scipy least_squares per iteration), "
                      f"{dt:.1f} s wall",
            "correspondences_per_s": len(sel) * res.iterations / dt,
            "match_s_per_iteration": split.get("match_s"),
            "seconds_per_iteration_by_stage": split}


if __name__ == "__main__":
    main()
