"""GPU side of the multi-GPU path that one GPU can exercise: the lexicographic reduce kernel, and the
full exchange plumbing with a 1-rank process group -- the library's own RCCL communicator (sicp_comm_init:
ncclUniqueId broadcast over torch.distributed, ncclAllGather / ncclAllReduce enqueued by the library), the
host-callback variant (device-pointer views for torch, all_gather on library-owned buffers), cloud shards and
query shards."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_lexmin_kernel_matches_reference():
    from simpleicp_amd import _lib
    rng = np.random.default_rng(3)
    world, Q = 5, 3000
    d2 = np.round(rng.uniform(0, 1, (world, Q)), 1)
    idx = rng.integers(0, 50, (world, Q)).astype(np.int64)
    none = rng.uniform(size=(world, Q)) < 0.3
    none[:, :10] = True                                           # queries nobody has a candidate for
    d2[none], idx[none] = np.inf, -1
    xyz = rng.normal(size=(world, Q, 3)); xyz[none] = 0
    G = np.concatenate((d2[..., None], idx.view(np.float64)[..., None], xyz), axis=2)
    with _lib.Context(0) as ctx:
        gd2, gidx, gxyz = ctx.lexmin_gathered(G)
    bd = np.full(Q, np.inf); bi = np.full(Q, -1, np.int64); bx = np.zeros((Q, 3))
    for r in range(world):
        better = (idx[r] >= 0) & ((bi < 0) | (d2[r] < bd) | ((d2[r] == bd) & (idx[r] < bi)))
        bd[better], bi[better], bx[better] = d2[r][better], idx[r][better], xyz[r][better]
    assert np.array_equal(gidx, bi) and np.array_equal(gd2, bd) and np.array_equal(gxyz, bx)
    assert np.all(gidx[:10] == -1) and np.all(np.isinf(gd2[:10]))


SCRIPT = r'''
import os, sys, numpy as np, torch, torch.distributed as td
sys.path.insert(0, "%(root)s"); sys.path.insert(0, "%(root)s/tests")
from conftest import load_golden, load_cloud
from simpleicp_amd import PointCloud, SimpleICP
def run(name="bunny"):
    g, files, kw = load_golden(name)
    pf = PointCloud(load_cloud(files[0]), columns=["x", "y", "z"]); pm = PointCloud(load_cloud(files[1]), columns=["x", "y", "z"])
    icp = SimpleICP(verbose=False); icp.add_point_clouds(pf, pm)
    H, X, rbp, res = icp.run(**kw)
    return H, X, res, icp.last_run_info["iterations"]
H0, X0, r0, it0 = run()                                # no process group: plain single-GPU path
B0 = run("dragon_q5000")                               # ... and a case above 2048 correspondences (the many-workgroup solver)
from simpleicp_amd import backend
os.environ["SICP_SOLVE"] = "host"; backend.reset_context()
Hh, Xh, rh, ith = run()                                # multi-kernel tail + host LM (what gn_shard builds on)
del os.environ["SICP_SOLVE"]; backend.reset_context()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="%(port)d", SICP_FORCE_EXCHANGE="1")
torch.cuda.set_device(0)
td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
H1, X1, r1, it1 = run()                                # same job through the library's own RCCL communicator (1 rank)
assert it0 == it1 and np.array_equal(H0, H1) and np.array_equal(X0, X1) and np.array_equal(r0, r1), (H0 - H1)
assert ith == it0 and np.abs(Hh - H0).max() < 1e-9     # device LM vs host LM: same minimiser
from simpleicp_amd import backend as _b
info = _b.get_context().comm_info()
assert info["communicator"] and info["backend"] == "none", info      # parked between runs, kept for the next one
os.environ["SICP_XCHG_KEYS_MIN_Q"] = "1"; backend.reset_context()
H8, X8, r8, it8 = run()                                # cloud shards' winners by three all-reduces on 8-byte keys (min d2, min index, max of the
B8 = run("dragon_q5000")                               # owner's coordinate bits) instead of the all-gather of 40-byte records: same bits
assert it8 == it0 and np.array_equal(H8, H0) and np.array_equal(X8, X0) and np.array_equal(r8, r0), (H8 - H0)
assert B8[3] == B0[3] and np.array_equal(B8[0], B0[0]) and np.array_equal(B8[2], B0[2]), (B8[0] - B0[0])
del os.environ["SICP_XCHG_KEYS_MIN_Q"]; backend.reset_context()
os.environ["SICP_GN_SHARD"] = "1"
H2, X2, r2, it2 = run()                                # sharded 6x6 reduction requested: the single-workgroup tail (Q <= 2048) ignores it
assert it2 == it0 and np.array_equal(H2, H0) and np.array_equal(r2, r0), (H2 - H0)
B2 = run("dragon_q5000")                               # Q = 5000: every evaluation's 8x8 Gram block through ncclAllReduce, then k_lm_advance
assert B2[3] == B0[3] and np.abs(B2[0] - B0[0]).max() < 1e-12 and np.abs(B2[2] - B0[2]).max() < 1e-12, (B2[0] - B0[0])
os.environ["SICP_SOLVE"] = "host"; backend.reset_context()
H7, X7, r7, it7 = run()                                # the host-side solve with its 30 sums all-reduced per step
assert it7 == ith and np.array_equal(H7, Hh) and np.array_equal(r7, rh), (H7 - Hh)
del os.environ["SICP_SOLVE"]; backend.reset_context()
del os.environ["SICP_GN_SHARD"]
os.environ["SICP_PARTITION"] = "queries"
H4, X4, r4, it4 = run()                                # query shards (cloud replicated): slices gathered in rank order
assert it4 == it0 and np.array_equal(H4, H0) and np.array_equal(r4, r0), (H4 - H0)
del os.environ["SICP_PARTITION"]
os.environ["SICP_XCHG"] = "callback"
H5, X5, r5, it5 = run()                                # collectives supplied by the host: torch.distributed callback
assert it5 == it0 and np.array_equal(H5, H0) and np.array_equal(r5, r0)
os.environ["SICP_GN_SHARD"] = "1"
B6 = run("dragon_q5000")                               # the sharded reduction with the SUM supplied by the host callback
assert B6[3] == B0[3] and np.abs(B6[0] - B0[0]).max() < 1e-12 and np.abs(B6[2] - B0[2]).max() < 1e-12
os.environ["SICP_XCHG_SYNC"] = "1"; del os.environ["SICP_GN_SHARD"]
H3, X3, r3, it3 = run()                                # blocking variant of the callback
td.destroy_process_group()
assert it3 == it0 and np.array_equal(H3, H0) and np.array_equal(r3, r0)
print("EXCHANGE_OK", it0)
'''


def test_exchange_plumbing_with_one_rank_process_group():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": str(ROOT), "port": port}], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "EXCHANGE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


WORKER2 = r'''
import os, sys, numpy as np
rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sys.path.insert(0, "%(root)s"); sys.path.insert(0, "%(root)s/tests")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SIMPLEICP_DEVICE="0")
import torch, torch.distributed as td
from conftest import load_golden, load_cloud
from simpleicp_amd import PointCloud, SimpleICP, backend
td.init_process_group("gloo", rank=rank, world_size=world)
def run(name="bunny", **extra):
    g, files, kw = load_golden(name)
    pf = PointCloud(load_cloud(files[0]), columns=["x", "y", "z"]); pm = PointCloud(load_cloud(files[1]), columns=["x", "y", "z"])
    icp = SimpleICP(verbose=False); icp.add_point_clouds(pf, pm)
    H, X, rbp, res = icp.run(**{**kw, **extra})
    info = icp.last_run_info
    assert info["ranks"] == world == 2 and info["exchange"] == "callback" and info["partition"] == os.environ.get("SICP_PARTITION", "cloud"), info
    return H, X, res, info["iterations"]
res = {}
res["cloud"] = run()                                    # the movable cloud in index shards, one all-gather + lexmin per iteration
os.environ["SICP_PARTITION"] = "queries"
res["queries"] = run()                                  # every rank the whole cloud, its slice of the queries
res["queries_odd"] = run(correspondences=999)           # ... of unequal length (the last rank's slice is one short)
del os.environ["SICP_PARTITION"]
os.environ["SICP_GN_SHARD"] = "1"
res["gn"] = run()                                       # + the 6x6 reduction sharded: ignored by the single-workgroup tail (Q <= 2048)
res["gn_q5000"] = run("dragon_q5000")                   # Q = 5000: two real slices, one SUM of the 8x8 Gram block per evaluation
del os.environ["SICP_GN_SHARD"]
res["dragon_q5000"] = run("dragon_q5000")               # Q > 2048: the large-Q chain between the exchanges
td.barrier()
td.destroy_process_group()
np.savez(out, **{k + "_" + n: v for k, t in res.items() for n, v in zip(("H", "X", "r", "it"), t)})
print("RANK_OK", rank)
'''


def test_two_ranks_sharing_one_gpu_agree_with_one_rank(tmp_path):
    """The multi-rank flow with REAL shards on hardware: two processes share cuda:0 (a gloo group, collectives staged through
    host memory -- RCCL refuses two ranks on one device), each uploads its index shard of the movable cloud (or matches its
    slice of the queries) and the per-iteration exchange merges them.  Cloud shards and query shards must reproduce the
    single-process run bit for bit; with the sharded 6x6 reduction the sums are grouped differently (1e-9)."""
    import socket
    sys.path.insert(0, str(ROOT / "tests"))
    from conftest import load_golden, load_cloud
    from simpleicp_amd import PointCloud, SimpleICP

    def run(name, **extra):
        g, files, kw = load_golden(name)
        pf = PointCloud(load_cloud(files[0]), columns=["x", "y", "z"]); pm = PointCloud(load_cloud(files[1]), columns=["x", "y", "z"])
        icp = SimpleICP(verbose=False); icp.add_point_clouds(pf, pm)
        H, X, rbp, res = icp.run(**{**kw, **extra})
        return H, X, res, icp.last_run_info["iterations"]

    ref = {"bunny": run("bunny"), "dragon_q5000": run("dragon_q5000"), "bunny_999": run("bunny", correspondences=999)}
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker2.py"
    script.write_text(WORKER2 % {"root": str(ROOT)})
    env = {k: v for k, v in os.environ.items() if k not in ("SICP_XCHG", "SICP_PARTITION", "SICP_GN_SHARD", "SICP_FORCE_EXCHANGE")}
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(port), str(tmp_path / f"rank{r}.npz")], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs) and all("RANK_OK" in o for o in outs), "\n".join(o[-3000:] for o in outs)
    for r in range(2):
        z = np.load(tmp_path / f"rank{r}.npz")
        for key, name in (("cloud", "bunny"), ("queries", "bunny"), ("queries_odd", "bunny_999"), ("dragon_q5000", "dragon_q5000")):
            H, X, res, it = ref[name]
            assert int(z[key + "_it"]) == it and np.array_equal(z[key + "_H"], H) and np.array_equal(z[key + "_X"], X) \
                and np.array_equal(z[key + "_r"], res), (key, r, np.abs(z[key + "_H"] - H).max())
        H, X, res, it = ref["bunny"]
        assert int(z["gn_it"]) == it and np.abs(z["gn_H"] - H).max() < 1e-9
        H, X, res, it = ref["dragon_q5000"]               # sums grouped by rank: equal to rounding, residuals of BOTH slices current
        assert int(z["gn_q5000_it"]) == it and np.abs(z["gn_q5000_H"] - H).max() < 1e-9 and np.abs(z["gn_q5000_r"] - res).max() < 1e-9


@pytest.mark.parametrize("partition,launcher", [("cloud", "self"), ("queries", "self"), ("cloud", "torchrun")])
def test_bench_two_ranks_self_launched(partition, launcher):
    """`python bench.py --gpus 2` from a plain interpreter: the file spawns its own ranks, shards the movable cloud (or the
    queries), times with barrier + MAX over ranks, and rank 0 prints ONE JSON line whose parity leg (every rank in the
    exchange, rank 0 against the oracle) is green.  SICP_BENCH_SHARE_GPU=1 puts both ranks on cuda:0 over gloo.
    `torchrun`: the same through the launcher command the round-end driver uses (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    from torch.distributed.run)."""
    import json
    import socket
    env = dict(os.environ, SICP_BENCH_SHARE_GPU="1")
    args = [str(ROOT / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "1", "--repeats", "2", "--points", "300000",
            "--partition", partition, "--no-cpu-baseline", "--no-end-to-end", "--no-bruteforce-leg", "--throughput-q", "40000",
            "--throughput-repeats", "2"]
    if launcher == "torchrun":
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + args
    else:
        cmd = [sys.executable] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "strong" and d["value"] > 0
    assert d["parity"]["ok"] is True, d["parity"]
    assert "callback" in d["config"]["parallelism"] and ("query" in d["config"]["parallelism"]) == (partition == "queries")
    # the record says what exchange ran, as the library counts it
    assert d["comm"]["backend"] == "callback" and d["comm"]["nranks"] == 2 and d["comm"]["partition"] == partition, d["comm"]
    assert d["comm"]["exchanges_timed"] == 6 and d["comm"]["exchange_us_per_iteration"] > 0
    # the throughput leg ran under query shards on both ranks (cloud shards when the caller pinned them: then the winners met by the
    # three reductions on 8-byte keys -- 40 000 queries are above the threshold) and met the oracle
    tp = d["throughput_point"]
    assert d["comm"]["winner_exchange"] == ("query_slices" if partition == "queries" else "records_allgather"), d["comm"]
    if partition == "cloud":
        assert tp["n_gpus"] == 2 and tp["comm"]["partition"] == "cloud" and tp["comm"]["queries_this_rank"] == tp["correspondences"]
        assert tp["comm"]["winner_exchange"] == "key_allreduces" and tp["comm"]["shard_rows_this_rank"] == 150_000, tp["comm"]
    else:
        assert tp["n_gpus"] == 2 and tp["comm"]["partition"] == "queries" and tp["comm"]["queries_this_rank"] == (tp["correspondences"] + 1) // 2
        assert tp["comm"]["winner_exchange"] == "query_slices", tp["comm"]
    assert tp["parity"]["ok"] is True and tp["roofline"]["kernel"] in ("k_grid_nn", "k_grid_nn16", "k_grid_nn16f"), tp


WORKER_KEYS = r"""
import os, sys, json, numpy as np
rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sys.path.insert(0, "%(root)s"); sys.path.insert(0, "%(root)s/tests")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SIMPLEICP_DEVICE="0", SICP_PARTITION="cloud")
cases = json.loads(os.environ["SICP_TEST_CASES"])
import bench
from simpleicp_amd import PointCloud, SimpleICP, backend
if world > 1:
    import torch, torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
res = {}
for case in cases:
    for k, v in case.get("env", {}).items():
        os.environ[k] = v
    backend.reset_context()
    Xf, Xm, _ = bench.synthetic_pair(case["n"])
    if case.get("sorted"):                       # index shards = slabs along x: spatially DISJOINT shards
        Xm = Xm[np.argsort(Xm[:, 0], kind="stable")]
    pf = PointCloud(Xf, columns=["x", "y", "z"]); pm = PointCloud(Xm, columns=["x", "y", "z"])
    icp = SimpleICP(verbose=False); icp.add_point_clouds(pf, pm)
    ctx = backend.get_context()
    ctx.timing_enable(True, count_work=True); ctx.timing_reset()
    H, X, rbp, r = icp.run(correspondences=case["q"], max_iterations=case["its"], min_change=0.0)
    info = icp.last_run_info
    work = ctx.match_work(); ctx.timing_enable(False)
    if world > 1:
        assert info["ranks"] == world and info["exchange"] == "callback" and info["partition"] == "cloud", info
        assert info["winner_exchange"] == case["form"] and info["exchanges"] == case["its"], info
        assert ctx.last_match_kernel() == case["kernel"], ctx.last_match_kernel()
    name = case["name"]
    res[name + "_H"] = H; res[name + "_r"] = r; res[name + "_X"] = X[:: max(1, len(X) // 5000)]
    res[name + "_cand"] = np.array([work["candidates"] / (case["q"] * case["its"])])
    for k in case.get("env", {}):
        del os.environ[k]
if world > 1:
    td.barrier(); td.destroy_process_group()
np.savez(out, **res)
print("RANK_OK", rank)
"""


def _run_key_workers(tmp_path, cases, world):
    import json
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / f"worker_keys{world}.py"
    script.write_text(WORKER_KEYS % {"root": str(ROOT)})
    env = {k: v for k, v in os.environ.items() if not k.startswith("SICP_")}
    env["SICP_TEST_CASES"] = json.dumps(cases)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(port), str(tmp_path / f"w{world}_rank{r}.npz")], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=1200)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs) and all("RANK_OK" in o for o in outs), "\n".join(o[-3000:] for o in outs)
    return [np.load(tmp_path / f"w{world}_rank{r}.npz") for r in range(world)]


def test_two_ranks_key_exchange_equals_one_rank(tmp_path):
    """The DEFAULT exchange of cloud shards from 32 768 queries -- three reductions on 8-byte keys (sicp_comm.cpp:
    exchange_best_keys_chained; min of the distance's bits, min of the index among the holders of that minimum, max of the owner's
    coordinate bits) -- with TWO real shards on hardware: two processes share cuda:0 over a gloo group whose callback serves
    SICP_XCHG_MIN_U64 / MAX_U64 (ABI 6).  Q = 40 000 (four queries per wave) and Q = 262 144 (the float32-filtered search, whose
    slot bounds the exchange refreshes), the solver replicated: H, the residuals and the transformed cloud bit-identical to the
    one-process run; with the sharded 6x6 reduction (the default from 262 144 correspondences): equal to rounding.  A callback
    that declines the new operations (an ABI-5 host) keeps the all-gather of records -- same bits."""
    cases = [
        dict(name="q40k", n=600_000, q=40_000, its=6, form="key_allreduces", kernel="k_grid_nn16", env={"SICP_GN_SHARD": "0"}),
        dict(name="q40k_declined", n=600_000, q=40_000, its=6, form="records_allgather", kernel="k_grid_nn16",
             env={"SICP_GN_SHARD": "0", "SICP_XCHG_U64": "0"}),
        dict(name="q262k", n=600_000, q=262_144, its=6, form="key_allreduces", kernel="k_grid_nn16f", env={"SICP_GN_SHARD": "0"}),
        dict(name="q262k_gn", n=600_000, q=262_144, its=6, form="key_allreduces", kernel="k_grid_nn16f", env={}),
    ]
    one = _run_key_workers(tmp_path, cases, 1)[0]
    two = _run_key_workers(tmp_path, cases, 2)
    for z in two:
        for c in cases:
            n = c["name"]
            if n.endswith("_gn"):
                assert np.abs(z[n + "_H"] - one[n + "_H"]).max() < 1e-9 and np.abs(z[n + "_r"] - one[n + "_r"]).max() < 1e-9
            else:
                assert np.array_equal(z[n + "_H"], one[n + "_H"]) and np.array_equal(z[n + "_r"], one[n + "_r"]) \
                    and np.array_equal(z[n + "_X"], one[n + "_X"]), (n, np.abs(z[n + "_H"] - one[n + "_H"]).max())


def test_spatially_disjoint_shards_search_from_the_job_wide_winner(tmp_path):
    """under cloud shards the filtered search bounds a slot by what IT left there -- this rank's own winner.
    With index shards that are slabs of the cloud (a cloud stored in scan order) half of every rank's queries have their answer on the
    other rank, and without the refresh (k_slot_bounds behind the exchange) each of those searches goes out to its far-away local
    neighbour in every iteration.  Same bits as one rank either way; the tallied candidates per query and iteration must stay at the
    one-rank figure's order (each rank reads its half of the cloud's cells: about half the candidates, not hundreds of times more)."""
    env = {"SICP_GN_SHARD": "0", "SICP_NN16F_MIN_Q": "50000"}
    cases = [dict(name="slabs1", n=600_000, q=100_000, its=1, form="key_allreduces", kernel="k_grid_nn16f", sorted=True, env=env),
             dict(name="slabs", n=600_000, q=100_000, its=8, form="key_allreduces", kernel="k_grid_nn16f", sorted=True, env=env)]
    one = _run_key_workers(tmp_path, cases, 1)[0]
    two = _run_key_workers(tmp_path, cases, 2)
    # candidates per query of iterations 1..7 (the cold iteration 0 has no bound on any rank: a rank's far queries walk out to its
    # own shard once; from then on the exchange's winner bounds them)
    later = lambda z: (8 * float(z["slabs_cand"][0]) - float(z["slabs1_cand"][0])) / 7
    for z in two:
        assert np.array_equal(z["slabs_H"], one["slabs_H"]) and np.array_equal(z["slabs_r"], one["slabs_r"])
        assert later(z) <= 1.25 * later(one), (later(z), later(one), float(z["slabs1_cand"][0]), float(one["slabs1_cand"][0]))
