"""N > 1 path on CPU: world_size-2 gloo run of the exchange step in simpleicp_amd/dist.py --
the same functions the RCCL path calls on device tensors.  Local shard results come from the
CPU oracle (tests may use it); the assertion is that sharded == unsharded, bit-exact."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import orc
        from simpleicp_amd import dist

        assert dist.is_distributed() and dist.rank_world() == (rank, world)
        rng = np.random.default_rng(7)                       # same data on every rank
        n, q = 30_001, 700
        Xm = np.round(rng.uniform(-5, 5, (n, 3)), 1)         # quantised -> exact ties across shards
        Xm[n // 2:n // 2 + 200] = Xm[:200]                   # duplicates living in different shards
        Xm[Xm == 0.0] = -0.0                                 # coordinates whose bit pattern a SUM would not carry (the key exchange's max must)
        Qp = np.round(rng.uniform(-5, 5, (q, 3)), 1)
        H = orc.params_to_H(np.array([0.1, -0.2, 0.05, 0.3, 0.1, -0.2]))
        lo, hi = dist.shard_bounds(n, rank, world)

        for max_dist in (np.inf, 0.15):
            idx, d2 = orc.knn(Xm[lo:hi], Qp, k=1, H=H, max_dist=max_dist, idx_base=lo)
            idx, d2 = idx[:, 0].copy(), d2[:, 0].copy()
            xyz = np.where((idx >= 0)[:, None], Xm[np.maximum(idx, 0)], 0.0)
            t_d2, t_idx, t_xyz = torch.from_numpy(d2), torch.from_numpy(idx), torch.from_numpy(xyz)
            dist.exchange_best_match(t_d2, t_idx, t_xyz)
            fidx, fd2 = orc.knn(Xm, Qp, k=1, H=H, max_dist=max_dist)
            assert np.array_equal(t_idx.numpy(), fidx[:, 0])
            assert np.array_equal(t_d2.numpy(), fd2[:, 0])
            ok = fidx[:, 0] >= 0
            assert np.array_equal(t_xyz.numpy()[ok], Xm[fidx[ok, 0]]) and np.all(t_xyz.numpy()[~ok] == 0)
            if np.isfinite(max_dist):
                assert (~ok).any() and ok.any()
            # the same winner by three reductions on 8-byte keys (cloud shards of many queries; sicp_comm.cpp:
            # exchange_best_keys_chained): min of the distance's bits, min of the index among the holders of that minimum, max of
            # the owner's coordinate bits -- equal to the gathered records' lexicographic minimum word for word, signs of zeros included
            k_d2, k_idx, k_xyz = torch.from_numpy(d2.copy()), torch.from_numpy(idx.copy()), torch.from_numpy(xyz.copy())
            dist.exchange_best_keys(k_d2, k_idx, k_xyz)
            assert np.array_equal(k_idx.numpy(), t_idx.numpy()) and np.array_equal(k_d2.numpy(), t_d2.numpy())
            assert np.array_equal(k_xyz.numpy().view(np.int64), t_xyz.numpy().view(np.int64))
            assert np.signbit(k_xyz.numpy()[ok]).any() and (k_xyz.numpy()[ok] == 0).any()          # (-0.0 did travel)
            # exact ties ACROSS shards were among the cases: a winner whose distance another rank holds too
            gd2 = [torch.empty_like(t_d2) for _ in range(world)]
            td.all_gather(gd2, torch.from_numpy(d2))
            holders = sum((g.numpy() == t_d2.numpy()) & np.isfinite(g.numpy()) for g in gd2)
            assert (holders >= 2).any()

        # query shards (SURVEY 8e "alternative"): every rank matches its slice of the queries in the WHOLE cloud;
        # gathering the slices in rank order restores the full result -- no reduction, bit-exact by construction
        fidx, fd2 = orc.knn(Xm, Qp, k=1, H=H)
        per = (q + world - 1) // world
        a, b = min(q, per * rank), min(q, per * rank + per)
        sidx, sd2 = orc.knn(Xm, Qp[a:b], k=1, H=H)
        t_d2 = torch.full((q,), float("nan"), dtype=torch.float64); t_idx = torch.full((q,), -7, dtype=torch.int64)
        t_xyz = torch.full((q, 3), float("nan"), dtype=torch.float64)
        t_d2[a:b] = torch.from_numpy(sd2[:, 0]); t_idx[a:b] = torch.from_numpy(sidx[:, 0]); t_xyz[a:b] = torch.from_numpy(Xm[sidx[:, 0]])
        dist.exchange_query_slices(t_d2, t_idx, t_xyz)
        assert np.array_equal(t_idx.numpy(), fidx[:, 0]) and np.array_equal(t_d2.numpy(), fd2[:, 0])
        assert np.array_equal(t_xyz.numpy(), Xm[fidx[:, 0]])

        # sharded 6x6 normal-equation reduction + SUM exchange == unsharded (to rounding)
        p1 = rng.uniform(-5, 5, (q, 3))
        n1 = rng.normal(size=(q, 3)); n1 = (n1 / np.linalg.norm(n1, axis=1, keepdims=True)).astype(np.float32)
        p2 = p1 + rng.normal(0, 0.05, (q, 3))
        keep = rng.uniform(size=q) < 0.8
        x = np.array([0.01, 0.02, -0.01, 0.1, 0.0, -0.1])
        per = (q + world - 1) // world
        a, b = min(q, per * rank), min(q, per * rank + per)
        part = orc.normal_equations(x, p1[a:b], n1[a:b], p2[a:b], keep[a:b])
        t = torch.from_numpy(part.copy())
        dist.allreduce_sum(t)
        full = orc.normal_equations(x, p1, n1, p2, keep)
        assert np.allclose(t.numpy(), full, rtol=1e-13, atol=1e-12)
        assert t.numpy()[29] == keep.sum()
        Path(tmp, f"ok{rank}").write_text("ok")
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_world(world, tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_shard_bounds_partition():
    sys.path.insert(0, str(ROOT))
    from simpleicp_amd.dist import shard_bounds
    for n in (0, 1, 7, 8, 10_000_001):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _worker_allgather(rank, world, port, tmp):
    """The production exchange = library packs (d2, idx bits, xyz) records -> ONE all_gather ->
    lexicographic reduce.  Here on CPU: numpy packs, gloo gathers (dist.allgather_into), and the
    reduce rule is checked against dist.exchange_best_match on the same data."""
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from simpleicp_amd import dist
        rng = np.random.default_rng(100 + rank)
        Q = 500
        d2 = np.round(rng.uniform(0, 1, Q), 1)                     # many cross-rank ties
        idx = rng.integers(0, 1000, Q).astype(np.int64) + 1000 * rank
        none = rng.uniform(size=Q) < 0.2
        d2[none], idx[none] = np.inf, -1
        xyz = rng.normal(size=(Q, 3)); xyz[none] = 0
        rec = np.column_stack((d2, idx.view(np.float64), xyz)).reshape(-1)
        send = torch.from_numpy(rec.copy())
        recv = torch.empty(world * 5 * Q, dtype=torch.float64)
        dist.allgather_into(recv, send)
        G = recv.numpy().reshape(world, Q, 5)
        # reduce exactly like k_lexmin_gathered
        bd = np.full(Q, np.inf); bi = np.full(Q, -1, np.int64); bx = np.zeros((Q, 3))
        for r in range(world):
            d, i = G[r, :, 0], G[r, :, 1].copy().view(np.int64)
            better = (i >= 0) & ((bi < 0) | (d < bd) | ((d == bd) & (i < bi)))
            bd[better], bi[better], bx[better] = d[better], i[better], G[r, better, 2:]
        t_d2, t_idx, t_xyz = torch.from_numpy(d2.copy()), torch.from_numpy(idx.copy()), torch.from_numpy(xyz.copy())
        dist.exchange_best_match(t_d2, t_idx, t_xyz)
        assert np.array_equal(t_idx.numpy(), bi) and np.array_equal(t_d2.numpy(), bd) and np.array_equal(t_xyz.numpy(), bx)
        Path(tmp, f"ag{rank}").write_text("ok")
    finally:
        td.destroy_process_group()


def test_allgather_exchange_world2(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker_allgather, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"ag{r}").exists() for r in range(2))


class _FakeCtx:
    """Stands in for _lib.Context in dist.attach: records what attach did; `fail_on` makes that rank's communicator fail."""

    def __init__(self, rank, fail_on):
        self.rank, self.fail_on, self.calls = rank, fail_on, []

    def set_partition(self, mode):
        self.calls.append(("partition", mode))

    def comm_unique_id(self):
        if self.fail_on == "id":
            raise RuntimeError("librccl could not be loaded")
        return b"\x01" * 128

    def comm_init(self, uid, rank, world, gn_shard=False):
        assert uid == b"\x01" * 128
        if self.fail_on == rank:
            raise RuntimeError("ncclCommInitRank failed")
        self.calls.append(("comm_init", rank, world))

    def comm_destroy(self):
        self._comm_key = self._comm_group = None           # (as _lib.Context.comm_destroy does)
        self.calls.append(("comm_destroy",))

    def comm_info(self):
        live = False
        for x in self.calls:
            live = True if x[0] == "comm_init" else (False if x[0] == "comm_destroy" else live)
        return {"communicator": live}

    def comm_activate(self, on=True, gn_shard=False):
        self.calls.append(("comm_activate", bool(on)))

    def set_exchange(self, fn, rank, world, gn_shard=False):
        self.calls.append(("callback", rank, world))


def _worker_attach(rank, world, port, tmp):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("SICP_XCHG", None)
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from simpleicp_amd import dist
        dist.make_exchange = lambda ctx, group=None: (lambda *a: 0)          # (the real one wraps device pointers)
        td.get_backend = lambda group=None: "nccl"                           # (a gloo group goes straight to the callback)
        # every rank's communicator comes up -> the library-owned RCCL path on every rank
        c = _FakeCtx(rank, fail_on=None)
        assert dist.attach(c, partition=1) == "rccl" and ("comm_init", rank, world) in c.calls and ("callback", rank, world) not in c.calls
        # the communicator stays with the context: end of the run parks it, the next run on the same group switches it back
        # on without another rendezvous
        dist.detach(c)
        assert c.calls[-3:] == [("comm_activate", False), ("callback", 0, 1), ("partition", 0)]
        n_init = sum(x[0] == "comm_init" for x in c.calls)
        assert dist.attach(c) == "rccl" and c.calls[-1] == ("comm_activate", True) and sum(x[0] == "comm_init" for x in c.calls) == n_init
        # a run that failed forgets the communicator (dist.forget; the library aborts it itself on an exchange timeout): the next
        # attach builds a new one instead of reviving a communicator that may be out of step -- also after a plain comm_destroy
        dist.detach(c)
        dist.forget(c)
        assert dist.attach(c) == "rccl" and sum(x[0] == "comm_init" for x in c.calls) == n_init + 1
        dist.detach(c)
        c.comm_destroy()
        assert dist.attach(c) == "rccl" and sum(x[0] == "comm_init" for x in c.calls) == n_init + 2
        # ONE rank fails -> ALL ranks drop their communicator and register the callback exchange
        c = _FakeCtx(rank, fail_on=1)
        assert dist.attach(c) == "callback" and ("comm_destroy",) in c.calls and c.calls[-1] == ("callback", rank, world)
        # rank 0 cannot even create an id (librccl missing)
        c = _FakeCtx(rank, fail_on="id" if rank == 0 else None)
        assert dist.attach(c) == "callback" and c.calls[-1] == ("callback", rank, world)
        os.environ["SICP_XCHG"] = "callback"
        c = _FakeCtx(rank, fail_on=None)
        assert dist.attach(c) == "callback" and not any(x[0] == "comm_init" for x in c.calls)
        # the queries partition replicates the cloud: taken only when EVERY rank has room for it (one rank short -> all say no)
        class _Mem:
            def __init__(self, free): self.free = free
            def device_memory(self): return self.free, 288 << 30
        roomy, tight = _Mem(200 << 30), _Mem(1 << 30)
        assert dist.queries_partition_fits(roomy, 100_000_000) and not dist.queries_partition_fits(tight, 100_000_000)
        assert dist.agree(dist.queries_partition_fits(roomy, 100_000_000)) is True
        assert dist.agree(dist.queries_partition_fits(roomy if rank == 0 else tight, 100_000_000)) is False
        Path(tmp, f"at{rank}").write_text("ok")
    finally:
        os.environ.pop("SICP_XCHG", None)
        td.destroy_process_group()


def test_attach_falls_back_to_the_callback_on_every_rank_together(tmp_path):
    """dist.attach: the library's own RCCL communicator by default; if it fails on ANY rank (or rank 0 cannot load
    librccl) every rank agrees to use the torch.distributed callback exchange -- no rank is left waiting in a collective
    the others never enter."""
    import torch.multiprocessing as mp
    mp.spawn(_worker_attach, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"at{r}").exists() for r in range(2))
