"""Multi-GPU exchange of the ICP iteration: one process per GPU, torch.distributed collectives
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

Partitioning: the MOVABLE cloud is sharded by contiguous index range,
the Q selected fixed points (+ normals, planarity) are replicated.  Per iteration there is ONE
exchange step after the local search:

    every rank holds, per query, its shard's best (d2, global idx, xyz of that point)
    -> ONE all_gather of the packed records (Q * 40 B per rank; the library packs them and calls
       back with device pointers, `make_exchange`)
    -> lexicographic (d2, idx) minimum over ranks (library kernel `k_lexmin_gathered`; the same
       rule is spelled out in torch ops in `exchange_best_match` below, which the CPU/gloo tests
       and the GPU test of the kernel use as the reference) == the single-GPU (d2, idx) rule

after which every rank owns the complete correspondence set, so the rejection statistics
(median / MAD are not sums) need no further collective.  The 6x6 normal-equation reduction can
either be replicated (no collective) or sharded over ranks with a SUM all-reduce of the 30
accumulators per solver step (``gn_shard=True``) -- the latter pays off only for very large Q.

The functions here work on torch tensors of any device so that the gloo/CPU tests exercise the
exact code the GPU path runs.
"""
from __future__ import annotations

import sys

from . import _lib


def _td():
    """torch.distributed if the caller's process has imported it -- a process group cannot have been
    initialised otherwise, and a single-GPU run() must not pay (seconds) for importing torch."""
    return sys.modules.get("torch.distributed")


def is_distributed() -> bool:
    td = _td()
    try:
        return bool(td) and td.is_available() and td.is_initialized() and td.get_world_size() > 1
    except Exception:  # noqa: BLE001
        return False


def is_initialized() -> bool:
    td = _td()
    try:
        return bool(td) and td.is_available() and td.is_initialized()
    except Exception:  # noqa: BLE001
        return False


def rank_world():
    import torch.distributed as td
    return td.get_rank(), td.get_world_size()


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of n rows owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def exchange_best_match(d2, idx, xyz, group=None):
    """In place: replace (d2[Q], idx[Q] int64, xyz[Q,3]) by the job-wide lexicographic
    (d2, idx) minimum and the coordinates that travel with it.  idx == -1 marks "no candidate"
    (d2 == +inf) and loses against any real candidate."""
    import torch
    import torch.distributed as td
    world = td.get_world_size(group)
    g_d2 = [torch.empty_like(d2) for _ in range(world)]
    g_idx = [torch.empty_like(idx) for _ in range(world)]
    g_xyz = [torch.empty_like(xyz) for _ in range(world)]
    td.all_gather(g_d2, d2.contiguous(), group=group)
    td.all_gather(g_idx, idx.contiguous(), group=group)
    td.all_gather(g_xyz, xyz.contiguous(), group=group)
    D = torch.stack(g_d2)                                   # (world, Q)
    I = torch.stack(g_idx)
    X = torch.stack(g_xyz)                                  # (world, Q, 3)
    dmin = D.min(dim=0).values
    big = torch.iinfo(torch.int64).max
    cand = torch.where((D == dmin) & (I >= 0), I, torch.full_like(I, big))
    imin = cand.min(dim=0).values
    winner = (cand == imin).to(torch.int64).argmax(dim=0)   # first rank holding the winner
    none = imin == big
    d2.copy_(torch.where(none, torch.full_like(dmin, float("inf")), dmin))
    idx.copy_(torch.where(none, torch.full_like(imin, -1), imin))
    picked = X.gather(0, winner.view(1, -1, 1).expand(1, -1, 3))[0]
    xyz.copy_(torch.where(none.unsqueeze(1), torch.zeros_like(picked), picked))
    return d2, idx, xyz


_SIGN = -(1 << 63)          # int64 whose bit pattern is 0x8000...: XOR with it maps unsigned order onto signed order


def allreduce_u64(bits, take_max, group=None):
    """In place: element-wise UNSIGNED minimum / maximum over the ranks of 8-byte words held as an int64 tensor (torch has no
    uint64 reductions): flipping the sign bit maps the unsigned order onto the signed one, the reduction runs on that, the flip is
    undone.  What SICP_XCHG_MIN_U64 / SICP_XCHG_MAX_U64 ask of a callback (include/simpleicp_hip.h)."""
    import torch.distributed as td
    bits ^= _SIGN
    td.all_reduce(bits, op=td.ReduceOp.MAX if take_max else td.ReduceOp.MIN, group=group)
    bits ^= _SIGN
    return bits


def exchange_best_keys(d2, idx, xyz, group=None):
    """The job-wide lexicographic (d2, idx) winner by THREE reductions on 8-byte keys instead of gathering every rank's record --
    the torch restatement of the library's `exchange_best_keys_chained` (sicp_comm.cpp; kernels k_xkey_* in sicp_kernels.hip),
    which cloud shards use from SICP_XCHG_KEYS_MIN_Q queries on.  In place on (d2[Q] f64, idx[Q] int64 GLOBAL indices, xyz[Q,3]);
    idx == -1 / d2 == +inf marks "no candidate here".
      1. min over ranks of the squared distance's bit pattern (non-negative doubles order like their bits as unsigned integers;
         all-ones = no candidate);
      2. min over ranks of the index, offered only by the ranks whose distance IS that minimum (all-ones otherwise): together the
         lexicographic minimum -- and since indices are global and the shards disjoint, exactly ONE rank owns the winner;
      3. max over ranks of the winner's coordinates as bit patterns, the owner's against zeros -- bit-exact where a sum would turn
         -0.0 into +0.0.
    Equals `exchange_best_match` word for word (tests/test_dist_gloo.py)."""
    import torch
    all_ones = -1                                            # int64 view of 0xffff...
    none = idx < 0
    key = d2.contiguous().view(torch.int64).clone()
    key[none] = all_ones
    gmin = allreduce_u64(key, False, group)
    mine = (~none) & (d2.contiguous().view(torch.int64) == gmin)
    ikey = torch.where(mine, idx, torch.full_like(idx, all_ones))
    gidx = allreduce_u64(ikey.clone(), False, group)
    own = (~none) & (idx == gidx)
    xb = torch.where(own.unsqueeze(1), xyz.contiguous().view(torch.int64), torch.zeros_like(xyz, dtype=torch.int64)).contiguous()
    allreduce_u64(xb.view(-1), True, group)
    nobody = gmin == all_ones
    d2.copy_(torch.where(nobody, torch.full_like(d2, float("inf")), gmin.view(torch.float64)))
    idx.copy_(torch.where(nobody, torch.full_like(idx, -1), gidx))
    xyz.copy_(torch.where(nobody.unsqueeze(1), torch.zeros_like(xyz), xb.view(torch.float64).view_as(xyz)))
    return d2, idx, xyz


def allreduce_sum(buf, group=None):
    import torch.distributed as td
    td.all_reduce(buf, op=td.ReduceOp.SUM, group=group)
    return buf


class _DevArray:
    """Zero-copy view of library-owned device memory for torch (CUDA array interface v2)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _wrap(ptr, shape, typestr, device):
    import torch
    return torch.as_tensor(_DevArray(ptr, shape, typestr), device=device)


def allgather_into(recv, send, group=None):
    """recv[(world*count)] <- every rank's send[(count)], rank order."""
    import torch.distributed as td
    td.all_gather_into_tensor(recv, send, group=group)
    return recv


def attach(ctx, gn_shard=False, partition=None, group=None):
    """Give `ctx` the job's collectives.  Default: the library's OWN RCCL communicator -- rank 0's ncclUniqueId is
    broadcast over the existing torch.distributed group and every rank calls sicp_comm_init (rendezvous + a first
    all-gather, both awaited with a deadline: it cannot hang); all-gather / all-reduce are then enqueued by the library on
    its stream between its kernels, no Python in the loop.  The communicator STAYS with the context: the next run on the
    same group switches it back on (sicp_comm_activate) instead of building another one.  SICP_XCHG=callback (or a group
    without GPUs) registers the torch.distributed callback of `make_exchange` instead.  Returns "rccl" or "callback"."""
    import os
    import torch.distributed as td
    rank, world = td.get_rank(group), td.get_world_size(group)
    if partition is not None:
        ctx.set_partition(partition)
    if os.environ.get("SICP_XCHG", "rccl") != "callback" and td.get_backend(group) != "gloo":
        pg = group if group is not None else td.group.WORLD
        key = (id(pg), rank, world)                  # (ctx._comm_group keeps `pg` alive, so the id cannot be recycled)
        if (getattr(ctx, "_comm_key", None) == key and getattr(ctx, "_comm_group", None) is pg
                and ctx.comm_info()["communicator"]):
            # an earlier run left its communicator parked on this context.  Every rank is in the same position: a
            # communicator is only ever kept when ALL ranks reported success below -- and a run that FAILS forgets it on
            # every rank (forget, below) -- so no agreement round is needed
            ctx.comm_activate(True, gn_shard=gn_shard)
            return "rccl"
        ctx._comm_key = ctx._comm_group = None
        err = None
        try:
            box = [ctx.comm_unique_id() if rank == 0 else None]
        except Exception as exc:  # noqa: BLE001  (librccl missing on rank 0: everybody must learn of it)
            box, err = [None], exc
        td.broadcast_object_list(box, src=td.get_global_rank(group, 0) if group is not None else 0, group=group)
        if box[0] is None:
            err = err or RuntimeError("rank 0 could not create an RCCL id")
        else:
            try:
                ctx.comm_init(box[0], rank, world, gn_shard=gn_shard)
            except Exception as exc:  # noqa: BLE001
                err = exc
        # all ranks take the same road: one failed communicator sends the whole job to the callback exchange
        verdicts = [None] * world
        td.all_gather_object(verdicts, None if err is None else f"rank {rank}: {err}", group=group)
        failed = [v for v in verdicts if v is not None]
        if not failed:
            ctx._comm_key, ctx._comm_group = key, pg
            return "rccl"
        ctx.comm_destroy()
        if rank == 0:
            print("simpleicp_amd: library-owned RCCL communicator unavailable (" + "; ".join(failed)
                  + "); using the torch.distributed callback exchange", file=sys.stderr, flush=True)
    ctx.set_exchange(make_exchange(ctx, group), rank, world, gn_shard=gn_shard)
    return "callback"


def forget(ctx):
    """After a sharded run raised: whatever communicator the context holds is not to be trusted again (the library aborts it
    itself on an exchange timeout; a rank that failed for another reason may be out of step with its peers) -- the next
    attach() builds a new one."""
    try:
        ctx.comm_destroy()
    except Exception:  # noqa: BLE001
        ctx._comm_key = ctx._comm_group = None


def detach(ctx):
    """Back to single-GPU behaviour (the exchange lives on the process-wide context); a communicator is parked, not
    destroyed -- Context.close() / comm_destroy() ends it."""
    ctx.comm_activate(False)
    ctx.set_exchange(None, 0, 1)
    ctx.set_partition(_lib.PART_CLOUD)


def agree(flag: bool, group=None) -> bool:
    """True iff `flag` is true on EVERY rank (one tiny object all-gather; only called when a job is about to choose a
    partition from a per-rank measurement)."""
    import torch.distributed as td
    votes = [None] * td.get_world_size(group)
    td.all_gather_object(votes, bool(flag), group=group)
    return all(votes)


def queries_partition_fits(ctx, n_points, headroom=0.5):
    """Query shards replicate the WHOLE searched cloud on every rank: coordinates (24 B / point), the grid's packed records
    (32 B), cell offsets and the subsample (about 8 B) -- take that road only when it fits comfortably in what is free now."""
    try:
        free, _total = ctx.device_memory()
    except Exception:  # noqa: BLE001
        return False
    return 64 * int(n_points) <= headroom * free


def exchange_query_slices(d2, idx, xyz, group=None):
    """Query-sharded mode in torch ops (the reference the tests hold the library's path against): rank r matched the
    queries [r * per, (r + 1) * per), per = ceil(Q / world); gathering the slices in rank order restores query order.
    In place on the full-size (Q) tensors."""
    import torch
    import torch.distributed as td
    world, rank = td.get_world_size(group), td.get_rank(group)
    Q = d2.shape[0]
    per = (Q + world - 1) // world
    lo = min(Q, per * rank)
    hi = min(Q, lo + per)

    def gather(t, fill):
        pad = torch.full((per,) + tuple(t.shape[1:]), fill, dtype=t.dtype, device=t.device)
        pad[: hi - lo] = t[lo:hi]
        parts = [torch.empty_like(pad) for _ in range(world)]
        td.all_gather(parts, pad, group=group)
        return torch.cat(parts)[:Q]

    d2.copy_(gather(d2, float("inf")))
    idx.copy_(gather(idx, -1))
    xyz.copy_(gather(xyz, 0.0))
    return d2, idx, xyz


def make_exchange(ctx, group=None, synchronous=None):
    """Callback for Context.set_exchange.  The library hands over DEVICE pointers it owns (stable
    across iterations, so the zero-copy tensor views are cached) and reduces the gathered records
    itself; the host side only issues the collective.  By default the collective is enqueued IN
    ORDER ON THE LIBRARY'S OWN STREAM (torch.cuda.ExternalStream), so nothing blocks the host
    between the pack kernel, the all-gather and the reduce kernel; SICP_XCHG_SYNC=1 (or
    synchronous=True) restores the blocking variant.  A gloo group (no GPU collectives; also what lets several ranks share
    ONE GPU in tests/test_gpu_exchange.py) is served by staging through host memory."""
    import os
    import torch
    import torch.distributed as td

    device = ctx.device
    dev = torch.device("cuda", device)
    world = td.get_world_size(group)
    if synchronous is None:
        synchronous = os.environ.get("SICP_XCHG_SYNC") == "1"
    lib_stream = torch.cuda.ExternalStream(ctx.stream_ptr(), device=dev)
    views = {}

    def view(ptr, count, typestr="<f8"):
        key = (ptr, count, typestr)
        t = views.get(key)
        if t is None:
            if len(views) > 64:
                views.clear()
            t = views[key] = _wrap(ptr, (count,), typestr, dev)
        return t

    serve_u64 = os.environ.get("SICP_XCHG_U64", "1") != "0"      # =0: answer like a callback written for ABI 5 (tests of the decline)

    host_staged = td.get_backend(group) == "gloo"      # a group without GPU collectives: stage through host memory, blocking

    def fn_host(what, a, b, c, count):
        if what in (_lib.XCHG_MIN_U64, _lib.XCHG_MAX_U64) and count == 0:
            return 0 if serve_u64 else 1                # the library's question at registration: is the operation served?
        with torch.cuda.device(dev):
            lib_stream.synchronize()
            send = view(a, count).cpu() if what in (_lib.XCHG_ALLGATHER_F64, _lib.XCHG_SUM_F64) else None
            if what == _lib.XCHG_ALLGATHER_F64:
                recv = torch.empty(count * world, dtype=torch.float64)
                allgather_into(recv, send, group)
                view(b, count * world).copy_(recv)
            elif what == _lib.XCHG_SUM_F64:
                allreduce_sum(send, group)
                view(a, count).copy_(send)
            elif what in (_lib.XCHG_MIN_U64, _lib.XCHG_MAX_U64) and serve_u64:
                bits = view(a, count, "<i8").cpu()
                allreduce_u64(bits, what == _lib.XCHG_MAX_U64, group)
                view(a, count, "<i8").copy_(bits)
            else:
                return 1
            torch.cuda.synchronize(dev)
        return 0

    def fn(what, a, b, c, count):
        if host_staged:
            return fn_host(what, a, b, c, count)
        if what in (_lib.XCHG_MIN_U64, _lib.XCHG_MAX_U64) and count == 0:
            return 0 if serve_u64 else 1
        with torch.cuda.device(dev), torch.cuda.stream(lib_stream):
            if synchronous:
                lib_stream.synchronize()
            if what == _lib.XCHG_ALLGATHER_F64:
                allgather_into(view(b, count * world), view(a, count), group)
            elif what == _lib.XCHG_SUM_F64:
                allreduce_sum(view(a, count), group)
            elif what in (_lib.XCHG_MIN_U64, _lib.XCHG_MAX_U64) and serve_u64:
                allreduce_u64(view(a, count, "<i8"), what == _lib.XCHG_MAX_U64, group)      # (two in-place XORs around the collective, same stream)
            else:
                return 1
            if synchronous:
                lib_stream.synchronize()
        return 0
    return fn
