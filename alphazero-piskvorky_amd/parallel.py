"""Multi-GPU sharding of self-play.

Games never interact (self_play.py:41-73), so ranks play disjoint game ids with no data-path
collective; the only exchange is at episode end.  It lives INSIDE the C-ABI library (include/az_engine.h az_dist_*:
ncclAllGather of the record counts, then grouped ncclSend / ncclRecv of the true sizes to the rank that trains -- or to
every rank -- on the engine's own stream, and the arena's tally as one ncclAllReduce); this module only bootstraps the
library's communicator from a torch.distributed "nccl" group (128 bytes of unique id over the group) and calls it.
With a gloo group (the CPU tests, ranks sharing one GPU) the same exchange runs on torch.distributed collectives staged
through the host, padded to the largest count.  This replaces result_queue.put/get (self_play.py:73,140).
"""
import numpy as np
import torch


def _dist_on(force):
    """True when collectives must run: a process group exists and has more than one rank -- or `force`, which drives a
    one-rank group through the very same collective calls (how the RCCL branch is exercised on a single GPU)."""
    import torch.distributed as td
    return td.is_available() and td.is_initialized() and (td.get_world_size() > 1 or force)


def rank_world():
    import torch.distributed as td
    return (td.get_rank(), td.get_world_size()) if td.is_available() and td.is_initialized() else (0, 1)


def all_gather_packed(packed, count, record_bytes, dst=None, force=False):
    """packed: uint8 tensor [count*record_bytes] on this rank's device.  Returns (list of per-rank uint8 tensors, counts).
    dst=None: all-gather, every rank receives every rank's records.  dst=r: gather to rank r only (the reference has a
    single trainer, train.py:95-104); the other ranks get an empty list of parts and the counts."""
    import torch.distributed as td
    if not _dist_on(force):
        return [packed[: count * record_bytes]], [int(count)]
    world, rank = td.get_world_size(), td.get_rank()
    dev = packed.device
    cnt = torch.tensor([count], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    td.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    buf = torch.zeros(mx * record_bytes, dtype=torch.uint8, device=dev)
    buf[: count * record_bytes] = packed[: count * record_bytes]
    if dst is None:
        outs = [torch.empty_like(buf) for _ in range(world)]
        td.all_gather(outs, buf)
    else:
        outs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
        td.gather(buf, gather_list=outs, dst=dst)
        if rank != dst:
            return [], counts
    return [o[: c * record_bytes] for o, c in zip(outs, counts)], counts


last_exchange = None      # which path the last gather_packed_records took: "az_dist" (RCCL inside the library) | "torch.distributed" | "local"


def engine_comm(engine, device, force=False):
    """True when `engine` holds an RCCL communicator over the ranks of the current torch.distributed group, creating it on
    first use: rank 0 draws the unique id and the group carries its 128 bytes to everybody.  Only for "nccl" groups (a gloo
    group means CPU tests or ranks sharing a GPU, which RCCL refuses)."""
    import torch.distributed as td
    if not _dist_on(force) or td.get_backend() != "nccl":
        return False
    world, rank = td.get_world_size(), td.get_rank()
    if engine.dist_world() == world and getattr(engine, "_dist_group_ok", False):
        return True
    if getattr(engine, "_dist_failed", False):
        return False
    from ._capi import AZ_DIST_ID_BYTES, AzError, dist_unique_id
    # every rank learns whether EVERY rank could join (a rank that cannot load RCCL must not leave the others waiting in a
    # collective): the id travels with a validity flag, and the outcome of az_dist_init is summed over the group
    idt = torch.zeros(AZ_DIST_ID_BYTES + 1, dtype=torch.uint8, device=device)
    if rank == 0:
        try:
            idt[:AZ_DIST_ID_BYTES].copy_(torch.frombuffer(bytearray(dist_unique_id()), dtype=torch.uint8))
            idt[AZ_DIST_ID_BYTES] = 1
        except AzError:
            pass
    td.broadcast(idt, 0)
    host = idt.cpu().numpy()
    ok = int(host[AZ_DIST_ID_BYTES]) == 1
    if ok:
        try:
            engine.dist_init(host[:AZ_DIST_ID_BYTES].tobytes(), rank, world)
        except AzError:
            ok = False
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=device)
    td.all_reduce(flag)
    if int(flag.item()) != world:
        engine._dist_failed = True        # fall back to torch.distributed's collectives (the same RCCL underneath)
        return False
    engine._dist_group_ok = True
    return True


def gather_packed_records(engine, device, dst=None, force=False):
    """Pack this rank's episode records on the device and exchange them.  Returns (uint8 tensor of all records
    in rank order -- empty on the ranks a gather-to-root leaves out --, per-rank counts)."""
    import torch.distributed as td
    global last_exchange
    if engine_comm(engine, device, force):
        # the library's own exchange: true sizes, only to the ranks that asked
        last_exchange = "az_dist"
        counts = engine.dist_counts()
        receive = dst is None or dst == engine.dist_rank()
        total = sum(counts) if receive else 0
        out = torch.empty(max(total, 1) * engine.record_bytes, dtype=torch.uint8, device=device)
        engine.dist_gather_records(-1 if dst is None else int(dst), out.data_ptr() if receive else 0)
        return out[: total * engine.record_bytes], counts
    last_exchange = "torch.distributed" if _dist_on(force) else "local"
    count = engine.last_records
    packed = torch.zeros(max(count, 1) * engine.record_bytes, dtype=torch.uint8, device=device)
    if count:
        engine.pack_into(packed.data_ptr())
    # RCCL moves device buffers directly; with a gloo group (CPU tests, shared-GPU rehearsal) stage through the host
    host_xchg = td.is_available() and td.is_initialized() and td.get_backend() != "nccl" and packed.device.type != "cpu"
    parts, counts = all_gather_packed(packed.cpu() if host_xchg else packed, count, engine.record_bytes, dst=dst, force=force)
    if not parts:
        return torch.zeros(0, dtype=torch.uint8, device=device), counts
    out = torch.cat(parts) if len(parts) > 1 else parts[0]
    return (out.to(device) if host_xchg else out), counts


def arena_block(num_games, rank, world):
    """Contiguous block [lo, hi) of arena game indices for this rank.  Blocks start at EVEN indices so a rank's local
    game parity equals the global one (odd game index => the baseline moves first, evaluator.py:64-69) and its
    per-game seeds are seed0 + global index with a plain offset."""
    per = (num_games + world - 1) // world
    per += per & 1
    lo = min(rank * per, num_games)
    return lo, min(lo + per, num_games)


def all_reduce_tally(wins, losses, draws, device, force=False, engine=None):
    """Arena tally exchange: sum of three integers over the ranks -- az_dist_allreduce_sum inside the library
    when an engine is given and the group is RCCL, torch.distributed otherwise (gloo)."""
    import torch.distributed as td
    if not _dist_on(force):
        return int(wins), int(losses), int(draws)
    if engine is not None and engine_comm(engine, device, force):
        w, l, d = engine.dist_allreduce_sum([wins, losses, draws])
        return w, l, d
    on_dev = td.get_backend() == "nccl"
    t = torch.tensor([wins, losses, draws], dtype=torch.int64, device=device if on_dev else "cpu")
    td.all_reduce(t, op=td.ReduceOp.SUM)
    w, l, d = (int(x) for x in t.tolist())
    return w, l, d


def broadcast_seed(seed, device, force=False):
    """Rank 0's seed for everybody (the reference is unseeded; ranks must still agree on the episode's game seeds)."""
    import torch.distributed as td
    if not _dist_on(force):
        return int(seed)
    on_dev = td.get_backend() == "nccl"
    t = torch.tensor([int(seed)], dtype=torch.int64, device=device if on_dev else "cpu")
    td.broadcast(t, 0)
    return int(t.item())


def broadcast_module_(module, src=0, force=False):
    """Rank `src`'s parameters and buffers for everybody, in place (the weight exchange after the optimizer
    steps, so that every rank searches with the same net in the next episode).  RCCL moves the device tensors directly;
    with a gloo group they are staged through the host."""
    import torch.distributed as td
    if not _dist_on(force):
        return module
    direct = td.get_backend() == "nccl"
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if direct or t.device.type == "cpu":
                td.broadcast(t.data, src)
            else:
                h = t.data.cpu()
                td.broadcast(h, src)
                t.data.copy_(h)
    return module
