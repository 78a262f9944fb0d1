"""world_size-2 gloo test of the episode-end record exchange (alphazero-piskvorky_amd/parallel.py)."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

from tests.util import ROOT


class FakeEngine:
    """Stands in for the HIP engine on CPU: `count` records of `record_bytes` bytes with a recognisable pattern."""

    def __init__(self, rank, count, record_bytes=168):
        self.last_records, self.record_bytes, self.rank = count, record_bytes, rank

    def payload(self):
        return ((np.arange(self.last_records * self.record_bytes) * 7 + self.rank * 13) % 251).astype(np.uint8)

    def pack_into(self, ptr):
        buf = self.payload()
        ctypes.memmove(ptr, buf.ctypes.data, buf.nbytes)


def _worker(rank, world, port, counts, q, dst=None):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    from alphazero_piskvorky_amd import parallel
    eng = FakeEngine(rank, counts[rank])
    packed, got = parallel.gather_packed_records(eng, torch.device("cpu"), dst=dst)
    q.put((rank, got, packed.numpy().copy()))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize("counts", [(5, 3), (0, 4), (2, 0)])
def test_gather_packed_records_world2_gloo(counts):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + sum(counts)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, counts, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect = np.concatenate([FakeEngine(r, counts[r]).payload() for r in range(2)])
    for rank, got, packed in outs:
        assert list(got) == list(counts)
        assert np.array_equal(packed, expect), f"rank {rank} received a different record stream"


@pytest.mark.parametrize("counts,dst", [((5, 3), 0), ((0, 4), 1), ((2, 0), 0)])
def test_gather_to_root_world2_gloo(counts, dst):
    """Gather to one rank (the reference's single trainer): the root holds every rank's records in rank order, the other
    rank receives the counts only."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29100 + (os.getpid() % 500) + sum(counts) + dst
    procs = [ctx.Process(target=_worker, args=(r, 2, port, counts, q, dst)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect = np.concatenate([FakeEngine(r, counts[r]).payload() for r in range(2)])
    for rank, got, packed in outs:
        assert list(got) == list(counts)
        if rank == dst:
            assert np.array_equal(packed, expect)
        else:
            assert packed.size == 0


def test_single_process_gather_is_identity():
    from alphazero_piskvorky_amd import parallel
    eng = FakeEngine(0, 6)
    packed, counts = parallel.gather_packed_records(eng, torch.device("cpu"))
    assert counts == [6] and np.array_equal(packed.numpy(), eng.payload())


def _tally_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    from alphazero_piskvorky_amd import parallel
    tally = parallel.all_reduce_tally(3 + rank, 1, 2 * rank, torch.device("cpu"))
    seed = parallel.broadcast_seed(1000 + 77 * rank, torch.device("cpu"))
    q.put((rank, tally, seed))
    td.barrier()
    td.destroy_process_group()


def test_arena_tally_and_seed_world2_gloo():
    """the arena's only exchange is the sum of (wins, losses, draws); every rank plays with rank 0's seed."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + (os.getpid() % 90)
    procs = [ctx.Process(target=_tally_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, tally, seed in outs:
        assert tuple(tally) == (7, 2, 2) and seed == 1000


def test_arena_blocks_are_even_aligned_and_cover_all_games():
    from alphazero_piskvorky_amd import parallel
    for games in (1, 2, 7, 20, 51, 64):
        for world in (1, 2, 3, 8):
            blocks = [parallel.arena_block(games, r, world) for r in range(world)]
            assert all(lo % 2 == 0 or lo == hi for lo, hi in blocks)       # local parity == global parity (evaluator.py:64-69)
            covered = [g for lo, hi in blocks for g in range(lo, hi)]
            assert covered == list(range(games))
    assert parallel.all_reduce_tally(4, 5, 6, torch.device("cpu")) == (4, 5, 6)     # no process group: identity
    assert parallel.broadcast_seed(99, torch.device("cpu")) == 99


def _bcast_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    from alphazero_piskvorky_amd import parallel
    from alphazero_piskvorky_amd.net import GomokuResNet
    torch.manual_seed(100 + rank)                     # different weights and BatchNorm statistics on every rank
    net = GomokuResNet(board_size=5)
    net.bn.running_mean.add_(rank + 1.0)
    parallel.broadcast_module_(net, src=0)
    q.put((rank, {k: v.numpy().copy() for k, v in net.state_dict().items()}))
    td.barrier()
    td.destroy_process_group()


def test_weight_broadcast_world2_gloo():
    """After the optimizer steps every rank continues with rank 0's parameters and buffers."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + (os.getpid() % 90)
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict((r, sd) for r, sd in (q.get(timeout=120) for _ in range(2)))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert set(outs[0]) == set(outs[1]) and len(outs[0]) > 20
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert float(outs[1]["bn.running_mean"][0]) == 1.0          # rank 0's buffer (0 + 1), not rank 1's
