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


def _worker(rank, world, port, counts, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    from alphazero_piskvorky_amd import parallel
    eng = FakeEngine(rank, counts[rank])
    packed, got = parallel.gather_packed_records(eng, torch.device("cpu"))
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


def test_single_process_gather_is_identity():
    from alphazero_piskvorky_amd import parallel
    eng = FakeEngine(0, 6)
    packed, counts = parallel.gather_packed_records(eng, torch.device("cpu"))
    assert counts == [6] and np.array_equal(packed.numpy(), eng.payload())
    assert parallel.shard_games(10, 1, 4) == [1, 5, 9]
