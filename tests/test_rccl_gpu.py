"""The RCCL branch of the episode-end exchange on ONE GPU: a one-rank `nccl` process group driven through the very same
collective calls a multi-GPU job makes (parallel.py `force=True` drops the world == 1 short-circuit), with device
tensors; and bench.py launched the way the driver launches it for N > 1 (torch.distributed.run), with one rank."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.util import ROOT

WORKER = r'''
import os, sys, json, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as td
torch.cuda.set_device(0)
td.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert td.get_backend() == "nccl" and td.get_world_size() == 1
import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd import parallel
from alphazero_piskvorky_amd.net import GomokuNet
dev = torch.device("cuda", 0)
out = {}
# an episode's records through the collective branch: all-gather and gather-to-root
e = az.Engine(5, 4, 16, 8, synthetic=True)
e.selfplay(6, seed0=5)
want = torch.zeros(e.last_records * e.record_bytes, dtype=torch.uint8, device=dev)
e.pack_into(want.data_ptr())
torch.cuda.synchronize()
for dst in (None, 0):
    packed, counts = parallel.gather_packed_records(e, dev, dst=dst, force=True)
    assert packed.is_cuda and counts == [e.last_records]
    assert torch.equal(packed, want)
out["records"] = int(e.last_records)
assert e.dist_world() == 1 and e.dist_rank() == 0        # parallel.py made the library's own communicator from the group
e.close()
# the same exchange with no torch.distributed in the call path: the C-ABI's az_dist_* on a communicator of its own
from alphazero_piskvorky_amd import _capi
e2 = az.Engine(5, 4, 16, 8, synthetic=True)
try:
    e2.dist_counts(); raise SystemExit("az_dist_counts before az_dist_init must fail")
except _capi.AzError:
    pass
e2.dist_init(_capi.dist_unique_id(), 0, 1)
assert e2.dist_counts() == [0]                           # no episode yet: this rank contributes nothing
e2.selfplay(5, seed0=9)
want2 = torch.zeros(e2.last_records * e2.record_bytes, dtype=torch.uint8, device=dev)
e2.pack_into(want2.data_ptr()); torch.cuda.synchronize()
assert e2.dist_counts() == [e2.last_records]
for dst in (0, -1):
    got = torch.zeros_like(want2)
    e2.dist_gather_records(dst, got.data_ptr())
    assert torch.equal(got, want2)
assert e2.dist_allreduce_sum([7, 2, 1]) == [7, 2, 1]
t = torch.arange(1000, dtype=torch.int32, device=dev); keep = t.clone()
e2.dist_broadcast(t.data_ptr(), t.numel() * 4, 0); assert torch.equal(t, keep)
e2.close()
assert parallel.all_reduce_tally(7, 2, 1, dev, force=True) == (7, 2, 1)
assert parallel.broadcast_seed(4242, dev, force=True) == 4242
net = GomokuNet(board_size=5).to(dev)
before = [p.detach().clone() for p in net.parameters()]
parallel.broadcast_module_(net, force=True)
assert all(torch.equal(a, b) for a, b in zip(before, net.parameters()))
td.barrier()
td.destroy_process_group()
json.dump(out, open(sys.argv[2], "w"))
'''


def test_one_rank_nccl_group_runs_every_collective_of_the_path(tmp_path):
    script = tmp_path / "rccl_worker.py"
    script.write_text(WORKER)
    res = tmp_path / "out.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(script), ROOT, str(res)], env=env, timeout=600, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.load(open(res))["records"] > 0


def _bench(cmd, env):
    p = subprocess.run(cmd, env=env, timeout=900, capture_output=True, text=True, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_under_torch_distributed_run_with_one_rank_agrees_with_plain_run():
    """bench.py --gpus 1 launched through torch.distributed.run initialises the nccl group and sends its barrier, the
    max/sum all-reduces and the record gather through RCCL; the deterministic quantities of the line (expansions and
    simulations played in the timed region are functions of the seeds alone) equal the plain N = 1 run's."""
    args = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--board", "5", "--win", "4", "--sims", "40", "--slots", "64",
            "--no-cpu"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    plain = _bench([sys.executable, "bench.py"] + args, env)
    dist = _bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                   "--master-addr", "127.0.0.1", "--master-port", "29743", "bench.py"] + args, env)
    assert dist["collectives"] == "nccl" and plain["collectives"] is None
    for key in ("metric", "unit", "n_gpus", "steps", "warmup", "dtype", "config"):
        assert plain[key] == dist[key], key
    for key in ("mean_select_depth", "terminal_hit_fraction"):
        assert plain[key] == dist[key], key
    assert plain["episode"]["games"] == dist["episode"]["games"] == 64
    assert plain["episode"]["records_gathered"] == dist["episode"]["records_gathered"]
    assert plain["episode"]["mean_plies_per_game"] == dist["episode"]["mean_plies_per_game"]
    assert dist["value"] > 0 and 0.3 < dist["value"] / plain["value"] < 3.0


def test_bench_gpus_n_launches_n_ranks_by_itself():
    """`python bench.py --gpus N` with no launcher in the environment (the driver's N = 1 command shape with N > 1) must
    run N ranks: it starts torch.distributed.run as a child before touching the GPU.  On this one-GPU box: refused without
    --share (a one-GPU line must never be labelled N GPUs), and with --share a 4-rank rehearsal (ranks share the device,
    gloo collectives; the box allows 6 GPU processes, so not the 8 the real node will run) whose line says n_gpus = 4, one
    distinct device, per-rank rates, the tape-wait counter, and an exchange that gathered every rank's records."""
    import torch
    if torch.cuda.device_count() >= 4:
        pytest.skip("a multi-GPU node: the real N-rank run is the driver's")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    args = ["--steps", "2", "--warmup", "1", "--board", "9", "--win", "5", "--sims", "24", "--slots", "128", "--steady-games", "0", "--no-cpu"]
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + args, env=env, timeout=600, capture_output=True, text=True, cwd=ROOT)
    assert p.returncode != 0 and "--share" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]
    line = _bench([sys.executable, "bench.py", "--gpus", "4", "--share"] + args, env)
    assert line["n_gpus"] == 4 and line["collectives"] == "gloo" and line["distinct_devices"] == 1
    assert [r["rank"] for r in line["ranks"]] == [0, 1, 2, 3]
    assert all(r["node_expansions_per_sec"] > 0 and r["tape_threads"] >= 1 and r["host_cpus"] >= 1 and r["tape_wait_seconds"] >= 0.0
               for r in line["ranks"])
    assert line["tape_wait_seconds"] == max(r["tape_wait_seconds"] for r in line["ranks"])
    assert abs(sum(r["node_expansions_per_sec"] for r in line["ranks"]) / line["value"] - 1.0) < 0.5
    ep = line["episode"]
    assert ep["games"] == 4 * 128 and ep["records_gathered"] == round(ep["mean_plies_per_game"] * ep["games"])
    # the same shard on one rank: per-game seeds make rank 0's block of the 4-rank episode this very episode
    one = _bench([sys.executable, "bench.py", "--gpus", "1"] + args, env)
    assert one["n_gpus"] == 1 and one["episode"]["games"] == 128 and one["ranks"][0]["device"] == line["ranks"][0]["device"]
