"""world_size-2 run of the real engine on ONE GPU (ranks share the device, gloo collectives): game sharding, per-game
seeding and the episode-end record exchange give every rank the same examples as a single process."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.util import ROOT

WORKER = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as td
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    td.init_process_group("gloo")
from alphazero_piskvorky_amd import net
from alphazero_piskvorky_amd.controller import NeuralNetworkController
from alphazero_piskvorky_amd.self_play import SelfPlayManager
from alphazero_piskvorky_amd.weights import synthetic_state_dict
m = net.GomokuNet(board_size=5)
m.load_state_dict({k: torch.tensor(v) for k, v in synthetic_state_dict(5).items()})
ctrl = NeuralNetworkController(m, device="cuda:0")
data = SelfPlayManager(ctrl, "cuda:0", mcts_params={"num_simulations": 24}, concurrent_games=8, seed=2024,
                       leaf_symmetry=len(sys.argv) > 3 and sys.argv[3] == "leafsym").generate_self_play(13)
rank = td.get_rank() if world > 1 else 0
np.savez(sys.argv[2] + f".{rank}.npz", s=np.stack([d[0].numpy() for d in data]), p=np.stack([d[1] for d in data]),
         z=np.array([d[2] for d in data]))
if world > 1:
    td.barrier(); td.destroy_process_group()
'''


@pytest.mark.parametrize("option", ["plain", "leafsym"])
def test_selfplay_manager_two_ranks_equals_single_process(tmp_path, option):
    """option = leafsym: random-symmetry leaf evaluation hashes the game's GLOBAL name (its seed), so the second rank's
    games -- local ids 0.. again -- draw the symmetries of the single-process games 7..12, not of games 0..5."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    single = str(tmp_path / "single")
    subprocess.run([sys.executable, str(script), ROOT, single, option], check=True, env=env, timeout=300, stdout=subprocess.DEVNULL)
    multi = str(tmp_path / "multi")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                    "--master-addr", "127.0.0.1", "--master-port", "29733", str(script), ROOT, multi, option],
                   check=True, env=env, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ref = np.load(single + ".0.npz")
    for rank in (0, 1):
        got = np.load(multi + f".{rank}.npz")
        assert got["z"].shape == ref["z"].shape and len(ref["z"]) > 0
        for key in ("s", "p", "z"):
            assert np.array_equal(got[key], ref[key]), f"rank {rank}: {key} differs from the single-process episode"


ARENA_WORKER = r'''
import os, sys, json, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as td
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    td.init_process_group("gloo")
from alphazero_piskvorky_amd import net, constants as C
from alphazero_piskvorky_amd.controller import NeuralNetworkController
from alphazero_piskvorky_amd.evaluator import ModelEvaluator
from alphazero_piskvorky_amd.weights import synthetic_state_dict
C.NUM_EVAL_SIMULATIONS = 24
def ctrl(seed):
    m = net.GomokuNet(board_size=5)
    m.load_state_dict({k: torch.tensor(v) for k, v in synthetic_state_dict(5, seed=seed).items()})
    return NeuralNetworkController(m, device="cuda:0")
wr, metrics = ModelEvaluator(None, False, "cuda:0", seed=77).evaluate(ctrl(1234), ctrl(99), num_games=11)
rank = td.get_rank() if world > 1 else 0
json.dump(dict(metrics, wr=wr), open(sys.argv[2] + f".{rank}.json", "w"))
if world > 1:
    td.barrier(); td.destroy_process_group()
'''


def test_arena_two_ranks_equals_single_process(tmp_path):
    """Arena games split over two ranks in even-aligned blocks + all-reduced tally == the single-process arena."""
    import json
    script = tmp_path / "arena_worker.py"
    script.write_text(ARENA_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    single = str(tmp_path / "single")
    subprocess.run([sys.executable, str(script), ROOT, single], check=True, env=env, timeout=300, stdout=subprocess.DEVNULL)
    multi = str(tmp_path / "multi")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                    "--master-addr", "127.0.0.1", "--master-port", "29735", str(script), ROOT, multi],
                   check=True, env=env, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ref = json.load(open(single + ".0.json"))
    assert ref["total"] == 11
    for rank in (0, 1):
        assert json.load(open(multi + f".{rank}.json")) == ref


TRAIN_WORKER = r'''
import os, sys, json, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as td
td.init_process_group("gloo")
from alphazero_piskvorky_amd import constants as C, train
C.BOARD_SIZE, C.WIN_LENGTH = 5, 4
C.BATCHES_PER_EPISODE, C.NUM_EPOCHS, C.BATCH_SIZE = 2, 1, 64
hist = train.run(episodes=2, games=10, sims=16, eval_games=6, device="cuda:0", seed=3, model_dir=sys.argv[2], log=lambda *a: None)
rank = td.get_rank()
json.dump([{k: h[k] for k in ("wins", "losses", "draws", "total", "promoted", "examples")} for h in hist],
          open(sys.argv[3] + f".{rank}.json", "w"))
td.barrier(); td.destroy_process_group()
'''


def test_train_loop_two_ranks_shared_model_dir(tmp_path):
    """The training loop under two ranks with ONE --model-dir: only rank 0 touches the directory (baseline created once,
    promotions saved atomically), the baseline and the trained weights are broadcast, the records are gathered to rank 0,
    and both ranks report the same arena tallies and promotion decisions."""
    import json
    import torch
    script = tmp_path / "train_worker.py"
    script.write_text(TRAIN_WORKER)
    models = tmp_path / "models"
    out = str(tmp_path / "hist")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29737", str(script), ROOT, str(models), out],
                       env=env, timeout=600, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    h0, h1 = json.load(open(out + ".0.json")), json.load(open(out + ".1.json"))
    assert len(h0) == 2 and all(h["total"] == 6 for h in h0)
    for a, b in zip(h0, h1):
        assert {k: a[k] for k in ("wins", "losses", "draws", "promoted")} == {k: b[k] for k in ("wins", "losses", "draws", "promoted")}
        assert a["examples"] > 0 and b["examples"] == 0           # gather-to-root: rank 1 keeps no examples
    files = sorted(f for f in os.listdir(models))
    assert all(f.endswith(".pt") for f in files), files           # no temporary files left behind
    assert len(files) == 1 + sum(h["promoted"] for h in h0)       # one baseline + one checkpoint per promotion
    for f in files:
        sd = torch.load(os.path.join(models, f), map_location="cpu", weights_only=True)
        assert "conv1.weight" in sd
