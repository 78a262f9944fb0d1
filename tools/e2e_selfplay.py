#!/usr/bin/env python3
"""Diagnostic: wall time of the drop-in seam SelfPlayManager.generate_self_play (tapes, episode, gather, tuples)
next to the engine's own episode time.  usage: e2e_selfplay.py [games] [board] [sims]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd.net import GomokuNet
from alphazero_piskvorky_amd.controller import NeuralNetworkController
from alphazero_piskvorky_amd.self_play import SelfPlayManager
from alphazero_piskvorky_amd.weights import synthetic_state_dict

games = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
sims = int(sys.argv[3]) if len(sys.argv) > 3 else 400
from alphazero_piskvorky_amd import constants as C
C.BOARD_SIZE, C.WIN_LENGTH = n, (4 if n <= 5 else 5)      # the reference's def-time constants (constants.py:2-3)
net = GomokuNet(board_size=n, device="cuda")
net.load_state_dict({k: torch.as_tensor(v) for k, v in synthetic_state_dict(n).items()})
ctl = NeuralNetworkController(net, device="cuda")
spm = SelfPlayManager(ctl, "cuda", mcts_params={"num_simulations": sims}, seed=1_000_000)
for rep in range(2):
    t0 = time.perf_counter()
    packed, total, eng, dev, _ = spm.generate_packed(games)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    c = spm.last_counters
    print(f"rep {rep}: generate_packed {t1 - t0:.2f} s (engine episode {c['seconds']:.2f} s, {total} records) "
          f"-> {games / (t1 - t0):.1f} games/s wall, {games / c['seconds']:.1f} games/s engine")
t0 = time.perf_counter()
ex = spm.generate_self_play(games)
t1 = time.perf_counter()
print(f"generate_self_play {t1 - t0:.2f} s, {len(ex)} tuples -> {games / (t1 - t0):.1f} games/s wall")
