#!/usr/bin/env python3
"""Diagnostic: latency of single-position search (MCTS.run seam) and of a 51-game arena, fused vs split trunk, for
GomokuNet and the ResidualBlock net (GPU box, from the repo root):  python tools/latency.py [json out]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd.weights import synthetic_state_dict, synthetic_resnet_state_dict
n, k = 15, 5
rows = []
for model in ("plain", "resnet"):
    gen = synthetic_resnet_state_dict if model == "resnet" else synthetic_state_dict
    sa, sb = gen(n), (gen(n, 2) if model == "resnet" else gen(n, seed=99))
    for split in ("0", "64"):
        os.environ["AZ_SPLIT_MAX"] = split
        e = az.Engine(n, k, 150, 1, model=model); e.load_weights(sa, 0)
        board = np.zeros(n * n, np.uint8)
        e.search(board, 1, -1, 1.0, None, 0.5)
        t = time.perf_counter()
        for _ in range(5): e.search(board, 1, -1, 1.0, None, 0.5)
        ts = (time.perf_counter() - t) / 5
        e.close()
        e = az.Engine(n, k, 200, 51, model=model)
        e.load_weights(sa, 0); e.load_weights(sb, 1)
        t = time.perf_counter(); r = e.arena(51, seed0=1); dt = time.perf_counter() - t
        e.close()
        rows.append({"model": model, "trunk": "fused only" if split == "0" else "split when <= 64 boards pend", "search_150_sims_ms": ts * 1e3,
                     "arena_51_games_200_sims_s": dt, "arena_plies": int(r["nply"].sum()), "arena_tally": [r["wins"], r["losses"], r["draws"]]})
        print(rows[-1], flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
