#!/usr/bin/env python3
"""Diagnostic: latency of single-position search (MCTS.run seam) and of a 51-game arena, fused vs split trunk."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd.weights import synthetic_state_dict
n, k = 15, 5
for split in ("0", "64"):
    os.environ["AZ_SPLIT_MAX"] = split
    e = az.Engine(n, k, 150, 1); e.load_weights(synthetic_state_dict(n), 0)
    board = np.zeros(n * n, np.uint8)
    e.search(board, 1, -1, 1.0, None, 0.5)
    t = time.perf_counter()
    for _ in range(5): e.search(board, 1, -1, 1.0, None, 0.5)
    print(f"AZ_SPLIT_MAX={split}: single search, 150 sims: {(time.perf_counter() - t) / 5 * 1e3:.1f} ms")
    e.close()
    e = az.Engine(n, k, 200, 51); sd = synthetic_state_dict(n)
    e.load_weights(sd, 0); e.load_weights(synthetic_state_dict(n, seed=99), 1)
    t = time.perf_counter(); r = e.arena(51, seed0=1); dt = time.perf_counter() - t
    print(f"AZ_SPLIT_MAX={split}: arena 51 games x 200 sims: {dt:.2f} s ({int(r['nply'].sum())} plies, W/L/D {r['wins']}/{r['losses']}/{r['draws']})")
    e.close()
