#!/usr/bin/env python3
"""Diagnostic: the latency path alone, for a rocprofv3 kernel trace (GPU box, from the repo root):
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o lat -- python3 tools/latency_prof.py [search|arena] [board] [sims]
search: 20 single-position searches (MCTS.run seam, 150 simulations); arena: one 51-game arena at 200 simulations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd.weights import synthetic_state_dict
what = sys.argv[1] if len(sys.argv) > 1 else "search"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
k = 5 if n >= 9 else 4
if what == "search":
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    e = az.Engine(n, k, S, 1); e.load_weights(synthetic_state_dict(n), 0)
    board = np.zeros(n * n, np.uint8)
    e.search(board, 1, -1, 1.0, None, 0.5)
    t = time.perf_counter()
    for _ in range(20): e.search(board, 1, -1, 1.0, None, 0.5)
    print(f"search {n}x{n} {S} sims: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms", flush=True)
else:
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    e = az.Engine(n, k, S, 51)
    e.load_weights(synthetic_state_dict(n), 0); e.load_weights(synthetic_state_dict(n, seed=99), 1)
    t = time.perf_counter(); r = e.arena(51, seed0=1); dt = time.perf_counter() - t
    print(f"arena 51 games {n}x{n} {S} sims: {dt:.3f} s, {int(r['nply'].sum())} plies", flush=True)
e.close()
