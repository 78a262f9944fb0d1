#!/usr/bin/env python3
"""Measures the two opt-in search upgrades (GPU box, from the repo root):  python tools/upgrades_gain.py > profiles/rNN_upgrades.json
  * evaluation cache: hit rate and games/s with the trained 5x5 checkpoint (weights from tests/golden/net_5.npz) and at 15x15
    with random-init weights, cache off / on; records must be identical.
  * virtual-loss batching: latency of one 150-simulation search and of the 51-game arena at 200 simulations (evaluator.py
    defaults) for batches of 1 / 4 / 8 / 16 leaves."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd.weights import synthetic_state_dict
from tests.util import weights_from_fixture

out = {"eval_cache": [], "virtual_loss": []}
for n, k, S, slots, games, sd, tag in ((5, 4, 100, 1024, 4096, weights_from_fixture(5, "ckpt_saved"), "trained 5x5 checkpoint (models/saved/5x5_4_in_a_row.pt)"),
                                       (15, 5, 400, 1024, 1024, synthetic_state_dict(15), "random-init 15x15")):
    ref = None
    for entries in (0, 1 << 21):
        e = az.Engine(n, k, S, slots)
        e.load_weights(sd, 0)
        e.set_eval_cache(entries)
        c = e.selfplay(games, seed0=1)
        rec = e.records()
        if ref is None:
            ref = rec
        same = all(np.array_equal(ref[key], rec[key]) for key in ref)
        out["eval_cache"].append({"board": n, "sims": S, "slots": slots, "games": games, "weights": tag, "cache_entries": entries,
                                  "games_per_sec": games / c["seconds"], "seconds": c["seconds"],
                                  "hit_rate": c["cache_hits"] / c["cache_lookups"] if c["cache_lookups"] else None,
                                  "net_evaluations": c["trunk_boards"], "evaluations_needed": c["expansions"] + c["root_evals"],
                                  "records_identical_to_cache_off": bool(same)})
        e.close()

n, k = 15, 5
sd = synthetic_state_dict(n)
sd2 = synthetic_state_dict(n, seed=99)
board = np.zeros(n * n, np.uint8)
for L in (1, 4, 8, 16):
    e = az.Engine(n, k, 150, 1)
    e.load_weights(sd, 0)
    e.set_virtual_loss(L)
    e.search(board, 1, -1, 1.0, None, 0.5)
    t0 = time.perf_counter()
    for _ in range(5):
        e.search(board, 1, -1, 1.0, None, 0.5)
    t_search = (time.perf_counter() - t0) / 5
    e.close()
    e = az.Engine(n, k, 200, 51)
    e.load_weights(sd, 0); e.load_weights(sd2, 1)
    e.set_virtual_loss(L)
    t0 = time.perf_counter()
    r = e.arena(51, seed0=3)
    t_arena = time.perf_counter() - t0
    e.close()
    out["virtual_loss"].append({"leaves_per_batch": L, "search_150_sims_ms": t_search * 1e3, "arena_51_games_200_sims_s": t_arena,
                                "arena_tally": [r["wins"], r["losses"], r["draws"]]})
print(json.dumps(out, indent=1))
