#!/usr/bin/env python3
"""Diagnostic: what opt-in subtree reuse saves.  5x5 with the trained checkpoint fixture (peaked priors) and 15x15 with
random-init weights (flat priors), full episodes with and without az_set_subtree_reuse."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd.weights import synthetic_state_dict
from tests.util import weights_from_fixture

for name, n, k, S, G, K, sd in (("5x5 trained checkpoint", 5, 4, 100, 1024, 1, weights_from_fixture(5, "ckpt_saved")),
                                ("15x15 random-init", 15, 5, 400, 1024, 4, synthetic_state_dict(15))):
    for reuse in (False, True):
        eng = az.MultiEngine(n, k, S, G, engines=K)
        eng.load_weights(sd, 0)
        eng.set_subtree_reuse(reuse)
        t0 = time.perf_counter()
        c = eng.selfplay(G, seed0=1_000_000)
        dt = time.perf_counter() - t0
        print(f"{name}, reuse={reuse}: {dt:.2f} s, {G / dt:.1f} games/s, plies {c['plies']}, simulations {c['simulations']} "
              f"({c['simulations'] / max(c['plies'], 1):.1f} per move), net evaluations {c['trunk_boards']}", flush=True)
        eng.close()
