#!/usr/bin/env python3
"""Diagnostic: games/s when an episode has more games than slots (slots are refilled as games end, so the episode tail is
paid once per episode instead of once per 1024 games).  usage: refill_rate.py [games] [slots]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd.weights import synthetic_state_dict
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
eng = az.MultiEngine(15, 5, 400, B, engines=4)
eng.load_weights(synthetic_state_dict(15), 0)
t0 = time.perf_counter()
c = eng.selfplay(G, seed0=1_000_000)
dt = time.perf_counter() - t0
print(f"15x15/5, 400 sims, {G} games on {B} slots: {dt:.1f} s, {G / dt:.1f} games/s, "
      f"{(c['expansions'] + c['root_evals']) / dt / 1e6:.3f} M expansions/s over the whole episode, {c['plies'] / G:.1f} plies/game")
eng.close()
