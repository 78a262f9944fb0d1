#!/usr/bin/env python3
"""Diagnostic: full-episode wall time vs how many plies the K engines play between host-side joins."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd.weights import synthetic_state_dict
eng = az.MultiEngine(15, 5, 400, 1024, engines=4)
eng.load_weights(synthetic_state_dict(15), 0)
for chunk in (1 << 20, 16, 4, 1):
    t0 = time.perf_counter()
    eng.selfplay_begin(1024, seed0=1_000_000)
    t1 = time.perf_counter()
    active = 1
    while active > 0:
        active, _ = eng.selfplay_step(chunk)
    c = eng.selfplay_end()
    t2 = time.perf_counter()
    print(f"chunk {chunk}: begin {t1 - t0:.2f} s, plies {t2 - t1:.2f} s, engine seconds {c['seconds']:.2f}, records {c['records']}", flush=True)
eng.close()
