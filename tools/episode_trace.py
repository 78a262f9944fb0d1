#!/usr/bin/env python3
"""Diagnostic: per-ply wall time vs number of active games over one full episode (15x15, 1024 games, 400 sims)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd.weights import synthetic_state_dict
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
eng = az.MultiEngine(15, 5, 400, 1024, engines=K)
eng.load_weights(synthetic_state_dict(15), 0)
eng.selfplay_begin(1024, seed0=1_000_000)
rows = []; active = 1024; t0 = time.perf_counter()
while active > 0:
    t = time.perf_counter(); a0 = active
    active, _ = eng.selfplay_step(1)
    rows.append((a0, time.perf_counter() - t))
tot = time.perf_counter() - t0
rows = np.array(rows)
print("plies", len(rows), "total", round(tot, 2), "s")
full = np.median(rows[rows[:, 0] == rows[0, 0], 1])          # ply time with every slot busy
print(f"full-occupancy ply {1e3 * full:.1f} ms; a ply with a active games costs a / {int(rows[0, 0])} of that at the throughput rate")
for lo, hi in ((513, 1024), (129, 512), (33, 128), (9, 32), (1, 8)):
    m = (rows[:, 0] >= lo) & (rows[:, 0] <= hi)
    if m.any():
        ideal = (rows[m, 0] / rows[0, 0] * full).sum()
        print(f"active {lo:4d}-{hi:4d}: plies {m.sum():4d}  time {rows[m, 1].sum():6.2f} s  mean ms/ply {1e3 * rows[m, 1].mean():7.1f}"
              f"  at the throughput rate {ideal:5.2f} s  (x{rows[m, 1].sum() / ideal:.2f})")
if len(sys.argv) > 2:
    for a, t in rows:
        print(int(a), round(1e3 * t, 2))
eng.selfplay_end(); eng.close()
