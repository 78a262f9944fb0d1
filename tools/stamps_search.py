#!/usr/bin/env python3
"""Diagnostic: where a ply of the persistent search kernel spends its time (build with `make variant NAME=stamps5 EXTRA=-DAZ_STAMPS
ONLY=5`, select with AZ_ENGINE_LIB): python tools/stamps_search.py [board] [win] [sims]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd import _capi
from alphazero_piskvorky_amd.weights import synthetic_state_dict
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S = int(sys.argv[3]) if len(sys.argv) > 3 else 100
B = 512                      # one workgroup per CU (two games each at 5x5)
e = az.Engine(n, k, S, B, engines=1)
e.load_weights(synthetic_state_dict(n), 0)
e.selfplay_begin(B, seed0=1)
e.selfplay_step(3)           # the third ply's stamps: every game still running, clocks warm
nwg = B // 2
buf = np.zeros((nwg * 2, 32), np.uint64)
assert _capi.lib().az_debug_stamps(e.h, buf.ctypes.data_as(C.c_void_p), nwg * 4) == 0
t = buf[:, :24].astype(np.float64).reshape(nwg, 2, 24)
names = ["kernel prologue", "iteration head (any_eval, 2-3 barriers)", "zero-fill + encode", "conv1 (+ plane re-zero)", "conv2", "conv3",
         "head convs", "policy_fc / value_fc1", "tree step (own game)", "wait for the other game's tree step"]
w0 = t[:, 0, :]
tot = w0[:, :10].sum(axis=1)
print(f"{n}x{n}, {S} simulations, {nwg} workgroups; s_memtime ticks per ITERATION of wave 0 (mean over workgroups), share of the ply")
for i, nm in enumerate(names):
    print(f"{nm:42s} {w0[:, i].mean() / (S + 1):9.1f}  {100 * w0[:, i].sum() / tot.sum():5.1f} %")
print(f"{'total':42s} {tot.mean() / (S + 1):9.1f}")
inner = ["softmax", "value_fc2 chain + tanh", "root noise, expand", "backup", "selection"]
lv = w0[:, 15].mean() / (S + 1)
print(f"inside the tree step (wave 0); selection descends {lv:.2f} levels per iteration on average:")
for i, nm in enumerate(inner):
    print(f"  {nm:40s} {w0[:, 10 + i].mean() / (S + 1):9.1f}")
lvn = max(w0[:, 15].mean(), 1)
print(f"  {'selection per level':40s} {w0[:, 14].mean() / lvn:9.1f}  (row read + PUCT {w0[:, 16].mean() / lvn:.1f}, argmax {w0[:, 17].mean() / lvn:.1f}, "
      f"move + terminal tests of the levels that go on {w0[:, 18].mean() / lvn:.1f}; the rest is the last level's exit and the leaf's hand-over)")
e.close()
