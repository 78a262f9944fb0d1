#!/usr/bin/env python3
"""Diagnostic: phase breakdown of k_trunk / k_trunk_emul from s_memtime stamps (build with `make EXTRA=-DAZ_STAMPS`):
python tools/stamps.py [board] [f32|bf16x3|f16x2]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd import _capi
from alphazero_piskvorky_amd.weights import synthetic_state_dict
n, B = int(sys.argv[1]) if len(sys.argv) > 1 else 15, 256      # one lane, one round of workgroups at n = 15
e = az.Engine(n, 5 if n > 5 else 4, 8, B, engines=1)
e.load_weights(synthetic_state_dict(n), 0)
if len(sys.argv) > 2:
    e.set_trunk_mode(sys.argv[2])
e.selfplay_begin(B, seed0=1)
e.selfplay_step(1)
G = {15: 1, 9: 2, 5: 4}[n]
ng = (B + G - 1) // G
buf = np.zeros((ng, 16), np.uint64)
rc = _capi.lib().az_debug_stamps(e.h, buf.ctypes.data_as(C.c_void_p), ng)
assert rc == 0, rc
t = buf[:, :6].astype(np.int64)
d = np.diff(t, axis=1)
names = ["prologue(zero+tables+encode)", "conv1", "conv2", "conv3(+out3 write)", "heads+feat"]
print("workgroups", ng, "stamp units = s_memtime ticks (100 MHz constant clock on gfx9? check total)")
for i, nm in enumerate(names):
    print(f"{nm:32s} mean {d[:, i].mean():10.1f}  min {d[:, i].min():8d}  max {d[:, i].max():8d}")
tot = t[:, 5] - t[:, 0]
print("total per WG mean", tot.mean(), " kernel span", t[:, 5].max() - t[:, 0].min())
starts = np.sort(t[:, 0] - t[:, 0].min())
print("start-time quartiles", starts[[0, ng // 4, ng // 2, 3 * ng // 4, ng - 1]])
e.close()
