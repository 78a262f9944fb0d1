#!/usr/bin/env python3
"""Diagnostic: per-wave phase breakdown of k_fc (build with -DAZ_STAMPS)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd import _capi
from alphazero_piskvorky_amd.weights import synthetic_state_dict
n, B = 15, 256
e = az.Engine(n, 5, 8, B, engines=1)
e.load_weights(synthetic_state_dict(n), 0)
e.selfplay_begin(B, seed0=1)
e.selfplay_step(1)
tot = B * 16 + 4096 * 32
buf = np.zeros(tot, np.uint64)
rc = _capi.lib().az_debug_stamps(e.h, buf.ctypes.data_as(C.c_void_p), -1)
assert rc == 0
fc = buf[B * 16:].reshape(-1, 8, 4).astype(np.int64)      # [wg][wave][stamp]
nwg = (B // 16) * 3                                     # the throughput shape <2, 4>: three workgroups per row of 16 boards
fc = fc[:nwg]
live = fc[:, :, 2] > 0
st = (fc[:, :, 1] - fc[:, :, 0])
ch = (fc[:, :, 2] - fc[:, :, 1])
print("staging cycles: mean %.0f max %d" % (st.mean(), st.max()))
print("chain   cycles (waves with a tile): mean %.0f max %d  n=%d" % (ch[live].mean(), ch[live].max(), live.sum()))
first = fc[:, :, 0][fc[:, :, 0] > 0]
print("WG start spread (cycles, within XCD counters not comparable): ", np.percentile(first - first.min(), [0, 50, 100]))
e.close()
