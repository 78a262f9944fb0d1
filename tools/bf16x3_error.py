#!/usr/bin/env python3
"""Feasibility check for an fp32-emulating conv trunk, CPU only: every conv operand is split
into three bf16 parts (hi + mid + lo), the six largest cross products are accumulated in float32, and the net outputs are
compared with the exact-order float32 oracle on random positions.  Prints the largest deviations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from alphazero_piskvorky_amd.weights import synthetic_state_dict


def bf16(x):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def split3(x):
    h = bf16(x); m = bf16(x - h); l = bf16(x - h - m)
    return h, m, l


def conv3x3_split(x, w, b):
    """x [cin, n, n], w [cout, cin, 3, 3] -> relu(conv + b) with 6 bf16 cross products accumulated in float32"""
    cin, n, _ = x.shape
    xp = np.zeros((cin, n + 2, n + 2), np.float32); xp[:, 1:-1, 1:-1] = x
    cols = np.stack([xp[:, ky:ky + n, kx:kx + n] for ky in range(3) for kx in range(3)], 0)      # [9, cin, n, n]
    A = cols.reshape(9 * cin, n * n)
    W = w.transpose(0, 2, 3, 1).reshape(w.shape[0], 9 * cin)                                      # k = (ky*3+kx)*cin + ci
    a, wgt = split3(A), split3(W)
    acc = np.zeros((w.shape[0], n * n), np.float32)
    for i, j in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):                                  # small terms first
        acc += wgt[i] @ a[j]
    return np.maximum(acc + b[:, None], 0).reshape(w.shape[0], n, n)


def forward_split(sd, planes):
    h = planes
    for i in (1, 2, 3):
        h = conv3x3_split(h, sd[f"conv{i}.weight"], sd[f"conv{i}.bias"])
    n = h.shape[1]
    flat = h.reshape(128, n * n)
    p = np.maximum(sd["policy_conv.weight"].reshape(4, 128) @ flat + sd["policy_conv.bias"][:, None], 0).reshape(-1)
    v = np.maximum(sd["value_conv.weight"].reshape(2, 128) @ flat + sd["value_conv.bias"][:, None], 0).reshape(-1)
    logits = sd["policy_fc.weight"] @ p + sd["policy_fc.bias"]
    hid = np.maximum(sd["value_fc1.weight"] @ v + sd["value_fc1.bias"], 0)
    return logits, float(np.tanh(sd["value_fc2.weight"] @ hid + sd["value_fc2.bias"])[0])


n = 15
sd = synthetic_state_dict(n)
net = orc.Net(n, sd)
o = orc.Oracle(n, 5, 1)
rs = np.random.RandomState(0)
dl = dp = dv = 0.0
for t in range(40):
    stones = rs.randint(0, 80)
    board = np.zeros(n * n, np.uint8)
    cells = rs.permutation(n * n)[:stones]
    board[cells[0::2]] = 1; board[cells[1::2]] = 2
    planes = o.encode(board, 1 + (stones & 1), int(cells[-1]) if stones else -1)
    lo, Po, vo = net.eval(planes)
    ls, vs = forward_split(sd, planes)
    e = np.exp(ls - ls.max()); Ps = e / e.sum()
    dl = max(dl, float(np.abs(ls - lo).max())); dp = max(dp, float(np.abs(Ps - Po).max())); dv = max(dv, abs(vs - vo))
print(f"15x15, 40 random positions, 3-way bf16 split with 6 cross products in the three 3x3 convs vs the fp32 oracle: "
      f"max |dlogit| {dl:.2e}, max |dP| {dp:.2e}, max |dvalue| {dv:.2e}")
