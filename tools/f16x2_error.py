#!/usr/bin/env python3
"""Feasibility check for the float16 emulation scheme (AZ_TRUNK_F16X2), CPU only: every conv operand is split into two float16
parts (hi + lo / 2048), three cross products are accumulated in float32 (hi*hi in one accumulator, the two mixed products in a
second one folded in as acc + accx / 2048), and the net outputs are compared with the exact-order float32 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from alphazero_piskvorky_amd.weights import synthetic_state_dict

def split2(x):
    h = x.astype(np.float16)
    r = (x - h.astype(np.float32)) * np.float32(2048.0)
    l = r.astype(np.float16)
    return h.astype(np.float32), l.astype(np.float32)

def conv3x3_split(x, w, b):
    cin, n, _ = x.shape
    xp = np.zeros((cin, n + 2, n + 2), np.float32); xp[:, 1:-1, 1:-1] = x
    cols = np.stack([xp[:, ky:ky + n, kx:kx + n] for ky in range(3) for kx in range(3)], 0)
    A = cols.reshape(9 * cin, n * n)
    W = w.transpose(0, 2, 3, 1).reshape(w.shape[0], 9 * cin)
    ah, al = split2(A); wh, wl = split2(W)
    acc_hh = (wh @ ah).astype(np.float32)
    acc_x = (wh @ al + wl @ ah).astype(np.float32)
    acc = acc_hh + acc_x * np.float32(1.0 / 2048.0)
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

for n, tag in ((15, "seeded"), (5, "ckpt")):
    if tag == "ckpt":
        from tests.util import weights_from_fixture
        sd = weights_from_fixture(5, "ckpt_saved")
    else:
        sd = synthetic_state_dict(n)
    net = orc.Net(n, sd); o = orc.Oracle(n, 5 if n > 5 else 4, 1)
    rs = np.random.RandomState(0)
    dl = dp = dv = 0.0; amax = 0
    for t in range(30):
        stones = rs.randint(0, n * n // 3)
        board = np.zeros(n * n, np.uint8)
        cells = rs.permutation(n * n)[:stones]
        board[cells[0::2]] = 1; board[cells[1::2]] = 2
        planes = o.encode(board, 1 + (stones & 1), int(cells[-1]) if stones else -1)
        lo, Po, vo = net.eval(planes)
        ls, vs = forward_split(sd, planes)
        e = np.exp(ls - ls.max()); Ps = e / e.sum()
        dl = max(dl, float(np.abs(ls - lo).max())); dp = max(dp, float(np.abs(Ps - Po).max())); dv = max(dv, abs(vs - vo))
    print(n, tag, f"fp16 2-way split, 3 products: max |dlogit| {dl:.2e}, |dP| {dp:.2e}, |dvalue| {dv:.2e}")
