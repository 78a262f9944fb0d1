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
S = int(os.environ.get("AZ_STAMPS_SIMS", "400"))     # long plies: the >= 2 s of back-to-back launches before the reading end before the games do
e = az.Engine(n, 5 if n > 5 else 4, S, B, engines=1)
e.load_weights(synthetic_state_dict(n), 0)
if len(sys.argv) > 2:
    e.set_trunk_mode(sys.argv[2])
e.selfplay_begin(B, seed0=1)
import time
t_end = time.time() + float(os.environ.get("AZ_STAMPS_WARM_SECONDS", "2.2"))     # the guide asks for >= 2 s of back-to-back launches before the reading
while time.time() < t_end:
    e.selfplay_step(1)
G = {15: 1, 9: 2, 5: 4}[n]
ng = (B + G - 1) // G
buf = np.zeros((ng, 16), np.uint64)
rc = _capi.lib().az_debug_stamps(e.h, buf.ctypes.data_as(C.c_void_p), ng)
assert rc == 0, rc
t = buf[:, :6].astype(np.int64)
rt = buf[:, 14:16].astype(np.int64)
# workgroups whose game has ended keep the stamps of an older launch: keep the rows that are one launch's (monotonic, < 1 ms)
ok = (np.diff(t, axis=1) > 0).all(axis=1) & (rt[:, 1] > rt[:, 0]) & (rt[:, 1] - rt[:, 0] < 100000)
print("workgroups with a complete set of stamps from one launch:", int(ok.sum()), "of", ng)
t, rt = t[ok], rt[ok]
ng = int(ok.sum())
d = np.diff(t, axis=1)
names = ["prologue(zero+tables+encode)", "conv1", "conv2", "conv3(+out3 write)", "heads+feat"]
print("workgroups", ng, "stamp units = s_memtime ticks (100 MHz constant clock on gfx9? check total)")
for i, nm in enumerate(names):
    print(f"{nm:32s} mean {d[:, i].mean():10.1f}  min {d[:, i].min():8d}  max {d[:, i].max():8d}")
if ng:
    ghz = (t[:, 5] - t[:, 0]) / (rt[:, 1] - rt[:, 0]) * 0.1
    print(f"in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz): median {np.median(ghz):.3f} GHz, min {ghz.min():.3f}, max {ghz.max():.3f}; "
          f"workgroup duration median {np.median(rt[:, 1] - rt[:, 0]) * 10:.0f} ns")
# conv3 per wave: entry, all MFMAs issued, past the barrier, epilogue written (slots behind k_fc's stamps)
allb = np.zeros(B * 16 + 4096 * 32 + B * 32, np.uint64)
if _capi.lib().az_debug_stamps(e.h, allb.ctypes.data_as(C.c_void_p), -1) == 0:
    c3 = allb[B * 16 + 4096 * 32:].reshape(B, 8, 4).astype(np.int64)[ok]
    if len(c3) and (np.diff(c3, axis=2) >= 0).all():
        dd = np.diff(c3, axis=2)
        print("conv3 per wave: main loop (entry -> last MFMA issued) mean %.0f  [min %d max %d];  wait at the barrier mean %.0f [max %d];  epilogue mean %.0f" %
              (dd[:, :, 0].mean(), dd[:, :, 0].min(), dd[:, :, 0].max(), dd[:, :, 1].mean(), dd[:, :, 1].max(), dd[:, :, 2].mean()))
        print("conv3: first wave in -> last wave out of the epilogue, mean %.0f" % (c3[:, :, 3].max(axis=1) - c3[:, :, 0].min(axis=1)).mean())
        print("conv3 main loop per wave (mean over workgroups):", np.round(dd[:, :, 0].mean(axis=0)).astype(int).tolist())
tot = t[:, 5] - t[:, 0]
print("total per WG mean", tot.mean(), " kernel span", t[:, 5].max() - t[:, 0].min())
starts = np.sort(t[:, 0] - t[:, 0].min())
print("start-time quartiles", starts[[0, ng // 4, ng // 2, 3 * ng // 4, ng - 1]])
e.close()
