#!/usr/bin/env python3
"""Full-size parity soak (not part of the regular suite: minutes of host time): complete self-play games at the BASELINE
configuration, engine vs the CPU oracle, every ply of every game bit for bit (boards, visit counts, pi, actions, z).
usage: soak_parity.py [games] [model plain|resnet] [sims] [out.json|-] [board] [reuse] [ckpt] [bf16x3|f16x2] [slots=N] [lanes=K]
bf16x3 / f16x2 = an opt-in emulated trunk: reports how many free-running games (and plies before the first difference) stay identical
to the exact-order oracle instead of demanding all of them."""
import json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd.weights import synthetic_state_dict, synthetic_resnet_state_dict
from alphazero_piskvorky_amd.net import fold_resnet_state_dict
from oracle import oracle as orc

G = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = sys.argv[2] if len(sys.argv) > 2 else "plain"
S = int(sys.argv[3]) if len(sys.argv) > 3 else 400
out = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] != "-" else None
n = int(sys.argv[5]) if len(sys.argv) > 5 else 15          # optional: board size, "reuse" for subtree reuse, "ckpt" = trained 5x5 weights
reuse = "reuse" in sys.argv[6:]
emul = "bf16x3" if "bf16x3" in sys.argv[6:] else ("f16x2" if "f16x2" in sys.argv[6:] else None)
k, seed0 = (4 if n <= 5 else 5), 1_000_000
if "ckpt" in sys.argv[6:]:
    from tests.util import weights_from_fixture
    sd = weights_from_fixture(5, "ckpt_saved")
else:
    sd = synthetic_resnet_state_dict(n) if model == "resnet" else synthetic_state_dict(n)
slots = max(4, G // 2)                                       # fewer slots than games: refills are part of the soak
lanes = 0
for a in sys.argv[6:]:
    if a.startswith("slots="): slots = int(a[6:])
    if a.startswith("lanes="): lanes = int(a[6:])
eng = az.Engine(n, k, S, slots, engines=lanes, log_table=orc.numpy_log_table(S), model=model)
eng.load_weights(sd, 0)
eng.set_subtree_reuse(reuse)
if emul:
    eng.set_trunk_mode(emul)
t0 = time.perf_counter()
c = eng.selfplay(G, seed0=seed0)
t_gpu = time.perf_counter() - t0
rec = eng.records(); nply, res = eng.games()
starts = np.concatenate([[0], np.cumsum(nply)])
onet = orc.Net(n, resnet_tensors=fold_resnet_state_dict(sd)) if model == "resnet" else orc.Net(n, sd)
o = orc.Oracle(n, k, S, reuse=reuse)

def check(g):
    noise, us = orc.selfplay_tape(seed0 + g, n)
    r = o.selfplay_game(onet, noise, us)
    sl = slice(int(starts[g]), int(starts[g + 1]))
    bad = [key for key in ("actions", "boards", "movers", "visits", "pis", "z", "lasts") if rec[key][sl].shape != r[key].shape or not np.array_equal(rec[key][sl], r[key])]
    L = min(int(nply[g]), r["nply"])
    same = 0                     # plies before the first one whose visit counts differ
    while same < L and np.array_equal(rec["visits"][sl][same], r["visits"][same]):
        same += 1
    return g, r["nply"], int(nply[g]), r["result"], int(res[g]), bad, r["counters"]["expansions"], same

t1 = time.perf_counter()
import threading
_stop = threading.Event()
def _heartbeat():                                  # a long run has to keep writing (gpurun kills a silent one)
    while not _stop.wait(60.0):
        print(f"... oracle replaying, {time.perf_counter() - t1:.0f} s", flush=True)
threading.Thread(target=_heartbeat, daemon=True).start()
rows, t_print = [], time.perf_counter()
with ThreadPoolExecutor(os.cpu_count() or 8) as pool:
    for row in pool.map(check, range(G)):
        rows.append(row)
        if time.perf_counter() - t_print > 30.0:
            print(f"... {len(rows)}/{G} games checked, {time.perf_counter() - t1:.0f} s", flush=True)
            t_print = time.perf_counter()
_stop.set()
t_cpu = time.perf_counter() - t1
fails = [r for r in rows if r[5] or r[1] != r[2] or r[3] != r[4]]
summary = {"config": f"{n}x{n}/{k}, {S} sims, {model} net{' (trained 5x5 checkpoint)' if 'ckpt' in sys.argv[6:] else ''}, {G} complete games on {slots} slots ({eng.lanes()} lanes)"
                     f"{', subtree reuse' if reuse else ''}, seeds {seed0}..", "plies": int(nply.sum()),
           "expansions_engine": int(c["expansions"]), "expansions_oracle": int(sum(r[6] for r in rows)),
           "games_bit_exact": G - len(fails), "games": G, "gpu_seconds": round(t_gpu, 2), "oracle_seconds": round(t_cpu, 1),
           "oracle_threads": os.cpu_count(), "compared": "actions, boards, movers, visit counts, pi (f32 bit patterns), z, last moves, per ply"}
if emul:
    summary["trunk"] = emul + " (opt-in emulated trunk: tolerance, not bit-exact)"
    summary["plies_identical_before_first_difference"] = int(sum(r[7] for r in rows))
    summary["games_with_identical_visit_counts_throughout"] = int(sum(1 for r in rows if r[7] == r[1] == r[2]))
print(json.dumps(summary))
if fails:
    print("MISMATCH", fails[:5])
if out:
    json.dump(summary, open(out, "w"), indent=1)
sys.exit(1 if (not emul) and (fails or summary["expansions_engine"] != summary["expansions_oracle"]) else 0)
