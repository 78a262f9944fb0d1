#!/usr/bin/env python3
"""Benchmark of the self-play hot path (BASELINE.json metric: MCTS node-expansions/s/GPU + self-play games/s,
15x15 / 5-in-a-row, 400 simulations per move).

  python bench.py --gpus N --steps K --warmup W
  N > 1 without a launcher's RANK in the environment: this process starts `python -m torch.distributed.run --nnodes=1
  --nproc-per-node N ... bench.py <same arguments>` as a child BEFORE touching the GPU, relays the child's JSON line and exit
  code (one rank per GPU, RCCL).  Launched by torch.distributed.run itself (the driver's N > 1 line) it is one of the ranks.
  Fewer GPUs than ranks is an error unless --share is given (rehearsal: ranks share devices, gloo collectives).

A "step" is one ply of every concurrent game on the GPU: MCTS.run (mcts.py:101-183) for each of the
`slots` games = 1 root evaluation + `sims` simulations, i.e. (sims + 1) lock-step batches of
{tree kernel, conv-trunk kernel, FC kernel}.  Inputs (weights, RNG tapes) are resident in HBM before
the timed region.  After the K timed steps the same games are played as ONE whole episode in one az_selfplay call
(timed separately) so that self-play games/s is a measured number, then a steady-state episode (games >> slots), and the
episode-end record exchange is timed.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def net_flops(n, model="plain"):
    """Algorithmic FLOPs (2*MAC) of one evaluation: trunk (convs + head convs) and FC tail.
    plain = SURVEY.md §8a a18; resnet = stem + 6 x (64->64) convs + 3 head channels (config 5)."""
    nn = n * n
    if model == "resnet":
        return 2 * nn * (36 * 64 + 6 * 576 * 64 + 64 * 3), 2 * (2 * nn * nn + nn * 64 + 64)
    trunk = 2 * nn * (36 * 32 + 288 * 64 + 576 * 128 + 128 * 6)
    fc = 2 * (4 * nn * nn + 2 * nn * 64 + 64)
    return trunk, fc


def usable_cpus():
    """CPUs this job may use: the affinity mask capped by the cgroup CPU quota (the GPU box grants a share of its hardware
    threads; the library sizes its own host threads the same way, az_counters.host_cpus)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and float(period) > 0:
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = max(1, min(n, q // p))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(n, k, sims, sd, budget_games):
    """The CPU oracle (a C restatement of the reference algorithm, see oracle/az_oracle.c) on the host cores:
    one game per thread like self_play.py:29-45, first 2 plies of `budget_games` games."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    cores = min(usable_cpus(), 64)
    o = orc.Oracle(n, k, sims)
    net = orc.Net(n, sd)
    plies = 2
    games = max(4 * cores, budget_games)     # about 10-30 s of CPU work on the box
    tapes = [orc.selfplay_tape(10_000 + g, n, maxply=plies) for g in range(games)]

    def one(g):
        return o.selfplay_game(net, tapes[g][0], tapes[g][1], maxply=plies)["counters"]

    # one game on one thread first ("1 thread and all-cores"), then one game per thread
    t1 = time.perf_counter()
    c1 = one(0)
    dt1 = time.perf_counter() - t1
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        cs = list(ex.map(one, range(games)))
    dt = time.perf_counter() - t0
    exp = sum(c["expansions"] + c["root_evals"] for c in cs)
    cpu_model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": exp / dt, "unit": "node-expansions/s", "cores": cores, "kind": "port",
            "single_thread_value": (c1["expansions"] + c1["root_evals"]) / dt1, "cpu_model": cpu_model,
            "host_threads_total": os.cpu_count(),
            "sample": f"CPU oracle (C restatement of mcts.py/net.py), first {plies} plies of {games} games, "
                      f"{n}x{n}/{k}, {sims} sims, one game per thread, {dt:.1f}s",
            "seconds": dt}


def launch_ranks(a, argv):
    """--gpus N > 1 from a plain `python bench.py`: one child `torch.distributed.run` with N ranks.  This parent never
    initialises the GPU (device_count() does not), relays the single JSON line and exits with the child's code."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if ndev < a.gpus and not a.share:
        print(f"bench.py: --gpus {a.gpus} needs {a.gpus} GPUs, this node shows {ndev} "
              "(--share lets ranks share devices with gloo collectives: a rehearsal, not a measurement)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in p.stdout:
        if out.startswith("{") and line is None:
            line = out
        else:
            sys.stderr.write(out)
    rc = p.wait()
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    elif rc == 0:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        rc = 1
    return rc


def device_name(index):
    """PCI address of a device (which physical GPU a rank played on), without initialising anything beyond torch's props."""
    pr = torch.cuda.get_device_properties(index)
    try:
        return f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    except AttributeError:
        return f"cuda:{index} {getattr(pr, 'uuid', '')}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--share", action="store_true", help="rehearsal on a node with fewer GPUs than ranks: ranks share devices (rank % devices) and "
                    "the collectives run on gloo; without it fewer GPUs than ranks is an error")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--board", type=int, default=15)
    ap.add_argument("--win", type=int, default=5)
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--slots", type=int, default=1024, help="concurrent games per GPU")
    ap.add_argument("--engines", type=int, default=0, help="lanes inside the engine (az_config.engines: the slots are split over this many HIP "
                    "streams, each driven by a host thread of the library, so that one lane's tree/FC kernels overlap another's conv trunk); "
                    "0 = the library's rule: 1 up to 5x5 (launch-bound), else one per 128 slots up to 4")
    ap.add_argument("--steady-games", type=int, default=-1, help="games of the second, steady-state episode (games >> slots, slots refilled "
                    "from the shared queue); -1 = 3 x slots, 0 = skip")
    ap.add_argument("--model", default="plain", choices=["plain", "resnet"], help="plain = GomokuNet (net.py); resnet = ResidualBlock variant (config 5)")
    ap.add_argument("--no-episode", action="store_true", help="skip playing the episode to its end")
    ap.add_argument("--subtree-reuse", action="store_true", help="opt-in search upgrade (not the reference's algorithm): keep the chosen child's subtree between plies")
    ap.add_argument("--eval-cache", type=int, default=0, help="opt-in search upgrade (results unchanged): device evaluation cache with this many entries")
    ap.add_argument("--virtual-loss", type=int, default=1, help="opt-in search upgrade (not the reference's algorithm): leaves per search and evaluation batch")
    ap.add_argument("--trunk", default="f32", choices=["f32", "bf16x3", "f16x2"], help="f32 = the canonical float32 conv trunk (bit-exact against the oracle, the "
                    "headline); bf16x3 / f16x2 = opt-in fp32-emulating trunks on the 16-bit matrix cores (tolerance instead of bit-exactness)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-emul", action="store_true", help="skip the short runs of the two opt-in fp32-emulating trunks that the default f32 line reports beside itself")
    ap.add_argument("--pmc-run", action="store_true", help="counter-collection run: 8 sims per move so the pass stays short")
    a = ap.parse_args()
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")

    # launched by torch.distributed.run (the driver's N > 1 launch line, also valid with one rank): collectives go through
    # the process group -- RCCL -- whatever the world size; plain `python bench.py` has no process group
    dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if a.gpus > 1 and not dist:
        sys.exit(launch_ranks(a, sys.argv[1:]))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if dist and world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher started {world} ranks", file=sys.stderr)
        sys.exit(2)
    ndev = torch.cuda.device_count()
    if dist and ndev < world and not a.share:
        if rank == 0:
            print(f"bench.py: {world} ranks but {ndev} GPUs on this node (--share: ranks share devices, gloo collectives -- a rehearsal)", file=sys.stderr)
        sys.exit(2)
    share = dist and a.share and ndev < world     # rehearsal on a box with fewer GPUs than ranks: ranks share devices, gloo collectives
    local = local % max(ndev, 1)
    if dist:
        import torch.distributed as td
        torch.cuda.set_device(local)
        if share:
            td.init_process_group("gloo")
        else:
            td.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if share else dev      # where collective buffers live

    import alphazero_piskvorky_amd as az
    from alphazero_piskvorky_amd.weights import synthetic_state_dict, synthetic_resnet_state_dict
    from alphazero_piskvorky_amd import parallel

    n, k, S, B = a.board, a.win, (8 if a.pmc_run else a.sims), a.slots
    sd = synthetic_resnet_state_dict(n) if a.model == "resnet" else synthetic_state_dict(n)
    eng = az.Engine(n, k, S, B, engines=a.engines, device=local, model=a.model)
    a.engines = eng.lanes()
    eng.load_weights(sd, 0)
    if a.subtree_reuse:
        eng.set_subtree_reuse(True)
    if a.virtual_loss > 1:
        eng.set_virtual_loss(a.virtual_loss)
    if a.eval_cache > 0:
        eng.set_eval_cache(a.eval_cache)
    if a.trunk != "f32":
        eng.set_trunk_mode(a.trunk)
    # every rank plays its own shard of the episode's games: ids rank*B .. rank*B+B-1 (seed = seed0 + id)
    eng.selfplay_begin(B, seed0=1_000_000 + rank * B)

    def barrier():
        torch.cuda.synchronize()
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    if a.warmup > 0:
        eng.selfplay_step(a.warmup)
    # roofline calibration: one ply with az_set_profiling on -- the lanes then play one after another, so the HIP
    # events around every launch time the kernel alone on the GPU (normally the launches of different lanes overlap)
    barrier()
    _, k0 = eng.selfplay_step(0)
    eng.set_profiling(True)
    _, k1 = eng.selfplay_step(1)
    eng.set_profiling(False)
    cal = {key: k1[key] - k0[key] for key in k1}
    barrier()
    _, c0 = eng.selfplay_step(0)
    t0 = time.perf_counter()
    active, c1 = eng.selfplay_step(a.steps)
    barrier()
    dt = dt_local = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        dt = float(tmax.item())

    d = {key: c1[key] - c0[key] for key in c1}
    exp_local = d["expansions"] + (d["plies"])           # leaf expansions + root expansions (mcts.py:120,136-138)
    sums = torch.tensor([exp_local, d["simulations"], d["plies"], d["depth_sum"], d["terminal_hits"]],
                        dtype=torch.float64, device=cdev)
    if dist:
        td.all_reduce(sums, op=td.ReduceOp.SUM)
    exp_all, sims_all, plies_all, depth_all, term_all = [float(x) for x in sums.tolist()]
    # per rank: expansions/s over the rank's own clock, how long its play streams waited for RNG tapes, and where it ran
    mine = {"rank": rank, "device": device_name(local), "node_expansions_per_sec": exp_local / dt_local,
            "tape_wait_seconds": d["tape_wait_seconds"], "tape_threads": c1["tape_threads"], "host_cpus": c1["host_cpus"],
            "lanes": a.engines}
    ranks = [mine]
    if dist:
        ranks = [None] * world
        td.all_gather_object(ranks, mine)

    # ---- one whole episode: measured games/s, then the episode-end record exchange ----
    # The instrumented episode above (warm-up, calibration ply with serialised lanes and HIP events, K timed plies) is put
    # aside and the SAME games (same seeds) are played again in one uninstrumented az_selfplay call -- what
    # SelfPlayManager.generate_self_play(B) costs.
    episode = None
    eng.selfplay_end()
    if not a.no_episode:
        cend = eng.selfplay(B, seed0=1_000_000 + rank * B)
        torch.cuda.synchronize()
        ep_local = cend["seconds"]
        tg0 = time.perf_counter()
        packed, counts = parallel.gather_packed_records(eng, dev, force=dist)
        torch.cuda.synchronize()
        tg = time.perf_counter() - tg0
        ep = torch.tensor([ep_local, tg, cend["tape_wait_seconds"]], dtype=torch.float64, device=cdev)
        tot = torch.tensor([cend["games"], cend["expansions"] + cend["plies"], cend["plies"]], dtype=torch.float64, device=cdev)
        if dist:
            td.all_reduce(ep, op=td.ReduceOp.MAX)
            td.all_reduce(tot, op=td.ReduceOp.SUM)
        ep_s, tg_s, tw_s = [float(x) for x in ep.tolist()]
        g_all, e_all, p_all = [float(x) for x in tot.tolist()]
        episode = {"what": f"one episode of {B} games per GPU on {B} slots (BASELINE's shape) in one az_selfplay call: no refill, the slots of "
                           "finished games stay empty until the longest game ends",
                   "games": int(g_all), "seconds": ep_s, "games_per_sec": g_all / ep_s,
                   "node_expansions_per_sec": e_all / ep_s, "mean_plies_per_game": p_all / g_all,
                   "record_gather_seconds": tg_s, "records_gathered": int(sum(counts)), "record_exchange": parallel.last_exchange,
                   "tape_wait_seconds": tw_s}
        # ---- steady state: games >> slots, every freed slot takes the next game of the engine's shared queue ----
        sg = a.steady_games if a.steady_games >= 0 else 3 * B
        if sg > 0:
            del packed
            cs = eng.selfplay(sg, seed0=2_000_000 + rank * sg)
            st = torch.tensor([cs["seconds"], cs["tape_wait_seconds"]], dtype=torch.float64, device=cdev)
            tot = torch.tensor([cs["games"], cs["expansions"] + cs["plies"]], dtype=torch.float64, device=cdev)
            if dist:
                td.all_reduce(st, op=td.ReduceOp.MAX)
                td.all_reduce(tot, op=td.ReduceOp.SUM)
            episode["steady_state"] = {"what": f"a second episode of {sg} games per GPU on the same {B} slots, freed slots refilled from the shared queue",
                                       "games": int(tot[0].item()), "seconds": float(st[0].item()),
                                       "games_per_sec": float(tot[0].item() / st[0].item()),
                                       "node_expansions_per_sec": float(tot[1].item() / st[0].item()),
                                       "tape_wait_seconds": float(st[1].item())}

    persist = eng.persistent()
    # ---- beside the headline (never instead of it): the same shard on the two opt-in fp32-emulating trunks, a few plies each ----
    emul = None
    if a.trunk == "f32" and world == 1 and not a.no_emul and not a.pmc_run:
        emul = {}
        for mode in ("bf16x3", "f16x2"):
            try:
                eng.set_trunk_mode(mode)
                eng.selfplay_begin(B, seed0=1_000_000 + rank * B)
                eng.selfplay_step(max(a.warmup, 1))
                torch.cuda.synchronize()
                _, e0 = eng.selfplay_step(0)
                te = time.perf_counter()
                _, e1 = eng.selfplay_step(min(a.steps, 6))
                torch.cuda.synchronize()
                de = time.perf_counter() - te
                eng.selfplay_end()
                ex = (e1["expansions"] - e0["expansions"]) + (e1["plies"] - e0["plies"])
                emul[mode] = {"node_expansions_per_sec": ex / de, "ms_per_step": de * 1e3 / min(a.steps, 6), "steps": min(a.steps, 6),
                              "dtype": f"{mode}: float32 emulated on the 16-bit matrix cores by split operands, float32 accumulate -- opt-in, a tolerance "
                                       "instead of bit-exactness (tests/test_emulated_trunk_gpu.py: within 2e-5 / 1e-6 / 2e-6 of the reference's torch "
                                       "logits / P / value, the reference's visit counts on all 120 recorded plies)"}
            except Exception as ex_:          # e.g. weights outside float16's range
                emul[mode] = {"error": str(ex_)}
        eng.set_trunk_mode("f32")

    if rank == 0:
        trunk_f, fc_f = net_flops(n, a.model)
        boards = cal["expansions"] + cal["plies"]        # boards the trunk kernel evaluated in the calibration ply
        launches = max(cal["trunk_launches"], 1)
        avg_ms = cal["trunk_seconds"] * 1e3 / launches
        achieved = (boards / launches) * trunk_f / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        # f32 MFMA spec peak; the emulated trunks issue 6 (bf16x3) or 3 (f16x2) 16-bit MFMA products per float32 product: dense peak / 6 or / 3
        peak = 157.3 if a.trunk == "f32" else 2500.0 / (6.0 if a.trunk == "bf16x3" else 3.0)
        agg = (d["expansions"] + d["plies"]) * trunk_f / dt / 1e12   # trunk FLOPs of this rank per wall second, timed region
        # HBM bytes per launch from the committed PMC passes (FETCH_SIZE doubled per the gfx950 note, WRITE_SIZE as is)
        traffic, traffic_src = None, None
        try:
            if a.model != "plain":
                raise KeyError("no PMC pass for this kernel yet")
            key = f"k_trunk<{n}>" if a.trunk == "f32" else f"k_trunk_emul<{n}, {1 if a.trunk == 'bf16x3' else 2}>"
            # the newest committed PMC passes of this kernel (round 3 re-collected the float32 trunk; the emulated trunks' kernels
            # are unchanged since round 2)
            pm, src = None, None
            for rnd in ("r03", "r02"):
                f = os.path.join(ROOT, "profiles", f"{rnd}_pmc_{a.trunk}_summary.json")
                if os.path.exists(f) and key in json.load(open(f)):
                    pm, src = json.load(open(f))[key], f"profiles/{rnd}_pmc_{a.trunk}_summary.json"
                    break
            # FETCH_SIZE / WRITE_SIZE in KiB per dispatch; FETCH_SIZE doubled per the gfx950 note (16-B-per-lane streams count half)
            per_board = (2.0 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0 / (pm["grid"] / pm["workgroup"])
            traffic = per_board * boards / launches
            traffic_src = f"{src} (separate --pmc passes, per board x boards per launch)"
        except Exception:
            pass
        # the rest of the path, priced against HBM with SURVEY 8(d)'s algorithmic bytes (reference semantics, fp32 edges):
        # per simulation  d*(12A+16) [P,N,W of A children per descended level + backup RMW] + 12A [expand]
        # + 2*ceil(n^2/8)+2 [leaf board] + 4n^2+4 [logits+value read]; A = legal cells at the calibration ply
        nn_cells = n * n
        ply_index = a.warmup                                  # plies already played when the calibration ply starts
        A = nn_cells - ply_index
        games_cal = max(cal["plies"] / max(a.engines, 1), 1)  # games per launch: every lane plays its own slots (one ply each)
        dbar = cal["depth_sum"] / max(cal["simulations"], 1)
        step_launches = max(cal["steps"], 1)
        bytes_sim = dbar * (12 * A + 16) + 12 * A + (2 * ((nn_cells + 7) // 8) + 2) + (4 * nn_cells + 4)
        step_ms = cal["step_seconds"] * 1e3 / step_launches
        step_gbs = games_cal * bytes_sim / (step_ms * 1e-3) / 1e9 if step_ms > 0 else 0.0
        # k_fc: the three FC layers; bytes = FC weights once per launch + per board (6n^2 head features in, n^2 logits + value out)
        fc_w_bytes = 4 * (4 * nn_cells * nn_cells + nn_cells + 2 * nn_cells * 64 + 64 + 64 + 1) if a.model == "plain" \
            else 4 * (2 * nn_cells * nn_cells + nn_cells + nn_cells * 64 + 64 + 64 + 1)
        feat_in = (6 if a.model == "plain" else 3) * nn_cells * 4
        fc_ms = (cal["nn_seconds"] - cal["trunk_seconds"]) * 1e3 / launches
        fc_bytes = fc_w_bytes + (boards / launches) * (feat_in + 4 * nn_cells + 4)
        fc_gbs = fc_bytes / (fc_ms * 1e-3) / 1e9 if fc_ms > 0 else 0.0
        rest = [] if persist else [
            {"kernel": f"k_step<{n}> (softmax/value tail, backup, PUCT select, expand; one wavefront per game)", "bound": "hbm",
             "achieved": step_gbs, "peak": 8000.0, "unit": "GB/s", "frac": step_gbs / 8000.0, "avg_launch_ms": step_ms,
             "games_per_launch": games_cal, "bytes_per_simulation": bytes_sim, "mean_select_depth": dbar,
             "note": "latency-bound: a serial select->backup chain per game, ~2 MB per 256-game launch; runs underneath another engine's trunk"},
            {"kernel": f"k_fc (policy_fc, value_fc1 on 16-board x 16-output MFMA tiles, four k-chains per output)", "bound": "hbm",
             "achieved": fc_gbs, "peak": 8000.0, "unit": "GB/s", "frac": fc_gbs / 8000.0, "avg_launch_ms": fc_ms,
             "bytes_per_launch": fc_bytes,
             "note": "FC weights are re-read per 16-board row from L2; ~1% of the net's FLOPs; what it costs the pipeline is CU slots, not bandwidth"},
        ]
        out = {
            "metric": "mcts_node_expansions_per_sec", "value": exp_all / dt, "unit": "node-expansions/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3 / max(a.steps, 1),
            "distinct_devices": len({r["device"] for r in ranks}), "ranks": ranks,
            "tape_wait_seconds": max(r["tape_wait_seconds"] for r in ranks),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.trunk == "f32" else ("bf16x3 (float32 emulated by three-way bf16 splits, float32 accumulate; NOT the bit-exact default)" if a.trunk == "bf16x3"
                      else "f16x2 (float32 emulated by two-way float16 splits, float32 accumulate; NOT the bit-exact default)"),
            "data": "synthetic",
            "collectives": (td.get_backend() if dist else None),
            "config": {"workload": f"{n}x{n} / {k}-in-a-row self-play, {B} concurrent games per GPU, {S} sims/move "
                                   f"{'(BASELINE.json configs[3] per-GPU shard)' if (n, k, S, B, a.model) == (15, 5, 400, 1024, 'plain') else '(custom)'}, {'GomokuNet' if a.model == 'plain' else 'ResidualBlock net'} random-init weights, numpy-compatible RNG tapes{', SUBTREE REUSE ON (not the reference algorithm)' if a.subtree_reuse else ''}{f', VIRTUAL-LOSS BATCHES OF {a.virtual_loss} (not the reference algorithm)' if a.virtual_loss > 1 else ''}{f', evaluation cache of {a.eval_cache} entries (results unchanged)' if a.eval_cache else ''}{f', FP32-EMULATING TRUNK ON THE 16-BIT MATRIX CORES, {a.trunk.upper()} (opt-in: tolerance, not bit-exact)' if a.trunk != 'f32' else ''}",
                       "board": n, "win_length": k, "sims_per_move": S, "games_per_gpu": B, "engines_per_gpu": a.engines,
                       "search_kernel": f"persistent ({persist} games per workgroup, trees in LDS)" if persist else "lock-step (k_trunk, k_fc, k_step per evaluation batch)", "parallelism": f"games sharded x{world}" + (" (ranks sharing GPUs, gloo rehearsal)" if share else "")},
            "per_gpu_node_expansions_per_sec": exp_all / dt / world,
            "simulations_per_sec": sims_all / dt, "plies_per_sec": plies_all / dt,
            "mean_select_depth": depth_all / max(sims_all, 1), "terminal_hit_fraction": term_all / max(sims_all, 1),
            "eval_cache_hit_rate": (d["cache_hits"] / d["cache_lookups"]) if d.get("cache_lookups") else None,
            "self_play_games_per_sec": None if episode is None else episode["games_per_sec"],
            "self_play_games_per_sec_steady_state": None if episode is None or "steady_state" not in episode else episode["steady_state"]["games_per_sec"],
            "episode": episode,
            "opt_in_emulated_trunks": emul,
            "roofline": {"kernel": (f"k_search<{n},{persist}> (persistent: one launch per ply = {S + 1} x [encode+conv trunk+heads (MFMA), FC layers, tree step], "
                                    f"{persist} games per workgroup, trees in LDS; priced with the trunk FLOPs only)" if persist
                                    else f"k_trunk_res_emul<{n}, {a.trunk}> (encode+stem+3 residual blocks+head convs, LDS-resident; the 64->64 convs = {6 if a.trunk == 'bf16x3' else 3} x v_mfma_f32_16x16x32_{'bf16' if a.trunk == 'bf16x3' else 'f16'} per tile and 32 k; peak = 16-bit dense peak / {6 if a.trunk == 'bf16x3' else 3})" if a.trunk != "f32" and a.model == "resnet"
                                    else f"k_trunk_emul<{n}, {a.trunk}> (encode+conv1+conv2+conv3+head convs, LDS-resident; conv2/conv3 = {6 if a.trunk == 'bf16x3' else 3} x v_mfma_f32_16x16x32_{'bf16' if a.trunk == 'bf16x3' else 'f16'} per tile and 32 k; peak = 16-bit dense peak / {6 if a.trunk == 'bf16x3' else 3})" if a.trunk != "f32"
                                    else f"k_trunk<{n}> (encode+conv1+conv2+conv3+head convs, LDS-resident, v_mfma_f32_16x16x4_f32)" if a.model == "plain"
                                    else f"k_trunk_res<{n}> (encode+stem+3 residual blocks+head convs, LDS-resident, v_mfma_f32_16x16x4_f32)"),
                         "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": avg_ms, "boards_per_launch": boards / launches,
                         "flops_per_board": trunk_f,
                         "measured": "HIP events around every k_trunk launch of one calibration ply (az_set_profiling: the lanes play "
                                     "one after another, every launch alone on the GPU), between warmup and the timed region",
                         "rest": rest,
                         "aggregate": {"achieved": agg, "frac": agg / peak, "unit": "TFLOP/s",
                                       "what": "trunk FLOPs per wall second over the timed region, all engines/streams overlapping "
                                               "(includes the time the FC and tree kernels take)"}},
        }
        if not a.no_cpu and world == 1 and a.model == "plain":          # reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(n, k, S, sd, 0)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    eng.close()
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
