#!/usr/bin/env python3
"""Golden-vector generator (BUILD CONTAINER ONLY).

Imports the Python reference from /root/reference/alphazero (read-only) and
records inputs/expected outputs of its hot path as small .npz fixtures in this
directory.  The reference itself never travels: only the data written here is
committed.  Re-run with:  python tests/golden/make_golden.py

One subprocess per board size, because the reference binds BOARD_SIZE /
WIN_LENGTH as def-time defaults (games.py:21-23, net.py:28).

Fixture families:
  G1 rules_{n}x{k}.npz      random play-outs: actions, per-ply terminal flag, result, encode planes
  G2 tree_{n}x{k}.npz       MCTS.run with the synthetic evaluator: visit counts / W / pi / action
  G2b synthgame_{n}x{k}.npz full self-play games (worker loop body) with the synthetic evaluator
  G3 net_{n}.npz            GomokuNet forward (seeded build-owned weights + real 5x5 checkpoint)
  G4 netgame_{n}x{k}.npz    seeded self-play games with the real net, per-ply records
  G5 augment.npz            SelfPlayManager._augment_symmetries on an asymmetric input
  G6 arena_5x4.npz          ModelEvaluator.evaluate between two 5x5 checkpoints, per-game seeds
  G7 zlabels.json           the z truth table of alphazero/tests/tests.py:11-22
  G4-full netgame_full_{n}x{k}.npz  real-net plies at the BASELINE search sizes: 15x15/400 sims (7 plies), 9x9/200 sims (24 plies)
  G4-complete netgame_complete_15x5.npz  ONE whole reference game at 15x15 / 400 sims (python make_golden.py complete)
  G8 resnet_ckpt_5.npz      the VALUES of one historical ResidualBlock checkpoint of the reference (weights-only load of
                            alphazero/models/old/model_20250728_225053.pt: data, no code) + the outputs of the BUILD's torch module
                            loaded with them (the reference ships no forward for this variant)  (python make_golden.py resnet)
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/alphazero"

# --------------------------------------------------------------------------
# Build-owned deterministic helpers shared (by definition, not by import) with
# the oracle and the HIP engine.  Their definitions are restated in
# oracle/az_oracle.c and csrc/; the fixtures pin all three to each other.
# --------------------------------------------------------------------------
import numpy as np

M32 = np.uint32(0xFFFFFFFF)


def fmix32(x):
    """murmur3 finaliser on uint32 numpy values (wrapping arithmetic)."""
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x85EBCA6B)
        x ^= x >> np.uint32(13)
        x *= np.uint32(0xC2B2AE35)
        x ^= x >> np.uint32(16)
    return x


def synth_eval_codes(codes, last_idx, n):
    """Synthetic evaluator on relative cell codes (0 empty, 1 mover, 2 opponent).

    Returns (P f32[n,n] dyadic, v python float dyadic). Pure integer hashing so
    Python, C and HIP give identical bits.
    """
    nn = n * n
    idx = np.arange(nn, dtype=np.uint32)
    with np.errstate(over="ignore"):
        h = fmix32(idx * np.uint32(3) + codes.astype(np.uint32) + np.uint32(0x9E3779B9))
        hs = np.bitwise_xor.reduce(h)
        hs ^= fmix32(np.uint32(0x51ED270B) + np.uint32(last_idx + 1))
        r = fmix32(hs + (idx + np.uint32(1)) * np.uint32(0x9E3779B1))
        p = (((r >> np.uint32(8)) & np.uint32(0xFFFF)) + np.uint32(1)).astype(np.float32) * np.float32(2.0 ** -23)
        vv = int(fmix32(hs ^ np.uint32(0x7F4A7C15)) & np.uint32(0x1FF))
    return p.reshape(n, n), float((vv - 256) / 256.0)


def build_weights(n, seed=1234):
    """Build-owned deterministic GomokuNet weights (state_dict layout of net.py:37-53)."""
    rs = np.random.RandomState(seed)
    shapes = [
        ("conv1.weight", (32, 4, 3, 3)), ("conv1.bias", (32,)),
        ("conv2.weight", (64, 32, 3, 3)), ("conv2.bias", (64,)),
        ("conv3.weight", (128, 64, 3, 3)), ("conv3.bias", (128,)),
        ("policy_conv.weight", (4, 128, 1, 1)), ("policy_conv.bias", (4,)),
        ("policy_fc.weight", (n * n, 4 * n * n)), ("policy_fc.bias", (n * n,)),
        ("value_conv.weight", (2, 128, 1, 1)), ("value_conv.bias", (2,)),
        ("value_fc1.weight", (64, 2 * n * n)), ("value_fc1.bias", (64,)),
        ("value_fc2.weight", (1, 64)), ("value_fc2.bias", (1,)),
    ]
    out = {}
    for name, shp in shapes:
        if name.endswith("weight"):
            fan_in = int(np.prod(shp[1:]))
            w = rs.standard_normal(shp) * (2.0 / fan_in) ** 0.5
        else:
            w = rs.standard_normal(shp) * 0.05
        out[name] = w.astype(np.float32)
    return out


# --------------------------------------------------------------------------
def worker(n, k):
    sys.path.insert(0, REF)
    import constants
    constants.BOARD_SIZE, constants.WIN_LENGTH = n, k
    constants.NUM_EVAL_SIMULATIONS = 40          # arena fixture size (evaluator.py:6)
    import random
    import torch
    torch.set_num_threads(1)
    import games
    import mcts as mcts_mod
    from games import Gomoku
    from mcts import MCTS
    from net import GomokuNet
    from controller import NeuralNetworkController, make_policy_value_fn
    from self_play import default_temperature_schedule, SelfPlayManager
    X, O, DRAW = constants.X, constants.O, constants.DRAW
    nn = n * n

    def codes_of(state):
        me = state.current_player
        c = np.zeros(nn, dtype=np.uint8)
        for r in range(n):
            for q in range(n):
                s = state.board[r][q]
                if s is not None:
                    c[r * n + q] = 1 if s == me else 2
        return c

    def abs_board(state):
        b = np.zeros(nn, dtype=np.uint8)
        for r in range(n):
            for q in range(n):
                s = state.board[r][q]
                if s is not None:
                    b[r * n + q] = 1 if s == X else 2
        return b

    def synth_fn(state):
        la = -1 if state.last_action is None else state.last_action[0] * n + state.last_action[1]
        return synth_eval_codes(codes_of(state), la, n)

    res_code = {None: 0, X: 1, O: 2, DRAW: 3}

    # ---------------- G1 rules ----------------
    rng = random.Random(1000 + n)
    ngames = {5: 200, 9: 60, 15: 30}[n]
    acts = -np.ones((ngames, nn), dtype=np.int16)
    nply = np.zeros(ngames, dtype=np.int16)
    result = np.zeros(ngames, dtype=np.uint8)
    enc_g, enc_p, enc_planes, legal_masks = [], [], [], []
    for g in range(ngames):
        s = Gomoku()
        m = 0
        while not s.is_terminal():
            legal = s.get_legal_actions()
            if rng.random() < 0.15 or m == 0:
                enc_g.append(g); enc_p.append(m)
                enc_planes.append(s.encode("cpu").numpy().astype(np.uint8))
                lm = np.zeros(nn, dtype=np.uint8)
                for (r, c) in legal:
                    lm[r * n + c] = 1
                legal_masks.append(lm)
            a = rng.choice(legal)
            acts[g, m] = a[0] * n + a[1]
            s = s.apply_action(a)
            m += 1
        nply[g] = m
        result[g] = res_code[s.get_game_result()]
    # crafted: overline / anti-diagonal / illegal move
    crafted = []
    def play(seq):
        s = Gomoku()
        flags = []
        for a in seq:
            flags.append(s.is_terminal())
            s = s.apply_action((a // n, a % n))
        return s, flags
    if n >= 9:
        # X builds 0,1,2,_,4,5 on row 0 then fills 3 -> 6-in-row (overline) must win
        xs = [0, 1, 2, 4, 5, 3]; os_ = [n * 3 + 0, n * 3 + 2, n * 5 + 4, n * 7 + 6, n * 8 + 1]
        seq = []
        for i in range(6):
            seq.append(xs[i])
            if i < 5: seq.append(os_[i])
        s, _ = play(seq)
        crafted.append((seq, res_code[s.get_game_result()]))
    # anti-diagonal through the top-right corner
    seq = []
    xs = [(i) * n + (n - 1 - i) for i in range(k)]
    os_ = [(n - 1) * n + i for i in range(k - 1)]
    for i in range(k):
        seq.append(xs[i])
        if i < k - 1: seq.append(os_[i])
    s, _ = play(seq)
    crafted.append((seq, res_code[s.get_game_result()]))
    # row that would only "win" by wrapping across the right edge: must NOT win
    seq = []
    xs = [0 * n + (n - 2), 0 * n + (n - 1), 1 * n + 0, 1 * n + 1] + ([1 * n + 2] if k == 5 else [])
    os_ = [(n - 1) * n + i for i in range(len(xs))]
    for i in range(len(xs)):
        seq.append(xs[i]); seq.append(os_[i]) if i < len(xs) - 1 else None
    s, _ = play(seq)
    crafted.append((seq, res_code[s.get_game_result()]))
    cr_len = max(len(c[0]) for c in crafted)
    cr_acts = -np.ones((len(crafted), cr_len), dtype=np.int16)
    for i, (sq, _) in enumerate(crafted):
        cr_acts[i, :len(sq)] = sq
    np.savez_compressed(
        os.path.join(HERE, f"rules_{n}x{k}.npz"),
        n=n, k=k, actions=acts, nply=nply, result=result,
        enc_game=np.array(enc_g, dtype=np.int32), enc_ply=np.array(enc_p, dtype=np.int32),
        enc_planes=np.array(enc_planes, dtype=np.uint8), legal_masks=np.array(legal_masks, dtype=np.uint8),
        crafted_actions=cr_acts, crafted_result=np.array([c[1] for c in crafted], dtype=np.uint8))

    # ---------------- helpers for tree capture ----------------
    class Capture:
        root = None
    OrigNode = mcts_mod.Node
    class RecNode(OrigNode):
        def __init__(self, state, parent=None, prior=1.0):
            super().__init__(state, parent, prior)
            if parent is None:
                Capture.root = self
    mcts_mod.Node = RecNode

    def tree_stats(root):
        cnt, maxd, stack = 0, 0, [(root, 0)]
        while stack:
            nd, d = stack.pop()
            if nd.children:
                cnt += 1
                maxd = max(maxd, d)
                for ch in nd.children.values():
                    if ch.N > 0:
                        stack.append((ch, d + 1))
        return cnt, maxd

    def run_capture(m, state, T, noise, seed):
        np.random.seed(seed)
        pi, a = m.run(state, temperature=T, add_root_noise=noise)
        root = Capture.root
        N = np.zeros(nn, dtype=np.int32); W = np.zeros(nn, dtype=np.float64); P = np.zeros(nn, dtype=np.float32)
        for (r, c), ch in root.children.items():
            N[r * n + c] = ch.N; W[r * n + c] = ch.W; P[r * n + c] = np.float32(ch.prior)
        nexp, maxd = tree_stats(root)
        return pi.astype(np.float32).reshape(nn), a[0] * n + a[1], N, W, P, nexp, maxd

    # ---------------- G2 tree (synthetic evaluator) ----------------
    S = {5: 100, 9: 200, 15: 400}[n]
    npos = {5: 6, 9: 3, 15: 2}[n]
    rng = random.Random(2000 + n)
    cases = []
    for i in range(npos):
        s = Gomoku()
        depth = 0 if i == 0 else rng.randrange(1, max(2, nn // 2))
        while True:
            s = Gomoku(); ok = True
            for _ in range(depth):
                s = s.apply_action(rng.choice(s.get_legal_actions()))
                if s.is_terminal():
                    ok = False; break
            if ok: break
        for noise in (True, False):
            ply = sum(1 for r in range(n) for c in range(n) if s.board[r][c] is not None)
            T = default_temperature_schedule(ply) if noise else np.float64(0.3 * np.exp(-ply / 4))
            m = MCTS(synth_fn, num_simulations=S, c_puct=2.0)
            seed = 77 + 13 * i + (1 if noise else 0)
            pi, a, N, W, P, nexp, maxd = run_capture(m, s, T, noise, seed)
            la = -1 if s.last_action is None else s.last_action[0] * n + s.last_action[1]
            cases.append(dict(board=abs_board(s), player=1 if s.current_player == X else 2, last=la,
                              noise=int(noise), seed=seed, T=float(T), pi=pi, action=a, N=N, W=W, P=P,
                              nexp=nexp, maxd=maxd))
    np.savez_compressed(
        os.path.join(HERE, f"tree_{n}x{k}.npz"), n=n, k=k, S=S, c_puct=2.0, alpha=0.3, w=0.25,
        **{key: np.array([c[key] for c in cases]) for key in cases[0]})

    # ---------------- G2b synthetic-evaluator self-play games ----------------
    Sg = {5: 100, 9: 100, 15: 100}[n]
    gcount = {5: 3, 9: 1, 15: 1}[n]
    maxply = {5: nn, 9: nn, 15: 6}[n]
    recs = []
    for g in range(gcount):
        seed = 500 + g
        np.random.seed(seed)
        m = MCTS(synth_fn, num_simulations=Sg, c_puct=2.0)
        s = Gomoku(); mv = 0
        while not s.is_terminal() and mv < maxply:
            T = default_temperature_schedule(mv)
            pi, a = m.run(s, temperature=T, add_root_noise=True)
            root = Capture.root
            N = np.zeros(nn, dtype=np.int32)
            for (r, c), ch in root.children.items():
                N[r * n + c] = ch.N
            recs.append(dict(game=g, ply=mv, board=abs_board(s), player=1 if s.current_player == X else 2,
                             pi=pi.astype(np.float32).reshape(nn), N=N, action=a[0] * n + a[1]))
            s = s.apply_action(a); mv += 1
        recs[-1]["final"] = res_code[s.get_game_result()]
    for r_ in recs:
        r_.setdefault("final", 255)
    np.savez_compressed(
        os.path.join(HERE, f"synthgame_{n}x{k}.npz"), n=n, k=k, S=Sg, seed0=500, maxply=maxply,
        **{key: np.array([c[key] for c in recs]) for key in recs[0]})

    # ---------------- G3 net forward ----------------
    def net_from(sd):
        net = GomokuNet(device="cpu")
        net.load_state_dict({kk: torch.tensor(v) for kk, v in sd.items()})
        net.eval()
        return net
    rng = random.Random(3000 + n)
    states = []
    for i in range(12):
        s = Gomoku()
        for _ in range(rng.randrange(0, nn - 1)):
            s2 = s.apply_action(rng.choice(s.get_legal_actions()))
            if s2.is_terminal(): break
            s = s2
        states.append(s)
    wsets = {"seeded": build_weights(n)}
    if n == 5:
        for tag, f in (("ckpt_saved", "models/saved/5x5_4_in_a_row.pt"), ("ckpt_0802", "models/model_20250802_083055.pt")):
            sd = torch.load(os.path.join(REF, f), map_location="cpu", weights_only=True)
            wsets[tag] = {kk: v.numpy().astype(np.float32) for kk, v in sd.items()}
    out = dict(n=n, boards=np.array([abs_board(s) for s in states]),
               players=np.array([1 if s.current_player == X else 2 for s in states], dtype=np.uint8),
               lasts=np.array([-1 if s.last_action is None else s.last_action[0] * n + s.last_action[1] for s in states], dtype=np.int16))
    for tag, sd in wsets.items():
        net = net_from(sd)
        pvf = make_policy_value_fn(NeuralNetworkController(net, device="cpu"))
        with torch.no_grad():
            x = torch.stack([s.encode("cpu") for s in states])
            logits, val = net(x)
        P = np.array([pvf(s)[0].reshape(nn) for s in states], dtype=np.float32)
        V = np.array([pvf(s)[1] for s in states], dtype=np.float64)
        out[f"{tag}_logits"] = logits.numpy(); out[f"{tag}_value"] = val.numpy().reshape(-1)
        out[f"{tag}_P"] = P; out[f"{tag}_v1"] = V
        if tag != "seeded":
            for kk, v in sd.items():
                out[f"{tag}__{kk}"] = v
    np.savez_compressed(os.path.join(HERE, f"net_{n}.npz"), **out)

    # ---------------- G4 real-net self-play games ----------------
    Sn = {5: 100, 9: 50, 15: 30}[n]
    gcount = {5: 3, 9: 1, 15: 1}[n]
    maxply = {5: nn, 9: 8, 15: 4}[n]
    sd = wsets["ckpt_saved"] if n == 5 else wsets["seeded"]
    net = net_from(sd)
    pvf = make_policy_value_fn(NeuralNetworkController(net, device="cpu"))
    recs = []
    for g in range(gcount):
        seed = 900 + g
        np.random.seed(seed)
        m = MCTS(pvf, num_simulations=Sn, c_puct=2.0)
        s = Gomoku(); mv = 0; hist = []
        while not s.is_terminal() and mv < maxply:
            T = default_temperature_schedule(mv)
            pi, a = m.run(s, temperature=T, add_root_noise=True)
            root = Capture.root
            N = np.zeros(nn, dtype=np.int32); W = np.zeros(nn, dtype=np.float64); P = np.zeros(nn, dtype=np.float32)
            for (r, c), ch in root.children.items():
                N[r * n + c] = ch.N; W[r * n + c] = ch.W; P[r * n + c] = np.float32(ch.prior)
            recs.append(dict(game=g, ply=mv, board=abs_board(s), player=1 if s.current_player == X else 2,
                             last=-1 if s.last_action is None else s.last_action[0] * n + s.last_action[1],
                             pi=pi.astype(np.float32).reshape(nn), N=N, W=W, P=P, action=a[0] * n + a[1], T=float(T)))
            hist.append(s.current_player)
            s = s.apply_action(a); mv += 1
        fin = s.get_game_result()
        for j, pl in enumerate(hist):
            recs[len(recs) - len(hist) + j]["z"] = 99 if fin is None else (0 if fin == DRAW else 1 if pl == fin else -1)
        recs[-1]["final"] = res_code[fin]
    for r_ in recs:
        r_.setdefault("final", 255)
    np.savez_compressed(
        os.path.join(HERE, f"netgame_{n}x{k}.npz"), n=n, k=k, S=Sn, seed0=900, maxply=maxply,
        weights="ckpt_saved" if n == 5 else "seeded",
        **{key: np.array([c[key] for c in recs]) for key in recs[0]})

    if n != 5:
        return
    # ---------------- G5 augmentation ----------------
    mgr = SelfPlayManager(controller=None, device="cpu")
    st = torch.arange(4 * nn, dtype=torch.float32).reshape(4, n, n)
    pi = (np.arange(nn, dtype=np.float32).reshape(n, n) + 1) / np.float32(nn * (nn + 1) / 2)
    sym = mgr._augment_symmetries(st, pi)
    np.savez_compressed(os.path.join(HERE, "augment.npz"), n=n, state=st.numpy(), pi=pi,
                        states=np.array([s_.numpy() for s_, _ in sym]), pis=np.array([p_ for _, p_ in sym]))

    # ---------------- G6 arena ----------------
    from evaluator import ModelEvaluator, temperature_schedule as eval_T
    cand = NeuralNetworkController(net_from(wsets["ckpt_saved"]), device="cpu")
    base = NeuralNetworkController(net_from(wsets["ckpt_0802"]), device="cpu")
    games_log = []
    class GameFactory:
        i = 0
        def __call__(self):
            np.random.seed(4000 + GameFactory.i)      # per-game seed (harness), evaluator.py:51 calls us once per game
            GameFactory.i += 1
            games_log.append([])
            return Gomoku()
    orig_run = MCTS.run
    def logged_run(self, root_state, temperature, add_root_noise=False):
        pi, a = orig_run(self, root_state, temperature, add_root_noise)
        games_log[-1].append((a[0] * n + a[1], float(temperature), 1 if root_state.current_player == X else 2))
        return pi, a
    MCTS.run = logged_run
    ev = ModelEvaluator(game_class=GameFactory(), print_games=False, device="cpu")
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        wr, metrics = ev.evaluate(cand, base, num_games=6, debug=False)
    MCTS.run = orig_run
    L = max(len(g_) for g_ in games_log)
    A = -np.ones((len(games_log), L), dtype=np.int16); TT = np.zeros((len(games_log), L)); PL = np.zeros((len(games_log), L), dtype=np.uint8)
    for i, g_ in enumerate(games_log):
        for j, (a, t, p) in enumerate(g_):
            A[i, j] = a; TT[i, j] = t; PL[i, j] = p
    np.savez_compressed(os.path.join(HERE, "arena_5x4.npz"), n=n, k=k, S=40, c_puct=2.0, seed0=4000,
                        actions=A, temps=TT, movers=PL, win_rate=wr,
                        wins=metrics["wins"], losses=metrics["losses"], draws=metrics["draws"], total=metrics["total"])
    # ---------------- G7 z truth table (alphazero/tests/tests.py:11-22) ----------------
    with open(os.path.join(HERE, "zlabels.json"), "w") as f:
        json.dump({"cases": [["D", "X", 0], ["D", "O", 0], ["X", "X", 1], ["X", "O", -1], ["O", "O", 1], ["O", "X", -1]],
                   "versions": {"numpy": np.__version__, "torch": torch.__version__, "python": sys.version.split()[0]}}, f, indent=1)


# --------------------------------------------------------------------------
# G4-full: the real net at the BASELINE workloads' search sizes -- 15x15 / 400 simulations (configs[3]) and
# 9x9 / 200 simulations (configs[2]) -- played by the Python reference itself.  Kept apart from worker() so that
# re-running it leaves the other fixtures untouched:  python tests/golden/make_golden.py full
# --------------------------------------------------------------------------
def worker_full(n, k, complete=False):
    """complete=True: ONE whole 15x15 / 400-simulation game of the Python reference from the empty board to its end
    (netgame_complete_15x5.npz, about 8 minutes of one container core), a different seed from the 7-ply fixture."""
    sys.path.insert(0, REF)
    import constants
    constants.BOARD_SIZE, constants.WIN_LENGTH = n, k
    import torch
    torch.set_num_threads(1)
    import mcts as mcts_mod
    from games import Gomoku
    from mcts import MCTS
    from net import GomokuNet
    from controller import NeuralNetworkController, make_policy_value_fn
    from self_play import default_temperature_schedule
    X, O, DRAW = constants.X, constants.O, constants.DRAW
    nn = n * n
    res_code = {None: 0, X: 1, O: 2, DRAW: 3}

    def abs_board(state):
        b = np.zeros(nn, dtype=np.uint8)
        for r in range(n):
            for q in range(n):
                s = state.board[r][q]
                if s is not None:
                    b[r * n + q] = 1 if s == X else 2
        return b

    class Capture:
        root = None
    OrigNode = mcts_mod.Node
    class RecNode(OrigNode):
        def __init__(self, state, parent=None, prior=1.0):
            super().__init__(state, parent, prior)
            if parent is None:
                Capture.root = self
    mcts_mod.Node = RecNode

    S = {9: 200, 15: 400}[n]
    maxply = n * n if complete else {9: 24, 15: 7}[n]
    seed = 1977 if complete else 1900
    net = GomokuNet(device="cpu")
    net.load_state_dict({kk: torch.tensor(v) for kk, v in build_weights(n).items()})
    net.eval()
    pvf = make_policy_value_fn(NeuralNetworkController(net, device="cpu"))
    np.random.seed(seed)
    m = MCTS(pvf, num_simulations=S, c_puct=2.0)
    s = Gomoku(); mv = 0; recs = []; hist = []
    while not s.is_terminal() and mv < maxply:
        T = default_temperature_schedule(mv)
        pi, a = m.run(s, temperature=T, add_root_noise=True)
        root = Capture.root
        N = np.zeros(nn, dtype=np.int32); W = np.zeros(nn, dtype=np.float64); P = np.zeros(nn, dtype=np.float32)
        for (r, c), ch in root.children.items():
            N[r * n + c] = ch.N; W[r * n + c] = ch.W; P[r * n + c] = np.float32(ch.prior)
        recs.append(dict(game=0, ply=mv, board=abs_board(s), player=1 if s.current_player == X else 2,
                         last=-1 if s.last_action is None else s.last_action[0] * n + s.last_action[1],
                         pi=pi.astype(np.float32).reshape(nn), N=N, W=W, P=P, action=a[0] * n + a[1], T=float(T)))
        hist.append(s.current_player)
        s = s.apply_action(a); mv += 1
        print(f"[full {n}x{n}] ply {mv}", flush=True)
    fin = s.get_game_result()
    for j, pl in enumerate(hist):
        recs[j]["z"] = 99 if fin is None else (0 if fin == DRAW else 1 if pl == fin else -1)
    for r_ in recs:
        r_["final"] = 255
    recs[-1]["final"] = res_code[fin]
    np.savez_compressed(
        os.path.join(HERE, f"netgame_{'complete' if complete else 'full'}_{n}x{k}.npz"), n=n, k=k, S=S, seed0=seed, maxply=maxply, weights="seeded",
        **{key: np.array([c[key] for c in recs]) for key in recs[0]})


def worker_resnet_checkpoint():
    """G8: values of a real ResidualBlock checkpoint + the build's own torch forward on seeded positions.  Nothing of the
    reference is imported or executed: torch.load(weights_only=True) reads tensors only."""
    import torch
    torch.set_num_threads(1)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from alphazero_piskvorky_amd.net import GomokuResNet
    src = "/root/reference/alphazero/models/old/model_20250728_225053.pt"
    sd = torch.load(src, map_location="cpu", weights_only=True)
    sd = {k: v for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    n = 5
    m = GomokuResNet(board_size=n)
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
    m.eval()
    rs = np.random.RandomState(58)
    nn = n * n
    boards, players, lasts, planes = [], [], [], []
    for t in range(24):
        stones = int(rs.randint(0, nn - 3))
        b = np.zeros(nn, np.uint8)
        cells = rs.permutation(nn)[:stones]
        b[cells[0::2]] = 1
        b[cells[1::2]] = 2
        pl = 1 + (stones & 1)
        la = int(cells[-1]) if stones else -1
        x = np.zeros((4, n, n), np.float32)                     # games.py:86-129 encode
        x[0] = (b == pl).reshape(n, n); x[1] = (b == 3 - pl).reshape(n, n)
        if la >= 0:
            x[2].reshape(-1)[la] = 1.0
        boards.append(b); players.append(pl); lasts.append(la); planes.append(x)
    with torch.no_grad():
        lg, v = m(torch.tensor(np.stack(planes)))
        P = torch.softmax(lg, 1)
    np.savez_compressed(os.path.join(HERE, "resnet_ckpt_5.npz"), n=n, source=os.path.basename(src),
                        boards=np.array(boards), players=np.array(players, np.uint8), lasts=np.array(lasts, np.int16),
                        logits=lg.numpy(), P=P.numpy(), value=v.numpy().reshape(-1),
                        versions=json.dumps({"numpy": np.__version__, "torch": torch.__version__}),
                        **{"w__" + k: t.numpy() for k, t in sd.items()})
    print("resnet_ckpt_5.npz written:", len(sd), "tensors,", float(v.abs().max()), "max |value|")


if __name__ == "__main__":
    if len(sys.argv) == 2 and sys.argv[1] == "resnet":
        worker_resnet_checkpoint()
    elif len(sys.argv) == 4 and sys.argv[3] in ("full", "complete"):
        worker_full(int(sys.argv[1]), int(sys.argv[2]), complete=sys.argv[3] == "complete")
    elif len(sys.argv) == 2 and sys.argv[1] == "complete":
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), "15", "5", "complete"], env=env, cwd="/tmp"))
    elif len(sys.argv) == 2 and sys.argv[1] == "full":
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(n), str(k), "full"], env=env, cwd="/tmp")
                 for n, k in ((9, 5), (15, 5))]
        rc = [p.wait() for p in procs]
        print("done", rc)
        sys.exit(max(rc))
    elif len(sys.argv) == 3:
        worker(int(sys.argv[1]), int(sys.argv[2]))
    else:
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(n), str(k)], env=env, cwd="/tmp")
                 for n, k in ((5, 4), (9, 5), (15, 5))]
        rc = [p.wait() for p in procs]
        print("done", rc)
        sys.exit(max(rc))
