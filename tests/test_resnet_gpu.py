"""ResidualBlock variant on the GPU: the HIP engine (model = resnet) against the oracle, bit for bit, plus the shims."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle as orc
import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd.net import GomokuResNet, fold_resnet_state_dict
from alphazero_piskvorky_amd.weights import synthetic_resnet_state_dict


SPLIT_MODES = ["0", "1000000"]     # AZ_SPLIT_MAX: fused k_trunk_res only | split (low-latency) trunk forced


def _positions(rs, n, cnt):
    boards = np.zeros((cnt, n * n), np.uint8); players = np.zeros(cnt, np.uint8); lasts = -np.ones(cnt, np.int16)
    for i in range(cnt):
        cells = rs.permutation(n * n)[:rs.randint(0, n * n // 2)]
        for j, c in enumerate(cells):
            boards[i, c] = 1 + j % 2
        players[i] = 1 + len(cells) % 2
        lasts[i] = cells[-1] if len(cells) else -1
    return boards, players, lasts


@pytest.mark.parametrize("split", SPLIT_MODES)
@pytest.mark.parametrize("n", [5, 9, 15])
def test_resnet_forward_bit_exact_vs_oracle_and_close_to_torch(n, split, monkeypatch):
    monkeypatch.setenv("AZ_SPLIT_MAX", split)
    sd = synthetic_resnet_state_dict(n)
    folded = fold_resnet_state_dict(sd)
    e = az.Engine(n, 5 if n > 5 else 4, 8, 16, model="resnet")
    e.load_weights(sd, 0)
    onet = orc.Net(n, resnet_tensors=folded)
    o = orc.Oracle(n, 5, 1)
    m = GomokuResNet(board_size=n)
    m.load_state_dict({k: torch.tensor(v) for k, v in sd.items()}, strict=False)
    m.eval()
    boards, players, lasts = _positions(np.random.RandomState(n), n, 23)      # more boards than slots, ragged tail
    logits, P, v = e.net_eval(boards, players, lasts)
    for i in range(len(players)):
        planes = o.encode(boards[i], int(players[i]), int(lasts[i]))
        ol, oP, ov = onet.eval(planes)
        assert np.array_equal(logits[i], ol), f"board {i}: max |d| = {np.abs(logits[i] - ol).max()}"
        assert np.array_equal(P[i], oP) and v[i] == np.float32(ov)
        with torch.no_grad():
            tl, tv = m(torch.tensor(planes)[None])
        np.testing.assert_allclose(logits[i], tl.numpy()[0], rtol=0, atol=1e-5)
        assert abs(float(v[i]) - float(tv)) <= 5e-6
    e.close()


@pytest.mark.parametrize("split", SPLIT_MODES)
def test_real_checkpoint_values_engine_equals_oracle_equals_build_torch(split, monkeypatch):
    """The VALUES of one of the reference's 20 historical ResidualBlock checkpoints (tests/golden/resnet_ckpt_5.npz, captured by
    a weights-only load) through the engine: logits / P / value bit for bit the oracle's, within 1e-5 / 1e-6 / 5e-6 of the
    build's torch module (outputs recorded in the fixture), and complete self-play games with these weights bit-exact
    against the oracle.  Parity against the reference itself: unpinned (no forward upstream)."""
    from tests.util import load
    monkeypatch.setenv("AZ_SPLIT_MAX", split)
    z = load("resnet_ckpt_5.npz")
    sd = {k[3:]: z[k] for k in z.files if k.startswith("w__")}
    n, k, S, G = int(z["n"]), 4, 60, 5
    e = az.Engine(n, k, S, 4, model="resnet", log_table=orc.numpy_log_table(S))
    e.load_weights(sd, 0)
    onet = orc.Net(n, resnet_tensors=fold_resnet_state_dict(sd))
    o = orc.Oracle(n, k, S)
    logits, P, v = e.net_eval(z["boards"], z["players"], z["lasts"])
    for i in range(len(z["players"])):
        ol, oP, ov = onet.eval(o.encode(z["boards"][i], int(z["players"][i]), int(z["lasts"][i])))
        assert np.array_equal(logits[i], ol) and np.array_equal(P[i], oP) and v[i] == np.float32(ov)
    np.testing.assert_allclose(logits, z["logits"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(P, z["P"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(v, z["value"], rtol=0, atol=5e-6)
    e.selfplay(G, seed0=4100)
    rec = e.records(); nply, res = e.games()
    off = 0
    for g in range(G):
        noise, us = orc.selfplay_tape(4100 + g, n)
        r = o.selfplay_game(onet, noise, us)
        L = int(nply[g]); sl = slice(off, off + L)
        assert L == r["nply"] and int(res[g]) == r["result"]
        for key in ("actions", "visits", "pis", "z"):
            assert np.array_equal(rec[key][sl], r[key]), f"game {g}: {key}"
        off += L
    e.close()


@pytest.mark.parametrize("split", SPLIT_MODES)
@pytest.mark.parametrize("n,k,S,G,cut", [(5, 4, 40, 5, 0), (9, 5, 24, 4, 5), (15, 5, 16, 3, 3)])
def test_resnet_selfplay_bit_exact_vs_oracle(n, k, S, G, cut, split, monkeypatch):
    monkeypatch.setenv("AZ_SPLIT_MAX", split)
    sd = synthetic_resnet_state_dict(n)
    e = az.Engine(n, k, S, 3, model="resnet", log_table=orc.numpy_log_table(S))
    e.load_weights(sd, 0)
    e.selfplay(G, seed0=77, max_plies=cut)
    rec = e.records(); nply, res = e.games()
    o = orc.Oracle(n, k, S); onet = orc.Net(n, resnet_tensors=fold_resnet_state_dict(sd))
    off = 0
    for g in range(G):
        noise, us = orc.selfplay_tape(77 + g, n)
        r = o.selfplay_game(onet, noise, us, maxply=cut if cut else None)
        L = int(nply[g]); sl = slice(off, off + L)
        assert L == r["nply"] and int(res[g]) == r["result"]
        for key in ("actions", "boards", "visits", "pis", "z"):
            assert np.array_equal(rec[key][sl], r[key]), f"game {g}: {key} differs from the oracle"
        off += L
    e.close()


def test_resnet_arena_and_shims():
    from alphazero_piskvorky_amd import constants, games
    from alphazero_piskvorky_amd.controller import NeuralNetworkController, make_policy_value_fn
    from alphazero_piskvorky_amd.evaluator import ModelEvaluator
    from alphazero_piskvorky_amd.self_play import SelfPlayManager
    n = 5

    def ctrl(seed):
        m = GomokuResNet(board_size=n)
        m.load_state_dict({k: torch.tensor(v) for k, v in synthetic_resnet_state_dict(n, seed).items()}, strict=False)
        m.eval()
        return NeuralNetworkController(m, device="cuda:0")

    a, b = ctrl(1), ctrl(2)
    # arena through the C-ABI vs the oracle
    S = 30
    e = az.Engine(n, 4, S, 4, model="resnet", log_table=orc.numpy_log_table(S))
    e.load_weights(a.net.state_dict(), 0); e.load_weights(b.net.state_dict(), 1)
    r = e.arena(5, seed0=900, temperature_table=orc.arena_T_table(n * n))
    o = orc.Oracle(n, 4, S)
    oa = orc.Net(n, resnet_tensors=fold_resnet_state_dict(a.net.state_dict()))
    ob = orc.Net(n, resnet_tensors=fold_resnet_state_dict(b.net.state_dict()))
    for g in range(5):
        ro = o.arena_game(oa, ob, g, np.random.RandomState(900 + g).random_sample(n * n))
        assert int(r["results"][g]) == ro["result"] and np.array_equal(r["actions"][g][:ro["nply"]], ro["actions"])
    e.close()
    # the Python seams pick the model kind from the controller's net
    P, v = make_policy_value_fn(a)(games.Gomoku(n, 4))
    assert P.shape == (n, n) and abs(float(P.sum()) - 1.0) < 1e-5 and -1.0 <= v <= 1.0
    data = SelfPlayManager(a, "cuda:0", mcts_params={"num_simulations": 20}, concurrent_games=8, seed=4).generate_self_play(8)
    assert len(data) > 0 and tuple(data[0][0].shape) == (4, n, n)
    saved = constants.NUM_EVAL_SIMULATIONS
    constants.NUM_EVAL_SIMULATIONS = 20
    try:
        wr, metrics = ModelEvaluator(games.Gomoku, False, "cuda:0", seed=3).evaluate(a, b, num_games=4)
    finally:
        constants.NUM_EVAL_SIMULATIONS = saved
    assert metrics["total"] == 4 and 0.0 <= wr <= 1.0
    out = a.train(data[:64], epochs=1)
    assert np.isfinite(out["loss"])


_ORACLE_CACHE = {}


def _host_threads():
    import os
    return max(1, min(16, os.cpu_count() or 1))


def test_config4_resnet_15x15_800sims_complete_games_bit_exact_vs_oracle():
    """BASELINE configs[4] at its workload: complete 15x15 / 5-in-a-row games, ResidualBlock net, 800 simulations per
    move.  Every ply of every game is searched again by the oracle from the engine's recorded position with the same
    tape (one search per host thread), and the recorded boards are the oracle-rules replay of the recorded moves, so by
    induction the games equal the oracle's free-running games: visit counts, pi bit patterns, moves, outcomes, z."""
    from concurrent.futures import ThreadPoolExecutor
    n, k, S, G, seed0 = 15, 5, 800, 2, 4100
    nn = n * n
    sd = synthetic_resnet_state_dict(n)
    e = az.Engine(n, k, S, G, model="resnet", log_table=orc.numpy_log_table(S))
    e.load_weights(sd, 0)
    c = e.selfplay(G, seed0=seed0)
    rec = e.records(); nply, res = e.games()
    e.close()
    assert c["simulations"] == S * c["plies"] and c["expansions"] + c["terminal_hits"] == c["simulations"]
    o = orc.Oracle(n, k, S)
    onet = orc.Net(n, resnet_tensors=fold_resnet_state_dict(sd))
    T = orc.selfplay_T_table(nn)
    jobs = []
    off = 0
    for g in range(G):
        L = int(nply[g])
        assert L >= 9 and int(res[g]) in (1, 2, 3)
        tape, us = orc.selfplay_tape(seed0 + g, n, maxply=L)
        rc, term, board, pl, result = o.replay(rec["actions"][off:off + L])
        assert rc == 0 and not term.any() and result == int(res[g])
        want_z = np.array([0 if result == 3 else (1 if m == result else -1) for m in rec["movers"][off:off + L]])
        assert np.array_equal(rec["z"][off:off + L], want_z)
        noff = 0
        for m in range(L):
            _, _, bm, pm, _ = o.replay(rec["actions"][off:off + m])
            assert np.array_equal(rec["boards"][off + m], bm) and int(rec["movers"][off + m]) == pm
            jobs.append((off + m, bm, pm, int(rec["lasts"][off + m]), float(T[m]), tape[noff:noff + nn - m], float(us[m])))
            noff += nn - m
        off += L

    def one(j):
        ri, board, pl, last, temp, noise, u = j
        r = o.search(onet, board, pl, last, temp, noise, u)
        return ri, r

    with ThreadPoolExecutor(_host_threads()) as ex:
        for ri, r in ex.map(one, jobs):
            assert np.array_equal(rec["visits"][ri], r["N"]), f"record {ri}: visit counts differ from the oracle"
            assert np.array_equal(rec["pis"][ri], r["pi"]), f"record {ri}: pi differs bit-wise from the oracle"
            assert int(rec["actions"][ri]) == r["action"]


@pytest.mark.parametrize("split", SPLIT_MODES)
def test_config4_resnet_15x15_arena_bit_exact_vs_oracle(split, monkeypatch):
    """configs[4]'s head-to-head arena (evaluator.py:50-104) on the 15x15 ResidualBlock net: 4 games at 200 simulations
    (NUM_EVAL_SIMULATIONS, constants.py) between two weight sets, engine vs the oracle's free-running games."""
    from concurrent.futures import ThreadPoolExecutor
    monkeypatch.setenv("AZ_SPLIT_MAX", split)
    n, k, S, G, seed0 = 15, 5, 200, 4, 8800
    a, b = synthetic_resnet_state_dict(n, 1), synthetic_resnet_state_dict(n, 2)
    e = az.Engine(n, k, S, G, model="resnet", log_table=orc.numpy_log_table(S))
    e.load_weights(a, 0); e.load_weights(b, 1)
    r = e.arena(G, seed0=seed0, temperature_table=orc.arena_T_table(n * n))
    e.close()
    o = orc.Oracle(n, k, S)
    oa = orc.Net(n, resnet_tensors=fold_resnet_state_dict(a))
    ob = orc.Net(n, resnet_tensors=fold_resnet_state_dict(b))

    def one(g):
        return o.arena_game(oa, ob, g, np.random.RandomState(seed0 + g).random_sample(n * n))

    if "arena" not in _ORACLE_CACHE:               # the oracle's games are the same for both trunk variants
        with ThreadPoolExecutor(min(G, _host_threads())) as ex:
            _ORACLE_CACHE["arena"] = list(ex.map(one, range(G)))
    outs = _ORACLE_CACHE["arena"]
    w = l = d = 0
    for g, ro in enumerate(outs):
        assert int(r["nply"][g]) == ro["nply"] and int(r["results"][g]) == ro["result"], f"arena game {g}"
        assert np.array_equal(r["actions"][g][:ro["nply"]], ro["actions"])
        w += ro["result"] == 1; l += ro["result"] == 2; d += ro["result"] == 3
    assert (r["wins"], r["losses"], r["draws"]) == (w, l, d) and r["total"] == G
