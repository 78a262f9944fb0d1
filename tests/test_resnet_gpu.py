"""ResidualBlock variant on the GPU: the HIP engine (model = resnet) against the oracle, bit for bit, plus the shims."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import oracle as orc
import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd.net import GomokuResNet, fold_resnet_state_dict
from alphazero_piskvorky_amd.weights import synthetic_resnet_state_dict


def _positions(rs, n, cnt):
    boards = np.zeros((cnt, n * n), np.uint8); players = np.zeros(cnt, np.uint8); lasts = -np.ones(cnt, np.int16)
    for i in range(cnt):
        cells = rs.permutation(n * n)[:rs.randint(0, n * n // 2)]
        for j, c in enumerate(cells):
            boards[i, c] = 1 + j % 2
        players[i] = 1 + len(cells) % 2
        lasts[i] = cells[-1] if len(cells) else -1
    return boards, players, lasts


@pytest.mark.parametrize("n", [5, 9, 15])
def test_resnet_forward_bit_exact_vs_oracle_and_close_to_torch(n):
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


@pytest.mark.parametrize("n,k,S,G,cut", [(5, 4, 40, 5, 0), (9, 5, 24, 4, 5), (15, 5, 16, 3, 3)])
def test_resnet_selfplay_bit_exact_vs_oracle(n, k, S, G, cut):
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
