"""Every board size the engine instantiates (3..15, the reference's BOARD_SIZE is a free integer, constants.py:2):
engine == oracle bit for bit for the plain net, the ResidualBlock net, the synthetic-evaluator tree and whole games."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc
import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd.net import fold_resnet_state_dict
from alphazero_piskvorky_amd.weights import synthetic_resnet_state_dict, synthetic_state_dict

SIZES = [(3, 3), (4, 3), (6, 4), (7, 4), (8, 5), (10, 5), (11, 5), (12, 5), (13, 5), (14, 5)]


def _positions(rs, n, cnt):
    boards = np.zeros((cnt, n * n), np.uint8); players = np.zeros(cnt, np.uint8); lasts = -np.ones(cnt, np.int16)
    for i in range(cnt):
        cells = rs.permutation(n * n)[:rs.randint(0, max(1, n * n // 2))]
        for j, c in enumerate(cells):
            boards[i, c] = 1 + j % 2
        players[i] = 1 + len(cells) % 2
        lasts[i] = cells[-1] if len(cells) else -1
    return boards, players, lasts


@pytest.mark.parametrize("n,k", SIZES)
def test_forward_both_models_bit_exact(n, k):
    o = orc.Oracle(n, k, 1)
    boards, players, lasts = _positions(np.random.RandomState(100 + n), n, 21)
    for model in ("plain", "resnet"):
        sd = synthetic_state_dict(n) if model == "plain" else synthetic_resnet_state_dict(n)
        onet = orc.Net(n, sd) if model == "plain" else orc.Net(n, resnet_tensors=fold_resnet_state_dict(sd))
        for split in (("0", "1000000") if model == "plain" else ("0",)):      # fused trunk | split trunk (plain net only)
            os.environ["AZ_SPLIT_MAX"] = split
            try:
                e = az.Engine(n, k, 4, 9, model=model)
            finally:
                os.environ.pop("AZ_SPLIT_MAX", None)
            e.load_weights(sd, 0)
            logits, P, v = e.net_eval(boards, players, lasts)
            for i in range(len(players)):
                ol, oP, ov = onet.eval(o.encode(boards[i], int(players[i]), int(lasts[i])))
                assert np.array_equal(logits[i], ol) and np.array_equal(P[i], oP) and v[i] == np.float32(ov), f"{model} split={split} n={n} board {i}"
            e.close()


@pytest.mark.parametrize("n,k", SIZES)
def test_selfplay_games_bit_exact(n, k):
    S, G = 40, 5
    cut = 0 if n <= 8 else 6
    for synthetic in (True, False):
        e = az.Engine(n, k, S, 3, synthetic=synthetic, log_table=orc.numpy_log_table(S))
        o = orc.Oracle(n, k, S, synthetic=synthetic)
        onet = None
        if not synthetic:
            sd = synthetic_state_dict(n)
            e.load_weights(sd, 0)
            onet = orc.Net(n, sd)
        e.selfplay(G, seed0=300 + n, max_plies=cut)
        rec = e.records(); nply, res = e.games()
        off = 0
        for g in range(G):
            noise, us = orc.selfplay_tape(300 + n + g, n)
            r = o.selfplay_game(onet, noise, us, maxply=cut if cut else None)
            L = int(nply[g]); sl = slice(off, off + L)
            assert L == r["nply"] and int(res[g]) == r["result"]
            for key in ("actions", "boards", "visits", "pis", "z"):
                assert np.array_equal(rec[key][sl], r[key]), f"n={n} synthetic={synthetic} game {g}: {key}"
            off += L
        e.close()


@pytest.mark.parametrize("n,k", SIZES)
def test_emulated_trunks_and_leaf_symmetry_on_every_size(n, k):
    """The opt-in modes on the sizes the dedicated tests do not visit: both emulation schemes within the tolerance of
    tests/test_emulated_trunk_gpu.py for both nets, and random-symmetry leaf evaluation bit for bit against the oracle."""
    o = orc.Oracle(n, k, 1)
    boards, players, lasts = _positions(np.random.RandomState(500 + n), n, 21)
    for model in ("plain", "resnet"):
        sd = synthetic_state_dict(n) if model == "plain" else synthetic_resnet_state_dict(n)
        onet = orc.Net(n, sd) if model == "plain" else orc.Net(n, resnet_tensors=fold_resnet_state_dict(sd))
        e = az.Engine(n, k, 4, 9, model=model)
        e.load_weights(sd, 0)
        want = [onet.eval(o.encode(boards[i], int(players[i]), int(lasts[i]))) for i in range(len(players))]
        for mode in ("bf16x3", "f16x2"):
            e.set_trunk_mode(mode)
            logits, P, v = e.net_eval(boards, players, lasts)
            for i, (ol, oP, ov) in enumerate(want):
                assert np.abs(logits[i] - ol).max() <= 2e-5 and np.abs(P[i] - oP).max() <= 1e-6, f"{model} {mode} n={n} board {i}"
                assert abs(float(v[i]) - float(ov)) <= (5e-6 if model == "resnet" else 2e-6), f"{model} {mode} n={n} board {i}"
            assert any(not np.array_equal(logits[i], want[i][0]) for i in range(len(want))), "the emulated trunk returned the canonical bits"
        e.close()
    S, G, cut = 24, 4, (0 if n <= 6 else 5)
    sd = synthetic_state_dict(n)
    e = az.Engine(n, k, S, 3, log_table=orc.numpy_log_table(S))
    e.load_weights(sd, 0)
    e.set_leaf_symmetry(True)
    e.selfplay(G, seed0=700 + n, max_plies=cut)
    rec = e.records(); nply, _ = e.games()
    ol = orc.Oracle(n, k, S, leaf_sym=True)
    onet = orc.Net(n, sd)
    off = 0
    for g in range(G):
        noise, us = orc.selfplay_tape(700 + n + g, n, maxply=cut or None)
        r = ol.selfplay_game(onet, noise, us, maxply=cut or None, game=700 + n + g)
        L = int(nply[g]); sl = slice(off, off + L)
        assert L == r["nply"]
        for key in ("actions", "visits", "pis"):
            assert np.array_equal(rec[key][sl], r[key]), f"leaf symmetry n={n} game {g}: {key}"
        off += L
    e.close()
