"""Pins the CPU oracle (oracle/az_oracle.c) to golden vectors captured from the Python reference.

Tolerances (floating point only; everything integer is bit-exact):
  net logits   |d| <= 2e-5   (torch CPU conv sums in a different order)
  softmax P    |d| <= 1e-6
  value        |d| <= 2e-6
  pi (MCTS)    |d| <= 1e-6 (custom exp vs numpy exp, 1 ulp in float64 before the float32 store)
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests.util import SIZES, load, weights_from_fixture, synth_eval_codes, GOLDEN


@pytest.mark.parametrize("n,k", SIZES)
def test_rules_replay_bit_exact(n, k):
    z = load(f"rules_{n}x{k}.npz")
    o = orc.Oracle(n, k, 1)
    for g in range(len(z["nply"])):
        m = int(z["nply"][g])
        rc, term, board, pl, res = o.replay(z["actions"][g, :m])
        assert rc == 0
        assert not term.any(), "is_terminal must be False before every played move"
        assert res == int(z["result"][g]) and res != 0
    # encode planes + legal masks at sampled plies
    for i in range(len(z["enc_game"])):
        g, p = int(z["enc_game"][i]), int(z["enc_ply"][i])
        rc, term, board, pl, res = o.replay(z["actions"][g, :p])
        last = int(z["actions"][g, p - 1]) if p > 0 else -1
        planes = o.encode(board, pl, last)
        assert np.array_equal(planes.astype(np.uint8), z["enc_planes"][i])
        assert np.array_equal((board == 0).astype(np.uint8), z["legal_masks"][i])
    # crafted: overline wins, anti-diagonal at the corner wins, no wrap-around across the edge
    for i in range(len(z["crafted_result"])):
        seq = z["crafted_actions"][i]; seq = seq[seq >= 0]
        rc, term, board, pl, res = o.replay(seq)
        assert rc == 0 and res == int(z["crafted_result"][i])


def test_illegal_move_is_rejected():
    o = orc.Oracle(5, 4, 1)
    rc, *_ = o.replay([3, 3])
    assert rc == -1   # games.py:76-77 ValueError("Invalid move")


@pytest.mark.parametrize("n,k", SIZES)
def test_synthetic_evaluator_matches_python_definition(n, k):
    z = load(f"tree_{n}x{k}.npz")
    o = orc.Oracle(n, k, 1, synthetic=True)
    for i in range(len(z["seed"])):
        board, pl, last = z["board"][i], int(z["player"][i]), int(z["last"][i])
        codes = np.where(board == 0, 0, np.where(board == pl, 1, 2)).astype(np.uint8)
        P_py, v_py = synth_eval_codes(codes, last, n)
        P, v = o.synth_eval(board, pl, last)
        assert np.array_equal(P, P_py.reshape(-1)) and v == v_py


@pytest.mark.parametrize("n,k", SIZES)
def test_tree_search_bit_exact_vs_reference(n, k):
    """G2: MCTS.run with the synthetic evaluator -> visit counts, W (float64), priors, action are bit-exact."""
    z = load(f"tree_{n}x{k}.npz")
    S = int(z["S"])
    o = orc.Oracle(n, k, S, synthetic=True)
    nn = n * n
    for i in range(len(z["seed"])):
        board = z["board"][i]
        A = int((board == 0).sum())
        rs = np.random.RandomState(int(z["seed"][i]))
        noise = rs.dirichlet([0.3] * A) if z["noise"][i] else None
        u = rs.random_sample()
        r = o.search(None, board, int(z["player"][i]), int(z["last"][i]), float(z["T"][i]), noise, u)
        assert np.array_equal(r["N"], z["N"][i]), f"visit counts differ in case {i}"
        assert r["N"].sum() == S
        assert np.array_equal(r["W"], z["W"][i])
        assert np.array_equal(r["P"], z["P"][i])
        assert r["nexp"] == int(z["nexp"][i]) and r["maxd"] == int(z["maxd"][i])
        assert r["action"] == int(z["action"][i])
        np.testing.assert_allclose(r["pi"], z["pi"][i], rtol=0, atol=1e-6)


@pytest.mark.parametrize("n,k", SIZES)
def test_synthetic_selfplay_games_bit_exact(n, k):
    """G2b: whole games of the _worker loop body with the synthetic evaluator: every board, count and move."""
    z = load(f"synthgame_{n}x{k}.npz")
    S, maxply = int(z["S"]), int(z["maxply"])
    o = orc.Oracle(n, k, S, synthetic=True)
    for g in np.unique(z["game"]):
        sel = np.where(z["game"] == g)[0]
        noise, us = orc.selfplay_tape(int(z["seed0"]) + int(g), n)
        r = o.selfplay_game(None, noise, us, maxply=maxply)
        assert r["nply"] == len(sel)
        assert np.array_equal(r["actions"], z["action"][sel])
        assert np.array_equal(r["boards"], z["board"][sel])
        assert np.array_equal(r["movers"], z["player"][sel])
        assert np.array_equal(r["visits"], z["N"][sel])
        np.testing.assert_allclose(r["pis"], z["pi"][sel], rtol=0, atol=1e-6)
        fin = int(z["final"][sel[-1]])
        assert r["result"] == (0 if fin == 255 else fin)


@pytest.mark.parametrize("n", [5, 9, 15])
def test_net_forward_vs_torch(n):
    z = load(f"net_{n}.npz")
    tags = ["seeded"] + (["ckpt_saved", "ckpt_0802"] if n == 5 else [])
    o = orc.Oracle(n, 5, 1)
    for tag in tags:
        net = orc.Net(n, weights_from_fixture(n, tag))
        for i in range(len(z["players"])):
            planes = o.encode(z["boards"][i], int(z["players"][i]), int(z["lasts"][i]))
            logits, P, v = net.eval(planes)
            np.testing.assert_allclose(logits, z[f"{tag}_logits"][i], rtol=0, atol=2e-5)
            np.testing.assert_allclose(P, z[f"{tag}_P"][i], rtol=0, atol=1e-6)
            assert abs(v - float(z[f"{tag}_value"][i])) <= 2e-6
            assert abs(P.sum() - 1.0) < 1e-5


NETGAMES = ["netgame_5x4", "netgame_9x5", "netgame_15x5",
            "netgame_full_9x5", "netgame_full_15x5",     # G4-full: 9x9 / 200 sims (24 plies), 15x15 / 400 sims (7 plies)
            "netgame_complete_15x5"]                     # G4-complete: one whole reference game at 15x15 / 400 sims (46 plies)


@pytest.mark.parametrize("fixture", NETGAMES)
def test_real_net_games_teacher_forced(fixture):
    """G4: per ply, search from the reference's recorded position with the reference's RNG draws.
    Priors/values differ from torch in the last bits, so visit counts are compared with a small budget
    and must be identical whenever the chosen action matches (it must always match here)."""
    z = load(fixture + ".npz")
    n, k = int(z["n"]), int(z["k"])
    S = int(z["S"])
    o = orc.Oracle(n, k, S)
    net = orc.Net(n, weights_from_fixture(n, str(z["weights"])))
    nn = n * n
    exact = total = 0
    for g in np.unique(z["game"]):
        sel = np.where(z["game"] == g)[0]
        noise_tape, us = orc.selfplay_tape(int(z["seed0"]) + int(g), n)
        off = 0
        for j, idx in enumerate(sel):
            ply = int(z["ply"][idx]); A = nn - ply
            noise = noise_tape[off:off + A]; off += A
            r = o.search(net, z["board"][idx], int(z["player"][idx]), int(z["last"][idx]), float(z["T"][idx]), noise, us[ply])
            np.testing.assert_allclose(r["P"], z["P"][idx], rtol=0, atol=1e-6)
            total += 1
            if np.array_equal(r["N"], z["N"][idx]):
                exact += 1
                np.testing.assert_allclose(r["W"], z["W"][idx], rtol=0, atol=1e-4)
                np.testing.assert_allclose(r["pi"], z["pi"][idx], rtol=0, atol=1e-6)
                assert r["action"] == int(z["action"][idx])
            else:
                assert np.abs(r["N"] - z["N"][idx]).sum() <= max(4, S // 10)
    # the fixtures are frozen and the search is deterministic: today every recorded ply reproduces the reference's visit counts,
    # and a regression of a single ply must not hide behind a budget
    assert exact == total, f"only {exact}/{total} plies had identical visit counts"


def test_real_net_full_game_free_running():
    """5x5 real checkpoint: free-running games reproduce the reference's trajectory and z labels."""
    z = load("netgame_5x4.npz")
    o = orc.Oracle(5, 4, int(z["S"]))
    net = orc.Net(5, weights_from_fixture(5, "ckpt_saved"))
    same = 0
    games = np.unique(z["game"])
    for g in games:
        sel = np.where(z["game"] == g)[0]
        noise, us = orc.selfplay_tape(int(z["seed0"]) + int(g), 5)
        r = o.selfplay_game(net, noise, us)
        if r["nply"] == len(sel) and np.array_equal(r["actions"], z["action"][sel]):
            same += 1
            assert np.array_equal(r["z"], z["z"][sel])
            assert r["result"] == int(z["final"][sel[-1]])
    assert same == len(games)       # frozen fixture, deterministic search: every game, not all but one


def test_complete_reference_game_15x15_free_running():
    """G4-complete: ONE whole game of the Python reference at the headline shape (15x15, 400 simulations, seeded weights,
    np.random.seed(1977)); the oracle, running free from the same RNG tape, plays the same 46 moves to the same result
    with the same z labels, pi within 1e-6 and the reference's visit counts on every ply."""
    z = load("netgame_complete_15x5.npz")
    n, S = 15, int(z["S"])
    o = orc.Oracle(n, 5, S)
    net = orc.Net(n, weights_from_fixture(n, str(z["weights"])))
    noise, us = orc.selfplay_tape(int(z["seed0"]), n)
    r = o.selfplay_game(net, noise, us)
    assert r["nply"] == len(z["ply"]) and np.array_equal(r["actions"], z["action"])
    assert r["result"] == int(z["final"][-1]) and r["result"] != 0 and np.array_equal(r["z"], z["z"])
    assert np.array_equal(r["visits"], z["N"]) and np.array_equal(r["boards"], z["board"])
    np.testing.assert_allclose(r["pis"], z["pi"], rtol=0, atol=1e-6)


def test_augmentation_bug_compatible():
    z = load("augment.npz")
    o = orc.Oracle(int(z["n"]), 4, 1)
    outs, outp = o.augment(z["state"], z["pi"])
    assert np.array_equal(outs, z["states"])
    assert np.array_equal(outp, z["pis"])


def test_arena_games_vs_reference():
    z = load("arena_5x4.npz")
    n, k, S = int(z["n"]), int(z["k"]), int(z["S"])
    o = orc.Oracle(n, k, S)
    cand = orc.Net(n, weights_from_fixture(n, "ckpt_saved"))
    base = orc.Net(n, weights_from_fixture(n, "ckpt_0802"))
    wins = losses = draws = 0
    same = 0
    for g in range(z["actions"].shape[0]):
        ref = z["actions"][g]; ref = ref[ref >= 0]
        us = np.random.RandomState(int(z["seed0"]) + g).random_sample(n * n)
        r = o.arena_game(cand, base, g, us)
        # temperature bookkeeping quirk and first mover
        L = min(len(ref), r["nply"])
        np.testing.assert_allclose(r["temps"][:L], z["temps"][g][:L], rtol=1e-15)
        if r["nply"] == len(ref) and np.array_equal(r["actions"], ref):
            same += 1
        wins += r["result"] == 1; losses += r["result"] == 2; draws += r["result"] == 3
    assert same == z["actions"].shape[0]      # frozen fixture, deterministic search: every game, not all but one
    if same == z["actions"].shape[0]:
        assert (wins, losses, draws) == (int(z["wins"]), int(z["losses"]), int(z["draws"]))
        assert abs((wins + 0.5 * draws) / (wins + losses + draws) - float(z["win_rate"])) < 1e-12


def test_z_label_truth_table():
    cases = json.load(open(os.path.join(GOLDEN, "zlabels.json")))["cases"]
    code = {"X": 1, "O": 2, "D": 3}
    L = orc.lib()
    for winner, mover, z in cases:
        assert L.orc_zlabel(code[winner], code[mover]) == z


def test_math_helpers():
    L = orc.lib()
    xs = np.linspace(-80, 5, 2001).astype(np.float32)
    got = np.array([L.orc_test_expf(float(x)) for x in xs])
    np.testing.assert_allclose(got, np.exp(xs.astype(np.float64)), rtol=3e-7)
    xs = np.linspace(-700, 0, 3001)
    got = np.array([L.orc_test_exp(float(x)) for x in xs])
    np.testing.assert_allclose(got, np.exp(xs), rtol=4e-16)
    xs = np.linspace(-6, 6, 1001).astype(np.float32)
    got = np.array([L.orc_test_tanhf(float(x)) for x in xs])
    np.testing.assert_allclose(got, np.tanh(xs.astype(np.float64)), rtol=0, atol=2e-7)
    rs = np.random.RandomState(3)
    for n in list(range(1, 230)) + [300]:
        a = rs.random_sample(n)
        assert L.orc_test_pwsum(a.ctypes.data, n) == float(a.sum())
