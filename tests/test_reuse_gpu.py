"""Opt-in subtree reuse (az_set_subtree_reuse; the reference lists it as a TODO, mcts.py:17-22,106): engine vs the
oracle's restatement of the same rule, bit for bit.  "Parity unpinned" by the reference -- it has no such mode; what the
reference pins is that the mode is OFF by default (every other test) and that turning it off again restores its results."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc
from tests.util import weights_from_fixture

import alphazero_piskvorky_amd as az


def _compare_games(e, o, onet, n, G, seed0, maxply=None):
    rec = e.records(); nply, res = e.games()
    off = 0; tot = dict(expansions=0, terminal_hits=0, depth_sum=0, sims=0, root_evals=0)
    for g in range(G):
        noise, us = orc.selfplay_tape(seed0 + g, n)
        r = o.selfplay_game(onet, noise, us, maxply=maxply)
        L = int(nply[g]); sl = slice(off, off + L)
        assert L == r["nply"] and (maxply is not None or int(res[g]) == r["result"])
        for key in ("actions", "boards", "movers", "visits", "pis", "z", "lasts"):
            assert np.array_equal(rec[key][sl], r[key]), f"game {g}: {key} differs from the oracle"
        assert (r["visits"].sum(axis=1) == o.S).all()        # pi is still a distribution over S visits (Q7)
        for k in tot:
            tot[k] += r["counters"][k]
        off += L
    return tot


@pytest.mark.parametrize("n,k,S,G,slots", [(5, 4, 60, 9, 4), (9, 5, 48, 4, 3)])
def test_subtree_reuse_synthetic_evaluator_vs_oracle(n, k, S, G, slots):
    """Refilled slots (G > slots) must start from a fresh root; retained roots skip the evaluation and idle until
    their carried visits are reached."""
    e = az.Engine(n, k, S, slots, synthetic=True, log_table=orc.numpy_log_table(S))
    e.set_subtree_reuse(True)
    c = e.selfplay(G, seed0=9100)
    o = orc.Oracle(n, k, S, synthetic=True, reuse=True)
    tot = _compare_games(e, o, None, n, G, 9100)
    assert (c["expansions"], c["terminal_hits"], c["depth_sum"], c["simulations"]) == \
           (tot["expansions"], tot["terminal_hits"], tot["depth_sum"], tot["sims"])
    assert c["root_evals"] == tot["root_evals"] < c["plies"]          # some roots were retained
    assert c["simulations"] < c["plies"] * S                           # and their searches were topped up, not rerun
    # off again: the reference's behaviour is back
    e.set_subtree_reuse(False)
    e.selfplay(G, seed0=9100)
    _compare_games(e, orc.Oracle(n, k, S, synthetic=True), None, n, G, 9100)
    e.close()


@pytest.mark.parametrize("split", ["0", "1000000"])
def test_subtree_reuse_real_net_vs_oracle(split, monkeypatch):
    monkeypatch.setenv("AZ_SPLIT_MAX", split)
    monkeypatch.setenv("AZ_PERSIST", "0" if split == "0" else "1")     # once on the lock-step pipeline, once in the persistent kernel
    n, k, S, G = 5, 4, 50, 5
    sd = weights_from_fixture(5, "ckpt_saved")                         # a trained net: peaked priors, deep reuse
    e = az.Engine(n, k, S, 3, log_table=orc.numpy_log_table(S))
    e.load_weights(sd, 0)
    e.set_subtree_reuse(True)
    c = e.selfplay(G, seed0=77)
    assert (e.persistent() > 0) == (split != "0")      # round 3: the LDS-tree kernel loads the retained rows and writes the whole tree back
    tot = _compare_games(e, orc.Oracle(n, k, S, reuse=True), orc.Net(n, sd), n, G, 77)
    assert (c["expansions"], c["root_evals"]) == (tot["expansions"], tot["root_evals"])
    e.close()


def test_subtree_reuse_15x15_cut_games_and_multi_engine():
    n, k, S, G, cut = 15, 5, 40, 6, 5
    sd = weights_from_fixture(n, "seeded")
    e = az.MultiEngine(n, k, S, 4, engines=2, log_table=orc.numpy_log_table(S))
    e.load_weights(sd, 0)
    e.set_subtree_reuse(True)
    e.selfplay(G, seed0=31, max_plies=cut)
    _compare_games(e, orc.Oracle(n, k, S, reuse=True), orc.Net(n, sd), n, G, 31, maxply=cut)
    e.close()


def test_subtree_reuse_argument_checks():
    e = az.Engine(5, 4, 1024, 1, synthetic=True)
    with pytest.raises(az.AzError):
        e.set_subtree_reuse(True)                                      # more rows than k_move can renumber
    e.close()
    e = az.Engine(5, 4, 20, 2, synthetic=True)
    e.selfplay_begin(2, seed0=1)
    with pytest.raises(az.AzError):
        e.set_subtree_reuse(True)                                      # not while an episode is open
    e.selfplay_end()
    e.close()


@pytest.mark.parametrize("n,k,S,G,slots,cut", [(5, 4, 50, 8, 4, 0), (9, 5, 30, 4, 3, 8)])
def test_subtree_reuse_residual_block_net_vs_oracle(n, k, S, G, slots, cut):
    """The ResidualBlock net with subtree reuse: in the persistent kernel at 5x5 (round 3), on the lock-step pipeline at 9x9."""
    from alphazero_piskvorky_amd.net import fold_resnet_state_dict
    from alphazero_piskvorky_amd.weights import synthetic_resnet_state_dict
    sd = synthetic_resnet_state_dict(n)
    onet = orc.Net(n, resnet_tensors=fold_resnet_state_dict(sd))
    e = az.Engine(n, k, S, slots, model="resnet", log_table=orc.numpy_log_table(S))
    e.load_weights(sd, 0)
    e.set_subtree_reuse(True)
    c = e.selfplay(G, seed0=610, max_plies=cut)
    assert (e.persistent() > 0) == (n <= 7)
    rec = e.records(); nply, res = e.games()
    e.close()
    o = orc.Oracle(n, k, S, reuse=True)
    off = 0
    for g in range(G):
        noise, us = orc.selfplay_tape(610 + g, n, maxply=cut or None)
        r = o.selfplay_game(onet, noise, us, maxply=cut or None)
        Lg = int(nply[g]); sl = slice(off, off + Lg)
        assert Lg == r["nply"], f"game {g}"
        for key in ("actions", "boards", "visits", "pis"):
            assert np.array_equal(rec[key][sl], r[key]), f"game {g}: {key}"
        off += Lg
