"""GPU tests of opt-in random-symmetry leaf evaluation (az_set_leaf_symmetry): the engine against the oracle's restatement
(orc_cfg.leaf_sym), bit for bit -- the symmetry of every evaluation is a fixed hash of (game, ply, evaluation index), the
forward pass is the canonical-order one, so nothing is left to tolerance.  "Parity unpinned" by the reference, which has no
such mode (tests/test_leaf_symmetry_cpu.py pins the oracle's symmetries to numpy's rot90 / fliplr)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc
from tests.util import build_weights, weights_from_fixture

import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd import _capi


def _nets(n, model):
    if model == "resnet":
        from alphazero_piskvorky_amd.net import fold_resnet_state_dict
        from alphazero_piskvorky_amd.weights import synthetic_resnet_state_dict
        sd = synthetic_resnet_state_dict(n)
        return sd, orc.Net(n, resnet_tensors=fold_resnet_state_dict(sd))
    sd = build_weights(n)
    return sd, orc.Net(n, sd)


@pytest.mark.parametrize("split", ["0", "1000000"])
@pytest.mark.parametrize("n,k,S,G,cut,model", [(5, 4, 40, 6, 0, "plain"), (9, 5, 30, 4, 6, "plain"), (15, 5, 16, 3, 4, "plain"),
                                               (9, 5, 24, 3, 5, "resnet")])
def test_selfplay_with_leaf_symmetry_bit_exact_vs_oracle(n, k, S, G, cut, model, split, monkeypatch):
    monkeypatch.setenv("AZ_SPLIT_MAX", split)
    monkeypatch.setenv("AZ_PERSIST", "0" if split == "0" else "1")        # 5x5: once on the lock-step pipeline, once in the persistent kernel
    seed0 = 7700
    sd, onet = _nets(n, model)
    e = az.Engine(n, k, S, 3, log_table=orc.numpy_log_table(S), model=model)       # fewer slots than games: refills included
    e.load_weights(sd, 0)
    e.set_leaf_symmetry(True)
    c = e.selfplay(G, seed0=seed0, max_plies=cut)
    assert (e.persistent() > 0) == (n <= 7 and model == "plain" and split != "0")     # round 3: the persistent kernel knows the option too
    rec = e.records(); nply, res = e.games()
    o = orc.Oracle(n, k, S, leaf_sym=True)
    plain = orc.Oracle(n, k, S)
    off, differs = 0, 0
    for g in range(G):
        noise, us = orc.selfplay_tape(seed0 + g, n, maxply=cut or None)
        r = o.selfplay_game(onet, noise, us, maxply=cut or None, game=seed0 + g)      # the hash names a game by its seed
        L = int(nply[g]); sl = slice(off, off + L)
        assert L == r["nply"], f"game {g}"
        for key in ("actions", "boards", "visits", "pis"):
            assert np.array_equal(rec[key][sl], r[key]), f"game {g}: {key} differs from the oracle"
        if not cut:
            assert int(res[g]) == r["result"] and np.array_equal(rec["z"][sl], r["z"])
        differs += not np.array_equal(plain.selfplay_game(onet, noise, us, maxply=cut or None)["visits"][:1], r["visits"][:1])
        off += L
    assert differs > 0, "the option changed nothing: no game's first search differs from the unrotated one"
    assert c["simulations"] == S * c["plies"]
    e.set_leaf_symmetry(False)            # and the default comes back bit-exact
    e.selfplay(1, seed0=seed0, max_plies=cut)
    r0 = plain.selfplay_game(onet, *orc.selfplay_tape(seed0, n, maxply=cut or None), maxply=cut or None)
    assert np.array_equal(e.records()["visits"], r0["visits"])
    e.close()


def test_single_search_and_arena_with_leaf_symmetry():
    n, k, S = 5, 4, 50
    cand, base = weights_from_fixture(n, "ckpt_saved"), weights_from_fixture(n, "ckpt_0802")
    e = az.Engine(n, k, S, 4, log_table=orc.numpy_log_table(S))
    e.load_weights(cand, 0); e.load_weights(base, 1)
    e.set_leaf_symmetry(True)
    o = orc.Oracle(n, k, S, leaf_sym=True)
    oc, ob = orc.Net(n, cand), orc.Net(n, base)
    board = np.zeros(n * n, np.uint8); board[[12, 7, 8]] = [1, 2, 1]
    noise = np.random.RandomState(4).dirichlet([0.3] * (n * n - 3))
    r = e.search(board, 2, 8, 0.9, noise, 0.61)
    ro = o.search(oc, board, 2, 8, 0.9, noise, 0.61, game=0)
    assert np.array_equal(r["N"], ro["N"]) and np.array_equal(r["P"], ro["P"]) and np.array_equal(r["W"], ro["W"])
    assert np.array_equal(r["pi"], ro["pi"]) and r["action"] == ro["action"]
    G = 6
    a = e.arena(G, seed0=31, temperature_table=orc.arena_T_table(n * n))
    for g in range(G):
        us = np.random.RandomState(31 + g).random_sample(n * n)
        rg = o.arena_game(oc, ob, g, us, key=31 + g)
        assert int(a["nply"][g]) == rg["nply"] and int(a["results"][g]) == rg["result"]
        assert np.array_equal(a["actions"][g][:rg["nply"]], rg["actions"])
    e.close()


def test_leaf_symmetry_with_the_emulated_trunk_and_rejected_evaluators():
    n, k, S = 9, 5, 40
    e = az.Engine(n, k, S, 4)
    e.load_weights(build_weights(n), 0)
    e.set_leaf_symmetry(True)
    e.set_trunk_mode("bf16x3")            # both opt-ins together: runs, every search makes its S simulations
    c = e.selfplay(4, seed0=3, max_plies=5)
    assert c["simulations"] == S * c["plies"] and int(e.records()["visits"].sum()) == S * c["plies"]
    e.set_trunk_mode("f32")
    e.close()
    s = az.Engine(5, 4, 8, 2, synthetic=True)
    with pytest.raises(_capi.AzError):
        s.set_leaf_symmetry(True)         # the synthetic evaluator is not a net
    s.close()


@pytest.mark.parametrize("n,k,S,G,cut,L", [(5, 4, 40, 5, 0, 4), (9, 5, 30, 3, 6, 8), (15, 5, 24, 2, 4, 5)])
def test_leaf_symmetry_with_virtual_loss_batching_bit_exact_vs_oracle(n, k, S, G, cut, L):
    """The two opt-ins together (round 3; they were mutually exclusive): simulation s of a batch is evaluation s + 1 of the
    search, exactly as in the sequential loop, so the symmetry hash needs no new definition.  Engine == oracle bit for bit
    (oracle: orc_cfg.vl + orc_cfg.leaf_sym), in either order of switching the options on."""
    seed0 = 8800
    sd, onet = _nets(n, "plain")
    o = orc.Oracle(n, k, S, leaf_sym=True, virtual_loss=L)
    for order in (0, 1):
        e = az.Engine(n, k, S, 3, log_table=orc.numpy_log_table(S))
        e.load_weights(sd, 0)
        if order == 0:
            e.set_leaf_symmetry(True); e.set_virtual_loss(L)
        else:
            e.set_virtual_loss(L); e.set_leaf_symmetry(True)
        c = e.selfplay(G, seed0=seed0, max_plies=cut)
        rec = e.records(); nply, res = e.games()
        e.close()
        assert c["simulations"] == S * c["plies"]
        off = 0
        for g in range(G):
            noise, us = orc.selfplay_tape(seed0 + g, n, maxply=cut or None)
            r = o.selfplay_game(onet, noise, us, maxply=cut or None, game=seed0 + g)
            Lg = int(nply[g]); sl = slice(off, off + Lg)
            assert Lg == r["nply"], f"game {g}"
            for key in ("actions", "boards", "visits", "pis"):
                assert np.array_equal(rec[key][sl], r[key]), f"game {g}: {key} differs from the oracle (order {order})"
            off += Lg


@pytest.mark.parametrize("n,k,S,G,slots,cut,L", [(5, 4, 60, 96, 32, 0, 1), (9, 5, 40, 24, 8, 8, 1), (5, 4, 40, 48, 16, 0, 4)])
def test_leaf_symmetry_with_the_evaluation_cache(n, k, S, G, slots, cut, L):
    """Round 3: the two opt-ins compose.  The net's outputs depend on the symmetry it was shown, so the symmetry is part of the
    cache key, and a hit lands in the logits row in image coordinates like a fresh evaluation.  Records with the cache on are the
    records with it off -- which are the oracle's (orc_cfg.leaf_sym) -- and the cache does hit (trained 5x5 weights)."""
    seed0 = 4242
    sd = weights_from_fixture(5, "ckpt_saved") if n == 5 else build_weights(n)
    onet = orc.Net(n, sd)
    out = []
    for entries in (0, 1 << 16):
        e = az.Engine(n, k, S, slots, log_table=orc.numpy_log_table(S))
        e.load_weights(sd, 0)
        if L > 1:
            e.set_virtual_loss(L)
        e.set_eval_cache(entries)
        e.set_leaf_symmetry(True)
        c = e.selfplay(G, seed0=seed0, max_plies=cut)
        out.append((e.records(), e.games(), c))
        e.close()
    (r0, g0, c0), (r1, g1, c1) = out
    for key in r0:
        assert np.array_equal(r0[key], r1[key]), key
    assert np.array_equal(g0[0], g1[0]) and np.array_equal(g0[1], g1[1])
    assert c1["cache_lookups"] > 0 and c1["cache_hits"] > 0 and c0["cache_hits"] == 0
    o = orc.Oracle(n, k, S, leaf_sym=True, virtual_loss=L) if L > 1 else orc.Oracle(n, k, S, leaf_sym=True)
    off = 0
    for g in range(3):
        noise, us = orc.selfplay_tape(seed0 + g, n, maxply=cut or None)
        r = o.selfplay_game(onet, noise, us, maxply=cut or None, game=seed0 + g)
        Lg = int(g1[0][g]); sl = slice(off, off + Lg)
        assert Lg == r["nply"]
        for key in ("actions", "visits", "pis"):
            assert np.array_equal(r1[key][sl], r[key]), f"game {g}: {key}"
        off += Lg


@pytest.mark.parametrize("n,k,S,G,slots,cut", [(5, 4, 60, 24, 8, 0), (9, 5, 40, 6, 4, 10)])
def test_leaf_symmetry_with_subtree_reuse_bit_exact_vs_oracle(n, k, S, G, slots, cut):
    """Round 3: the two opt-ins compose without a new definition -- a retained root is not evaluated again, and simulation s
    of its search is evaluation s + 1 whether the search starts at 0 or at the visits the root carried over.  Engine == oracle
    (orc_cfg.reuse + orc_cfg.leaf_sym) bit for bit, and roots are in fact retained."""
    seed0 = 5151
    sd = weights_from_fixture(5, "ckpt_saved") if n == 5 else build_weights(n)
    onet = orc.Net(n, sd)
    e = az.Engine(n, k, S, slots, log_table=orc.numpy_log_table(S))
    e.load_weights(sd, 0)
    e.set_leaf_symmetry(True)
    e.set_subtree_reuse(True)
    c = e.selfplay(G, seed0=seed0, max_plies=cut)
    rec = e.records(); nply, res = e.games()
    e.close()
    assert c["simulations"] < S * c["plies"]          # retained roots carried visits over
    o = orc.Oracle(n, k, S, leaf_sym=True, reuse=True)
    off = 0
    for g in range(G):
        noise, us = orc.selfplay_tape(seed0 + g, n, maxply=cut or None)
        r = o.selfplay_game(onet, noise, us, maxply=cut or None, game=seed0 + g)
        Lg = int(nply[g]); sl = slice(off, off + Lg)
        assert Lg == r["nply"], f"game {g}"
        for key in ("actions", "boards", "visits", "pis"):
            assert np.array_equal(rec[key][sl], r[key]), f"game {g}: {key}"
        off += Lg
