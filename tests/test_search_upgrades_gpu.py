"""The two search upgrades the reference only lists as TODOs (mcts.py:17-22), both opt-in:

* evaluation cache (az_set_eval_cache): must change NOTHING -- records, visit counts, pi bit patterns and the work
  counters are identical with the cache on or off; only the number of boards the net kernels evaluate drops.
* virtual-loss batching (az_set_virtual_loss): changes the visit counts by design, so its parity is against the
  oracle's restatement of the rule (oracle/az_oracle.c, orc_cfg.vl) -- "parity unpinned" by the reference; batches of
  one reproduce the reference's sequential loop exactly.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc
from tests.util import weights_from_fixture

import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd.net import fold_resnet_state_dict
from alphazero_piskvorky_amd.weights import synthetic_resnet_state_dict

WORK = ("games", "plies", "records", "simulations", "expansions", "root_evals", "terminal_hits", "depth_sum")


def _same_records(a, b):
    for key in a:
        assert np.array_equal(a[key], b[key]), key


@pytest.mark.parametrize("n,k,S,G,slots,engines,tag,cut", [(5, 4, 100, 96, 32, 1, "ckpt_saved", 0), (9, 5, 60, 40, 16, 3, "seeded", 12),
                                                           (15, 5, 48, 12, 8, 2, "seeded", 4)])
def test_eval_cache_changes_nothing_but_the_net_work(n, k, S, G, slots, engines, tag, cut):
    sd = weights_from_fixture(n, tag)
    out = {}
    for entries in (0, 1 << 16):
        e = az.Engine(n, k, S, slots, engines=engines, log_table=orc.numpy_log_table(S))
        e.load_weights(sd, 0)
        e.set_eval_cache(entries)
        c = e.selfplay(G, seed0=606, max_plies=cut)
        out[entries] = (e.records(), e.games(), c)
        if entries:
            c2 = e.selfplay(G, seed0=606, max_plies=cut)          # the same episode again: now nearly everything is known
            out["again"] = (e.records(), e.games(), c2)
        e.close()
    (r0, g0, c0), (r1, g1, c1), (r2, g2, c2) = out[0], out[1 << 16], out["again"]
    _same_records(r0, r1); _same_records(r0, r2)
    assert np.array_equal(g0[0], g1[0]) and np.array_equal(g0[1], g1[1]) and np.array_equal(g0[0], g2[0])
    for key in WORK:
        assert c0[key] == c1[key] == c2[key], key
    assert c0["cache_lookups"] == 0 and c0["cache_hits"] == 0 and c0["trunk_boards"] == c0["expansions"] + c0["root_evals"]
    assert c1["cache_lookups"] == c1["expansions"] + c1["root_evals"]          # every evaluation is looked up first
    assert c1["trunk_boards"] == c1["expansions"] + c1["root_evals"] - c1["cache_hits"]
    assert c1["cache_hits"] > 0, "games share their opening positions: there must be hits"
    assert c2["cache_hits"] > c1["cache_hits"] and c2["cache_hits"] >= 0.5 * c2["cache_lookups"]
    print(f"{n}x{n} S={S}: hit rate {c1['cache_hits'] / c1['cache_lookups']:.3f} (first episode), "
          f"{c2['cache_hits'] / c2['cache_lookups']:.3f} (repeated episode)")


def test_eval_cache_is_invalidated_by_new_weights_and_keyed_by_net():
    """az_load_weights bumps the cache generation (old entries never match); the arena's two nets do not share entries."""
    n, k, S = 5, 4, 40
    a, b = weights_from_fixture(n, "ckpt_saved"), weights_from_fixture(n, "ckpt_0802")
    ref = az.Engine(n, k, S, 8, log_table=orc.numpy_log_table(S))
    e = az.Engine(n, k, S, 8, log_table=orc.numpy_log_table(S))
    e.set_eval_cache(1 << 14)
    for sd in (a, b, a):
        for eng in (ref, e):
            eng.load_weights(sd, 0)
        ref.selfplay(12, seed0=9)
        c = e.selfplay(12, seed0=9)
        _same_records(ref.records(), e.records())
        assert c["cache_hits"] > 0
    for eng in (ref, e):
        eng.load_weights(a, 0); eng.load_weights(b, 1)
    T = orc.arena_T_table(n * n)
    r0, r1 = ref.arena(9, seed0=31, temperature_table=T), e.arena(9, seed0=31, temperature_table=T)
    assert np.array_equal(r0["actions"], r1["actions"]) and np.array_equal(r0["results"], r1["results"])
    # a position searched with the cache on (single search, both weight slots)
    board = np.zeros(n * n, np.uint8); board[12] = 1; board[7] = 2
    for slot in (0, 1):
        s0, s1 = ref.search(board, 1, 7, 0.9, None, 0.4, slot=slot), e.search(board, 1, 7, 0.9, None, 0.4, slot=slot)
        assert np.array_equal(s0["N"], s1["N"]) and np.array_equal(s0["W"], s1["W"]) and np.array_equal(s0["pi"], s1["pi"])
    ref.close(); e.close()


def test_eval_cache_resnet_and_tiny_table():
    """ResidualBlock net with a table so small (1024 entries) that entries are overwritten all the time."""
    n, k, S, G = 9, 5, 40, 10
    sd = synthetic_resnet_state_dict(n)
    out = []
    for entries in (0, 1):
        e = az.Engine(n, k, S, 4, model="resnet", log_table=orc.numpy_log_table(S))
        e.load_weights(sd, 0)
        e.set_eval_cache(entries)
        c = e.selfplay(G, seed0=12, max_plies=6)
        out.append((e.records(), c))
        e.close()
    _same_records(out[0][0], out[1][0])
    assert out[1][1]["cache_hits"] > 0


def _random_position(rs, n, k, stones):
    o = orc.Oracle(n, k, 1)
    while True:
        acts = list(rs.permutation(n * n)[:stones])
        rc, term, board, pl, res = o.replay(acts)
        if rc == 0 and res == 0 and not term.any():
            return board, pl, (acts[-1] if acts else -1)


@pytest.mark.parametrize("L", [2, 5, 8, 32])
def test_virtual_loss_search_bit_exact_vs_oracle(L):
    """Single searches, synthetic evaluator: open boards, and 5x5 boards with 2-6 empty cells where batches run into
    terminal leaves and into leaves that are already pending (duplicates)."""
    rs = np.random.RandomState(40 + L)
    dups = 0
    for n, k, S, stones_list in ((9, 5, 70, (0, 9, 40)), (5, 4, 50, (19, 21, 22, 23)), (15, 5, 33, (0, 30))):
        e = az.Engine(n, k, S, 2, synthetic=True, log_table=orc.numpy_log_table(S))
        e.set_virtual_loss(L)
        o = orc.Oracle(n, k, S, synthetic=True, virtual_loss=L)
        for stones in stones_list:
            for _ in range(3):
                board, pl, last = _random_position(rs, n, k, stones)
                noise = rs.dirichlet([0.3] * (n * n - stones))
                r = e.search(board, pl, last, 0.8, noise, 0.37)
                ro = o.search(None, board, pl, last, 0.8, noise, 0.37)
                assert np.array_equal(r["N"], ro["N"]), f"{n}x{n}, {stones} stones, L={L}"
                assert np.array_equal(r["W"], ro["W"]) and np.array_equal(r["P"], ro["P"])
                assert np.array_equal(r["pi"], ro["pi"]) and r["action"] == ro["action"]
                assert r["N"].sum() == S
        e.close()


@pytest.mark.parametrize("L,synthetic", [(4, True), (8, False), (32, False), (3, False)])
def test_virtual_loss_selfplay_games_bit_exact_vs_oracle(L, synthetic):
    n, k, S, G = (5, 4, 60, 6) if L != 3 else (9, 5, 50, 4)
    cut = 0 if n == 5 else 7
    sd = weights_from_fixture(n, "ckpt_saved" if n == 5 else "seeded")
    e = az.Engine(n, k, S, 4, synthetic=synthetic, log_table=orc.numpy_log_table(S))
    if not synthetic:
        e.load_weights(sd, 0)
    e.set_virtual_loss(L)
    c = e.selfplay(G, seed0=321, max_plies=cut)
    rec = e.records(); nply, res = e.games()
    e.close()
    o = orc.Oracle(n, k, S, synthetic=synthetic, virtual_loss=L)
    onet = None if synthetic else orc.Net(n, sd)
    off = 0
    tot = dict(expansions=0, terminal_hits=0, depth_sum=0, dup_sims=0, sims=0)
    for g in range(G):
        noise, us = orc.selfplay_tape(321 + g, n)
        r = o.selfplay_game(onet, noise, us, maxply=cut if cut else None)
        Lg = int(nply[g]); sl = slice(off, off + Lg)
        assert Lg == r["nply"] and int(res[g]) == r["result"]
        for key in ("actions", "boards", "visits", "pis", "z"):
            assert np.array_equal(rec[key][sl], r[key]), f"game {g}: {key} differs from the oracle (L={L})"
        for key in tot:
            tot[key] += r["counters"][key]
        off += Lg
    assert (c["expansions"], c["terminal_hits"], c["depth_sum"], c["duplicate_leaves"], c["simulations"]) == \
        (tot["expansions"], tot["terminal_hits"], tot["depth_sum"], tot["dup_sims"], tot["sims"])
    assert c["simulations"] == S * c["plies"]
    # the point of the exercise: ceil(S / L) + 1 dependent evaluation batches per move instead of S + 1
    assert c["steps"] % (1 + -(-S // L)) == 0


def test_virtual_loss_batches_of_one_equal_the_sequential_kernel(monkeypatch):
    """AZ_VL_FORCE=1 routes L = 1 through the batched tree kernel: it must reproduce k_step bit for bit."""
    n, k, S, G = 9, 5, 40, 6
    sd = weights_from_fixture(n, "seeded")
    out = []
    for force in ("0", "1"):
        monkeypatch.setenv("AZ_VL_FORCE", force)
        e = az.Engine(n, k, S, 4, log_table=orc.numpy_log_table(S))
        e.load_weights(sd, 0)
        e.set_virtual_loss(1)
        c = e.selfplay(G, seed0=55, max_plies=8)
        out.append((e.records(), c))
        e.close()
    _same_records(out[0][0], out[1][0])
    for key in WORK:
        assert out[0][1][key] == out[1][1][key], key


def test_virtual_loss_arena_resnet_cache_and_lanes_together():
    """Everything at once: ResidualBlock net, two lanes, virtual-loss batches of 6, evaluation cache on -- the arena and a
    self-play episode against the oracle (whose search is the same with or without a cache)."""
    n, k, S, L = 9, 5, 45, 6
    a, b = synthetic_resnet_state_dict(n, 1), synthetic_resnet_state_dict(n, 2)
    e = az.Engine(n, k, S, 8, engines=2, model="resnet", log_table=orc.numpy_log_table(S))
    e.load_weights(a, 0); e.load_weights(b, 1)
    e.set_virtual_loss(L); e.set_eval_cache(1 << 15)
    r = e.arena(5, seed0=70, temperature_table=orc.arena_T_table(n * n))
    c = e.selfplay(7, seed0=900, max_plies=6)
    rec = e.records(); nply, _ = e.games()
    e.close()
    o = orc.Oracle(n, k, S, virtual_loss=L)
    oa = orc.Net(n, resnet_tensors=fold_resnet_state_dict(a)); ob = orc.Net(n, resnet_tensors=fold_resnet_state_dict(b))
    for g in range(5):
        ro = o.arena_game(oa, ob, g, np.random.RandomState(70 + g).random_sample(n * n))
        assert int(r["results"][g]) == ro["result"] and np.array_equal(r["actions"][g][:ro["nply"]], ro["actions"])
    off = 0
    for g in range(7):
        noise, us = orc.selfplay_tape(900 + g, n)
        ro = o.selfplay_game(oa, noise, us, maxply=6)
        sl = slice(off, off + int(nply[g]))
        for key in ("actions", "visits", "pis"):
            assert np.array_equal(rec[key][sl], ro[key]), f"game {g}: {key}"
        off += int(nply[g])
    assert c["cache_lookups"] > 0


def test_upgrade_switches_are_validated():
    e = az.Engine(5, 4, 20, 2, synthetic=True)
    with pytest.raises(az.AzError):
        e.set_virtual_loss(0)
    with pytest.raises(az.AzError):
        e.set_virtual_loss(33)
    e.set_subtree_reuse(True)
    with pytest.raises(az.AzError):
        e.set_virtual_loss(4)                     # not combinable with subtree reuse
    e.set_subtree_reuse(False)
    e.set_virtual_loss(4)
    with pytest.raises(az.AzError):
        e.set_subtree_reuse(True)
    e.selfplay_begin(2, seed0=1)
    with pytest.raises(az.AzError):
        e.set_virtual_loss(2)                     # not while an episode is open
    with pytest.raises(az.AzError):
        e.set_eval_cache(1024)
    e.selfplay_end()
    e.set_eval_cache(0)
    e.close()
