"""Gomoku rules on the HIP path (games.py:64-82,133-179,212-227): the device win test `wins_through` and the board
bit-planes, through the C-ABI (az_rules_replay, az_search), against the Python reference's golden vectors and the oracle.

Bar: everything here is integer work -> bit-exact.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc
from tests.util import SIZES, load

import alphazero_piskvorky_amd as az


def _engine(n, k, S=8, **kw):
    return az.Engine(n, k, S, 4, synthetic=True, log_table=orc.numpy_log_table(S), **kw)


@pytest.mark.parametrize("n,k", SIZES)
def test_rules_replay_on_device_vs_reference_golden(n, k):
    """G1 fixtures (random play-outs of the Python reference + its crafted overline / corner anti-diagonal / no-wrap
    cases) replayed by the device kernel: outcome, terminal flags before every move, final board."""
    z = load(f"rules_{n}x{k}.npz")
    e = _engine(n, k)
    o = orc.Oracle(n, k, 1)
    r = e.rules_replay(z["actions"])
    assert np.array_equal(r["results"], z["result"].astype(np.int32))
    assert (r["first_illegal"] == -1).all()
    for g in range(len(z["nply"])):
        m = int(z["nply"][g])
        assert not r["term_before"][g, :m].any(), "is_terminal must be False before every played move"
        _, _, board, pl, _ = o.replay(z["actions"][g, :m])
        assert np.array_equal(r["boards"][g], board) and int(r["players"][g]) == pl
    c = e.rules_replay(z["crafted_actions"])
    assert np.array_equal(c["results"], z["crafted_result"].astype(np.int32)), "crafted overline / corner / no-wrap cases"
    assert (c["first_illegal"] == -1).all()
    e.close()


def _line_cases(n, k):
    """Crafted sequences per direction: a k-line touching every edge/corner, an overline (k+1 completed in the middle),
    and 'lines' that only exist if the board wrapped around an edge.  O plays far-away filler cells."""
    cases = []
    dirs = [(0, 1), (1, 0), (1, 1), (1, -1)]

    def seq_of(xs):
        taken = set(xs)
        fill = [c for c in range(n * n - 1, -1, -1) if c not in taken]
        # filler cells for O chosen so that O never builds k in a row: alternate two far rows/columns
        os_ = []
        for c in fill:
            r, q = divmod(c, n)
            if (r + 2 * q) % 5 == 0:              # sparse pattern, no 2 adjacent in any direction
                os_.append(c)
            if len(os_) >= len(xs) - 1:
                break
        s = []
        for i, x in enumerate(xs):
            s.append(x)
            if i < len(xs) - 1:
                s.append(os_[i])
        return s

    for dr, dc in dirs:
        for r0 in (0, n - k if dr else n - 1):
            for c0 in ((0, n - k) if dc == 1 else ((k - 1, n - 1) if dc == -1 else (0, n - 1))):
                cells = [(r0 + i * dr, c0 + i * dc) for i in range(k)]
                if all(0 <= r < n and 0 <= c < n for r, c in cells):
                    xs = [r * n + c for r, c in cells]
                    cases.append(seq_of(xs))                           # completed at an end
                    cases.append(seq_of(xs[:k // 2] + xs[k // 2 + 1:] + [xs[k // 2]]))   # completed in the middle
        if n >= k + 1:
            # overline: k+1 cells, the middle one last
            r0, c0 = (0 if dr == 0 else 1), (1 if dc >= 0 else n - 2)
            cells = [(r0 + i * dr, c0 + i * dc) for i in range(k + 1)]
            if all(0 <= r < n and 0 <= c < n for r, c in cells):
                xs = [r * n + c for r, c in cells]
                mid = xs.pop(k // 2)
                cases.append(seq_of(xs + [mid]))
    # wrap-around: consecutive INDICES across a row end (horizontal), and diagonals stepping over the left/right edge
    for start in (n - 2, n - 1, 2 * n - 3):
        xs = [start + i for i in range(k)]
        if len({x // n for x in xs}) > 1:
            cases.append(seq_of(xs))
    for start, step in ((n - 2, n + 1), (1, n - 1), (n + 1, n - 1)):
        xs = [start + i * step for i in range(k)]
        cols = [x % n for x in xs]
        if max(xs) < n * n and any(abs(cols[i + 1] - cols[i]) != 1 for i in range(k - 1)):
            cases.append(seq_of(xs))
    return cases


@pytest.mark.parametrize("n,k", [(5, 4), (9, 5), (15, 5), (7, 3), (15, 6)])
def test_rules_edge_corner_overline_and_wrap_cases_vs_oracle(n, k):
    cases = _line_cases(n, k)
    L = max(len(c) for c in cases)
    acts = -np.ones((len(cases), L), np.int16)
    for i, c in enumerate(cases):
        acts[i, :len(c)] = c
    e = _engine(n, k)
    o = orc.Oracle(n, k, 1)
    r = e.rules_replay(acts)
    wins = 0
    for i, c in enumerate(cases):
        rc, term, board, pl, res = o.replay(c)
        assert rc == 0
        assert int(r["results"][i]) == res, f"case {i}: {c}"
        assert np.array_equal(r["term_before"][i, :len(c)], term.astype(bool)) and np.array_equal(r["boards"][i], board)
        wins += res == 1
    assert 0 < wins < len(cases)            # both winning lines and non-lines (wrap cases) are present
    e.close()


@pytest.mark.parametrize("n", list(range(3, 16)))
def test_rules_random_playouts_every_size_vs_oracle(n):
    """Random legal play-outs continued to a FULL board (past the end of the game: the winner must stay the first one,
    games.py:140-141,206), every board size, several win lengths; plus an illegal move."""
    rs = np.random.RandomState(100 + n)
    nn = n * n
    for k in sorted({min(n, 3), min(n, 4), min(n, 5)}):
        G = 24
        acts = np.stack([rs.permutation(nn) for _ in range(G)]).astype(np.int16)
        e = _engine(n, k)
        o = orc.Oracle(n, k, 1)
        r = e.rules_replay(acts)
        for g in range(G):
            rc, term, board, pl, res = o.replay(acts[g])
            assert rc == 0 and int(r["results"][g]) == res and res != 0
            assert np.array_equal(r["term_before"][g], term.astype(bool))
            assert np.array_equal(r["boards"][g], board) and int(r["players"][g]) == pl
        bad = acts[:2].copy()
        bad[0, 5] = bad[0, 2]                    # occupied cell: games.py:76-77 ValueError("Invalid move")
        bad[1, 0] = nn                           # out of range
        rb = e.rules_replay(bad)
        assert list(rb["first_illegal"]) == [5, 0]
        assert o.replay(bad[0][:6])[0] == -1
        e.close()


@pytest.mark.parametrize("n,k", SIZES)
def test_search_scores_crafted_terminal_children_like_the_oracle(n, k):
    """One move before each crafted terminal case: MCTS.run from that position, with a root-noise vector that puts its
    whole mass on the crafted last move, so the search is certain to try it.  A child that completes a line is a
    terminal leaf worth -1 for the side to move there, i.e. +1 per visit on the parent's edge (mcts.py:132-134,141):
    W == N on that edge; a move that only 'completes' a line across the board edge must not score like that."""
    z = load(f"rules_{n}x{k}.npz")
    cases = [list(s[s >= 0]) for s in z["crafted_actions"]] + _line_cases(n, k)
    S = 96
    e = _engine(n, k, S)
    o = orc.Oracle(n, k, S, synthetic=True)
    seen_win = seen_nonwin = 0
    for seq in cases:
        rc, term, board, pl, res = o.replay(seq[:-1])
        assert rc == 0 and res == 0
        last_move = int(seq[-1])
        full = o.replay(seq)[4]
        legal = np.flatnonzero(board == 0)
        noise = (legal == last_move).astype(np.float64)
        args = (board, pl, int(seq[-2]), 1.0, noise, 0.5)
        r = e.search(*args)
        ro = o.search(None, *args)
        assert np.array_equal(r["N"], ro["N"]) and np.array_equal(r["W"], ro["W"]) and r["action"] == ro["action"]
        assert np.array_equal(r["pi"], ro["pi"]) and np.array_equal(r["P"], ro["P"])
        assert r["N"][last_move] > 0
        if full == pl:
            assert r["W"][last_move] == float(r["N"][last_move]), "a winning move is +1 on every visit"
            seen_win += 1
        else:
            assert r["W"][last_move] != float(r["N"][last_move]), "not a line: must not be scored as a win"
            seen_nonwin += 1
    assert seen_win >= 8 and seen_nonwin >= 2
    e.close()
