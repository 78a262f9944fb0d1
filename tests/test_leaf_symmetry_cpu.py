"""CPU checks of the oracle's restatement of random-symmetry leaf evaluation (include/az_engine.h az_set_leaf_symmetry; SURVEY
8f-2's optional half).  The reference has no such code (games.py:183-197 rot90 / flip are unused), so the rule is pinned to
numpy's own rot90 / fliplr -- the functions the reference's augmentation uses (self_play.py:103-105)."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.util import build_weights


def sym_np(t, x):
    """dihedral symmetry t of a 2-D array: t < 4 = np.rot90 t times, t >= 4 = rot90(fliplr(x), t - 4)"""
    return np.rot90(x, t) if t < 4 else np.rot90(np.fliplr(x), t - 4)


def inv(t):
    return (4 - t) % 4 if t < 4 else t


@pytest.mark.parametrize("n", [5, 9])
def test_symmetric_evaluation_equals_numpy_rot90_fliplr(n):
    sd = build_weights(n)
    net = orc.Net(n, sd)
    o = orc.Oracle(n, 4 if n == 5 else 5, 1)
    rs = np.random.RandomState(n)
    for trial in range(6):
        stones = int(rs.randint(1, n * n // 2))
        board = np.zeros(n * n, np.uint8)
        cells = rs.permutation(n * n)[:stones]
        board[cells[0::2]] = 1; board[cells[1::2]] = 2
        planes = o.encode(board, 1 + (stones & 1), int(cells[-1]))
        for t in range(8):
            shown = np.stack([sym_np(t, planes[ch]) for ch in range(4)])          # what the net sees
            l_img, _, v_img = net.eval(np.ascontiguousarray(shown))
            want = sym_np(inv(t), l_img.reshape(n, n)).reshape(-1)                # policy brought back to board order
            got, v = net.eval_sym(planes, t)
            assert np.array_equal(got, np.ascontiguousarray(want)), f"symmetry {t}"
            assert np.float32(v) == np.float32(v_img)
        l0, _, v0 = net.eval(planes)
        g0, vv = net.eval_sym(planes, 0)
        assert np.array_equal(g0, l0) and np.float32(vv) == np.float32(v0)


def test_symmetry_hash_covers_all_eight_and_is_stable():
    from oracle.oracle import lib
    seen = {lib().orc_leaf_sym_of(g, p, i) for g in range(4) for p in range(6) for i in range(20)}
    assert seen == set(range(8))
    assert [lib().orc_leaf_sym_of(3, 7, i) for i in range(6)] == [lib().orc_leaf_sym_of(3, 7, i) for i in range(6)]


def test_search_with_leaf_symmetry_keeps_the_search_invariants():
    n, k, S = 5, 4, 60
    net = orc.Net(n, build_weights(n))
    off = orc.Oracle(n, k, S)
    on = orc.Oracle(n, k, S, leaf_sym=True)
    board = np.zeros(n * n, np.uint8); board[[6, 7]] = [1, 2]
    noise = np.random.RandomState(1).dirichlet([0.3] * (n * n - 2))
    a = off.search(net, board, 1, 7, 1.0, noise, 0.3)
    b = on.search(net, board, 1, 7, 1.0, noise, 0.3, game=5)
    c = on.search(net, board, 1, 7, 1.0, noise, 0.3, game=5)
    assert int(b["N"].sum()) == S and int(a["N"].sum()) == S
    assert np.array_equal(b["N"], c["N"]) and np.array_equal(b["pi"], c["pi"])        # reproducible
    assert abs(float(b["pi"].sum()) - 1.0) < 1e-5 and (b["N"][[6, 7]] == 0).all()
    assert not np.array_equal(a["P"], b["P"]) or not np.array_equal(a["N"], b["N"])    # the option does something
