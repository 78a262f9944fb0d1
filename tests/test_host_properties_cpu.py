"""Property tests (hypothesis) of the host-side Gomoku class against the oracle's rules restatement, and of the
C-ABI host RNG against numpy for arbitrary seeds."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import oracle as orc
from alphazero_piskvorky_amd import _capi, games

CODE = {None: 0, "X": 1, "O": 2, "D": 3}


@settings(max_examples=120, deadline=None)
@given(n=st.integers(3, 15), k=st.integers(3, 5), seed=st.integers(0, 2 ** 31 - 1))
def test_random_playouts_agree_with_the_oracle_rules(n, k, seed):
    k = min(k, n)
    rs = np.random.RandomState(seed)
    order = list(rs.permutation(n * n))
    s = games.Gomoku(n, k)
    played = []
    for a in order:
        if s.is_terminal():
            break
        s = s.apply_action((int(a) // n, int(a) % n))
        played.append(int(a))
    o = orc.Oracle(n, k, 1)
    rc, term, board, pl, res = o.replay(played)
    assert rc == 0 and not term.any()
    assert np.array_equal(board, s.cells)
    assert res == CODE[s.get_game_result()]
    assert pl == s.player_code()
    # encode agrees with the oracle's planes
    planes = o.encode(board, pl, played[-1] if played else -1)
    assert np.array_equal(planes, s.encode("cpu").numpy())
    # legal actions = empties in row-major order (games.py:35-47)
    assert [r * n + c for r, c in s.get_legal_actions()] == [int(i) for i in np.flatnonzero(board == 0)]


@settings(max_examples=40, deadline=None)
@given(seed=st.integers(0, 2 ** 32 - 1), n=st.sampled_from([3, 5, 9]), plies=st.integers(1, 4))
def test_host_rng_any_seed(seed, n, plies):
    nn = n * n
    noise, u = _capi.rng_selfplay_tape(seed, n, max_plies=plies)
    rs = np.random.RandomState(seed)
    off = 0
    for m in range(plies):
        assert np.array_equal(rs.dirichlet([0.3] * (nn - m)), noise[off:off + nn - m])
        assert rs.random_sample() == u[m]
        off += nn - m
