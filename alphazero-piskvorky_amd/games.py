"""Host-side Gomoku state with the reference's public surface (games.py:20-243), used by the shims that
take a state object (MCTS.run, make_policy_value_fn, ModelEvaluator's game_class).  The device keeps its own
bit-planed boards; this class only carries positions across the Python boundary."""
import numpy as np
import torch

from . import constants as _c

_DIRS = ((0, 1), (1, 0), (1, 1), (1, -1))


class Gomoku:
    def __init__(self, board_size=None, win_length=None):
        board_size = _c.BOARD_SIZE if board_size is None else board_size
        if not isinstance(board_size, int):
            raise ValueError("Board size must be an integer.")
        self.board_size = board_size
        self.win_length = _c.WIN_LENGTH if win_length is None else win_length
        self.cells = np.zeros(board_size * board_size, dtype=np.uint8)   # 0 empty, 1 X, 2 O
        self.current_player = _c.X
        self.winner = None
        self.last_action = None

    # -- views ------------------------------------------------------------------------------
    @property
    def board(self):
        sym = {0: None, 1: _c.X, 2: _c.O}
        n = self.board_size
        return [[sym[int(self.cells[r * n + c])] for c in range(n)] for r in range(n)]

    def player_code(self, player=None):
        return 1 if (self.current_player if player is None else player) == _c.X else 2

    def last_index(self):
        return -1 if self.last_action is None else self.last_action[0] * self.board_size + self.last_action[1]

    # -- rules ------------------------------------------------------------------------------
    def get_legal_actions(self):
        n = self.board_size
        return [(int(i // n), int(i % n)) for i in np.flatnonzero(self.cells == 0)]   # row-major (games.py:42-47)

    def get_other_player(self, player):
        if player not in (_c.X, _c.O):
            raise ValueError(f"Invalid player: {player}. Must be 'X' or 'O'.")
        return _c.O if player == _c.X else _c.X

    def apply_action(self, action):
        r, c = action
        idx = r * self.board_size + c
        if self.cells[idx] != 0:
            raise ValueError("Invalid move")                     # games.py:76-77
        nxt = self.clone()
        nxt.cells[idx] = self.player_code()
        nxt.current_player = self.get_other_player(self.current_player)
        nxt.last_action = (int(r), int(c))
        return nxt

    def encode(self, device="cpu"):
        n = self.board_size
        me = self.player_code()
        planes = np.zeros((4, n, n), dtype=np.float32)
        grid = self.cells.reshape(n, n)
        planes[0] = grid == me
        planes[1] = (grid != me) & (grid != 0)
        if self.last_action is not None:
            planes[2, self.last_action[0], self.last_action[1]] = 1.0
        return torch.from_numpy(planes).to(device)

    def _line_from(self, r, c, dr, dc, who):
        n = self.board_size
        for i in range(self.win_length):
            rr, cc = r + i * dr, c + i * dc
            if not (0 <= rr < n and 0 <= cc < n) or self.cells[rr * n + cc] != who:
                return False
        return True

    def is_terminal(self):
        if self.winner is not None:
            return True
        n = self.board_size
        for idx in np.flatnonzero(self.cells):                  # row-major scan, games.py:144-158
            r, c = divmod(int(idx), n)
            who = int(self.cells[idx])
            if any(self._line_from(r, c, dr, dc, who) for dr, dc in _DIRS):
                self.winner = _c.X if who == 1 else _c.O
                return True
        if not (self.cells == 0).any():
            self.winner = _c.DRAW
            return True
        return False

    def get_game_result(self):
        return self.winner if self.is_terminal() else None

    # -- copies -----------------------------------------------------------------------------
    def clone(self):
        g = Gomoku(self.board_size, self.win_length)
        g.cells = self.cells.copy()
        g.current_player = self.current_player
        g.winner = self.winner
        g.last_action = None if self.last_action is None else tuple(self.last_action)
        return g

    def rot90(self):
        g = self.clone()
        n = self.board_size
        g.cells = np.rot90(self.cells.reshape(n, n), -1).reshape(-1).copy()   # clockwise, games.py:183-189
        return g

    def flip(self):
        g = self.clone()
        n = self.board_size
        g.cells = self.cells.reshape(n, n)[:, ::-1].reshape(-1).copy()
        return g

    def __repr__(self):
        rows = [" | ".join(x if x is not None else " " for x in row) for row in self.board]
        return "Gomoku(\n" + "\n".join(rows) + "\n)"
