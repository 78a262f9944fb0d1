"""MCTS.run seam (mcts.py:87-183) over az_search: one position, the whole search on the GPU.

The reference's randomness comes from numpy's global legacy RNG in a fixed order per call --
np.random.dirichlet([alpha]*len(legal)) when add_root_noise, then one random_sample() inside
np.random.choice (mcts.py:114,177).  The shim makes the same two draws from the same global RNG and hands
them to the engine, so np.random.seed(s) reproduces the reference's choices."""
import numpy as np

from . import constants as _c
from ._capi import Engine
from .controller import PolicyValueFn, device_index, model_kind, weights_version


def numpy_log_table(S):
    """float32 log(N + 1e-8) exactly as mcts.py:161 evaluates it."""
    return np.log(np.arange(S + 1, dtype=np.float32) + 1e-8).astype(np.float32)


class MCTS:
    def __init__(self, policy_value_fn, num_simulations, c_puct, dirichlet_alpha=0.3, dirichlet_weight=0.25, virtual_loss=1):
        if not callable(policy_value_fn):
            raise TypeError("policy_value_fn must be callable: state -> (policy [n, n], value)")
        # make_policy_value_fn(controller): the leaves are evaluated inside the HIP engine.  Any other callable keeps the
        # reference's plugin seam (mcts.py:87-93): the tree stays on the GPU and every evaluation is one host round trip
        # (az_search_callback) -- compatible, not fast.
        self._external = not isinstance(policy_value_fn, PolicyValueFn)
        self.policy_value_fn = policy_value_fn
        self.num_simulations = num_simulations
        self.c_puct = c_puct
        self.dirichlet_alpha = dirichlet_alpha
        self.dirichlet_weight = dirichlet_weight
        self.virtual_loss = virtual_loss      # opt-in: leaves per evaluation batch (az_set_virtual_loss); 1 = the reference's loop
        self._engine = None
        self._version = None

    def _eng_external(self, n, k):
        if self._engine is None or (self._engine.n, self._engine.k) != (n, k):
            self._engine = Engine(n, k, self.num_simulations, 1, c_puct=self.c_puct, dirichlet_alpha=self.dirichlet_alpha,
                                  dirichlet_weight=self.dirichlet_weight, log_table=numpy_log_table(self.num_simulations))
        return self._engine

    def _run_external(self, root_state, temperature, noise, u):
        from .games import Gomoku
        n, k = root_state.board_size, root_state.win_length

        def evaluate(cells, player, last):
            g = Gomoku(n, k)
            g.cells = cells
            g.current_player = _c.X if player == 1 else _c.O
            g.last_action = None if last < 0 else (last // n, last % n)
            return self.policy_value_fn(g)

        return self._eng_external(n, k).search_callback(root_state.cells, root_state.player_code(), root_state.last_index(),
                                                        float(temperature), evaluate, noise, u)

    def _eng(self, n, k):
        ctrl = self.policy_value_fn.controller
        if self._engine is None or (self._engine.n, self._engine.k) != (n, k):
            self._engine = Engine(n, k, self.num_simulations, 1, c_puct=self.c_puct,
                                  dirichlet_alpha=self.dirichlet_alpha, dirichlet_weight=self.dirichlet_weight,
                                  device=device_index(ctrl.device), log_table=numpy_log_table(self.num_simulations),
                                  model=model_kind(ctrl.net))
            self._engine.set_virtual_loss(self.virtual_loss)
            self._version = None
        ver = weights_version(ctrl.net)
        if ver != self._version:
            self._engine.load_weights(ctrl.net.state_dict(), 0)
            self._version = ver
        return self._engine

    def run(self, root_state, temperature, add_root_noise=False):
        n = root_state.board_size
        legal = int((root_state.cells == 0).sum())
        if legal == 0:
            return np.zeros((n, n), dtype=np.float32), None            # mcts.py:152-153
        noise = np.random.dirichlet([self.dirichlet_alpha] * legal) if add_root_noise else None
        u = np.random.random_sample()
        if self._external:
            r = self._run_external(root_state, temperature, noise, u)
        else:
            eng = self._eng(n, root_state.win_length)
            r = eng.search(root_state.cells, root_state.player_code(), root_state.last_index(), float(temperature), noise, u)
        a = r["action"]
        self.last_visits = r["N"].reshape(n, n)
        return r["pi"].reshape(n, n), (a // n, a % n)
