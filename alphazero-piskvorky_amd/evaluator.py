"""ModelEvaluator seam (evaluator.py:14-122): all arena games run concurrently in the HIP engine with two
weight sets (candidate = X = slot 0, baseline = O = slot 1, odd games start with O)."""
import numpy as np
import torch

from . import constants as _c
from ._capi import Engine
from .controller import device_index, model_kind
from .mcts import numpy_log_table


def temperature_schedule(move: int) -> float:
    return _c.EVAL_TEMPERATURE * np.exp(-move / _c.EVAL_TEMPERATURE_SCHEDULE_HALFTIME)


class ModelEvaluator:
    def __init__(self, game_class=None, print_games=False, device=None, seed=None):
        self.game_class = game_class
        self.print_games = print_games
        self.device = device if device is not None else torch.device("cuda")
        self.seed = seed
        self._engine = None

    def evaluate(self, candidate_controller, baseline_controller, num_games=20, debug=False):
        candidate_controller.net.eval()
        baseline_controller.net.eval()
        n = candidate_controller.net.board_size
        k = min(_c.WIN_LENGTH, n)
        S = _c.NUM_EVAL_SIMULATIONS
        key = (n, k, S, num_games, model_kind(candidate_controller.net))
        if self._engine is None or self._key != key:
            if self._engine is not None:
                self._engine.close()
            self._engine = Engine(n, k, S, max(1, min(num_games, _c.CONCURRENT_GAMES)), c_puct=_c.EVAL_EXPLORATION_CONSTANT,
                                  device=device_index(self.device), log_table=numpy_log_table(S), model=key[4])
            self._key = key
        eng = self._engine
        eng.load_weights(candidate_controller.net.state_dict(), 0)
        eng.load_weights(baseline_controller.net.state_dict(), 1)
        T = np.array([float(temperature_schedule(m)) for m in range(n * n + 2)], dtype=np.float64)
        seed0 = self.seed if self.seed is not None else int(np.random.randint(0, 2 ** 31 - 1))
        r = eng.arena(num_games, seed0=seed0, temperature_table=T)
        if debug:
            print(f"[Evaluator]: Candidate Win Rate: {r['win_rate']:.2%} (W:{r['wins']} L:{r['losses']} D:{r['draws']})")
        self.last_result = r
        return r["win_rate"], {"wins": r["wins"], "losses": r["losses"], "draws": r["draws"], "total": r["total"],
                               "win_rate": r["win_rate"]}
