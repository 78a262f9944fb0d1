"""ModelEvaluator seam (evaluator.py:14-122): all arena games run concurrently in the HIP engine with two
weight sets (candidate = X = slot 0, baseline = O = slot 1, odd games start with O).  With torch.distributed
initialised the games are split over the ranks in even-aligned blocks and the W/L/D tally is all-reduced, so every
rank returns the same result (and takes the same promotion decision)."""
import numpy as np
import torch

from . import constants as _c
from . import parallel
from ._capi import Engine
from .controller import device_index, model_kind
from .mcts import numpy_log_table


def temperature_schedule(move: int) -> float:
    return _c.EVAL_TEMPERATURE * np.exp(-move / _c.EVAL_TEMPERATURE_SCHEDULE_HALFTIME)


class ModelEvaluator:
    def __init__(self, game_class=None, print_games=False, device=None, seed=None, virtual_loss=1, eval_cache=0):
        self.game_class = game_class
        self.print_games = print_games
        self.device = device if device is not None else torch.device("cuda")
        self.seed = seed
        self.virtual_loss = virtual_loss      # opt-in search upgrades (mcts.py:17-22 TODO), see include/az_engine.h
        self.eval_cache = eval_cache
        self._engine = None

    def evaluate(self, candidate_controller, baseline_controller, num_games=20, debug=False):
        candidate_controller.net.eval()
        baseline_controller.net.eval()
        n = candidate_controller.net.board_size
        k = min(_c.WIN_LENGTH, n)
        S = _c.NUM_EVAL_SIMULATIONS
        import torch.distributed as td
        rank, world = (td.get_rank(), td.get_world_size()) if td.is_available() and td.is_initialized() else (0, 1)
        lo, hi = parallel.arena_block(num_games, rank, world)
        mine = hi - lo
        dev = torch.device("cuda", device_index(self.device))
        key = (n, k, S, max(mine, 1), model_kind(candidate_controller.net), self.virtual_loss, self.eval_cache)
        if self._engine is None or self._key != key:
            if self._engine is not None:
                self._engine.close()
            self._engine = Engine(n, k, S, max(1, min(mine, _c.CONCURRENT_GAMES)), c_puct=_c.EVAL_EXPLORATION_CONSTANT,
                                  device=device_index(self.device), log_table=numpy_log_table(S), model=key[4])
            self._engine.set_virtual_loss(self.virtual_loss)
            self._engine.set_eval_cache(self.eval_cache)
            self._key = key
        eng = self._engine
        eng.load_weights(candidate_controller.net.state_dict(), 0)
        eng.load_weights(baseline_controller.net.state_dict(), 1)
        T = np.array([float(temperature_schedule(m)) for m in range(n * n + 2)], dtype=np.float64)
        seed0 = self.seed if self.seed is not None else int(np.random.randint(0, 2 ** 31 - 1))
        if self.seed is None:
            seed0 = parallel.broadcast_seed(seed0, dev)
        r = eng.arena(mine, seed0=seed0 + lo, temperature_table=T) if mine > 0 else {"wins": 0, "losses": 0, "draws": 0, "total": 0, "win_rate": 0.0}
        if world > 1:
            w, l, d = parallel.all_reduce_tally(r["wins"], r["losses"], r["draws"], dev, engine=eng)
            tot = w + l + d
            r = dict(r, wins=w, losses=l, draws=d, total=tot, win_rate=(w + 0.5 * d) / tot if tot else 0.0)   # evaluator.py:106-109
        if debug:
            print(f"[Evaluator]: Candidate Win Rate: {r['win_rate']:.2%} (W:{r['wins']} L:{r['losses']} D:{r['draws']})")
        self.last_result = r
        return r["win_rate"], {"wins": r["wins"], "losses": r["losses"], "draws": r["draws"], "total": r["total"],
                               "win_rate": r["win_rate"]}
