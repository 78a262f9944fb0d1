"""SelfPlayManager seam (self_play.py:24-159): the whole episode runs in the HIP engine, one engine per GPU,
games sharded over the ranks of torch.distributed when it is initialised, records exchanged once at episode end."""
from collections.abc import Callable

import numpy as np
import torch

from . import constants as _c
from . import parallel
from ._capi import AZ_AUG_REFERENCE4, Engine
from .controller import device_index, model_kind
from .mcts import numpy_log_table


def default_temperature_schedule(move: int) -> float:
    norm = 1 + _c.TEMPERATURE_BASELINE
    return (np.exp(-move / _c.TEMPERATURE_SCHEDULE_HALFTIME) + _c.TEMPERATURE_BASELINE) / norm


class SelfPlayManager:
    def __init__(self, controller, device, mcts_params: dict = None,
                 temperature_schedule: Callable[[int], float] = default_temperature_schedule,
                 concurrent_games: int = None, augmentation: int = AZ_AUG_REFERENCE4, seed: int = None,
                 engines_per_gpu: int = None, subtree_reuse: bool = False, gather_to: int = None,
                 eval_cache: int = 0, virtual_loss: int = 1, trunk: str = "f32", leaf_symmetry: bool = False):
        self.controller = controller
        self.device = device
        self.mcts_params = mcts_params or {"num_simulations": 100}
        self.temperature_schedule = temperature_schedule
        self.concurrent_games = concurrent_games or _c.CONCURRENT_GAMES
        self.augmentation = augmentation      # 4 = the reference's rotations (self_play.py:94-108), 8 = full dihedral group, 1 = none
        self.seed = seed
        self.engines_per_gpu = engines_per_gpu      # None: az_config.engines = 0, the library chooses
        self.subtree_reuse = subtree_reuse    # opt-in search upgrade (mcts.py:17-22 TODO); off = the reference's fresh root every move
        self.eval_cache = eval_cache          # opt-in: positions kept in the device evaluation cache (mcts.py:17,22 TODO); results unchanged
        self.virtual_loss = virtual_loss      # opt-in: leaves per search and evaluation batch (mcts.py:17-22 TODO); 1 = sequential like the reference
        self.trunk = trunk                    # opt-in: "bf16x3" / "f16x2" = fp32-emulating conv trunks on the 16-bit matrix cores (tolerance, not bit-exact)
        self.leaf_symmetry = leaf_symmetry    # opt-in: every net evaluation sees a pseudo-random dihedral symmetry of the position (README.md:61,82)
        self.gather_to = gather_to            # multi-rank: None = every rank receives all records (all-gather); r = only rank r does
        self.last_counters = None
        self._engine = None

    def _eng(self, n, k, slots):
        p = self.mcts_params
        key = (n, k, p.get("num_simulations", 100), slots, p.get("c_puct", _c.SELF_PLAY_EXPLORATION_CONSTANT),
               p.get("dirichlet_alpha", 0.3), p.get("dirichlet_weight", 0.25), model_kind(self.controller.net),
               self.virtual_loss, self.eval_cache, self.leaf_symmetry)
        if self._engine is None or self._engine_key != key:
            if self._engine is not None:
                self._engine.close()
            # lanes per GPU: the library's own rule (1 on small boards, one per 128 slots up to 4) unless the caller fixed it
            self._engine = Engine(n, k, key[2], slots, engines=self.engines_per_gpu or 0, c_puct=key[4], dirichlet_alpha=key[5],
                                  dirichlet_weight=key[6], device=device_index(self.device),
                                  log_table=numpy_log_table(key[2]), model=key[7])
            self._engine.set_virtual_loss(self.virtual_loss)
            self._engine.set_eval_cache(self.eval_cache)
            self._engine.set_leaf_symmetry(self.leaf_symmetry)
            self._engine_key = key
        return self._engine

    def generate_self_play(self, num_games: int, num_workers: int = None, flatten=False) -> list:
        """Returns list[(state f32 (4,n,n) CPU tensor, pi f32 (n,n) ndarray, z int)] in (game, ply, k) order
        (self_play.py:110-159; num_workers / flatten are accepted and unused like the reference's `flatten`)."""
        packed, total, eng, dev, n = self.generate_packed(num_games)
        aug = self.augmentation
        states = torch.empty((total * aug, 4, n, n), dtype=torch.float32, device=dev)
        pis = torch.empty((total * aug, n, n), dtype=torch.float32, device=dev)
        zs = torch.empty(total * aug, dtype=torch.float32, device=dev)
        if total:
            eng.examples_from_packed(packed.data_ptr(), total, aug, states.data_ptr(), pis.data_ptr(), zs.data_ptr())
        states, pis, zs = states.cpu(), pis.cpu().numpy(), zs.cpu().numpy().astype(np.int64)
        print(f"[SelfPlayManager] Collected {len(zs)} examples from {num_games} games.")
        return list(zip(states.unbind(0), list(pis), zs.tolist()))     # (tensor view, ndarray view, int) per example

    def generate_packed(self, num_games: int):
        """The same episode, but the result stays on the device as packed records (one per position, all ranks'
        records after the exchange): (uint8 tensor, record count, engine, device, n).  Feed it to
        device_replay.DeviceReplayBuffer.extend_packed to train without materialising Python tuples."""
        n = self.controller.net.board_size
        k = min(_c.WIN_LENGTH, n)
        rank, world = parallel.rank_world()
        per = (num_games + world - 1) // world
        lo, hi = min(rank * per, num_games), min((rank + 1) * per, num_games)
        mine = hi - lo
        seed0 = self.seed if self.seed is not None else int(np.random.randint(0, 2 ** 31 - 1))
        if self.seed is None:
            seed0 = parallel.broadcast_seed(seed0, torch.device("cuda", device_index(self.device)))
        dev = torch.device("cuda", device_index(self.device))
        eng = self._eng(n, k, max(1, min(self.concurrent_games, max(mine, 1))))
        eng.load_weights(self.controller.net.state_dict(), 0)
        eng.set_subtree_reuse(self.subtree_reuse)
        if eng.trunk_mode() != self.trunk:
            eng.set_trunk_mode(self.trunk)
        T = np.array([float(self.temperature_schedule(m)) for m in range(n * n + 1)], dtype=np.float64)
        if mine > 0:
            self.last_counters = eng.selfplay(mine, seed0=seed0 + lo, temperature_table=T)
        else:
            eng.last_records = 0
        packed, counts = parallel.gather_packed_records(eng, dev, dst=self.gather_to)
        total = int(sum(counts)) if (self.gather_to is None or rank == self.gather_to) else 0
        return packed, total, eng, dev, n
