#!/usr/bin/env python3
"""The reference's training loop (train.py:39-131) on the GPU path: self-play -> replay buffer -> AdamW steps ->
arena -> promote.  Everything expensive runs in the HIP engine; this file is glue and mirrors the reference's
episode structure so that the drop-in can be exercised end to end:

    python -m alphazero_piskvorky_amd.train --episodes 2 --games 64 --sims 50
"""
import argparse
import os
import tempfile
import time

import torch

from . import constants as C
from . import parallel
from .controller import NeuralNetworkController
from .evaluator import ModelEvaluator
from .games import Gomoku
from .net import GomokuNet
from .promoter import ModelPromoter
from .replay_buffer import ReplayBuffer
from .self_play import SelfPlayManager

PROMOTION_THRESHOLD = 0.55          # promoter.py:19, strict ">" with draws counted one half


def run(episodes, games, sims, eval_games, device="cuda:0", seed=0, model_dir=None, log=print, device_replay=False,
        subtree_reuse=False, trunk="f32", eval_sims=None):
    """eval_sims: simulations per move in the arena (evaluator.py:53-62 takes NUM_EVAL_SIMULATIONS = 200 whatever the
    self-play count is; main() passes that constant); None = as many as self-play, which keeps small test runs short.
    device_replay=True keeps the examples on the GPU from the episode-end gather to the optimizer step (packed records
    in a device ring, batches unpacked + augmented by az_examples_gather) instead of materialising Python tuples."""
    torch.manual_seed(seed)
    n = C.BOARD_SIZE
    rank, world = parallel.rank_world()
    candidate = NeuralNetworkController(GomokuNet(board_size=n), device=device)
    parallel.broadcast_module_(candidate.net)            # every rank starts from rank 0's initialisation
    model_dir = model_dir or tempfile.mkdtemp(prefix="az_models_")
    # multi-rank: every rank plays its shard of the games; the records go to rank 0 only (the reference has one trainer,
    # train.py:95-104), rank 0 takes the optimizer steps and its weights are broadcast; only rank 0 writes model_dir
    manager = SelfPlayManager(candidate, device, mcts_params={"num_simulations": sims, "c_puct": C.SELF_PLAY_EXPLORATION_CONSTANT},
                              seed=seed, subtree_reuse=subtree_reuse, gather_to=0 if world > 1 else None, trunk=trunk)
    evaluator = ModelEvaluator(game_class=Gomoku, print_games=False, device=device, seed=seed)
    promoter = ModelPromoter(model_dir, evaluator, lambda: GomokuNet(board_size=n), device, threshold=PROMOTION_THRESHOLD)
    buffer = ReplayBuffer(capacity=C.BUFFER_CAPACITY)
    ring = None
    history = []
    saved_eval_sims = C.NUM_EVAL_SIMULATIONS
    C.NUM_EVAL_SIMULATIONS = sims if eval_sims is None else eval_sims
    try:
        for ep in range(episodes):
            t0 = time.perf_counter()
            # disjoint seed ranges per episode and per use: game g of the episode's self-play draws from
            # RandomState(base + g), arena game g from RandomState(base + 500_000 + g)
            base = seed + 1_000_003 * ep
            manager.seed = base
            losses = []
            if device_replay:
                from .device_replay import DeviceReplayBuffer
                packed, records, eng, dev, _ = manager.generate_packed(games)
                data = range(records * manager.augmentation)
                if rank == 0:
                    if ring is None:
                        ring = DeviceReplayBuffer(eng, capacity=C.BUFFER_CAPACITY, aug=manager.augmentation, device=dev, seed=seed)
                    ring.extend_packed(packed, records)
                    for _ in range(C.BATCHES_PER_EPISODE):
                        losses.append(candidate.train_tensors(*ring.sample_batch(C.BATCH_SIZE), epochs=C.NUM_EPOCHS))
            else:
                data = manager.generate_self_play(num_games=games, num_workers=C.NUM_WORKERS)   # train.py:89-92
                if rank == 0:
                    buffer.extend(data)                                                          # train.py:95
                    for _ in range(C.BATCHES_PER_EPISODE):                                       # train.py:100-104
                        losses.append(candidate.train(buffer.sample_batch(C.BATCH_SIZE), epochs=C.NUM_EPOCHS))
            parallel.broadcast_module_(candidate.net)          # multi-rank: everybody continues with rank 0's weights
            evaluator.seed = base + 500_000
            win_rate, metrics, promoted = promoter.evaluate_and_maybe_promote(candidate, num_games=eval_games)   # train.py:114-119
            loss = losses[-1].get("loss") if losses else None
            history.append(dict(metrics, episode=ep, examples=len(data), loss=loss,
                                promoted=promoted, seconds=time.perf_counter() - t0))
            log(f"[train] episode {ep}: {len(data)} examples, loss {'n/a' if loss is None else format(loss, '.4f')}, "
                f"arena {metrics['wins']}/{metrics['losses']}/{metrics['draws']} -> {win_rate:.2%}"
                f"{' (promoted)' if promoted else ''}, {history[-1]['seconds']:.1f}s")
    finally:
        C.NUM_EVAL_SIMULATIONS = saved_eval_sims
    return history


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=C.NUM_EPISODES)
    ap.add_argument("--games", type=int, default=C.NUM_SELF_PLAY_GAMES)
    ap.add_argument("--sims", type=int, default=C.NUM_SELF_PLAY_SIMULATIONS)
    ap.add_argument("--eval-games", type=int, default=C.EVALUATION_GAMES)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--model-dir", default=None)
    ap.add_argument("--trunk", default="f32", choices=["f32", "bf16x3", "f16x2"], help="self-play conv trunk: f32 (bit-exact default) or an fp32-emulating trunk on the 16-bit matrix cores (opt-in: bf16x3, or the faster f16x2 with float16's range)")
    ap.add_argument("--subtree-reuse", action="store_true", help="self-play keeps the chosen child's subtree between plies (opt-in search upgrade)")
    ap.add_argument("--device-replay", action="store_true", help="keep examples on the GPU (packed ring + on-device batch unpacking)")
    a = ap.parse_args()
    # one process per GPU under torch.distributed.run: games and arena games are sharded over the ranks (self_play.py,
    # evaluator.py), the examples are gathered to rank 0, which trains, writes the checkpoints and broadcasts the weights
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as td
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        td.init_process_group("nccl", device_id=torch.device("cuda", local))
        a.device = f"cuda:{local}"
    run(a.episodes, a.games, a.sims, a.eval_games, a.device, model_dir=a.model_dir, device_replay=a.device_replay,
        subtree_reuse=a.subtree_reuse, trunk=a.trunk, eval_sims=C.NUM_EVAL_SIMULATIONS)
    if world > 1:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
