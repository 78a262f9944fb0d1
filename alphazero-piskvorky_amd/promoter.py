"""ModelPromoter with the reference's signature (promoter.py:18-62): arena against the newest checkpoint, save the
candidate as model_<timestamp>.pt when win_rate > threshold (strict, draws count one half)."""
import os
from datetime import datetime

import torch

from . import parallel
from .controller import NeuralNetworkController
from .model_loader import ModelLoader, save_state_dict_atomic


class ModelPromoter:
    def __init__(self, model_dir, evaluator, net_class, device, threshold=0.55):
        self.model_dir = model_dir
        self.evaluator = evaluator
        self.net_class = net_class
        self.threshold = threshold
        self.device = device
        self.best_path = None
        os.makedirs(model_dir, exist_ok=True)      # harmless on every rank; nothing else is written off rank 0

    def evaluate_and_maybe_promote(self, candidate_controller, num_games=20, metadata=None, debug=False):
        # multi-rank (one process per GPU): only rank 0 touches model_dir; the baseline it loads (or creates) is
        # broadcast, and since the arena tally is all-reduced every rank takes the same decision below
        rank, _ = parallel.rank_world()
        baseline_net = ModelLoader(self.model_dir, self.net_class).get_best_model() if rank == 0 else self.net_class().float()
        baseline = NeuralNetworkController(baseline_net, device=self.device)
        parallel.broadcast_module_(baseline.net)
        win_rate, metrics = self.evaluator.evaluate(candidate_controller, baseline, num_games=num_games, debug=debug)
        was_promoted = win_rate > self.threshold                       # promoter.py:47
        if was_promoted:
            path = os.path.join(self.model_dir, f"model_{datetime.now().strftime('%Y%m%d_%H%M%S_%f')}.pt")
            if rank == 0:
                save_state_dict_atomic(candidate_controller.net.state_dict(), path)
            self.best_path = path
            print(f"[Promoter]: promoted new model with win rate {win_rate:.2%}: {path}")
            if metadata:
                print("Metadata:", metadata)
        else:
            print(f"[Promoter]: candidate rejected (win rate: {win_rate:.2%})")
        return win_rate, metrics, was_promoted
