"""ModelPromoter with the reference's signature (promoter.py:18-62): arena against the newest checkpoint, save the
candidate as model_<timestamp>.pt when win_rate > threshold (strict, draws count one half)."""
import os
from datetime import datetime

import torch

from .controller import NeuralNetworkController
from .model_loader import ModelLoader


class ModelPromoter:
    def __init__(self, model_dir, evaluator, net_class, device, threshold=0.55):
        self.model_dir = model_dir
        self.evaluator = evaluator
        self.net_class = net_class
        self.threshold = threshold
        self.device = device
        self.best_path = None
        os.makedirs(model_dir, exist_ok=True)

    def evaluate_and_maybe_promote(self, candidate_controller, num_games=20, metadata=None, debug=False):
        baseline_net = ModelLoader(self.model_dir, self.net_class).get_best_model()
        baseline = NeuralNetworkController(baseline_net, device=self.device)
        win_rate, metrics = self.evaluator.evaluate(candidate_controller, baseline, num_games=num_games, debug=debug)
        was_promoted = win_rate > self.threshold                       # promoter.py:47
        if was_promoted:
            path = os.path.join(self.model_dir, f"model_{datetime.now().strftime('%Y%m%d_%H%M%S_%f')}.pt")
            torch.save(candidate_controller.net.state_dict(), path)
            self.best_path = path
            print(f"[Promoter]: promoted new model with win rate {win_rate:.2%}: {path}")
            if metadata:
                print("Metadata:", metadata)
        else:
            print(f"[Promoter]: candidate rejected (win rate: {win_rate:.2%})")
        return win_rate, metrics, was_promoted
