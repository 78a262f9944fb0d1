"""ModelLoader with the reference's behaviour (model_loader.py:9-56): newest models/*.pt by ctime, or a freshly
initialised net written to disk when the directory is empty.  Checkpoints are plain state_dicts (weights_only)."""
import os
from datetime import datetime

import torch

from . import constants as _c
from .net import GomokuNet


def save_state_dict_atomic(state_dict, path):
    """Write to a temporary name in the same directory, then rename: a reader that lists the directory never sees a
    half-written checkpoint."""
    tmp = f"{path}.tmp{os.getpid()}"
    torch.save(state_dict, tmp)
    os.replace(tmp, path)


class ModelLoader:
    def __init__(self, model_dir=None, net_class=GomokuNet):
        self.model_dir = model_dir or _c.MODEL_DIR
        self.net_class = net_class
        os.makedirs(self.model_dir, exist_ok=True)
        self.best_path = self._find_latest_model()

    def get_best_model(self):
        net = self.net_class()
        if self.best_path is None:
            path = os.path.join(self.model_dir, f"model_{datetime.now().strftime('%Y%m%d_%H%M%S')}.pt")
            save_state_dict_atomic(net.state_dict(), path)            # model_loader.py:29-35
            self.best_path = path
            return net.float()
        net.load_state_dict(torch.load(self.best_path, map_location="cpu", weights_only=True))
        return net.float()

    def _find_latest_model(self):
        models = [f for f in os.listdir(self.model_dir) if f.endswith(".pt")]
        if not models:
            return None
        return os.path.join(self.model_dir, max(models, key=lambda f: os.path.getctime(os.path.join(self.model_dir, f))))
