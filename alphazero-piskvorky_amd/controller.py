"""controller.py seam (controller.py:33-207): make_policy_value_fn evaluates on the GPU engine;
NeuralNetworkController keeps the reference's training surface (torch AdamW, soft-target CE + MSE)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import constants as _c
from ._capi import Engine
from .games import Gomoku


class PolicyValueFn:
    """Callable state -> (P float32[n,n], float v) (controller.py:39-53).  Carries the controller so that
    MCTS / SelfPlayManager can hand its weights to the engine instead of calling back into Python per leaf."""

    def __init__(self, controller):
        self.controller = controller
        self._engine = None
        self._version = None

    def _eng(self, n):
        ver = weights_version(self.controller.net)
        if self._engine is None or self._engine.n != n:
            self._engine = Engine(n, min(_c.WIN_LENGTH, n), 1, 1, device=device_index(self.controller.device),
                                  model=model_kind(self.controller.net))
            self._version = None
        if ver != self._version:
            self._engine.load_weights(self.controller.net.state_dict(), 0)
            self._version = ver
        return self._engine

    def __call__(self, state):
        if not isinstance(state, Gomoku):
            raise TypeError("policy_value_fn expects a Gomoku state")
        n = state.board_size
        _, P, v = self._eng(n).net_eval(state.cells[None], [state.player_code()], [state.last_index()])
        return P[0].reshape(n, n), float(v[0])


def make_policy_value_fn(controller):
    return PolicyValueFn(controller)


def model_kind(net):
    from .net import GomokuResNet
    return "resnet" if isinstance(net, GomokuResNet) else "plain"


def device_index(device):
    d = torch.device(device) if device is not None else torch.device("cuda")
    return d.index if d.index is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)


def weights_version(net):
    return tuple((p.data_ptr(), p._version) for p in net.state_dict().values())


class AlphaZeroDataset(torch.utils.data.Dataset):
    def __init__(self, examples):
        self.examples = examples

    def __len__(self):
        return len(self.examples)

    def __getitem__(self, i):
        s, p, z = self.examples[i]
        return s.float(), p, (z.float() if torch.is_tensor(z) else torch.tensor(z, dtype=torch.float32))


class NeuralNetworkController:
    def __init__(self, net, device=None, lr=None, batch_size=None):
        self.net = net
        self.device = device if device is not None else "cuda"
        self.batch_size = batch_size or _c.BATCH_SIZE
        self.net.to(self.device)
        self.optimizer = torch.optim.AdamW(self.net.parameters(), lr=lr or _c.LEARNING_RATE, weight_decay=1e-4)
        self.training_history = []

    def train_step(self, states, pis, zs):
        """One optimizer step (controller.py:100-131): loss = CE(soft pi) + MSE(z)."""
        self.net.train()
        states, pis, zs = states.to(self.device), pis.to(self.device), zs.to(self.device)
        logits, value = self.net(states)
        policy_loss = -(pis.flatten(1) * F.log_softmax(logits, dim=1)).sum(dim=1).mean()
        value_loss = F.mse_loss(value.squeeze(-1), zs)
        loss = policy_loss + value_loss
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return {"loss": loss.item(), "policy_loss": policy_loss.item(), "value_loss": value_loss.item()}

    def train(self, examples, epochs=1):
        n = self.net.board_size
        examples = list(examples)                       # sample_batch may hand over the deque itself
        for s, _, _ in examples[:1]:
            if tuple(s.shape) != (4, n, n):
                raise ValueError(f"example state has shape {tuple(s.shape)}, expected {(4, n, n)}")   # controller.py:136-138
        loader = torch.utils.data.DataLoader(AlphaZeroDataset(examples), batch_size=self.batch_size, shuffle=True)
        for _ in range(epochs):
            sums, batches = {}, 0
            for states, pis, zs in loader:
                out = self.train_step(states, torch.as_tensor(pis), zs)
                for k, v in out.items():
                    sums[k] = sums.get(k, 0.0) + v
                batches += 1
            self.training_history.append({k: v / max(batches, 1) for k, v in sums.items()})
        self.net.eval()
        return self.training_history[-1] if self.training_history else {}

    def train_tensors(self, states, pis, zs, epochs=1):
        """train() for a batch that is already on the device as tensors (device_replay.DeviceReplayBuffer.sample_batch):
        the same epochs of shuffled mini-batches of `batch_size` (controller.py:140-194), no Dataset / DataLoader hop."""
        n = self.net.board_size
        if tuple(states.shape[1:]) != (4, n, n):
            raise ValueError(f"example state has shape {tuple(states.shape[1:])}, expected {(4, n, n)}")
        total = states.shape[0]
        for _ in range(epochs):
            perm = torch.randperm(total, device=states.device)
            sums, batches = {}, 0
            for lo in range(0, total, self.batch_size):
                idx = perm[lo:lo + self.batch_size]
                out = self.train_step(states[idx], pis[idx], zs[idx])
                for k, v in out.items():
                    sums[k] = sums.get(k, 0.0) + v
                batches += 1
            self.training_history.append({k: v / max(batches, 1) for k, v in sums.items()})
        self.net.eval()
        return self.training_history[-1] if self.training_history else {}

    def save(self, path):
        torch.save(self.net.state_dict(), path)

    def load(self, path):
        self.net.load_state_dict(torch.load(path, map_location=self.device, weights_only=True))
