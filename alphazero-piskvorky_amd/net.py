"""GomokuNet as a torch module with the reference's state_dict ABI (net.py:37-53): the optimizer step stays in
PyTorch-ROCm; inference during search runs in the HIP engine from the same tensors (az_load_weights)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import constants as _c


class GomokuNet(nn.Module):
    TRUNK = ((4, 32), (32, 64), (64, 128))

    def __init__(self, board_size=None, device=None):
        super().__init__()
        self.board_size = _c.BOARD_SIZE if board_size is None else board_size
        self.device = device if device else torch.device("cpu")
        cells = self.board_size ** 2
        for i, (cin, cout) in enumerate(self.TRUNK, start=1):
            setattr(self, f"conv{i}", nn.Conv2d(cin, cout, kernel_size=3, padding=1))
        self.policy_conv = nn.Conv2d(128, 4, kernel_size=1)
        self.policy_fc = nn.Linear(4 * cells, cells)
        self.value_conv = nn.Conv2d(128, 2, kernel_size=1)
        self.value_fc1 = nn.Linear(2 * cells, 64)
        self.value_fc2 = nn.Linear(64, 1)

    def forward(self, x):
        h = x
        for i in range(1, 4):
            h = F.relu(getattr(self, f"conv{i}")(h))
        p = F.relu(self.policy_conv(h)).flatten(1)
        v = F.relu(self.value_conv(h)).flatten(1)
        return self.policy_fc(p), torch.tanh(self.value_fc2(F.relu(self.value_fc1(v))))
