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


class ResidualBlock(nn.Module):
    """conv-BN-ReLU-conv-BN, + skip, ReLU (legacy/resnet/example.py:9-27); convs carry no bias in the reference's
    checkpoints (alphazero/models/old/model_20250728_*.pt)."""

    def __init__(self, channels):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, channels, kernel_size=3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(channels)
        self.conv2 = nn.Conv2d(channels, channels, kernel_size=3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(channels)

    def forward(self, x):
        h = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(h)) + x)


class GomokuResNet(nn.Module):
    """The ResidualBlock variant of BASELINE config 5.  The reference no longer has a forward for it;
    the parameter names and shapes are those of its historical checkpoints, so they load with load_state_dict."""

    CHANNELS = 64

    def __init__(self, board_size=None, device=None):
        super().__init__()
        self.board_size = _c.BOARD_SIZE if board_size is None else board_size
        self.device = device if device else torch.device("cpu")
        cells, ch = self.board_size ** 2, self.CHANNELS
        self.conv = nn.Conv2d(4, ch, kernel_size=3, padding=1)
        self.bn = nn.BatchNorm2d(ch)
        self.res1, self.res2, self.res3 = ResidualBlock(ch), ResidualBlock(ch), ResidualBlock(ch)
        self.policy_conv = nn.Conv2d(ch, 2, kernel_size=1)
        self.policy_bn = nn.BatchNorm2d(2)
        self.policy_fc = nn.Linear(2 * cells, cells)
        self.value_conv = nn.Conv2d(ch, 1, kernel_size=1)
        self.value_bn = nn.BatchNorm2d(1)
        self.value_fc1 = nn.Linear(cells, 64)
        self.value_fc2 = nn.Linear(64, 1)

    def forward(self, x):
        h = F.relu(self.bn(self.conv(x)))
        h = self.res3(self.res2(self.res1(h)))
        p = F.relu(self.policy_bn(self.policy_conv(h))).flatten(1)
        v = F.relu(self.value_bn(self.value_conv(h))).flatten(1)
        return self.policy_fc(p), torch.tanh(self.value_fc2(F.relu(self.value_fc1(v))))


def fold_resnet_state_dict(sd, eps=1e-5):
    """Eval-mode BatchNorm folded into the preceding conv (float64 arithmetic, float32 result): the 24 tensors of
    az_load_weights_resnet (include/az_engine.h)."""
    import numpy as np

    def get(k):
        t = sd[k]
        return (t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)).astype(np.float64)

    def fold(conv, bn):
        w = get(conv + ".weight")
        b = get(conv + ".bias") if (conv + ".bias") in sd else np.zeros(w.shape[0])
        scale = get(bn + ".weight") / np.sqrt(get(bn + ".running_var") + eps)
        wf = w * scale.reshape(-1, 1, 1, 1)
        bf = (b - get(bn + ".running_mean")) * scale + get(bn + ".bias")
        return wf.astype(np.float32), bf.astype(np.float32)

    out = list(fold("conv", "bn"))
    for r in ("res1", "res2", "res3"):
        out += fold(f"{r}.conv1", f"{r}.bn1")
        out += fold(f"{r}.conv2", f"{r}.bn2")
    pw, pb = fold("policy_conv", "policy_bn")
    vw, vb = fold("value_conv", "value_bn")
    out += [pw.reshape(2, -1), pb, vw.reshape(1, -1), vb]
    for k in ("policy_fc.weight", "policy_fc.bias", "value_fc1.weight", "value_fc1.bias", "value_fc2.weight", "value_fc2.bias"):
        out.append(get(k).astype(np.float32))
    out[22] = out[22].reshape(-1)
    return [np.ascontiguousarray(t) for t in out]
