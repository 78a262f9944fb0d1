"""Deterministic synthetic GomokuNet weights (state_dict layout of the reference's net.py:37-53).

There is no network access for checkpoints, so benchmarks and size-independent tests use this
generator: numpy RandomState(seed) normal * sqrt(2/fan_in) for weights, 0.05 * normal for biases.
tests/golden/make_golden.py holds the same definition for the golden fixtures (the tests assert
the two agree)."""
import numpy as np


def synthetic_state_dict(n, seed=1234):
    rs = np.random.RandomState(seed)
    shapes = [
        ("conv1.weight", (32, 4, 3, 3)), ("conv1.bias", (32,)),
        ("conv2.weight", (64, 32, 3, 3)), ("conv2.bias", (64,)),
        ("conv3.weight", (128, 64, 3, 3)), ("conv3.bias", (128,)),
        ("policy_conv.weight", (4, 128, 1, 1)), ("policy_conv.bias", (4,)),
        ("policy_fc.weight", (n * n, 4 * n * n)), ("policy_fc.bias", (n * n,)),
        ("value_conv.weight", (2, 128, 1, 1)), ("value_conv.bias", (2,)),
        ("value_fc1.weight", (64, 2 * n * n)), ("value_fc1.bias", (64,)),
        ("value_fc2.weight", (1, 64)), ("value_fc2.bias", (1,)),
    ]
    out = {}
    for name, shp in shapes:
        if name.endswith("weight"):
            fan_in = int(np.prod(shp[1:]))
            w = rs.standard_normal(shp) * (2.0 / fan_in) ** 0.5
        else:
            w = rs.standard_normal(shp) * 0.05
        out[name] = w.astype(np.float32)
    return out


def synthetic_resnet_state_dict(n, seed=4321):
    """Deterministic weights for the ResidualBlock variant (keys/shapes of the reference's historical checkpoints,
    alphazero/models/old/model_20250728_*.pt), with non-trivial BatchNorm statistics."""
    rs = np.random.RandomState(seed)
    nn = n * n
    out = {}

    def conv(name, co, ci, k, bias):
        out[name + ".weight"] = (rs.standard_normal((co, ci, k, k)) * (2.0 / (ci * k * k)) ** 0.5).astype(np.float32)
        if bias:
            out[name + ".bias"] = (rs.standard_normal(co) * 0.05).astype(np.float32)

    def bn(name, c):
        out[name + ".weight"] = rs.uniform(0.6, 1.4, c).astype(np.float32)
        out[name + ".bias"] = (rs.standard_normal(c) * 0.1).astype(np.float32)
        out[name + ".running_mean"] = (rs.standard_normal(c) * 0.2).astype(np.float32)
        out[name + ".running_var"] = rs.uniform(0.5, 1.5, c).astype(np.float32)

    def lin(name, o, i):
        out[name + ".weight"] = (rs.standard_normal((o, i)) * (2.0 / i) ** 0.5).astype(np.float32)
        out[name + ".bias"] = (rs.standard_normal(o) * 0.05).astype(np.float32)

    conv("conv", 64, 4, 3, True); bn("bn", 64)
    for r in (1, 2, 3):
        conv(f"res{r}.conv1", 64, 64, 3, False); bn(f"res{r}.bn1", 64)
        conv(f"res{r}.conv2", 64, 64, 3, False); bn(f"res{r}.bn2", 64)
    conv("policy_conv", 2, 64, 1, True); bn("policy_bn", 2); lin("policy_fc", nn, 2 * nn)
    conv("value_conv", 1, 64, 1, True); bn("value_bn", 1); lin("value_fc1", 64, nn); lin("value_fc2", 1, 64)
    return out
