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
