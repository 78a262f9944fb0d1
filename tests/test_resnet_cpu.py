"""ResidualBlock variant (BASELINE config 5) on CPU: the oracle's forward against the build's torch definition.

The reference ships no forward for this variant, only historical checkpoints that fix the topology:
parity against the reference is UNPINNED; the oracle is pinned to the build-owned torch module (logits 1e-5, softmax 1e-6, value 5e-6) and
the module's parameter names/shapes to the checkpoint key list recorded below (data captured from
alphazero/models/old/model_20250728_225053.pt)."""
import numpy as np
import torch

from oracle import oracle as orc
from alphazero_piskvorky_amd.net import GomokuResNet, fold_resnet_state_dict
from alphazero_piskvorky_amd.weights import synthetic_resnet_state_dict

CHECKPOINT_KEYS_5x5 = {
    "conv.weight": (64, 4, 3, 3), "conv.bias": (64,), "bn.weight": (64,), "bn.bias": (64,), "bn.running_mean": (64,), "bn.running_var": (64,),
    **{f"res{r}.{c}.weight": (64, 64, 3, 3) for r in (1, 2, 3) for c in ("conv1", "conv2")},
    **{f"res{r}.{b}.{k}": (64,) for r in (1, 2, 3) for b in ("bn1", "bn2") for k in ("weight", "bias", "running_mean", "running_var")},
    "policy_conv.weight": (2, 64, 1, 1), "policy_conv.bias": (2,),
    **{f"policy_bn.{k}": (2,) for k in ("weight", "bias", "running_mean", "running_var")},
    "policy_fc.weight": (25, 50), "policy_fc.bias": (25,),
    "value_conv.weight": (1, 64, 1, 1), "value_conv.bias": (1,),
    **{f"value_bn.{k}": (1,) for k in ("weight", "bias", "running_mean", "running_var")},
    "value_fc1.weight": (64, 25), "value_fc1.bias": (64,), "value_fc2.weight": (1, 64), "value_fc2.bias": (1,),
}


def _module(n, seed=4321):
    m = GomokuResNet(board_size=n)
    sd = synthetic_resnet_state_dict(n, seed)
    missing = m.load_state_dict({k: torch.tensor(v) for k, v in sd.items()}, strict=False)
    assert not missing.unexpected_keys and all(k.endswith("num_batches_tracked") for k in missing.missing_keys)
    m.eval()
    return m, sd


def test_module_has_the_checkpoint_parameter_abi():
    m = GomokuResNet(board_size=5)
    have = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.endswith("num_batches_tracked")}
    assert have == CHECKPOINT_KEYS_5x5
    assert set(synthetic_resnet_state_dict(5)) == set(CHECKPOINT_KEYS_5x5)


def test_oracle_resnet_forward_matches_torch_definition():
    for n in (5, 9, 15):
        m, sd = _module(n)
        net = orc.Net(n, resnet_tensors=fold_resnet_state_dict(sd))
        o = orc.Oracle(n, 5 if n > 5 else 4, 1)
        rs = np.random.RandomState(n)
        for _ in range(6):
            b = np.zeros(n * n, np.uint8)
            cells = rs.permutation(n * n)[:rs.randint(0, n * n // 2)]
            for j, c in enumerate(cells):
                b[c] = 1 + j % 2
            planes = o.encode(b, 1 + len(cells) % 2, int(cells[-1]) if len(cells) else -1)
            lg, P, v = net.eval(planes)
            with torch.no_grad():
                tl, tv = m(torch.tensor(planes)[None])
            np.testing.assert_allclose(lg, tl.numpy()[0], rtol=0, atol=1e-5)
            np.testing.assert_allclose(P, torch.softmax(tl[0], 0).numpy(), rtol=0, atol=1e-6)
            assert abs(v - float(tv)) <= 5e-6


def _checkpoint_fixture():
    from tests.util import load
    z = load("resnet_ckpt_5.npz")
    return z, {k[3:]: z[k] for k in z.files if k.startswith("w__")}


def test_real_checkpoint_values_through_the_oracle_and_the_build_torch_module():
    """tests/golden/resnet_ckpt_5.npz: the VALUES of the reference's historical checkpoint model_20250728_225053.pt (a
    weights-only load in the build container) and the outputs the build's GomokuResNet gives with them.  The parameter
    set is exactly the ABI above; the module reproduces the recorded outputs here, and the oracle (BatchNorm folded on the
    host) meets them within the ResidualBlock tolerances.  Parity against the REFERENCE stays unpinned: it has no forward."""
    z, sd = _checkpoint_fixture()
    assert {k: tuple(v.shape) for k, v in sd.items()} == CHECKPOINT_KEYS_5x5
    n = int(z["n"])
    m = GomokuResNet(board_size=n)
    m.load_state_dict({k: torch.tensor(v) for k, v in sd.items()}, strict=False)
    m.eval()
    net = orc.Net(n, resnet_tensors=fold_resnet_state_dict(sd))
    o = orc.Oracle(n, 4, 1)
    assert float(np.abs(z["value"]).max()) > 0.05 and float(z["P"].max()) > 0.1       # a trained net, not an initialisation
    for i in range(len(z["players"])):
        planes = o.encode(z["boards"][i], int(z["players"][i]), int(z["lasts"][i]))
        with torch.no_grad():
            tl, tv = m(torch.tensor(planes)[None])
        np.testing.assert_allclose(tl.numpy()[0], z["logits"][i], rtol=0, atol=2e-6)     # the module here == the module at fixture time
        lg, P, v = net.eval(planes)
        np.testing.assert_allclose(lg, z["logits"][i], rtol=0, atol=1e-5)
        np.testing.assert_allclose(P, z["P"][i], rtol=0, atol=1e-6)
        assert abs(v - float(z["value"][i])) <= 5e-6


def test_bn_folding_is_exact_in_float64():
    sd = synthetic_resnet_state_dict(5)
    t = fold_resnet_state_dict(sd)
    assert len(t) == 24 and t[0].shape == (64, 4, 3, 3) and t[2].shape == (64, 64, 3, 3) and t[14].shape == (2, 64)
    assert t[16].shape == (1, 64) and t[18].shape == (25, 50) and t[20].shape == (64, 25) and t[22].shape == (64,)
    scale = sd["res2.bn1.weight"].astype(np.float64) / np.sqrt(sd["res2.bn1.running_var"].astype(np.float64) + 1e-5)
    np.testing.assert_array_equal(t[6], (sd["res2.conv1.weight"].astype(np.float64) * scale[:, None, None, None]).astype(np.float32))
