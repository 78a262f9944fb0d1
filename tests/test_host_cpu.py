"""CPU tests of the host layer: the C-ABI library loads and exports every symbol of include/az_engine.h,
the host RNG reproduces numpy's legacy RandomState bit for bit, the host Gomoku class follows the golden
rules vectors, and the product refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd import _capi, games, net, self_play, evaluator, weights
from tests.util import ROOT, SIZES, build_weights, load


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "az_engine.h")).read()
    declared = set(re.findall(r"\b(az_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = ctypes.CDLL(_capi.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, f"declared in include/az_engine.h but not exported: {missing}"
    assert set(_capi.EXPORTS) <= declared


def test_no_cpu_fallback_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(az.AzError) as ei:
        az.Engine(5, 4, 10, 4)
    assert "no HIP device" in str(ei.value) or "-2" in str(ei.value)


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under the product package may import, link or open it."""
    pkg = os.path.join(ROOT, "alphazero-piskvorky_amd")
    bad = re.compile(r"(^|\W)(import\s+oracle|from\s+oracle|liboracle|oracle/|orc_[a-z_]+\s*\()")
    for dp, dn, fs in os.walk(pkg):
        dn[:] = [d for d in dn if not d.startswith("build")]
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                src = open(os.path.join(dp, f)).read()
                m = bad.search(src)
                assert not m, f"{os.path.join(dp, f)} references the oracle: {m.group(0)!r}"


@pytest.mark.parametrize("n", [5, 9, 15])
def test_host_rng_matches_numpy_randomstate(n):
    nn = n * n
    for seed in (0, 7, 900, 2 ** 31 + 5):
        noise, u = _capi.rng_selfplay_tape(seed, n)
        rs = np.random.RandomState(seed & 0xFFFFFFFF)
        off = 0
        for m in range(nn):
            d = rs.dirichlet([0.3] * (nn - m))
            assert np.array_equal(d, noise[off:off + nn - m])
            assert rs.random_sample() == u[m]
            off += nn - m


def test_host_rng_other_alphas_and_uniforms():
    for alpha in (1.0, 2.5, 0.03):
        noise, u = _capi.rng_selfplay_tape(42, 5, alpha=alpha, max_plies=3)
        rs = np.random.RandomState(42)
        off = 0
        for m in range(3):
            assert np.array_equal(rs.dirichlet([alpha] * (25 - m)), noise[off:off + 25 - m])
            assert rs.random_sample() == u[m]
            off += 25 - m
    assert np.array_equal(_capi.rng_uniforms(5, 1000), np.random.RandomState(5).random_sample(1000))


@pytest.mark.parametrize("n,k", SIZES)
def test_host_gomoku_follows_golden_rules(n, k):
    z = load(f"rules_{n}x{k}.npz")
    code = {None: 0, "X": 1, "O": 2, "D": 3}
    for g in range(min(len(z["nply"]), 25)):
        s = games.Gomoku(n, k)
        for m in range(int(z["nply"][g])):
            assert not s.is_terminal()
            a = int(z["actions"][g, m])
            s = s.apply_action((a // n, a % n))
        assert s.is_terminal() and code[s.get_game_result()] == int(z["result"][g])
    for i in range(min(len(z["enc_game"]), 40)):
        g, p = int(z["enc_game"][i]), int(z["enc_ply"][i])
        s = games.Gomoku(n, k)
        for m in range(p):
            a = int(z["actions"][g, m]); s = s.apply_action((a // n, a % n))
        assert np.array_equal(s.encode("cpu").numpy().astype(np.uint8), z["enc_planes"][i])
        legal = np.zeros(n * n, np.uint8)
        for r, c in s.get_legal_actions():
            legal[r * n + c] = 1
        assert np.array_equal(legal, z["legal_masks"][i])
    s = games.Gomoku(n, k).apply_action((0, 0))
    with pytest.raises(ValueError):
        s.apply_action((0, 0))


def test_synthetic_weights_equal_fixture_generator():
    for n in (5, 15):
        a, b = weights.synthetic_state_dict(n), build_weights(n)
        assert list(a) == list(b) == _capi.STATE_DICT_ORDER
        assert all(np.array_equal(a[k], b[k]) for k in a)


def test_torch_net_has_reference_state_dict_abi_and_numbers():
    z = load("net_5.npz")
    m = net.GomokuNet(board_size=5)
    assert list(m.state_dict().keys()) == _capi.STATE_DICT_ORDER
    m.load_state_dict({k[len("ckpt_saved__"):]: torch.tensor(z[k]) for k in z.files if k.startswith("ckpt_saved__")})
    m.eval()
    xs = []
    for i in range(len(z["players"])):
        s = games.Gomoku(5, 4)
        s.cells = z["boards"][i].copy(); s.current_player = "X" if z["players"][i] == 1 else "O"
        la = int(z["lasts"][i]); s.last_action = None if la < 0 else (la // 5, la % 5)
        xs.append(s.encode("cpu"))
    with torch.no_grad():
        logits, v = m(torch.stack(xs))
    np.testing.assert_allclose(logits.numpy(), z["ckpt_saved_logits"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(v.numpy().reshape(-1), z["ckpt_saved_value"], rtol=0, atol=1e-6)


def test_temperature_schedules_match_reference_values():
    # 1.0 at m=0, 0.906 at 10, 0.374 at 100; a23: 0.3*exp(-step/4)
    assert self_play.default_temperature_schedule(0) == 1.0
    assert abs(self_play.default_temperature_schedule(10) - 0.9058) < 1e-3
    assert abs(self_play.default_temperature_schedule(100) - 0.3741) < 1e-3
    z = load("arena_5x4.npz")
    t = z["temps"][0]
    assert t[0] == evaluator.temperature_schedule(0) and t[1] == evaluator.temperature_schedule(1) and t[2] == t[1]


def test_replay_buffer_and_model_loader_file_formats(tmp_path):
    """Reference file formats: buffer.pkl = pickled deque of tuples (replay_buffer.py:50-71), models/*.pt = state_dict
    picked by ctime (model_loader.py:43-56)."""
    import pickle
    from collections import deque
    from alphazero_piskvorky_amd.replay_buffer import ReplayBuffer
    from alphazero_piskvorky_amd.model_loader import ModelLoader
    buf = ReplayBuffer(capacity=5)
    ex = [(torch.zeros(4, 5, 5), np.full((5, 5), 1 / 25, np.float32), z) for z in (1, -1, 0, 1, -1, 0, 1)]
    buf.extend(ex)
    assert len(buf) == 5 and buf.all()[0][2] == 0                      # FIFO: the two oldest fell out
    assert buf.sample_batch(10) is buf.buffer and len(buf.sample_batch(3)) == 3
    path = str(tmp_path / "buffer.pkl")
    buf.save(path)
    raw = pickle.load(open(path, "rb"))
    assert isinstance(raw, deque) and len(raw) == 5
    other = ReplayBuffer(capacity=4)
    other.load(path)
    assert len(other) == 4 and other.all()[-1][2] == 1
    mdir = str(tmp_path / "models")
    ml = ModelLoader(mdir, lambda: net.GomokuNet(board_size=5))
    assert ml.best_path is None
    m = ml.get_best_model()                                             # writes a random-init checkpoint
    files = os.listdir(mdir)
    assert len(files) == 1 and files[0].startswith("model_") and files[0].endswith(".pt")
    m2 = ModelLoader(mdir, lambda: net.GomokuNet(board_size=5)).get_best_model()
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_emulated_trunk_weight_split_is_exact_enough():
    """az_emul_split (host side of az_set_trunk_mode, no GPU needed): the parts a weight is split into reproduce it to
    float32 precision -- bf16x3: hi + mid + lo, error <= 2^-24 |x|; f16x2: hi + lo / 2048, error <= 2^-22 |x| -- each part is
    the round-to-nearest value of what the previous parts left over, and float16's range is enforced."""
    import ctypes as C
    from alphazero_piskvorky_amd import _capi
    L = _capi.lib()
    L.az_emul_split.argtypes = [C.c_int, C.c_float, C.POINTER(C.c_uint16)]
    rs = np.random.RandomState(5)
    xs = np.concatenate([rs.standard_normal(2000) * 10.0 ** rs.uniform(-6, 3, 2000), [0.0, 1.0, -1.0, 65503.0, 1e-30, 3.0e38]]).astype(np.float32)
    parts = (C.c_uint16 * 3)()
    for x in xs:
        x = np.float32(x)
        assert L.az_emul_split(_capi.AZ_TRUNK_BF16X3, float(x), parts) == 3
        hi, mid, lo = [np.array([int(parts[i]) << 16], np.uint32).view(np.float32)[0] for i in range(3)]
        assert abs(np.float64(hi) + np.float64(mid) + np.float64(lo) - np.float64(x)) <= 2.0 ** -24 * abs(np.float64(x)) + 1e-45
        want_hi = (np.array([x]).view(np.uint32) + 0x7FFF + ((np.array([x]).view(np.uint32) >> 16) & 1)) >> 16
        assert int(parts[0]) == int(want_hi[0])                      # round to nearest even
        if abs(x) < 65504.0:
            assert L.az_emul_split(_capi.AZ_TRUNK_F16X2, float(x), parts) == 2
            h, l = [np.array([int(parts[i])], np.uint16).view(np.float16)[0] for i in range(2)]
            assert h == np.float16(x)
            assert abs(np.float64(h) + np.float64(l) / 2048.0 - np.float64(x)) <= 2.0 ** -22 * abs(np.float64(x)) + 2.0 ** -36
        else:
            assert L.az_emul_split(_capi.AZ_TRUNK_F16X2, float(x), parts) == -1       # AZ_ERR_INVALID: outside float16's range
    assert L.az_emul_split(7, 1.0, parts) < 0


def test_bench_gpus_n_refuses_to_measure_fewer_devices_than_ranks():
    """`python bench.py --gpus N` starts N ranks itself (a child torch.distributed.run, before any GPU call); on a node with
    fewer than N GPUs it must exit non-zero instead of printing a one-GPU line labelled N -- here: 0 GPUs, N = 2."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has the GPUs the launcher asks for")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-cpu"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "--share" in p.stderr and not p.stdout.strip()


def test_counters_struct_mirrors_the_header():
    """az_counters grew tape_wait_seconds / tape_threads / host_cpus this round: the ctypes mirror must have every field
    of the header's struct, in order."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "az_engine.h")).read()
    body = re.search(r"typedef struct \{((?:(?!typedef struct).)*?)\} az_counters;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            names += [x.strip() for x in decl.split(None, 1)[1].split(",")]
    assert names == [f for f, _ in _capi.az_counters._fields_]
