"""GPU tests of the drop-in Python seams: same call signatures and contracts as the reference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.util import load, weights_from_fixture

import alphazero_piskvorky_amd as az
from alphazero_piskvorky_amd import constants, games, net
from alphazero_piskvorky_amd.controller import NeuralNetworkController, make_policy_value_fn
from alphazero_piskvorky_amd.evaluator import ModelEvaluator
from alphazero_piskvorky_amd.mcts import MCTS
from alphazero_piskvorky_amd.self_play import SelfPlayManager, default_temperature_schedule


def _controller(tag, n=5):
    m = net.GomokuNet(board_size=n)
    m.load_state_dict({k: torch.tensor(v) for k, v in weights_from_fixture(n, tag).items()})
    m.eval()
    return NeuralNetworkController(m, device="cuda:0")


def test_make_policy_value_fn_matches_reference_numbers():
    z = load("net_5.npz")
    pvf = make_policy_value_fn(_controller("ckpt_saved"))
    for i in range(len(z["players"])):
        s = games.Gomoku(5, 4)
        s.cells = z["boards"][i].copy(); s.current_player = "X" if z["players"][i] == 1 else "O"
        la = int(z["lasts"][i]); s.last_action = None if la < 0 else (la // 5, la % 5)
        P, v = pvf(s)
        assert P.shape == (5, 5) and P.dtype == np.float32 and isinstance(v, float)
        np.testing.assert_allclose(P.reshape(-1), z["ckpt_saved_P"][i], rtol=0, atol=1e-6)
        assert abs(v - float(z["ckpt_saved_v1"][i])) <= 2e-6


def test_mcts_run_reproduces_reference_games_under_np_random_seed():
    """The _worker loop body (self_play.py:48-65) driven through the MCTS.run shim with np.random.seed,
    against the reference's recorded trajectories."""
    z = load("netgame_5x4.npz")
    m = MCTS(make_policy_value_fn(_controller("ckpt_saved")), num_simulations=int(z["S"]), c_puct=2.0)
    matched = 0
    games_ = np.unique(z["game"])
    for g in games_:
        sel = np.where(z["game"] == g)[0]
        np.random.seed(int(z["seed0"]) + int(g))
        s = games.Gomoku(5, 4); mv = 0; ok = True
        while not s.is_terminal():
            pi, a = m.run(s, temperature=default_temperature_schedule(mv), add_root_noise=True)
            assert pi.shape == (5, 5) and pi.dtype == np.float32 and abs(float(pi.sum()) - 1.0) < 1e-5
            if mv < len(sel) and a[0] * 5 + a[1] == int(z["action"][sel[mv]]) and ok:
                assert np.array_equal(m.last_visits.reshape(-1), z["N"][sel[mv]])
                np.testing.assert_allclose(pi.reshape(-1), z["pi"][sel[mv]], rtol=0, atol=1e-6)
            else:
                ok = False
            s = s.apply_action(a); mv += 1
        matched += ok and mv == len(sel)
    assert matched >= len(games_) - 1


def test_generate_self_play_contract():
    ctrl = _controller("ckpt_saved")
    mgr = SelfPlayManager(ctrl, "cuda:0", mcts_params={"num_simulations": 40, "c_puct": 2.0}, concurrent_games=8, seed=77)
    data = mgr.generate_self_play(num_games=12, num_workers=3)
    c = mgr.last_counters
    assert len(data) == 4 * c["plies"]                                   # 4 "symmetries" per position (self_play.py:146-148)
    s, p, zval = data[0]
    assert isinstance(s, torch.Tensor) and s.dtype == torch.float32 and tuple(s.shape) == (4, 5, 5) and s.is_contiguous() and s.device.type == "cpu"
    assert isinstance(p, np.ndarray) and p.dtype == np.float32 and p.shape == (5, 5) and p.flags["C_CONTIGUOUS"]
    assert isinstance(zval, int)
    assert all(abs(float(pp.sum()) - 1.0) < 1e-5 for _, pp, _ in data[::7])
    assert set(zz for _, _, zz in data) <= {-1, 0, 1}
    assert float(data[0][0].sum()) == 0.0                                # first position of game 0: empty board
    # k = 0 record equals the engine's raw record; k = 1..3 follow the reference's rotation rule
    raw = mgr._engine.records()
    for r in (0, 5, c["plies"] - 1):
        assert np.array_equal(data[4 * r + 2][0].numpy(), np.rot90(data[4 * r][0].numpy(), 2, (1, 2)))
        for kk in range(4):
            assert np.array_equal(data[4 * r + kk][1], np.rot90(raw["pis"][r].reshape(5, 5)))
    # training consumes the examples (controller.py:133-178 shape guard + AdamW step)
    out = ctrl.train(data[:256], epochs=1)
    assert np.isfinite(out["loss"])
    # same seed -> same examples whatever the number of concurrent slots (per-game RNG streams)
    ctrl2 = _controller("ckpt_saved")
    d2 = SelfPlayManager(ctrl2, "cuda:0", mcts_params={"num_simulations": 40, "c_puct": 2.0}, concurrent_games=5, seed=77).generate_self_play(12)
    ctrl3 = _controller("ckpt_saved")
    d3 = SelfPlayManager(ctrl3, "cuda:0", mcts_params={"num_simulations": 40, "c_puct": 2.0}, concurrent_games=12, seed=77).generate_self_play(12)
    assert len(d2) == len(d3) and all(np.array_equal(a[1], b[1]) and a[2] == b[2] for a, b in zip(d2, d3))


def test_generate_self_play_with_subtree_reuse_keeps_the_example_contract():
    """SelfPlayManager(subtree_reuse=True): same record contract, fewer simulations, slot-count independent."""
    outs = []
    for slots in (4, 9):
        mgr = SelfPlayManager(_controller("ckpt_saved"), "cuda:0", mcts_params={"num_simulations": 40, "c_puct": 2.0},
                              concurrent_games=slots, seed=78, subtree_reuse=True)
        data = mgr.generate_self_play(num_games=9)
        c = mgr.last_counters
        assert len(data) == 4 * c["plies"] and c["simulations"] < 40 * c["plies"] and c["root_evals"] < c["plies"]
        assert all(abs(float(pp.sum()) - 1.0) < 1e-5 for _, pp, _ in data[::5]) and set(zz for _, _, zz in data) <= {-1, 0, 1}
        raw = mgr._engine.records()
        assert (raw["visits"].sum(axis=1) == 40).all()                   # every root ends the ply with S visits below it
        outs.append(data)
    assert len(outs[0]) == len(outs[1]) and all(np.array_equal(a[1], b[1]) and a[2] == b[2] for a, b in zip(*outs))


def test_model_evaluator_matches_reference_arena():
    z = load("arena_5x4.npz")
    constants.NUM_EVAL_SIMULATIONS = int(z["S"])
    ev = ModelEvaluator(game_class=games.Gomoku, print_games=False, device="cuda:0", seed=int(z["seed0"]))
    wr, metrics = ev.evaluate(_controller("ckpt_saved"), _controller("ckpt_0802"), num_games=z["actions"].shape[0])
    constants.NUM_EVAL_SIMULATIONS = 200
    assert set(metrics) == {"wins", "losses", "draws", "total", "win_rate"}
    r = ev.last_result
    same = sum(np.array_equal(r["actions"][g][:int(r["nply"][g])], z["actions"][g][z["actions"][g] >= 0]) for g in range(metrics["total"]))
    assert same == metrics["total"]      # frozen fixture, deterministic search: every game, not all but one
    if same == metrics["total"]:
        assert (metrics["wins"], metrics["losses"], metrics["draws"]) == (int(z["wins"]), int(z["losses"]), int(z["draws"]))
        assert abs(wr - float(z["win_rate"])) < 1e-12


@pytest.mark.parametrize("n,k,S", [(5, 4, 100), (9, 5, 60), (15, 5, 40)])
def test_mcts_with_an_arbitrary_python_evaluator_matches_the_oracle(n, k, S):
    """The policy_value_fn plugin seam (mcts.py:87-93): MCTS takes ANY callable state -> (policy [n,n], value).  The shim
    keeps the tree on the GPU and calls the evaluator on the host for every leaf (az_search_callback).  Here the callable
    is the build's deterministic synthetic evaluator written in Python, so the whole search must equal the oracle's (and
    the reference's own MCTS.run with that evaluator, pinned in tree_*.npz) bit for bit."""
    from oracle import oracle as orc
    from tests.util import synth_eval_codes
    calls = []

    def python_evaluator(state):
        cells = state.cells
        me = state.player_code()
        codes = np.where(cells == 0, 0, np.where(cells == me, 1, 2)).astype(np.uint8)
        calls.append(1)
        return synth_eval_codes(codes, state.last_index(), n)          # (float32 [n,n], python float)

    m = MCTS(python_evaluator, num_simulations=S, c_puct=2.0)
    o = orc.Oracle(n, k, S, synthetic=True)
    rs = np.random.RandomState(17)
    g = games.Gomoku(n, k)
    for ply in range(4):
        T = 0.8
        np.random.seed(1000 + ply)
        pi, action = m.run(g, temperature=T, add_root_noise=True)
        np.random.seed(1000 + ply)
        noise = np.random.dirichlet([0.3] * int((g.cells == 0).sum())); u = np.random.random_sample()
        ro = o.search(None, g.cells, g.player_code(), g.last_index(), T, noise, u)
        assert np.array_equal(m.last_visits.reshape(-1), ro["N"]), f"ply {ply}: visit counts"
        assert np.array_equal(pi.reshape(-1), ro["pi"]) and action[0] * n + action[1] == ro["action"]
        g = g.apply_action(action)
    assert len(calls) >= 4 * (S + 1) - 4 * 3       # one call per evaluation (terminal leaves need none)
    # the golden trees of the Python reference itself (its MCTS.run with this evaluator)
    z = load(f"tree_{n}x{k}.npz")
    if int(z["S"]) == S:
        e = az.Engine(n, k, S, 1, log_table=orc.numpy_log_table(S))
        for i in range(len(z["seed"])):
            board = z["board"][i]
            rsz = np.random.RandomState(int(z["seed"][i]))
            noise = rsz.dirichlet([0.3] * int((board == 0).sum())) if z["noise"][i] else None
            u = rsz.random_sample()

            def ev(cells, player, last):
                codes = np.where(cells == 0, 0, np.where(cells == player, 1, 2)).astype(np.uint8)
                return synth_eval_codes(codes, last, n)

            r = e.search_callback(board, int(z["player"][i]), int(z["last"][i]), float(z["T"][i]), ev, noise, u)
            assert np.array_equal(r["N"], z["N"][i]) and np.array_equal(r["W"], z["W"][i]) and np.array_equal(r["P"], z["P"][i])
            assert r["action"] == int(z["action"][i])
        e.close()


def test_mcts_external_evaluator_errors_surface():
    def broken(state):
        raise RuntimeError("evaluator failed")
    m = MCTS(broken, num_simulations=5, c_puct=2.0)
    with pytest.raises(RuntimeError, match="evaluator failed"):
        m.run(games.Gomoku(5, 4), temperature=1.0)
    with pytest.raises(TypeError):
        MCTS(None, num_simulations=5, c_puct=2.0)


@pytest.mark.parametrize("device_replay", [False, True])
def test_training_loop_plumbing_config1(device_replay):
    """BASELINE.json configs[0] plumbing on the GPU path: self-play -> buffer -> AdamW -> arena -> promote (train.py:85-119);
    device_replay=True keeps the examples in the device ring instead of Python tuples."""
    from alphazero_piskvorky_amd import train
    saved = (constants.BATCHES_PER_EPISODE, constants.NUM_EPOCHS, constants.BATCH_SIZE)
    constants.BATCHES_PER_EPISODE, constants.NUM_EPOCHS, constants.BATCH_SIZE = 2, 1, 256
    try:
        hist = train.run(episodes=2, games=16, sims=24, eval_games=6, device="cuda:0", seed=3, log=lambda *_: None,
                         device_replay=device_replay)
    finally:
        constants.BATCHES_PER_EPISODE, constants.NUM_EPOCHS, constants.BATCH_SIZE = saved
    assert len(hist) == 2
    for h in hist:
        assert h["examples"] > 0 and h["examples"] % 4 == 0 and np.isfinite(h["loss"])
        assert h["total"] == 6 and 0.0 <= h["win_rate"] <= 1.0 and h["wins"] + h["losses"] + h["draws"] == 6


def test_training_loop_config0_at_its_stated_size():
    """BASELINE.json configs[0] as the reference runs it (train.py:62-119 with constants.py's defaults): 5x5 / 4-in-a-row,
    20 episodes of 100 self-play games at 100 simulations, 10 batches x 3 epochs of AdamW and a 51-game arena per episode,
    promotion at a win rate above 0.55 -- about 12 s on one MI355X (the reference's README quotes half an hour on a laptop)."""
    from alphazero_piskvorky_amd import train
    hist = train.run(episodes=20, games=100, sims=100, eval_games=constants.EVALUATION_GAMES, eval_sims=constants.NUM_EVAL_SIMULATIONS, device="cuda:0", seed=11,
                     log=lambda *_: None)
    assert len(hist) == 20 and constants.EVALUATION_GAMES == 51
    for h in hist:
        assert h["examples"] >= 100 * 7 * 4 and h["examples"] % 4 == 0 and np.isfinite(h["loss"])
        assert h["total"] == 51 and h["wins"] + h["losses"] + h["draws"] == 51
        assert h["promoted"] == (h["win_rate"] > train.PROMOTION_THRESHOLD)


def test_device_replay_ring_matches_host_examples():
    """sampling from the device ring gives exactly the examples generate_self_play would have produced."""
    from alphazero_piskvorky_amd import Engine, parallel
    from alphazero_piskvorky_amd.device_replay import DeviceReplayBuffer
    n = 5
    e = Engine(n, 4, 16, 8, synthetic=True)
    c = e.selfplay(6, seed0=5)
    R = c["records"]
    dev = torch.device("cuda:0")
    packed, counts = parallel.gather_packed_records(e, dev)
    st = torch.empty((R * 4, 4, n, n), device=dev); pi = torch.empty((R * 4, n, n), device=dev); zz = torch.empty(R * 4, device=dev)
    e.examples_from_packed(packed.data_ptr(), R, 4, st.data_ptr(), pi.data_ptr(), zz.data_ptr())
    torch.cuda.synchronize()
    buf = DeviceReplayBuffer(e, capacity=4 * R, device="cuda:0", seed=1)
    buf.extend_packed(packed, R)
    assert len(buf) == 4 * R
    s, p, z = buf.sample_batch(4 * R)                       # without replacement: a permutation of all examples
    torch.cuda.synchronize()
    key = lambda a, b, c_: [(tuple(x.flatten().tolist()), tuple(y.flatten().tolist()), float(w)) for x, y, w in zip(a.cpu(), b.cpu(), c_.cpu())]
    assert sorted(key(s, p, z)) == sorted(key(st, pi, zz))
    # FIFO overwrite: a ring of half the size keeps only the newest positions
    small = DeviceReplayBuffer(e, capacity=4 * (R // 2), device="cuda:0", seed=2)
    small.extend_packed(packed, R)
    s2, p2, z2 = small.sample_batch(10 ** 6)
    keep = R // 2
    assert s2.shape[0] == 4 * keep
    assert sorted(key(s2, p2, z2)) == sorted(key(st[4 * (R - keep):], pi[4 * (R - keep):], zz[4 * (R - keep):]))
    # feeds train_step directly (controller.py:100-131)
    ctrl = _controller("ckpt_saved")
    out = ctrl.train_step(s[:64], p[:64], z[:64])
    assert np.isfinite(out["loss"])
    # export in the reference's example format, oldest first == the (record, k) order of examples_from_packed
    ex = buf.export_examples()
    assert len(ex) == 4 * R and isinstance(ex[0][0], torch.Tensor) and isinstance(ex[0][1], np.ndarray) and isinstance(ex[0][2], int)
    assert all(torch.equal(ex[i][0], st[i].cpu()) and np.array_equal(ex[i][1], pi[i].cpu().numpy()) and ex[i][2] == int(zz[i]) for i in range(0, 4 * R, 7))
    e.close()
