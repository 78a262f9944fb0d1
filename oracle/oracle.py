"""ctypes binding of the CPU oracle (oracle/az_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STATE_DICT_ORDER = [
    "conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "conv3.weight", "conv3.bias",
    "policy_conv.weight", "policy_conv.bias", "policy_fc.weight", "policy_fc.bias",
    "value_conv.weight", "value_conv.bias", "value_fc1.weight", "value_fc1.bias",
    "value_fc2.weight", "value_fc2.bias",
]


def build(force=False):
    if os.environ.get("AZ_ORACLE_LIB"):          # e.g. the sanitizer build (make -C oracle liboracle_asan.so)
        return os.environ["AZ_ORACLE_LIB"]
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "az_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


class _Cfg(C.Structure):
    _fields_ = [("n", C.c_int), ("k", C.c_int), ("S", C.c_int),
                ("c_puct", C.c_double), ("alpha", C.c_double), ("w", C.c_double),
                ("eval_kind", C.c_int), ("log_table", C.POINTER(C.c_float)), ("reuse", C.c_int), ("vl", C.c_int),
                ("leaf_sym", C.c_int), ("game_id", C.c_int)]


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_net_create.restype = C.c_void_p
        L.orc_net_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.orc_resnet_create.restype = C.c_void_p
        L.orc_resnet_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.orc_net_free.argtypes = [C.c_void_p]
        L.orc_net_eval.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.orc_net_eval_sym.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_leaf_sym_of.restype = C.c_int; L.orc_leaf_sym_of.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_search.restype = C.c_int
        L.orc_selfplay_game.restype = C.c_int
        L.orc_arena_game.restype = C.c_int
        L.orc_test_expf.restype = C.c_float; L.orc_test_expf.argtypes = [C.c_float]
        L.orc_test_tanhf.restype = C.c_float; L.orc_test_tanhf.argtypes = [C.c_float]
        L.orc_test_exp.restype = C.c_double; L.orc_test_exp.argtypes = [C.c_double]
        L.orc_test_pwsum.restype = C.c_double; L.orc_test_pwsum.argtypes = [C.c_void_p, C.c_int]
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def numpy_log_table(S):
    """log_table[N] = np.log(float32(N) + 1e-8) exactly as mcts.py:161 evaluates it (float32)."""
    return np.log(np.arange(S + 1, dtype=np.float32) + 1e-8).astype(np.float32)


def selfplay_T_table(nn):
    """self_play.py:24-26 default_temperature_schedule, evaluated with numpy like the reference."""
    return np.array([(np.exp(-m / 100) + 0.01) / 1.01 for m in range(nn + 1)], dtype=np.float64)


def arena_T_table(nn):
    """evaluator.py:14-19 temperature_schedule."""
    return np.array([0.3 * np.exp(-m / 4) for m in range(nn + 2)], dtype=np.float64)


def selfplay_tape(seed, n, maxply=None):
    """Per-game RNG tape: the draws mcts.py:114,177 make per ply, from numpy's own legacy RNG."""
    nn = n * n
    rs = np.random.RandomState(seed)
    noise, us = [], []
    for m in range(nn if maxply is None else maxply):
        noise.append(rs.dirichlet([0.3] * (nn - m)))
        us.append(rs.random_sample())
    return np.concatenate(noise), np.array(us, dtype=np.float64)


class Net:
    def __init__(self, n, sd=None, resnet_tensors=None):
        """sd: GomokuNet state_dict (numpy); resnet_tensors: the 24 folded tensors of az_load_weights_resnet."""
        self.n = n
        if resnet_tensors is not None:
            self._keep = [np.ascontiguousarray(np.asarray(t, dtype=np.float32)) for t in resnet_tensors]
            assert len(self._keep) == 24
            arr = (C.c_void_p * 24)(*[t.ctypes.data for t in self._keep])
            self.h = lib().orc_resnet_create(n, arr)
            return
        self._keep = [np.ascontiguousarray(np.asarray(sd[k], dtype=np.float32)) for k in STATE_DICT_ORDER]
        arr = (C.c_void_p * 16)(*[t.ctypes.data for t in self._keep])
        self.h = lib().orc_net_create(n, arr)

    def eval(self, planes):
        nn = self.n * self.n
        planes = np.ascontiguousarray(planes, dtype=np.float32)
        logits = np.zeros(nn, np.float32); P = np.zeros(nn, np.float32); v = np.zeros(1, np.float32)
        lib().orc_net_eval(self.h, _p(planes), _p(logits), _p(P), _p(v))
        return logits, P, float(v[0])

    def eval_sym(self, planes, t):
        """raw outputs for the position shown to the net under dihedral symmetry t (0..7); logits in board order"""
        nn = self.n * self.n
        planes = np.ascontiguousarray(planes, dtype=np.float32)
        logits = np.zeros(nn, np.float32); v = np.zeros(1, np.float32)
        lib().orc_net_eval_sym(self.h, int(self.n), _p(planes), int(t), _p(logits), _p(v))
        return logits, float(v[0])

    def __del__(self):
        try:
            lib().orc_net_free(self.h)
        except Exception:
            pass


class Oracle:
    def __init__(self, n, k, S, c_puct=2.0, alpha=0.3, w=0.25, synthetic=False, log_table=None, reuse=False, virtual_loss=0,
                 leaf_sym=False):
        self.n, self.k, self.S = n, k, S
        self.log_table = numpy_log_table(max(S, 1)) if log_table is None else np.ascontiguousarray(log_table, np.float32)
        self.cfg = _Cfg(n, k, S, c_puct, alpha, w, 1 if synthetic else 0,
                        self.log_table.ctypes.data_as(C.POINTER(C.c_float)), 1 if reuse else 0, int(virtual_loss),
                        1 if leaf_sym else 0, 0)

    def _cfg_for(self, game):
        """the configuration with the game id the leaf-symmetry hash uses (a copy: one Oracle serves many threads)"""
        if not self.cfg.leaf_sym:
            return self.cfg
        c = _Cfg.from_buffer_copy(self.cfg)
        c.game_id = C.c_int32(int(game) & 0xFFFFFFFF).value      # the engine's key: low 32 bits of seed0 + game id
        return c

    # ---- rules ----
    def replay(self, actions):
        nn = self.n * self.n
        a = np.ascontiguousarray(actions, dtype=np.int16)
        term = np.zeros(len(a), np.uint8); board = np.zeros(nn, np.uint8)
        pl = C.c_int(); res = C.c_int()
        rc = lib().orc_replay(self.n, self.k, _p(a), len(a), _p(term), _p(board), C.byref(pl), C.byref(res))
        return rc, term, board, pl.value, res.value

    def encode(self, board, player, last):
        nn = self.n * self.n
        planes = np.zeros(4 * nn, np.float32)
        lib().orc_encode(self.n, _p(np.ascontiguousarray(board, np.uint8)), int(player), int(last), _p(planes))
        return planes.reshape(4, self.n, self.n)

    def synth_eval(self, board, player, last):
        nn = self.n * self.n
        P = np.zeros(nn, np.float32); v = np.zeros(1, np.float32)
        lib().orc_synth_eval(self.n, _p(np.ascontiguousarray(board, np.uint8)), int(player), int(last), _p(P), _p(v))
        return P, float(v[0])

    # ---- search ----
    def search(self, net, board, player, last, T, noise, u, game=0):
        nn = self.n * self.n
        pi = np.zeros(nn, np.float32); N = np.zeros(nn, np.int32); W = np.zeros(nn, np.float64); P = np.zeros(nn, np.float32)
        nexp = C.c_int(); maxd = C.c_int()
        nz = None if noise is None else np.ascontiguousarray(noise, np.float64)
        a = lib().orc_search(C.byref(self._cfg_for(game)), C.c_void_p(net.h if net else None),
                             _p(np.ascontiguousarray(board, np.uint8)), int(player), int(last), C.c_double(T),
                             _p(nz), C.c_double(u), _p(pi), _p(N), _p(W), _p(P), C.byref(nexp), C.byref(maxd))
        return dict(action=a, pi=pi, N=N, W=W, P=P, nexp=nexp.value, maxd=maxd.value)

    def selfplay_game(self, net, noise_tape, u_tape, T_table=None, maxply=None, game=0):
        nn = self.n * self.n
        maxply = nn if maxply is None else maxply
        T_table = selfplay_T_table(nn) if T_table is None else np.ascontiguousarray(T_table, np.float64)
        boards = np.zeros((maxply, nn), np.uint8); movers = np.zeros(maxply, np.uint8); lasts = np.zeros(maxply, np.int16)
        pis = np.zeros((maxply, nn), np.float32); visits = np.zeros((maxply, nn), np.int32)
        actions = np.zeros(maxply, np.int16); z = np.zeros(maxply, np.int8)
        res = C.c_int(); counters = np.zeros(8, np.int64)
        nz = None if noise_tape is None else np.ascontiguousarray(noise_tape, np.float64)
        ut = np.ascontiguousarray(u_tape, np.float64)
        m = lib().orc_selfplay_game(C.byref(self._cfg_for(game)), C.c_void_p(net.h if net else None), _p(nz), _p(ut), _p(T_table),
                                    int(maxply), _p(boards), _p(movers), _p(lasts), _p(pis), _p(visits), _p(actions),
                                    _p(z), C.byref(res), _p(counters))
        return dict(nply=m, boards=boards[:m], movers=movers[:m], lasts=lasts[:m], pis=pis[:m], visits=visits[:m],
                    actions=actions[:m], z=z[:m], result=res.value,
                    counters=dict(expansions=int(counters[0]), sims=int(counters[1]), terminal_hits=int(counters[2]),
                                  depth_sum=int(counters[3]), root_evals=int(counters[4]), dup_sims=int(counters[5])))

    def arena_game(self, cand, base, game_index, u_tape, T_table=None, key=None):
        """key: what the leaf-symmetry hash names the game by (the engine: its seed, seed0 + game_index); default game_index"""
        nn = self.n * self.n
        T_table = arena_T_table(nn) if T_table is None else np.ascontiguousarray(T_table, np.float64)
        actions = np.zeros(nn, np.int16); temps = np.zeros(nn, np.float64); nply = C.c_int()
        ut = np.ascontiguousarray(u_tape, np.float64)
        r = lib().orc_arena_game(C.byref(self._cfg_for(game_index if key is None else key)), C.c_void_p(cand.h), C.c_void_p(base.h), int(game_index), _p(ut),
                                 _p(T_table), _p(actions), _p(temps), C.byref(nply))
        return dict(result=r, nply=nply.value, actions=actions[:nply.value], temps=temps[:nply.value])

    def augment(self, state4, pi):
        n = self.n
        outs = np.zeros((4, 4, n, n), np.float32); outp = np.zeros((4, n, n), np.float32)
        lib().orc_augment(n, _p(np.ascontiguousarray(state4, np.float32)), _p(np.ascontiguousarray(pi, np.float32)),
                          _p(outs), _p(outp))
        return outs, outp
