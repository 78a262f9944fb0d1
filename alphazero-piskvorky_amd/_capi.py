"""ctypes binding of libaz_engine.so (include/az_engine.h).

This module is the only place the Python host layer touches the C-ABI.  It fails loudly when
the HIP library is missing or no GPU is present: there is no CPU fallback for the path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AZ_ENGINE_LIB") or os.path.join(_HERE, "libaz_engine.so")   # override: experiment builds

STATE_DICT_ORDER = [
    "conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "conv3.weight", "conv3.bias",
    "policy_conv.weight", "policy_conv.bias", "policy_fc.weight", "policy_fc.bias",
    "value_conv.weight", "value_conv.bias", "value_fc1.weight", "value_fc1.bias",
    "value_fc2.weight", "value_fc2.bias",
]

AZ_EVAL_NET, AZ_EVAL_SYNTHETIC = 0, 1
AZ_RES_NONE, AZ_RES_X, AZ_RES_O, AZ_RES_DRAW = 0, 1, 2, 3
AZ_AUG_NONE, AZ_AUG_REFERENCE4, AZ_AUG_DIHEDRAL8 = 1, 4, 8
AZ_MODEL_PLAIN, AZ_MODEL_RESNET = 0, 1
AZ_TRUNK_F32, AZ_TRUNK_BF16X3, AZ_TRUNK_F16X2 = 0, 1, 2

EXPORTS = [
    "az_create", "az_destroy", "az_last_error", "az_load_weights", "az_load_weights_resnet", "az_net_eval", "az_search", "az_search_callback", "az_selfplay",
    "az_selfplay_begin", "az_selfplay_step", "az_selfplay_end", "az_selfplay_games", "az_selfplay_records", "az_record_bytes", "az_selfplay_pack", "az_examples_from_packed",
    "az_examples_gather", "az_arena", "az_rules_replay", "az_rng_selfplay_tape", "az_rng_uniforms", "az_set_profiling", "az_set_subtree_reuse", "az_get_counters", "az_get_lanes", "az_get_persistent", "az_set_virtual_loss", "az_set_eval_cache",
    "az_set_trunk_mode", "az_get_trunk_mode", "az_set_leaf_symmetry", "az_emul_split",
    "az_dist_unique_id", "az_dist_init", "az_dist_rank", "az_dist_world", "az_dist_counts", "az_dist_gather_records",
    "az_dist_allreduce_sum", "az_dist_broadcast",
]
AZ_DIST_ID_BYTES = 128


class AzError(RuntimeError):
    pass


class az_config(C.Structure):
    _fields_ = [("board_size", C.c_int32), ("win_length", C.c_int32), ("num_simulations", C.c_int32),
                ("slots", C.c_int32), ("c_puct", C.c_double), ("dirichlet_alpha", C.c_double),
                ("dirichlet_weight", C.c_double), ("eval_kind", C.c_int32), ("device", C.c_int32),
                ("log_table", C.POINTER(C.c_float)), ("model", C.c_int32), ("engines", C.c_int32)]


class az_selfplay_args(C.Structure):
    _fields_ = [("seed0", C.c_uint64), ("num_games", C.c_int32), ("max_plies", C.c_int32),
                ("temperature_table", C.POINTER(C.c_double)), ("noise_tape", C.POINTER(C.c_double)),
                ("u_tape", C.POINTER(C.c_double)), ("tape_stride", C.c_int64)]


class az_counters(C.Structure):
    _fields_ = [("games", C.c_int64), ("plies", C.c_int64), ("records", C.c_int64), ("simulations", C.c_int64),
                ("expansions", C.c_int64), ("root_evals", C.c_int64), ("terminal_hits", C.c_int64),
                ("depth_sum", C.c_int64), ("steps", C.c_int64), ("seconds", C.c_double), ("nn_seconds", C.c_double),
                ("trunk_seconds", C.c_double), ("trunk_launches", C.c_int64), ("trunk_boards", C.c_int64),
                ("step_seconds", C.c_double), ("duplicate_leaves", C.c_int64), ("cache_lookups", C.c_int64),
                ("cache_hits", C.c_int64), ("tape_wait_seconds", C.c_double), ("tape_threads", C.c_int64),
                ("host_cpus", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class az_arena_args(C.Structure):
    _fields_ = [("seed0", C.c_uint64), ("num_games", C.c_int32), ("temperature_table", C.POINTER(C.c_double)),
                ("u_tape", C.POINTER(C.c_double))]


class az_arena_result(C.Structure):
    _fields_ = [("wins", C.c_int32), ("losses", C.c_int32), ("draws", C.c_int32), ("total", C.c_int32),
                ("win_rate", C.c_double)]


_LIB = None


def build(force=False):
    """Compile the HIP engine in-tree (hipcc --offload-arch=gfx950); cross-compiles without a GPU."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".hip", ".h", ".cpp"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "az_engine.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        if not os.path.exists("/opt/rocm/bin/hipcc") and os.path.exists(LIB_PATH):
            return LIB_PATH
        subprocess.check_call(["make", "-C", src_dir], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise AzError(f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # torch ships its own libamdhip64.so.7; load it FIRST so that this library binds to the same HIP
        # runtime (same soname) -- two runtimes in one process cannot both own the GPU.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the pure C-ABI user
            pass
        L = C.CDLL(LIB_PATH)
        L.az_last_error.restype = C.c_char_p
        L.az_last_error.argtypes = [C.c_void_p]
        L.az_create.argtypes = [C.POINTER(az_config), C.POINTER(C.c_void_p)]
        L.az_destroy.argtypes = [C.c_void_p]
        L.az_destroy.restype = None
        L.az_record_bytes.restype = C.c_int64
        L.az_record_bytes.argtypes = [C.c_void_p]
        L.az_rng_selfplay_tape.argtypes = [C.c_uint64, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        L.az_rng_uniforms.argtypes = [C.c_uint64, C.c_int, C.c_void_p]
        L.az_dist_init.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        L.az_dist_gather_records.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.az_dist_counts.argtypes = [C.c_void_p, C.c_void_p]
        L.az_dist_allreduce_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.az_dist_broadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.az_dist_unique_id.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def rng_selfplay_tape(seed, n, alpha=0.3, max_plies=0):
    nn = n * n
    plies = max_plies if 0 < max_plies < nn else nn
    total = sum(nn - m for m in range(plies))
    noise = np.zeros(total, np.float64)
    u = np.zeros(nn, np.float64)
    rc = lib().az_rng_selfplay_tape(int(seed), n, float(alpha), int(max_plies), _p(noise), _p(u))
    if rc:
        raise AzError(f"az_rng_selfplay_tape failed ({rc})")
    return noise, u[:plies]


def dist_unique_id():
    """AZ_DIST_ID_BYTES bytes naming a new RCCL communicator (az_dist_unique_id): create on one rank, hand to all."""
    buf = C.create_string_buffer(AZ_DIST_ID_BYTES)
    rc = lib().az_dist_unique_id(buf)
    if rc:
        raise AzError(f"az_dist_unique_id failed ({rc}): {lib().az_last_error(None).decode()}")
    return buf.raw


def rng_uniforms(seed, count):
    u = np.zeros(count, np.float64)
    rc = lib().az_rng_uniforms(int(seed), int(count), _p(u))
    if rc:
        raise AzError(f"az_rng_uniforms failed ({rc})")
    return u


class Engine:
    """One engine per GPU (az_create .. az_destroy).  engines = lanes inside the engine (az_config.engines): the slots
    are split over that many HIP streams driven by the library's own host threads; 0 lets the library choose."""

    def __init__(self, board_size, win_length, num_simulations, slots, c_puct=2.0, dirichlet_alpha=0.3,
                 dirichlet_weight=0.25, synthetic=False, device=0, log_table=None, model="plain", engines=0):
        self.n, self.k, self.S, self.slots = board_size, win_length, num_simulations, slots
        self.device = int(device)
        if model not in ("plain", "resnet"):
            raise ValueError("model must be 'plain' or 'resnet'")
        self.model = model
        self.nn = board_size * board_size
        self._log_table = None if log_table is None else np.ascontiguousarray(log_table, np.float32)
        if self._log_table is not None and len(self._log_table) < num_simulations + 1:
            raise ValueError("log_table must have num_simulations + 1 entries")
        cfg = az_config(board_size, win_length, num_simulations, slots, c_puct, dirichlet_alpha, dirichlet_weight,
                        AZ_EVAL_SYNTHETIC if synthetic else AZ_EVAL_NET, device,
                        None if self._log_table is None else self._log_table.ctypes.data_as(C.POINTER(C.c_float)),
                        AZ_MODEL_RESNET if model == "resnet" else AZ_MODEL_PLAIN, int(engines))
        self.h = C.c_void_p()
        rc = lib().az_create(C.byref(cfg), C.byref(self.h))
        if rc:
            raise AzError(f"az_create failed ({rc}): {lib().az_last_error(None).decode()}")
        self.record_bytes = int(lib().az_record_bytes(self.h))
        self.last_records = 0

    def _check(self, rc, what):
        if rc:
            raise AzError(f"{what} failed ({rc}): {lib().az_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            lib().az_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights ----
    def load_weights(self, state_dict, slot=0):
        if self.model == "resnet":
            from .net import fold_resnet_state_dict
            keep = state_dict if isinstance(state_dict, (list, tuple)) else fold_resnet_state_dict(state_dict)
            keep = [np.ascontiguousarray(t, dtype=np.float32) for t in keep]
            arr = (C.c_void_p * 24)(*[t.ctypes.data for t in keep])
            self._check(lib().az_load_weights_resnet(self.h, int(slot), arr), "az_load_weights_resnet")
            return
        keep = []
        for k in STATE_DICT_ORDER:
            t = state_dict[k]
            if hasattr(t, "detach"):
                t = t.detach().cpu().numpy()
            keep.append(np.ascontiguousarray(t, dtype=np.float32))
        arr = (C.c_void_p * 16)(*[t.ctypes.data for t in keep])
        self._check(lib().az_load_weights(self.h, int(slot), arr), "az_load_weights")

    # ---- net ----
    def net_eval(self, boards, players, lasts, slot=0):
        boards = np.ascontiguousarray(boards, np.uint8).reshape(-1, self.nn)
        cnt = boards.shape[0]
        players = np.ascontiguousarray(players, np.uint8).reshape(cnt)
        lasts = np.ascontiguousarray(lasts, np.int16).reshape(cnt)
        logits = np.zeros((cnt, self.nn), np.float32)
        policy = np.zeros((cnt, self.nn), np.float32)
        value = np.zeros(cnt, np.float32)
        self._check(lib().az_net_eval(self.h, int(slot), cnt, _p(boards), _p(players), _p(lasts), _p(logits),
                                      _p(policy), _p(value)), "az_net_eval")
        return logits, policy, value

    # ---- single search ----
    def search(self, board, player, last, temperature, noise=None, u=0.5, slot=0):
        board = np.ascontiguousarray(board, np.uint8).reshape(self.nn)
        nz = None if noise is None else np.ascontiguousarray(noise, np.float64)
        if nz is not None and len(nz) != int((board == 0).sum()):
            raise ValueError("noise must have one entry per legal cell")
        pi = np.zeros(self.nn, np.float32); N = np.zeros(self.nn, np.int32)
        W = np.zeros(self.nn, np.float64); P = np.zeros(self.nn, np.float32)
        a = C.c_int32(-1)
        self._check(lib().az_search(self.h, int(slot), _p(board), int(player), int(last), C.c_double(temperature),
                                    _dp(nz), C.c_double(u), _p(pi), C.byref(a), _p(N), _p(W), _p(P)), "az_search")
        return dict(action=int(a.value), pi=pi, N=N, W=W, P=P)

    _EVAL_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float))

    def search_callback(self, board, player, last, temperature, fn, noise=None, u=0.5):
        """MCTS.run with the evaluator on the host (az_search_callback): fn(cells uint8[n*n], player, last) -> (policy
        float32[n*n] or [n,n], value).  One host round trip per simulation: the compatibility path for arbitrary
        policy_value_fn callables (mcts.py:87-93)."""
        board = np.ascontiguousarray(board, np.uint8).reshape(self.nn)
        nz = None if noise is None else np.ascontiguousarray(noise, np.float64)
        if nz is not None and len(nz) != int((board == 0).sum()):
            raise ValueError("noise must have one entry per legal cell")
        nn = self.nn
        err = []

        def _cb(user, b, pl, la, pol, val):
            try:
                cells = np.ctypeslib.as_array(b, shape=(nn,)).copy()
                P, v = fn(cells, int(pl), int(la))
                if hasattr(P, "detach"):
                    P = P.detach().cpu().numpy()
                np.ctypeslib.as_array(pol, shape=(nn,))[:] = np.asarray(P, dtype=np.float32).reshape(nn)
                val[0] = float(v)
                return 0
            except Exception as ex:       # nothing may propagate through the C frames
                err.append(ex)
                return 1

        cb = self._EVAL_CB(_cb)
        pi = np.zeros(nn, np.float32); N = np.zeros(nn, np.int32)
        W = np.zeros(nn, np.float64); P = np.zeros(nn, np.float32)
        a = C.c_int32(-1)
        rc = lib().az_search_callback(self.h, _p(board), int(player), int(last), C.c_double(temperature), _dp(nz), C.c_double(u),
                                      cb, None, _p(pi), C.byref(a), _p(N), _p(W), _p(P))
        if err:
            raise err[0]
        self._check(rc, "az_search_callback")
        return dict(action=int(a.value), pi=pi, N=N, W=W, P=P)

    # ---- self-play ----
    def selfplay(self, num_games, seed0=0, max_plies=0, temperature_table=None, noise_tape=None, u_tape=None):
        T = None if temperature_table is None else np.ascontiguousarray(temperature_table, np.float64)
        if T is not None and len(T) < self.nn + 1:
            raise ValueError("temperature_table needs n*n + 1 entries")
        nz = None if noise_tape is None else np.ascontiguousarray(noise_tape, np.float64)
        ut = None if u_tape is None else np.ascontiguousarray(u_tape, np.float64)
        stride = 0 if nz is None else nz.shape[1]
        args = az_selfplay_args(int(seed0), int(num_games), int(max_plies), _dp(T), _dp(nz), _dp(ut), stride)
        c = az_counters()
        self._check(lib().az_selfplay(self.h, C.byref(args), C.byref(c)), "az_selfplay")
        self.last_records = int(c.records)
        self.last_games = int(num_games)
        return c.as_dict()

    def selfplay_begin(self, num_games, seed0=0, max_plies=0, temperature_table=None):
        T = None if temperature_table is None else np.ascontiguousarray(temperature_table, np.float64)
        args = az_selfplay_args(int(seed0), int(num_games), int(max_plies), _dp(T), None, None, 0)
        self._check(lib().az_selfplay_begin(self.h, C.byref(args)), "az_selfplay_begin")
        self.last_games = int(num_games)

    def selfplay_step(self, max_steps=1):
        active = C.c_int32(0)
        c = az_counters()
        self._check(lib().az_selfplay_step(self.h, int(max_steps), C.byref(active), C.byref(c)), "az_selfplay_step")
        return int(active.value), c.as_dict()

    def selfplay_end(self):
        c = az_counters()
        self._check(lib().az_selfplay_end(self.h, C.byref(c)), "az_selfplay_end")
        self.last_records = int(c.records)
        return c.as_dict()

    def games(self):
        nply = np.zeros(self.last_games, np.int32); res = np.zeros(self.last_games, np.int32)
        self._check(lib().az_selfplay_games(self.h, _p(nply), _p(res)), "az_selfplay_games")
        return nply, res

    def records(self):
        R, nn = self.last_records, self.nn
        out = dict(boards=np.zeros((R, nn), np.uint8), movers=np.zeros(R, np.uint8), lasts=np.zeros(R, np.int16),
                   actions=np.zeros(R, np.int16), pis=np.zeros((R, nn), np.float32), visits=np.zeros((R, nn), np.int32),
                   z=np.zeros(R, np.int8))
        self._check(lib().az_selfplay_records(self.h, _p(out["boards"]), _p(out["movers"]), _p(out["lasts"]),
                                              _p(out["actions"]), _p(out["pis"]), _p(out["visits"]), _p(out["z"])),
                    "az_selfplay_records")
        return out

    def _torch_sync(self):
        """The engine works on its own non-blocking HIP stream and synchronises it before every call returns; device
        buffers handed in by the caller (torch tensors on the engine's device) must be complete on the caller's side too."""
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.current_stream(self.device).synchronize()

    def pack_into(self, dev_ptr):
        self._torch_sync()
        self._check(lib().az_selfplay_pack(self.h, C.c_void_p(dev_ptr)), "az_selfplay_pack")

    def examples_from_packed(self, packed_ptr, records, aug, states_ptr, pis_ptr, z_ptr):
        self._torch_sync()
        self._check(lib().az_examples_from_packed(self.h, C.c_void_p(packed_ptr), C.c_int64(records), int(aug),
                                                  C.c_void_p(states_ptr), C.c_void_p(pis_ptr), C.c_void_p(z_ptr)),
                    "az_examples_from_packed")

    def examples_gather(self, packed_ptr, idx_ptr, sym_ptr, count, reference_pi, states_ptr, pis_ptr, z_ptr):
        self._torch_sync()
        self._check(lib().az_examples_gather(self.h, C.c_void_p(packed_ptr), C.c_void_p(idx_ptr), C.c_void_p(sym_ptr),
                                             int(count), int(reference_pi), C.c_void_p(states_ptr), C.c_void_p(pis_ptr),
                                             C.c_void_p(z_ptr)), "az_examples_gather")

    # ---- rules ----
    def rules_replay(self, actions):
        """Gomoku rules on the device for whole action lists (az_rules_replay): actions [games][max_len] int16, -1 padded."""
        acts = np.ascontiguousarray(actions, np.int16)
        if acts.ndim == 1:
            acts = acts.reshape(1, -1)
        G, L = acts.shape
        term = np.zeros((G, max(L, 1)), np.uint8); boards = np.zeros((G, self.nn), np.uint8)
        players = np.zeros(G, np.int32); results = np.zeros(G, np.int32); bad = np.zeros(G, np.int32)
        self._check(lib().az_rules_replay(self.h, G, L, _p(acts), _p(term), _p(boards), _p(players), _p(results), _p(bad)),
                    "az_rules_replay")
        return dict(term_before=term[:, :L].astype(bool), boards=boards, players=players, results=results, first_illegal=bad)

    # ---- arena ----
    def arena(self, num_games, seed0=0, temperature_table=None, u_tape=None):
        T = None if temperature_table is None else np.ascontiguousarray(temperature_table, np.float64)
        ut = None if u_tape is None else np.ascontiguousarray(u_tape, np.float64)
        args = az_arena_args(int(seed0), int(num_games), _dp(T), _dp(ut))
        res = az_arena_result()
        results = np.zeros(num_games, np.int32); actions = np.zeros((num_games, self.nn), np.int16)
        nply = np.zeros(num_games, np.int32)
        self._check(lib().az_arena(self.h, C.byref(args), C.byref(res), _p(results), _p(actions), _p(nply)), "az_arena")
        return dict(wins=res.wins, losses=res.losses, draws=res.draws, total=res.total, win_rate=res.win_rate,
                    results=results, actions=actions, nply=nply)

    def set_subtree_reuse(self, on):
        """Opt-in: keep the chosen child's subtree for the next ply (mcts.py:17-22 TODO); see include/az_engine.h."""
        self._check(lib().az_set_subtree_reuse(self.h, 1 if on else 0), "az_set_subtree_reuse")

    def set_virtual_loss(self, leaves):
        """Opt-in: virtual-loss batching, `leaves` leaves per search and evaluation batch (mcts.py:17-22 TODO); 1 = the
        reference's sequential loop.  See include/az_engine.h."""
        self._check(lib().az_set_virtual_loss(self.h, int(leaves)), "az_set_virtual_loss")

    def set_eval_cache(self, entries):
        """Opt-in: evaluation cache of `entries` positions in HBM (mcts.py:17,22 TODO), 0 = off; results are bit-identical."""
        self._check(lib().az_set_eval_cache(self.h, C.c_int64(int(entries))), "az_set_eval_cache")

    def set_leaf_symmetry(self, on):
        """Opt-in: every net evaluation of a search sees a pseudo-random dihedral symmetry of the position (SURVEY 8f-2's
        optional half); see include/az_engine.h."""
        self._check(lib().az_set_leaf_symmetry(self.h, 1 if on else 0), "az_set_leaf_symmetry")

    def set_trunk_mode(self, mode):
        """Opt-in: "bf16x3" / "f16x2" = fp32-emulating conv trunks on the 16-bit matrix cores (tolerance instead of
        bit-exactness; f16x2 is the faster one and has float16's range), "f32" = the default canonical float32 trunk.
        See include/az_engine.h."""
        code = {"f32": AZ_TRUNK_F32, "bf16x3": AZ_TRUNK_BF16X3, "f16x2": AZ_TRUNK_F16X2}.get(mode, mode)
        self._check(lib().az_set_trunk_mode(self.h, int(code)), "az_set_trunk_mode")

    def trunk_mode(self):
        return {AZ_TRUNK_BF16X3: "bf16x3", AZ_TRUNK_F16X2: "f16x2"}.get(int(lib().az_get_trunk_mode(self.h)), "f32")

    # ---- multi-GPU exchange inside the library (RCCL on the engine's stream; include/az_engine.h az_dist_*) ----
    def dist_init(self, unique_id, rank, world):
        if len(unique_id) != AZ_DIST_ID_BYTES:
            raise ValueError("unique_id must be AZ_DIST_ID_BYTES bytes (dist_unique_id())")
        self._check(lib().az_dist_init(self.h, C.c_char_p(bytes(unique_id)), int(rank), int(world)), "az_dist_init")

    def dist_rank(self):
        return int(lib().az_dist_rank(self.h))

    def dist_world(self):
        return int(lib().az_dist_world(self.h))

    def dist_counts(self):
        c = np.zeros(self.dist_world(), np.int64)
        self._check(lib().az_dist_counts(self.h, _p(c)), "az_dist_counts")
        return [int(x) for x in c]

    def dist_gather_records(self, dst, dev_ptr):
        """dst = rank that receives every rank's packed records (-1: all ranks); dev_ptr = device buffer (0 where not receiving)."""
        self._torch_sync()
        self._check(lib().az_dist_gather_records(self.h, int(dst), C.c_void_p(dev_ptr or None)), "az_dist_gather_records")

    def dist_allreduce_sum(self, values):
        v = np.ascontiguousarray(values, np.int64).copy()
        self._check(lib().az_dist_allreduce_sum(self.h, _p(v), len(v)), "az_dist_allreduce_sum")
        return [int(x) for x in v]

    def dist_broadcast(self, dev_ptr, nbytes, root=0):
        self._torch_sync()
        self._check(lib().az_dist_broadcast(self.h, C.c_void_p(dev_ptr), C.c_int64(int(nbytes)), int(root)), "az_dist_broadcast")

    def set_profiling(self, on):
        lib().az_set_profiling(self.h, 1 if on else 0)

    def lanes(self):
        """Number of lanes (streams + driver threads) inside the engine, az_config.engines after the library's choice."""
        return int(lib().az_get_lanes(self.h))

    def persistent(self):
        """Games per workgroup of the persistent search kernel in the last / open episode, 0 = lock-step pipeline."""
        return int(lib().az_get_persistent(self.h))

    def counters(self):
        c = az_counters()
        lib().az_get_counters(self.h, C.byref(c))
        return c.as_dict()


class MultiEngine(Engine):
    """An engine with an explicit number of lanes (kept for callers written against the host-thread version: the
    lanes, their streams and their driver threads now live inside the library, az_config.engines)."""

    def __init__(self, board_size, win_length, num_simulations, slots, engines=1, **kw):
        super().__init__(board_size, win_length, num_simulations, slots, engines=max(1, min(int(engines), slots)), **kw)
