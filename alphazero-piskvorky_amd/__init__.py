"""MI355X-native batched self-play engine for AlphaZero-Piskvorky (Gomoku).

Host layer mirroring the reference's Python seams over the C-ABI in
include/az_engine.h.  The compute path is HIP only; nothing here falls back to the CPU.

    from alphazero_piskvorky_amd.self_play import SelfPlayManager      # self_play.py:80-159
    from alphazero_piskvorky_amd.evaluator import ModelEvaluator       # evaluator.py:22-122
    from alphazero_piskvorky_amd.mcts import MCTS                      # mcts.py:86-183
    from alphazero_piskvorky_amd.controller import make_policy_value_fn, NeuralNetworkController
"""
from . import _capi  # noqa: F401
from ._capi import AzError, Engine, MultiEngine  # noqa: F401
