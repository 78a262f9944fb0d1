"""MI355X-native batched self-play engine for AlphaZero-Piskvorky (Gomoku).

Host layer mirroring the reference's Python seams over the C-ABI in
include/az_engine.h.  The compute path is HIP only; nothing here falls back to the CPU.
"""
from . import _capi  # noqa: F401
from ._capi import AzError, Engine  # noqa: F401
