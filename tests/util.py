"""Shared test helpers (no reference imports; make_golden's top-level helpers are build-owned)."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
_mg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mg)
build_weights = _mg.build_weights
synth_eval_codes = _mg.synth_eval_codes
fmix32 = _mg.fmix32

SIZES = [(5, 4), (9, 5), (15, 5)]


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def weights_from_fixture(n, tag):
    """state_dict (numpy) for a weight set named in a fixture: 'seeded' or a 5x5 checkpoint tag."""
    if tag == "seeded":
        return build_weights(n)
    z = load(f"net_{n}.npz")
    pre = tag + "__"
    return {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
