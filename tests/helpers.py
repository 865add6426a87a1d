"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

from oracle.digest import compare_digest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_gold(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def rel_err(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b)).double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


ERRORS = []  # (what, measured error, tolerance) -- dumped by conftest at session end


def grad_floor(store, prefix):
    """1e-2 x the largest fixture gradient under ``prefix``: the error floor for gradients that are
    mathematically zero (a bias feeding a train-mode norm) and therefore pure rounding noise on both sides."""
    m = 0.0
    for k in store.files:
        if k.startswith(prefix):
            m = max(m, float(store[k][3]) if k.endswith("@digest") else float(np.abs(store[k]).max()))
    return 1e-2 * m


def check_against(store, key, t, rtol, what=None, floor=0.0):
    """Compare tensor ``t`` with fixture entry ``key`` (full tensor or ``@digest``)."""
    what = what or key
    if key in store.files:
        ref = store[key]
        if floor > 0.0 and float(np.abs(ref).max()) < floor:
            tt = np.asarray(t.detach().float().cpu())
            e = float(np.abs(tt - ref).max()) / floor
        else:
            e = rel_err(t, ref)
        ERRORS.append((what, e, rtol))
        assert e <= rtol, f"{what}: max error {e:.3e} (rel to max-abs) > {rtol}"
        return e
    try:
        e = compare_digest(t, store[key + "@digest"], rtol, what)
    except AssertionError:
        ERRORS.append((what, float("nan"), rtol))
        raise
    ERRORS.append((what, e, rtol))
    return e
