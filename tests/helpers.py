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


def check_against(store, key, t, rtol, what=None):
    """Compare tensor ``t`` with fixture entry ``key`` (full tensor or ``@digest``)."""
    what = what or key
    if key in store.files:
        e = rel_err(t, store[key])
        assert e <= rtol, f"{what}: max error {e:.3e} (rel to max-abs) > {rtol}"
        return e
    return compare_digest(t, store[key + "@digest"], rtol, what)
