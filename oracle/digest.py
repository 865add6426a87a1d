"""TEST INFRASTRUCTURE -- tensor digests for fixtures too large to commit whole.

digest(t) -> float64 vector [numel, sum, sum of squares, max-abs, 512 strided samples...].
compare_digest() checks a candidate tensor against a stored digest with a tolerance that is
relative to the tensor's max-abs (samples) or to sqrt(numel)*rms (sums)."""
from __future__ import annotations

import numpy as np
import torch

NSAMP = 512


def _flat64(t):
    if isinstance(t, torch.Tensor):
        t = t.detach().to("cpu", torch.float64).contiguous().numpy()
    return np.asarray(t, dtype=np.float64).reshape(-1)


def sample_index(n):
    return np.unique(np.linspace(0, n - 1, NSAMP).astype(np.int64))


def digest(t) -> np.ndarray:
    v = _flat64(t)
    idx = sample_index(v.size)
    head = np.array([v.size, v.sum(), np.square(v).sum(), np.abs(v).max()], dtype=np.float64)
    return np.concatenate([head, v[idx]])


def compare_digest(t, d, rtol=1e-3, what="tensor"):
    """Raise AssertionError unless ``t`` matches digest ``d`` within ``rtol`` (relative to max-abs)."""
    v = _flat64(t)
    n, s, ss, amax = int(d[0]), d[1], d[2], d[3]
    assert v.size == n, f"{what}: numel {v.size} != {n}"
    idx = sample_index(n)
    scale = max(amax, 1e-30)
    err = np.abs(v[idx] - d[4:]).max() / scale
    assert err <= rtol, f"{what}: sampled max error {err:.3e} (rel to max-abs {amax:.3e}) > {rtol}"
    rms = np.sqrt(ss / n)
    # the sum of n values each off by <= rtol*scale can drift by n*rtol*scale in the worst case; errors are
    # not that coherent -- allow sqrt(n)-scaled drift plus a small coherent part
    tol_sum = rtol * (np.sqrt(n) * rms * 4.0 + 0.05 * n * rms) + 1e-12
    assert abs(v.sum() - s) <= tol_sum, f"{what}: sum {v.sum():.6e} vs {s:.6e} (tol {tol_sum:.3e})"
    e_ss = abs(np.square(v).sum() - ss) / max(ss, 1e-30)
    assert e_ss <= 4 * rtol, f"{what}: sum-of-squares rel err {e_ss:.3e}"
    return float(err)
