"""TEST INFRASTRUCTURE (oracle side) -- deterministic, counter-based generators.

Shared by the oracle, the golden-fixture generator, the parity tests and the
benchmark so that the CPU container and the GPU box build bit-identical
weights and synthetic inputs without ever shipping checkpoints:

* ``det_state_dict`` fills a ``state_dict`` template *by key name* (splitmix64
  of ``fnv1a(key) + element index``), so 175 MB of IR-50 weights never need
  committing (SURVEY.md section 7 step 1, section 8c "How the oracle is used").
* ``synth_faces`` / ``synth_heatmap`` / ``synth_parsing`` / ``synth_pairs`` follow
  the tensor contracts of the reference loaders:
  ``SUPER_RESOLUTION/FHN_loader.py:30-35,65-66,100,119-137`` and
  ``helen_loader.py:45,118``.

Nothing here is on the product path.
"""
from __future__ import annotations

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 counters."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(key: str, n: int, seed: int = 0) -> np.ndarray:
    """n float64 uniforms in [0,1) keyed by (key, seed)."""
    base = np.uint64((fnv1a64(key) + 0x632BE59BD9B4E019 * (seed + 1)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) + base) & _MASK
    bits = splitmix64(ctr) >> np.uint64(11)
    return bits.astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(key: str, n: int, seed: int = 0) -> np.ndarray:
    """Box-Muller on two uniform streams (float64)."""
    u1 = uniform01(key + "/u1", n, seed)
    u2 = uniform01(key + "/u2", n, seed)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def det_tensor(key: str, shape, dtype=torch.float32, seed: int = 0) -> torch.Tensor:
    """Deterministic fill rule chosen from the state_dict key alone."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    u = uniform01(key, n, seed)
    if leaf == "running_mean":
        v = (u - 0.5) * 0.2
    elif leaf == "running_var":
        v = 0.5 + u
    elif leaf == "bias":
        v = (u - 0.5) * 0.2
    elif len(shape) >= 2:  # conv / linear / deconv weight: uniform, variance 1/fan_in
        fan_in = int(np.prod(shape[1:]))
        bound = np.sqrt(3.0 / fan_in)
        v = (2.0 * u - 1.0) * bound
    else:  # 1-D ".weight": norm gamma or PReLU slope -- both get (0.25, 1.0)
        v = 0.25 + 0.75 * u
    return torch.from_numpy(v.reshape(shape)).to(dtype)


def det_state_dict(template: dict, seed: int = 0) -> dict:
    """Fill every entry of ``template`` (name -> tensor) deterministically."""
    out = {}
    for k, t in template.items():
        out[k] = det_tensor(k, t.shape, t.dtype if t.dtype.is_floating_point else torch.int64, seed)
        if t.dtype.is_floating_point:
            out[k] = out[k].to(t.dtype)
    return out


# --------------------------------------------------------------------------- inputs

def synth_faces(n: int, size: int = 112, seed: int = 1, start: int = 0) -> torch.Tensor:
    """hr_img: N x 3 x size x size fp32 in [-1,1]: a few low-frequency Gaussian blobs per
    channel + 0.1 * white noise (range of Normalize(0.5,0.5), FHN_loader.py:30-35)."""
    ys = np.arange(size, dtype=np.float64)[:, None]
    xs = np.arange(size, dtype=np.float64)[None, :]
    out = np.empty((n, 3, size, size), dtype=np.float64)
    for i in range(n):
        g = start + i
        p = uniform01(f"face/{g}/blobs", 3 * 6 * 4, seed).reshape(3, 6, 4)
        for c in range(3):
            img = np.zeros((size, size), dtype=np.float64)
            for b in range(6):
                cx, cy, sg, amp = p[c, b]
                cx = cx * size
                cy = cy * size
                sg = (0.08 + 0.25 * sg) * size
                amp = 2.0 * amp - 1.0
                img += amp * np.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2.0 * sg * sg))
            noise = uniform01(f"face/{g}/noise/{c}", size * size, seed).reshape(size, size)
            img = img + 0.1 * (2.0 * noise - 1.0)
            out[i, c] = img
    out = np.clip(out / 1.6, -1.0, 1.0)
    return torch.from_numpy(out.astype(np.float32))


def synth_lr_from_hr(hr: torch.Tensor, scale: int = 7) -> torch.Tensor:
    """lr_img: average-pool ``scale``x then bicubic back to the hr size on the host
    (mirrors the down/up-sampling of FHN_loader.py:65-66; 112/7 = 16 px)."""
    import torch.nn.functional as F
    lo = F.avg_pool2d(hr, scale)
    up = F.interpolate(lo, size=hr.shape[-2:], mode="bicubic", align_corners=False)
    return up.clamp_(-1.0, 1.0).contiguous()


def synth_heatmap(n: int, size: int, n_landmarks: int, sigma: float, seed: int = 2, start: int = 0) -> torch.Tensor:
    """N x size x size sum of Gaussians at seeded landmark coordinates
    (FHN_loader.py:119-137 generate_hm / gaussian_k)."""
    ys = np.arange(size, dtype=np.float64)[:, None]
    xs = np.arange(size, dtype=np.float64)[None, :]
    out = np.zeros((n, size, size), dtype=np.float64)
    for i in range(n):
        pts = uniform01(f"lmk/{start + i}", 2 * n_landmarks, seed).reshape(n_landmarks, 2)
        pts = (0.15 + 0.7 * pts) * size
        for x0, y0 in pts:
            out[i] += np.exp(-((xs - x0) ** 2 + (ys - y0) ** 2) / (2.0 * sigma * sigma))
    return torch.from_numpy(out.astype(np.float32))


def synth_parsing(n: int, size: int, n_classes: int, seed: int = 2, start: int = 0) -> torch.Tensor:
    """N x 1 x size x size int64 labels in [0, n_classes): blocky regions
    (helen_loader.py:45 / FHN_loader.py:39 label range)."""
    out = np.empty((n, 1, size, size), dtype=np.int64)
    cell = max(1, size // 7)
    g = (size + cell - 1) // cell
    for i in range(n):
        u = uniform01(f"parse/{start + i}", g * g, seed).reshape(g, g)
        lab = np.floor(u * n_classes).astype(np.int64)
        out[i, 0] = np.kron(lab, np.ones((cell, cell), dtype=np.int64))[:size, :size]
    return torch.from_numpy(out)


def synth_labels(n: int, n_classes: int, seed: int = 2, start: int = 0) -> torch.Tensor:
    u = uniform01(f"cls/{start}", n, seed)
    return torch.from_numpy(np.floor(u * n_classes).astype(np.int64))


def synth_pairs(p: int, dim: int = 512, seed: int = 0):
    """Verification pairs for config 5: e1 ~ N(0,1); label ~ Bernoulli(.5);
    e2 = e1 + 0.5 N(0,1) if same else fresh N(0,1).  Returns (e1, e2, issame)."""
    e1 = normal("pairs/e1", p * dim, seed).reshape(p, dim)
    nz = normal("pairs/e2", p * dim, seed).reshape(p, dim)
    same = uniform01("pairs/lab", p, seed) < 0.5
    e2 = np.where(same[:, None], e1 + 0.5 * nz, nz)
    return e1.astype(np.float32), e2.astype(np.float32), same
