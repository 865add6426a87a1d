"""Loader-side tensor synthesis on the HIP path: the two per-sample computations of the reference's Dataset classes
that produce network inputs / targets -- /root/reference SUPER_RESOLUTION/FHN_loader.py:65-66 (low-resolution image: PIL bicubic down
and up), :119-137 and helen_loader.py:124-143 (summed-Gaussian landmark heat-map).  Same function names and argument meaning as the
reference methods (``generate_hm`` / ``gaussian_k``); batched, device tensors in and out, no CPU fallback."""
from __future__ import annotations

import numpy as np
import torch

from .._lib import lib, ptr, stream


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def lr_from_hr(hr_u8, scale=8, base=128, normalize=True):
    """FHN_loader.py:65-66 for a batch of uint8 crops [N][H][W][3] (what ``sr_img.crop(...)`` holds):
    ``sr_img.resize((int(base / scale),) * 2).resize((W, H), Image.BICUBIC)`` -- bit-identical to PIL.  ``scale``: one number or one
    per image (the loader draws it from scale_list).  Returns (lr uint8 [N][H][W][3], lr float32 [N][3][H][W] after ToTensor +
    Normalize(0.5, 0.5), or None when normalize is False)."""
    hr = torch.as_tensor(hr_u8)
    assert hr.dtype == torch.uint8 and hr.dim() == 4 and hr.shape[3] == 3, "hr_u8: uint8 [N][H][W][3]"
    hr = hr.to(_dev()).contiguous()
    N, H, W, _ = hr.shape
    low = np.broadcast_to(np.asarray([int(base / s) for s in np.atleast_1d(scale)], dtype=np.int32), (N,)).copy()
    if low.min() * 16 < max(H, W) or low.max() > min(H, W):
        raise ValueError(f"lr_from_hr: low-resolution edge {low.min()}..{low.max()} outside [{-(-max(H, W) // 16)}, {min(H, W)}]")
    low_d = torch.from_numpy(low).to(hr.device)
    lr_u8 = torch.empty_like(hr)
    lr_n = torch.empty((N, 3, H, W), dtype=torch.float32, device=hr.device) if normalize else None
    lib.xr_lr_synth(ptr(hr), ptr(low_d), int(low.max()), ptr(lr_u8), ptr(lr_n), N, H, W, stream())
    return lr_u8, lr_n


def generate_hm(height, width, landmark, s=2.0):
    """FHN_loader.py:119-129 / helen_loader.py:132-143: sum of one Gaussian per landmark (x, y).  ``landmark``: [L][2] -> float32
    [height][width], or a batch [N][L][2] -> [N][height][width] (device tensor)."""
    lm = torch.as_tensor(np.asarray(landmark, dtype=np.float64) if not isinstance(landmark, torch.Tensor) else landmark).double()
    single = lm.dim() == 2
    if single:
        lm = lm[None]
    assert lm.dim() == 3 and lm.shape[2] == 2, "landmark: [L][2] or [N][L][2]"
    lm = lm.to(_dev()).contiguous()
    N, L, _ = lm.shape
    hm = torch.empty((N, height, width), dtype=torch.float32, device=lm.device)
    lib.xr_heatmap(ptr(lm), ptr(hm), N, L, height, width, float(s), stream())
    return hm[0] if single else hm


def gaussian_k(x0, y0, sigma, width=224, height=224):
    """FHN_loader.py:131-137: one bump = the heat-map of a single landmark."""
    return generate_hm(height, width, [[x0, y0]], s=sigma)
