"""xrface -- MI355X-native (gfx950) hot path of the cross-resolution face-recognition system.

Host side of the C ABI in include/xrface.h: nn.Module mirrors of the reference's model/loss/eval
interfaces (same class names, constructor arguments, forward tuples, state_dict keys) whose compute
runs on hand-written HIP kernels.  See DESIGN.md.
"""
from .ops import get_compute_dtype, invalidate_weight_cache, set_compute_dtype, set_deterministic  # noqa: F401

__all__ = ["set_compute_dtype", "get_compute_dtype", "invalidate_weight_cache", "set_deterministic"]
