"""Loss modules of the SR variant on the HIP path -- mirror of /root/reference SUPER_RESOLUTION/loss/loss.py."""
import torch.nn as nn

from .. import ops
from .loss import CrossEntropyLoss2d  # noqa: F401  (identical to the root variant, SUPER_RESOLUTION/loss/loss.py:20-26)


class Landmark_Loss(nn.Module):
    """SUPER_RESOLUTION/loss/loss.py:7-17.  Upstream ``forward`` raises (``torch.pow`` without an exponent, :16); its evident
    intent -- the commented line above it and the root variant loss/loss.py:28-31 -- is the mean squared error between the
    channel-summed prediction and the ONE summed-Gaussian heat-map: mean((sum_c input[:, c] - target)^2)."""

    def forward(self, input, target):
        return ops.landmark_loss(input, target, 1.0)
