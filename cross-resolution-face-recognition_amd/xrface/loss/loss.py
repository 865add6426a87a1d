"""Loss modules on the HIP path -- mirror of /root/reference loss/loss.py."""
import torch.nn as nn

from .. import ops


class MSELossFunc(nn.Module):
    """97 * mean((input - target)^2)   (reference loss/loss.py:7-15)."""

    def forward(self, input, target):
        return ops.mse_loss(input, target, 97.0)


class MSELoss_Landmark(nn.Module):
    """97 * mean((sum_c input[:, c] - target)^2)   (reference loss/loss.py:17-32)."""

    def forward(self, input, target):
        return ops.landmark_loss(input, target, 97.0)


class CrossEntropyLoss2d(nn.Module):
    """NLLLoss(log_softmax(outputs, 1), squeeze(targets))   (reference loss/loss.py:34-62)."""

    def __init__(self, weight=None):
        super().__init__()

    def forward(self, outputs, targets):
        return ops.cross_entropy_2d(outputs, targets)


class MSELoss(nn.Module):
    """nn.MSELoss drop-in used as ``criterion`` by distill_main.py:210 / train_FHN.py:189."""

    def forward(self, input, target):
        return ops.mse_loss(input, target, 1.0)


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss drop-in for (N, C) logits (main.py:132)."""

    def forward(self, input, target):
        return ops.cross_entropy(input, target)


class ArcFaceHead(nn.Module):
    """Build-defined ArcFace margin head (named by BASELINE.json's north_star, absent from the reference tree;
    l2_norm as in model_irse.py:16-20).  forward(emb, target) -> margin logits (N, classes); pair with
    CrossEntropyLoss.  Parity is pinned only against this repo's fp64 restatement (oracle/cpu_ref.py:arcface_logits)."""

    def __init__(self, in_features=512, classes=10572, s=64.0, m=0.5):
        super().__init__()
        import torch
        self.weight = nn.Parameter(torch.empty(classes, in_features))
        nn.init.xavier_uniform_(self.weight)
        self.s, self.m = s, m

    def forward(self, emb, target):
        return ops.arcface_logits(emb, self.weight, target, self.s, self.m)


def MMD(source, target, sigmas=(1.0, 2.0, 4.0, 8.0, 16.0)):
    """``from loss.loss import MMD`` (Face_Hallucination_sub_Net.py:25) -- undefined upstream; build-defined as the biased
    multi-bandwidth Gaussian-kernel MMD^2 (fp64 restatement: oracle/cpu_ref.py:mmd_gaussian)."""
    return ops.mmd(source, target, sigmas)
