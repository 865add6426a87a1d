"""Verification metrics on the HIP path -- mirror of /root/reference utils/utils.py:14-87,132-169.

``calculate_roc`` keeps the reference signature and return tuple.  The pair distances and the
2*K*T threshold re-scans (80 000 numpy passes in the reference) collapse to one HBM pass for the
distances plus one histogram pass; fold membership is drawn exactly like the reference does it
(sklearn KFold(shuffle=True) on the global numpy RNG) unless ``fold_id`` is given.  The unused
O(P^2 log P) ``margin_list`` (utils/utils.py:47-49) is dropped.
"""
from __future__ import annotations

import numpy as np
import torch

from .._lib import lib, ptr, stream


def pair_dist(embeddings1, embeddings2, device=None):
    """dist_i = sum_d (e1[i,d] - e2[i,d])^2 on the GPU; accepts numpy arrays or tensors."""
    dev = device or torch.device("cuda", torch.cuda.current_device())
    e1 = torch.as_tensor(embeddings1, dtype=torch.float32).to(dev).contiguous()
    e2 = torch.as_tensor(embeddings2, dtype=torch.float32).to(dev).contiguous()
    assert e1.shape == e2.shape and e1.dim() == 2
    dist = torch.empty(e1.shape[0], dtype=torch.float32, device=dev)
    lib.xr_pairdist_l2(ptr(e1), ptr(e2), ptr(dist), e1.shape[0], e1.shape[1], stream())
    return dist


def roc_histograms(dist, actual_issame, fold_id, thresholds, nrof_folds):
    """int64 [F][2][T+1]: per fold / label, the count of pairs whose first predicted-same threshold index is j."""
    dev = dist.device
    thr = torch.as_tensor(np.asarray(thresholds, dtype=np.float32)).to(dev)
    same = torch.as_tensor(np.asarray(actual_issame).astype(np.uint8)).to(dev)
    fid = torch.as_tensor(np.asarray(fold_id).astype(np.int32)).to(dev)
    T = thr.numel()
    hist = torch.zeros((nrof_folds, 2, T + 1), dtype=torch.int64, device=dev)
    lib.xr_roc_hist(ptr(dist), ptr(same), ptr(fid), ptr(thr), ptr(hist), dist.numel(), T, nrof_folds, stream())
    return hist


def _rates(tp, fp, tn, fn, n):
    with np.errstate(divide="ignore", invalid="ignore"):
        tpr = np.where(tp + fn == 0, 0.0, tp.astype(np.float64) / np.maximum(tp + fn, 1))
        fpr = np.where(fp + tn == 0, 0.0, fp.astype(np.float64) / np.maximum(fp + tn, 1))
    acc = (tp + tn).astype(np.float64) / float(n)
    return tpr, fpr, acc


def calculate_accuracy(threshold, dist, actual_issame):
    """(tpr, fpr, acc) at ONE threshold, predict_same = dist < threshold (reference utils/utils.py:14-24; host arrays).
    Expressed through the same confusion counts the histogram path produces for a whole threshold grid."""
    dist = np.asarray(dist)
    same = np.asarray(actual_issame).astype(bool)
    pred = dist < threshold
    tp = np.array([np.count_nonzero(pred & same)])
    fp = np.array([np.count_nonzero(pred & ~same)])
    fn = np.array([np.count_nonzero(same)]) - tp
    tn = np.array([np.count_nonzero(~same)]) - fp
    tpr, fpr, acc = _rates(tp, fp, tn, fn, dist.size)
    return float(tpr[0]), float(fpr[0]), float(acc[0])


def calculate_roc(thresholds, embeddings1, embeddings2, actual_issame, nrof_folds=50, pca=0, fold_id=None):
    """reference utils/utils.py:26-87 -> (mean tpr[T], mean fpr[T], mean accuracy, best_thresholds[K])."""
    if pca != 0:
        raise NotImplementedError("pca > 0 is not on the hot path (distill_main.py:121-136 always passes pca=0)")
    assert embeddings1.shape[0] == embeddings2.shape[0] and embeddings1.shape[1] == embeddings2.shape[1]
    thresholds = np.asarray(thresholds)
    assert np.all(np.diff(thresholds) > 0), "thresholds must be ascending"
    issame = np.asarray(actual_issame).astype(bool)
    nrof_pairs = min(len(issame), embeddings1.shape[0])
    if nrof_folds < 2:
        # the reference builds KFold(n_splits=nrof_folds) (utils/utils.py:31), which rejects a single fold: there is no train split
        raise ValueError("k-fold cross-validation requires at least one train/test split by setting nrof_folds=2 or more, "
                         f"got nrof_folds={nrof_folds}.")
    if fold_id is None:
        from sklearn.model_selection import KFold
        fold_id = np.empty(nrof_pairs, dtype=np.int32)
        for f, (_, test_set) in enumerate(KFold(n_splits=nrof_folds, shuffle=True).split(np.arange(nrof_pairs))):
            fold_id[test_set] = f
    dist = pair_dist(embeddings1[:nrof_pairs], embeddings2[:nrof_pairs])
    hist = roc_histograms(dist, issame[:nrof_pairs], fold_id, thresholds, nrof_folds)
    tpr, fpr, acc, best = roc_sweep(hist, len(thresholds), nrof_folds)
    return tpr, fpr, acc.mean(), thresholds[best].astype(np.float64)


def roc_sweep(hist, T, nrof_folds):
    """The K-fold sweep of utils/utils.py:51-83 on the device (xr_roc_sweep, one workgroup) over the int64 histogram [F][2][T+1]
    (overwritten with its prefix sums): (mean tpr [T], mean fpr [T], per-fold accuracy [F], per-fold best threshold index [F]).
    Integer confusion counts, fp64 rates: bit-identical to the numpy evaluation (oracle/cpu_ref.py:calculate_roc)."""
    dev = hist.device
    out = torch.empty(2 * T + nrof_folds, dtype=torch.float64, device=dev)
    best = torch.empty(nrof_folds, dtype=torch.int32, device=dev)
    lib.xr_roc_sweep(ptr(hist), T, nrof_folds, ptr(out), ptr(out[T:]), ptr(out[2 * T:]), ptr(best), stream())
    res = out.cpu().numpy()
    return res[:T], res[T:2 * T], res[2 * T:], best.cpu().numpy().astype(np.int64)


class AverageMeter(object):
    """Running mean of a scalar stream (reference utils/utils.py:132-147: attributes val / avg / sum / count,
    methods reset() / update(val, n))."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val, self.sum, self.count = 0, 0, 0

    @property
    def avg(self):
        return self.sum / self.count if self.count else 0

    def update(self, val, n=1):
        self.val = val
        self.sum, self.count = self.sum + val * n, self.count + n


def accuracy(output, target, topk=(1,)):
    """Top-k precision in percent, one 1-element tensor per k (reference utils/utils.py:156-169; host-side bookkeeping)."""
    n = target.shape[0]
    top = output.float().topk(max(topk), dim=1).indices            # (N, maxk), best first
    hit = top.eq(target.reshape(-1, 1))                             # (N, maxk)
    return [hit[:, :k].any(dim=1).float().sum().reshape(1) * (100.0 / n) for k in topk]
