"""TEST INFRASTRUCTURE -- CPU restatement (oracle) of the reference hot path.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
It restates, in a functional style over a plain ``state_dict`` (name -> tensor),
what the reference's ``nn.Module`` graphs compute, using stock fp32 ``torch`` CPU
ops.  It is pinned against the imported reference modules by
``oracle/make_golden.py`` (run in the build container, where ``/root/reference``
exists) and against the committed fixtures in ``tests/golden`` by
``tests/test_oracle_golden.py``.

Reference citations (relative to /root/reference):
  residual block            model/FSRnet.py:75-98
  hourglass BasicBlock      model/FSRnet.py:105-135
  Hourglass                 model/FSRnet.py:176-215
  Course_SR_Network         model/FSRnet.py:308-340
  Fine_SR_Encoder           model/FSRnet.py:342-379
  Prior_Estimation_Network  model/FSRnet.py:381-426
  Fine_SR_Decoder           model/FSRnet.py:428-459
  bottleneck_IR(_SE)        SUPER_RESOLUTION/model/model_irse.py:23-91
  Backbone                  SUPER_RESOLUTION/model/model_irse.py:129-189
  ResNet / BasicBlock       model/resnet.py:18-47,152-225
  losses                    loss/loss.py:7-62
  pair distance + ROC       utils/utils.py:14-87
  KD step                   distill_main.py:42-96
  C4 composed step          SUPER_RESOLUTION/train_FHN.py:274-279 + distill_main.py:59-70
  LR synthesis / heat-maps  SUPER_RESOLUTION/FHN_loader.py:65-66,119-137
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-5
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------- helpers

def _inorm(sd, p, x):
    """InstanceNorm2d, no running stats; affine iff the key exists (model/FSRnet.py:81,112)."""
    w = sd.get(p + ".weight")
    b = sd.get(p + ".bias")
    return F.instance_norm(x, None, None, w, b, True, 0.0, EPS)


def _bnorm(sd, p, x, train, new_stats=None):
    """BatchNorm (1d/2d), eps 1e-5, momentum 0.1.  In train mode the updated running
    statistics are written to ``new_stats`` (dict) instead of mutating ``sd``."""
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    if train:
        rm, rv = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], train, BN_MOMENTUM, EPS)
    if train and new_stats is not None:
        new_stats[p + ".running_mean"] = rm
        new_stats[p + ".running_var"] = rv
    return y


def _conv(sd, p, x, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride, padding)


# ----------------------------------------------------------------------------- FSRNet (root)

def residual_block(sd, p, x):
    """PReLU_out(IN2(conv2(PReLU(IN1(conv1(x))))) + x) -- model/FSRnet.py:90-98."""
    y = _conv(sd, p + ".conv1", x, 1, 1)
    y = F.prelu(_inorm(sd, p + ".in1", y), sd[p + ".relu.weight"])
    y = _conv(sd, p + ".conv2", y, 1, 1)
    y = _inorm(sd, p + ".in2", y) + x
    return F.prelu(y, sd[p + ".relu_out.weight"])


def _shared_trunk(sd, p, x, times=3, blocks=3):
    """The same ``blocks``-block Sequential applied ``times`` times (model/FSRnet.py:331-333)."""
    for _ in range(times):
        for j in range(blocks):
            x = residual_block(sd, f"{p}.{j}", x)
    return x


def coarse_sr(sd, x, p=""):
    """Course_SR_Network.forward -> (feat64, coarse_img) (model/FSRnet.py:328-340)."""
    y = F.prelu(_inorm(sd, p + "bn_mid", _conv(sd, p + "conv_input", x, 1, 1)), sd[p + "relu.weight"])
    y = _shared_trunk(sd, p + "residual", y)
    y = _inorm(sd, p + "bn_mid", y)
    return y, _conv(sd, p + "conv_mid", y, 1, 1)


def fine_encoder(sd, x, p=""):
    """Fine_SR_Encoder.forward (model/FSRnet.py:359-379): conv7x7 s4 -> IN -> PReLU -> trunk -> conv_end -> IN -> PReLU."""
    y = F.prelu(_inorm(sd, p + "bn_mid", _conv(sd, p + "conv_input", x, 4, 3)), sd[p + "relu.weight"])
    y = _shared_trunk(sd, p + "residual", y)
    return F.prelu(_inorm(sd, p + "bn_mid", _conv(sd, p + "conv_end", y, 1, 1)), sd[p + "relu.weight"])


def hg_basic_block(sd, p, x):
    """Hourglass BasicBlock (model/FSRnet.py:119-135): non-affine IN, one PReLU used twice."""
    a = sd[p + ".relu.weight"]
    y = F.prelu(_inorm(sd, p + ".bn1", _conv(sd, p + ".conv1", x, 1, 1)), a)
    y = _inorm(sd, p + ".bn2", _conv(sd, p + ".conv2", y, 1, 1)) + x
    return F.prelu(y, a)


def _hg_seq(sd, p, x, nblocks=2):
    for j in range(nblocks):
        x = hg_basic_block(sd, f"{p}.{j}", x)
    return x


def hourglass(sd, p, x, n):
    """Hourglass._hour_glass_forward (model/FSRnet.py:200-212), depth n (keys hg.<n-1>.<branch>)."""
    up1 = _hg_seq(sd, f"{p}.{n - 1}.0", x)
    low = F.max_pool2d(x, 2, stride=2)
    low = _hg_seq(sd, f"{p}.{n - 1}.1", low)
    low = hourglass(sd, p, low, n - 1) if n > 1 else _hg_seq(sd, f"{p}.{n - 1}.3", low)
    low = _hg_seq(sd, f"{p}.{n - 1}.2", low)
    return up1 + F.interpolate(low, scale_factor=2)


def prior_net(sd, x, p=""):
    """Prior_Estimation_Network.forward -> (feat128, landmark97, parsing11) (model/FSRnet.py:408-426)."""
    y = F.prelu(_inorm(sd, p + "bn", _conv(sd, p + "conv", x, 4, 3)), sd[p + "relu.weight"])
    for j in range(3):
        y = residual_block(sd, f"{p}residual.{j}", y)
    y = hourglass(sd, p + "hg.hg", y, 2)
    return y, _conv(sd, p + "fc_landmark", y), _conv(sd, p + "fc", y)


def fine_decoder(sd, x, p=""):
    """Fine_SR_Decoder.forward (model/FSRnet.py:448-459)."""
    a = sd[p + "relu.weight"]
    y = F.prelu(_inorm(sd, p + "bn_mid", _conv(sd, p + "conv_input", x, 1, 1)), a)
    y = F.conv_transpose2d(y, sd[p + "deconv.weight"], sd[p + "deconv.bias"], stride=4, padding=2, output_padding=1)
    y = F.prelu(_inorm(sd, p + "bn_mid", y), a)
    y = _shared_trunk(sd, p + "residual", y)
    y = _inorm(sd, p + "bn_mid", y)
    return _conv(sd, p + "conv_out", y, 1, 1)


def fhn_forward(sds, lr_img):
    """112x112 composition of the four generators as SUPER_RESOLUTION/train_FHN.py:274-279 does:
    coarse_img -> {prior, encoder} -> cat(prior_feat, enc_feat) -> decoder.
    ``sds`` = dict(coarse=..., prior=..., encoder=..., decoder=...) of state_dicts."""
    _, coarse_img = coarse_sr(sds["coarse"], lr_img)
    pf, lmk, par = prior_net(sds["prior"], coarse_img)
    ef = fine_encoder(sds["encoder"], coarse_img)
    sr = fine_decoder(sds["decoder"], torch.cat((pf, ef), 1))
    return sr, coarse_img, lmk, par


def discriminator(sd, x, train, new_stats=None, p=""):
    """Discriminator.forward (model/FSRnet.py:477-485): conv3x3(192->64) -> bn_mid -> PReLU -> bn_mid AGAIN -> flatten ->
    Linear(64*56*56 -> 512) -> BatchNorm1d.  ``bn_mid`` runs twice per call, so in train mode its running statistics are
    updated twice (the second update starts from the first one's result)."""
    ns = {} if new_stats is None else new_stats
    cur = dict(sd)

    def bn(key, t):
        y = _bnorm(cur, key, t, train, ns)
        if train:   # later uses of the same layer continue from the updated statistics
            cur[key + ".running_mean"], cur[key + ".running_var"] = ns[key + ".running_mean"], ns[key + ".running_var"]
        return y
    y = F.prelu(bn(p + "bn_mid", _conv(sd, p + "conv_input", x, 1, 1)), sd[p + "relu.weight"])
    y = bn(p + "bn_mid", y)
    y = F.linear(y.flatten(1), sd[p + "fc.weight"], sd[p + "fc.bias"])
    return bn(p + "bn_end", y)


def gan_forward(sd, lr_img, hr_img, train):
    """OverallNetwork_GAN.forward (model/FSRnet.py:525-545), 224x224 only: coarse -> forward_once(coarse) and
    forward_once(hr) (encoder, prior, cat, discriminator) -> decoder(cat of the coarse branch).
    Returns (sr, coarse, landmark1, parsing1, emb1, emb2, new discriminator running stats)."""
    stats = {}
    cur = dict(sd)
    _, coarse = coarse_sr(sd, lr_img, "_coarse_sr_network.")

    def once(x):
        ef = fine_encoder(sd, x, "_fine_sr_encoder.")
        pf, lmk, par = prior_net(sd, x, "_prior_estimation_network.")
        cat = torch.cat((pf, ef), 1)
        emb = discriminator(cur, cat, train, stats, "_discriminator.")
        if train:
            cur.update(stats)
        return cat, lmk, par, emb
    cat1, lmk1, par1, e1 = once(coarse)
    _, _, _, e2 = once(hr_img)
    sr = fine_decoder(sd, cat1, "_fine_sr_decoder.")
    return sr, coarse, lmk1, par1, e1, e2, stats


# ----------------------------------------------------------------------------- FSRNet (SR variant)
# SUPER_RESOLUTION/model/FSRnet.py:12-35,77-156,251-416 restated as a small interpreter over layer specs; the spec
# positions ARE the reference's nn.Sequential indices, so state_dict keys line up ("model.<i>.weight").

def _rconv(sd, key, x, pad, stride=1):
    if pad:
        x = F.pad(x, (pad,) * 4, mode="reflect")
    return F.conv2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride)


def _sr_resblock(sd, p, x):
    y = F.prelu(_inorm(sd, p + ".in1", _conv(sd, p + ".conv1", x, 1, 1)), sd[p + ".relu.weight"])
    return _inorm(sd, p + ".in2", _conv(sd, p + ".conv2", y, 1, 1)) + x


def _sr_bottleneck(sd, p, x):
    y = _conv(sd, p + ".conv1", F.relu(F.instance_norm(x, eps=EPS)))
    y = _conv(sd, p + ".conv2", F.relu(F.instance_norm(y, eps=EPS)), 1, 1)
    y = _conv(sd, p + ".conv3", F.relu(F.instance_norm(y, eps=EPS)))
    return y + x


def _sr_hourglass(sd, p, x, n, nblocks=3):
    def seq(q, t):
        for j in range(nblocks):
            t = _sr_bottleneck(sd, f"{q}.{j}", t)
        return t
    up1 = seq(f"{p}.{n - 1}.0", x)
    low = seq(f"{p}.{n - 1}.1", F.max_pool2d(x, 2, stride=2))
    low = _sr_hourglass(sd, p, low, n - 1, nblocks) if n > 1 else seq(f"{p}.{n - 1}.3", low)
    low = seq(f"{p}.{n - 1}.2", low)
    return up1 + F.interpolate(low, scale_factor=2)


def _sr_generator(sd, x, head, n_blocks=6, tail_out=False):
    """head: list of (seq index, reflect pad, stride, followed by IN+ReLU); then n_blocks residual blocks, then the
    shared up-sampling tail {convT, pad+conv, IN, ReLU} x 2; optional `out` = pad + conv + tanh."""
    i = 0
    for idx, pad, stride, norm in head:
        x = _rconv(sd, f"model.{idx}", x, pad, stride)
        if norm:
            x = F.relu(F.instance_norm(x, eps=EPS))
        i = idx
    i = i + 3 if head[-1][3] else i + 1          # skip IN, ReLU positions
    for _ in range(n_blocks):
        x = _sr_resblock(sd, f"model.{i}", x)
        i += 1
    for _ in range(2):
        x = F.conv_transpose2d(x, sd[f"model.{i}.weight"], None, stride=2, padding=1, output_padding=1)
        x = F.relu(F.instance_norm(_rconv(sd, f"model.{i + 2}", x, 1), eps=EPS))
        i += 5
    if tail_out:
        x = torch.tanh(_rconv(sd, "out.1", x, 1))
    return x


def sr_coarse(sd, x):
    """Coarse_SR_Network (SUPER_RESOLUTION/model/FSRnet.py:251-298)."""
    return _sr_generator(sd, x, [(1, 3, 1, True), (5, 1, 2, False), (7, 1, 1, True), (11, 1, 2, False), (13, 1, 1, True)],
                         tail_out=True)


def sr_encoder(sd, x):
    """Fine_SR_Encoder (:301-342)."""
    return _sr_generator(sd, x, [(1, 1, 1, False), (3, 1, 2, False), (5, 1, 1, True), (9, 1, 2, False), (11, 1, 1, True)])


def sr_decoder(sd, x):
    """Fine_SR_Decoder (:373-416)."""
    return _sr_generator(sd, x, [(1, 1, 2, False), (3, 1, 1, True), (7, 1, 2, False), (9, 1, 1, True)], tail_out=True)


def sr_prior(sd, x, n_hourglass=4, n_blocks=2):
    """Prior_Estimation_Network (:345-370) -> (feat, landmark, parsing)."""
    y = F.relu(F.instance_norm(_rconv(sd, "model.1", x, 3), eps=EPS))
    i = 4
    for _ in range(n_blocks):
        y = _sr_resblock(sd, f"model.{i}", y)
        i += 1
    for _ in range(n_hourglass):
        y = _sr_hourglass(sd, f"model.{i}.hg", y, 4)
        i += 1
    return y, _conv(sd, "fc_landmark", y), _conv(sd, "fc", y)


# ----------------------------------------------------------------------------- IR / IR-SE backbone

IR50_UNITS = ((64, 64, 3), (64, 128, 4), (128, 256, 14), (256, 512, 3))  # model_irse.py:104-110
IR50_STAGE_ENDS = (2, 6, 20, 23)


def ir50_blocks():
    out = []
    for cin, depth, n in IR50_UNITS:
        out.append((cin, depth, 2))
        out += [(depth, depth, 1)] * (n - 1)
    return out


def ir_block(sd, p, x, cin, depth, stride, se, train, new_stats=None):
    """bottleneck_IR / bottleneck_IR_SE (model_irse.py:49-91)."""
    if cin == depth:
        sc = x[:, :, ::stride, ::stride] if stride > 1 else x  # MaxPool2d(1, stride)
    else:
        sc = _bnorm(sd, p + ".shortcut_layer.1", _conv(sd, p + ".shortcut_layer.0", x, stride), train, new_stats)
    r = _bnorm(sd, p + ".res_layer.0", x, train, new_stats)
    r = F.prelu(_conv(sd, p + ".res_layer.1", r, 1, 1), sd[p + ".res_layer.2.weight"])
    r = _bnorm(sd, p + ".res_layer.4", _conv(sd, p + ".res_layer.3", r, stride, 1), train, new_stats)
    if se:
        s = r.mean((2, 3), keepdim=True)
        s = torch.sigmoid(_conv(sd, p + ".res_layer.5.fc2", F.relu(_conv(sd, p + ".res_layer.5.fc1", s))))
        r = r * s
    return r + sc


def ir_input_layer(sd, x, train, new_stats=None, p=""):
    y = _bnorm(sd, p + "input_layer.1", _conv(sd, p + "input_layer.0", x, 1, 1), train, new_stats)
    return F.prelu(y, sd[p + "input_layer.2.weight"])


def ir_backbone(sd, x, se=False, train=False, drop_mask=None, taps=(), new_stats=None, p=""):
    """Backbone.forward (model_irse.py:167-172).  ``drop_mask`` (N x 512 x 7 x 7 of {0,1}) replaces the
    Dropout(0.5) RNG in train mode (kept entries are scaled by 2); None disables dropout.
    Returns (embedding, [tapped body outputs in order of ``taps`` (block indices)])."""
    y = ir_input_layer(sd, x, train, new_stats, p)
    tapped = []
    for i, (cin, depth, stride) in enumerate(ir50_blocks()):
        y = ir_block(sd, f"{p}body.{i}", y, cin, depth, stride, se, train, new_stats)
        if i in taps:
            tapped.append(y)
    y = _bnorm(sd, p + "output_layer.0", y, train, new_stats)
    if train and drop_mask is not None:
        y = y * drop_mask * 2.0
    y = F.linear(y.flatten(1), sd[p + "output_layer.3.weight"], sd[p + "output_layer.3.bias"])
    return _bnorm(sd, p + "output_layer.4", y, train, new_stats), tapped


def ir_teacher5(sd, x, se=False):
    """The 5-output teacher wrapper distill_main.py:59 expects (not in the reference tree):
    eval-mode IR-50 with taps after body blocks 2/6/20/23."""
    e, t = ir_backbone(sd, x, se=se, train=False, taps=IR50_STAGE_ENDS)
    return (e, *t)


# ----------------------------------------------------------------------------- ResNet-34

R34_LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))  # model/resnet.py:160-163,231-236


def resnet34(sd, x, train=False, new_stats=None, p=""):
    """ResNet.forward -> (emb, x1, x2, x3, x4) (model/resnet.py:208-225; max-pool skipped :212)."""
    y = F.relu(_bnorm(sd, p + "bn1", _conv(sd, p + "conv1", x, 2, 3), train, new_stats))
    feats = []
    inpl = 64
    for li, (planes, n, stride) in enumerate(R34_LAYERS, start=1):
        for b in range(n):
            q = f"{p}layer{li}.{b}"
            st = stride if b == 0 else 1
            o = F.relu(_bnorm(sd, q + ".bn1", _conv(sd, q + ".conv1", y, st, 1), train, new_stats))
            o = _bnorm(sd, q + ".bn2", _conv(sd, q + ".conv2", o, 1, 1), train, new_stats)
            if b == 0 and (st != 1 or inpl != planes):
                y = _bnorm(sd, q + ".downsample.1", _conv(sd, q + ".downsample.0", y, st), train, new_stats)
            y = F.relu(o + y)
        inpl = planes
        feats.append(y)
    z = _bnorm(sd, p + "bn_o1", y, train, new_stats)
    z = F.linear(z.flatten(1), sd[p + "fc.weight"], sd[p + "fc.bias"])
    return (_bnorm(sd, p + "bn_o2", z, train, new_stats), *feats)


# ----------------------------------------------------------------------------- losses

def mse97(a, b):
    """MSELossFunc (loss/loss.py:13-15)."""
    return ((a.float() - b.float()) ** 2).mean() * 97.0


def landmark_loss(pred, target):
    """MSELoss_Landmark (loss/loss.py:20-32): channel-summed prediction vs ONE heat-map."""
    return ((pred.sum(1).float() - target.float()) ** 2).mean() * 97.0


def nll2d(logits, target):
    """CrossEntropyLoss2d (loss/loss.py:61-62)."""
    return F.nll_loss(F.log_softmax(logits, 1), torch.squeeze(target))


def cross_entropy(logits, target):
    """nn.CrossEntropyLoss of the teacher trainer (main.py:132, train_teacher_model.py:190)."""
    return F.cross_entropy(logits, target)


def mmd_gaussian(a, b, sigmas=(1.0, 2.0, 4.0, 8.0, 16.0)):
    """Build-defined (reference imports an undefined ``MMD``, Face_Hallucination_sub_Net.py:25):
    biased Gaussian-kernel MMD^2 between two N x D batches, multi-bandwidth.  fp64."""
    a, b = a.double(), b.double()
    z = torch.cat((a, b), 0)
    d2 = torch.cdist(z, z).pow(2)
    k = sum(torch.exp(-d2 / (2.0 * s * s)) for s in sigmas)
    n = a.shape[0]
    return k[:n, :n].mean() + k[n:, n:].mean() - 2.0 * k[:n, n:].mean()


def arcface_logits(emb, weight, target, s=64.0, m=0.5):
    """Build-defined ArcFace head (absent from the reference; l2_norm as model_irse.py:16-20):
    logits = s*cos(theta + m*onehot), easy-margin off, with the usual cos(pi-m) fallback.  fp64 in."""
    e = emb / emb.norm(2, 1, True)
    w = weight / weight.norm(2, 1, True)
    cos = (e @ w.t()).clamp(-1.0, 1.0)
    sin = (1.0 - cos * cos).clamp_min(0).sqrt()
    cm, sm = float(np.cos(m)), float(np.sin(m))
    th, mm = float(np.cos(np.pi - m)), float(np.sin(np.pi - m) * m)
    phi = torch.where(cos > th, cos * cm - sin * sm, cos - mm)
    onehot = F.one_hot(target, weight.shape[0]).bool()
    return s * torch.where(onehot, phi, cos)


# ----------------------------------------------------------------------------- evaluation (utils/utils.py)

def pair_dist(e1, e2):
    """utils/utils.py:41-43: squared L2 per pair (fp32 in, numpy)."""
    d = np.subtract(e1, e2)
    return np.sum(np.square(d), 1)


def confusion_at(thresholds, dist, issame):
    """Vectorised calculate_accuracy (utils/utils.py:14-24) over all thresholds at once.
    predict_same = dist < thr.  Returns integer arrays (tp, fp, tn, fn) of len(thresholds)."""
    thresholds = np.asarray(thresholds)
    issame = np.asarray(issame).astype(bool)
    ds = np.sort(dist[issame])
    dd = np.sort(dist[~issame])
    tp = np.searchsorted(ds, thresholds, side="left").astype(np.int64)
    fp = np.searchsorted(dd, thresholds, side="left").astype(np.int64)
    fn = ds.size - tp
    tn = dd.size - fp
    return tp, fp, tn, fn


def rates_from_counts(tp, fp, tn, fn, n):
    with np.errstate(divide="ignore", invalid="ignore"):
        tpr = np.where(tp + fn == 0, 0.0, tp.astype(np.float64) / np.maximum(tp + fn, 1))
        fpr = np.where(fp + tn == 0, 0.0, fp.astype(np.float64) / np.maximum(fp + tn, 1))
    acc = (tp + tn).astype(np.float64) / float(n)
    return tpr, fpr, acc


def calculate_roc(thresholds, e1, e2, issame, folds):
    """calculate_roc (utils/utils.py:26-87) with the fold index arrays given explicitly
    (the reference draws them from an unseeded sklearn KFold) and the unused
    O(P^2 log P) ``margin_list`` (:47-49) dropped.  ``folds`` = list of (train_idx, test_idx)."""
    return calculate_roc_from_dist(thresholds, pair_dist(e1, e2), issame, folds)


def calculate_roc_from_dist(thresholds, dist, issame, folds):
    """The K-fold part of calculate_roc (utils/utils.py:51-87) on given distances."""
    thresholds = np.asarray(thresholds)
    issame = np.asarray(issame)
    dist = np.asarray(dist)
    k, t = len(folds), len(thresholds)
    tprs, fprs = np.zeros((k, t)), np.zeros((k, t))
    acc, best = np.zeros(k), np.zeros(k)
    for f, (tr, te) in enumerate(folds):
        a = rates_from_counts(*confusion_at(thresholds, dist[tr], issame[tr]), len(tr))[2]
        bi = int(np.argmax(a))
        best[f] = thresholds[bi]
        tprs[f], fprs[f], acc_te = rates_from_counts(*confusion_at(thresholds, dist[te], issame[te]), len(te))
        acc[f] = acc_te[bi]
    return tprs.mean(0), fprs.mean(0), acc.mean(), best


# ----------------------------------------------------------------------------- step restatements

def params_of(sd, skip_buffers=True):
    """Trainable leaves of a state_dict: everything except BN buffers."""
    return {k: v for k, v in sd.items()
            if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var"))}


def with_grad(sd):
    """Clone a state_dict making every trainable tensor a grad-requiring leaf."""
    out = {}
    for k, v in sd.items():
        v = v.clone()
        if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
        out[k] = v
    return out


def grads_of(loss, sd, retain=False):
    """d loss / d theta for every trainable leaf; None where unused."""
    names = [k for k, v in sd.items() if v.requires_grad]
    gs = torch.autograd.grad(loss, [sd[k] for k in names], retain_graph=retain, allow_unused=True)
    return dict(zip(names, gs))


def coarse_step_grads(sd, lr_img, hr_img):
    """BASELINE config 1: L = 12 * mse97(coarse, hr) (Face_Hallucination_sub_Net.py:225)."""
    sd = with_grad(sd)
    _, coarse = coarse_sr(sd, lr_img)
    loss = 12.0 * mse97(coarse, hr_img)
    return loss.detach(), coarse.detach(), grads_of(loss, sd)


def teacher_step_grads(sd, x, target, se=True, drop_mask=None):
    """train_teacher_model.py:189-202 with the IR-(SE-)50 as ``model``: CE on the 512-d output."""
    sd = with_grad(sd)
    stats = {}
    emb, _ = ir_backbone(sd, x, se=se, train=True, drop_mask=drop_mask, new_stats=stats)
    loss = cross_entropy(emb, target)
    return loss.detach(), emb.detach(), grads_of(loss, sd), stats


def kd_step_grads(t_sd, s_sd, a_sd, x, se=False):
    """distill_main.py:59-70 at pre-step weights: student loss -> student grads; assistant loss ->
    assistant grads (the intended pairing; :74's second student step is a reference bug)."""
    with torch.no_grad():
        t = ir_teacher5(t_sd, x, se=se)
    s_sd, a_sd = with_grad(s_sd), with_grad(a_sd)
    st_s, st_a = {}, {}
    s = resnet34(s_sd, x, train=True, new_stats=st_s)
    a = resnet34(a_sd, x, train=True, new_stats=st_a)
    s_loss = F.mse_loss(s[0], t[0])
    a_loss = sum(F.mse_loss(t[k] - s[k], a[k]) for k in range(1, 5)) + F.mse_loss(t[0] - s[0], a[0])
    gs = grads_of(s_loss, s_sd, retain=True)
    ga = grads_of(a_loss, a_sd)
    return (s_loss.detach(), a_loss.detach()), gs, ga, (st_s, st_a), [v.detach() for v in s], [v.detach() for v in a]


def fhn_step_grads(sds, lr_img, hr_img, heatmap, parsing):
    """One forward; per-(loss_k, theta_k) gradients at pre-step weights, discriminator
    terms dropped (112x112 composition): L_coarse = 12*mse97(coarse,hr) -> coarse;
    L_enc = 10*mse97(sr,hr) -> encoder; L_prior = mse97(sr,hr)+lmk+CE -> prior; L_dec = 10*mse97(sr,hr) -> decoder."""
    sds = {k: with_grad(v) for k, v in sds.items()}
    sr, coarse, lmk, par = fhn_forward(sds, lr_img)
    pix = mse97(sr, hr_img)
    losses = {
        "coarse": 12.0 * mse97(coarse, hr_img),
        "encoder": 10.0 * pix,
        "prior": pix + landmark_loss(lmk, heatmap) + nll2d(par, parsing),
        "decoder": 10.0 * pix,
    }
    grads = {k: grads_of(losses[k], sds[k], retain=True) for k in ("coarse", "encoder", "prior", "decoder")}
    outs = dict(sr=sr.detach(), coarse=coarse.detach(), landmark=lmk.detach(), parsing=par.detach())
    return {k: v.detach() for k, v in losses.items()}, outs, grads


def gan_step_grads(sd, lr_img, hr_img, heatmap, parsing, dist=None):
    """Face_Hallucination_sub_Net.py:218-247 on OverallNetwork_GAN (224x224) at pre-step weights: one forward, then
      L_disc  = -dist(emb1, emb2)                                   -> _discriminator
      L_coarse = 12 * mse97(coarse, hr)                             -> _coarse_sr_network
      L_enc   = 10 * mse97(sr, hr) - L_disc                         -> _fine_sr_encoder
      L_prior = -L_disc + mse97(sr, hr) + landmark + CE             -> _prior_estimation_network
      L_dec   = 10 * mse97(sr, hr)                                  -> _fine_sr_decoder
    ``dist`` stands in for the reference's undefined ``MMD`` import (:25); default F.mse_loss.
    Returns (losses, outputs, {subnet: {param name without the sub-network prefix: grad or None}})."""
    dist = dist or F.mse_loss
    sdg = with_grad(sd)
    sr, coarse, lmk, par, e1, e2, _ = gan_forward(sdg, lr_img, hr_img, train=True)
    l_disc = -dist(e1, e2)
    pix = mse97(sr, hr_img)
    losses = {"disc": l_disc, "coarse": 12.0 * mse97(coarse, hr_img), "encoder": 10.0 * pix - l_disc,
              "prior": -l_disc + pix + landmark_loss(lmk, heatmap) + nll2d(par, parsing), "decoder": 10.0 * pix}
    prefix = {"disc": "_discriminator.", "coarse": "_coarse_sr_network.", "encoder": "_fine_sr_encoder.",
              "prior": "_prior_estimation_network.", "decoder": "_fine_sr_decoder."}
    grads = {}
    for k, pre in prefix.items():
        sub = {n[len(pre):]: v for n, v in sdg.items() if n.startswith(pre)}
        grads[k] = grads_of(losses[k], sub, retain=True)
    outs = dict(sr=sr.detach(), coarse=coarse.detach(), landmark=lmk.detach(), parsing=par.detach(), emb1=e1.detach(),
                emb2=e2.detach())
    return {k: v.detach() for k, v in losses.items()}, outs, grads


def fhn_perceptual_grads(sds, bb_sd, lr_img, hr_img, heatmap, parsing, taps=(21, 22), lam_feature=1.0, lam_landmark=1.0,
                         lam_parsing=1.0):
    """SUPER_RESOLUTION/train_FHN.py:251-308 at pre-step weights (SR-variant generators, frozen eval-mode IR-50 as the
    perceptual backbone, features tapped after body blocks 21 and 22 = layer_list[-3:-1]):
      L_coarse = lam_F * sum_l MSE(feat_l(hr), feat_l(coarse))            -> coarse
      L_prior  = lam_L * landmark(lmk, heatmap) + lam_P * CE(parsing)      -> prior
      L_encdec = lam_F * sum_l MSE(feat_l(hr), feat_l(sr))                 -> encoder + decoder
    The upstream SR Landmark_Loss raises (torch.pow without exponent, SUPER_RESOLUTION/loss/loss.py:16); its evident
    intent mean((sum_c in - target)^2) is used."""
    sds = {k: with_grad(v) for k, v in sds.items()}
    coarse = sr_coarse(sds["coarse"], lr_img)
    with torch.no_grad():
        _, f_hr = ir_backbone(bb_sd, hr_img, train=False, taps=taps)
    _, f_c = ir_backbone(bb_sd, coarse, train=False, taps=taps)
    l_coarse = lam_feature * sum(F.mse_loss(a, b) for a, b in zip(f_hr, f_c))
    pf, lmk, par = sr_prior(sds["prior"], coarse)
    ef = sr_encoder(sds["encoder"], coarse)
    sr = sr_decoder(sds["decoder"], torch.cat((pf, ef), 1))
    l_prior = lam_landmark * ((lmk.sum(1) - heatmap) ** 2).mean() + lam_parsing * nll2d(par, parsing)
    _, f_sr = ir_backbone(bb_sd, sr, train=False, taps=taps)
    l_ed = lam_feature * sum(F.mse_loss(a, b) for a, b in zip(f_hr, f_sr))
    grads = {"coarse": grads_of(l_coarse, sds["coarse"], retain=True), "prior": grads_of(l_prior, sds["prior"], retain=True),
             "encoder": grads_of(l_ed, sds["encoder"], retain=True), "decoder": grads_of(l_ed, sds["decoder"], retain=True)}
    losses = dict(coarse=l_coarse.detach(), prior=l_prior.detach(), encdec=l_ed.detach())
    return losses, dict(coarse=coarse.detach(), sr=sr.detach()), grads


def c4_step_grads(fhn_sds, s_sd, a_sd, t_sd, lr_img, hr_img, taps=IR50_STAGE_ENDS):
    """BASELINE config 4: the two halves composed.  sr = FHN(lr) with the four root generators composed
    as SUPER_RESOLUTION/train_FHN.py:274-279; IR-SE-50 student and assistant (train mode, Dropout pinned off) both see
    ``sr``; the frozen eval-mode IR-SE-50 teacher sees ``hr``; losses of distill_main.py:63-70 with the stage taps
    after body blocks 2/6/20/23.  Pinned (loss_k, theta_k) pairs at pre-step weights:
      student_loss   = MSE(s_out, t_out)                                        -> student AND the four FHN generators
                                                                                   (the gradient flows through the student
                                                                                   into the hallucination net)
      assistant_loss = sum_k MSE(t_k - s_k, a_k) + MSE(t_out - s_out, a_out)    -> assistant
    Returns ((s_loss, a_loss), outputs dict, grads dict(student, assistant, coarse, prior, encoder, decoder), stats)."""
    with torch.no_grad():
        t = ir_teacher5(t_sd, hr_img, se=True)
    fhn = {k: with_grad(v) for k, v in fhn_sds.items()}
    s_sd, a_sd = with_grad(s_sd), with_grad(a_sd)
    sr, coarse, lmk, par = fhn_forward(fhn, lr_img)
    st_s, st_a = {}, {}
    s_e, s_t = ir_backbone(s_sd, sr, se=True, train=True, taps=taps, new_stats=st_s)
    a_e, a_t = ir_backbone(a_sd, sr, se=True, train=True, taps=taps, new_stats=st_a)
    s, a = (s_e, *s_t), (a_e, *a_t)
    s_loss = F.mse_loss(s[0], t[0])
    a_loss = sum(F.mse_loss(t[k] - s[k], a[k]) for k in range(1, 5)) + F.mse_loss(t[0] - s[0], a[0])
    grads = {"student": grads_of(s_loss, s_sd, retain=True)}
    for k in ("coarse", "prior", "encoder", "decoder"):
        grads[k] = grads_of(s_loss, fhn[k], retain=True)
    grads["assistant"] = grads_of(a_loss, a_sd)
    outs = dict(sr=sr.detach(), coarse=coarse.detach(), s_emb=s[0].detach(), a_emb=a[0].detach(), t_emb=t[0],
                s_taps=[v.detach() for v in s[1:]], a_taps=[v.detach() for v in a[1:]], t_taps=list(t[1:]))
    return (s_loss.detach(), a_loss.detach()), outs, grads, (st_s, st_a)


# ----------------------------------------------------------------------------- loader-side synthesis
# SUPER_RESOLUTION/FHN_loader.py:65-66 makes the low-resolution input with Pillow:
#     lr_img = sr_img.resize((int(128 / scale), int(128 / scale))).resize((112, 112), Image.BICUBIC)
# (Image.resize defaults to BICUBIC).  Pillow is a third-party dependency of the reference (unpinned there; 12.2.0 in this
# image); its 8-bit resampler (src/libImaging/Resample.c) is restated below from its published algorithm -- separable
# convolution with the a = -0.5 cubic, support 2 * max(scale, 1) (antialiasing when shrinking), per-output-pixel windows,
# coefficients normalised in double then rounded to 22-bit fixed point, the horizontal pass first, each pass rounding to
# uint8 -- and pinned against PIL itself by oracle/make_golden.py (bit-exact) and the committed tests/golden/loader.npz.
_PIL_PRECISION_BITS = 32 - 8 - 2


def _pil_bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_coeffs(in_size, out_size):
    """(bounds [out][2] = (first input index, tap count), kk [out][ksize] int32 fixed-point taps) of Resample.c:precompute_coeffs
    + normalize_coeffs_8bpc for the full-image box."""
    import math
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_pil_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            kk[xx, x] = int(-0.5 + v * (1 << _PIL_PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PIL_PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pil_pass(img, out_size, axis):
    """One 8-bit resampling pass along ``axis`` (0 = vertical, 1 = horizontal) of a uint8 [H][W][C] image."""
    src = np.moveaxis(img.astype(np.int64), axis, 0)
    bounds, kk = pil_coeffs(src.shape[0], out_size)
    out = np.empty((out_size,) + src.shape[1:], np.int64)
    for xx in range(out_size):
        x0, n = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (_PIL_PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[x0 + x] * int(kk[xx, x])
        out[xx] = acc >> _PIL_PRECISION_BITS
    return np.moveaxis(np.clip(out, 0, 255).astype(np.uint8), 0, axis)


def pil_resize_bicubic(img, out_w, out_h):
    """PIL.Image.resize((out_w, out_h), BICUBIC) of a uint8 [H][W][C] image: horizontal pass, then vertical (Resample.c)."""
    h, w = img.shape[:2]
    if out_w != w:
        img = _pil_pass(img, out_w, 1)
    if out_h != h:
        img = _pil_pass(img, out_h, 0)
    return img


def lr_from_hr_u8(hr_u8, low):
    """FHN_loader.py:65-66 on a uint8 [H][W][3] crop: down to low x low, back up to H x W, both bicubic."""
    h, w = hr_u8.shape[:2]
    return pil_resize_bicubic(pil_resize_bicubic(hr_u8, low, low), w, h)


def to_tensor_normalize(img_u8):
    """transforms.ToTensor() + Normalize(0.5, 0.5) (FHN_loader.py:30-35): uint8 HWC -> float32 CHW in [-1, 1]."""
    t = torch.from_numpy(np.ascontiguousarray(img_u8)).permute(2, 0, 1).to(torch.float32).div(255)
    return (t - 0.5) / 0.5


def gaussian_k(x0, y0, sigma, width, height):
    """FHN_loader.py:131-137 / helen_loader.py:124-130: one Gaussian bump, float64 [height][width]."""
    x = np.arange(0, width, 1, float)
    y = np.arange(0, height, 1, float)[:, np.newaxis]
    return np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))


def generate_hm(height, width, landmark, s=2.0):
    """FHN_loader.py:119-129 / helen_loader.py:132-143: the sum of one Gaussian per landmark (x, y), accumulated into a
    float32 map (each float64 bump is added to the float32 running sum, as the in-place += of the reference does)."""
    hm = np.zeros((height, width), dtype=np.float32)
    for i in range(np.shape(landmark)[0]):
        hm += gaussian_k(landmark[i][0], landmark[i][1], s, width, height)
    return hm
