"""Training-step counterparts of the reference's loop bodies (the scripts themselves cannot run; SURVEY.md
section 0).  Pinned semantics: evaluate every forward once, compute each
(loss_k, theta_k) gradient at the PRE-step weights, then step every optimizer.

  fhn_step      Face_Hallucination_sub_Net.py:218-247 loss->optimizer map on the 112x112 composition of
                SUPER_RESOLUTION/train_FHN.py:274-279 (discriminator/MMD terms dropped: MMD is undefined upstream)
  kd_step       distill_main.py:59-74 (the second student_optimizer.step() at :74 is a reference bug; the
                assistant is stepped)
  teacher_step  train_teacher_model.py:189-202
"""
from __future__ import annotations

import torch

from . import ops
from .loss.loss import CrossEntropyLoss, CrossEntropyLoss2d, MSELoss, MSELoss_Landmark, MSELossFunc


def _assign_grads(params, grads):
    for p, g in zip(params, grads):
        p.grad = g  # None stays None: stock optimizers then skip the parameter (incl. weight decay)


def _pair_grads(loss, module, retain=True):
    params = [p for p in module.parameters() if p.requires_grad]
    grads = torch.autograd.grad(loss, params, retain_graph=retain, allow_unused=True)
    _assign_grads(params, grads)


def fhn_step(nets, lr_img, hr_img, heatmap, parsing, optimizers=None):
    """nets: dict(coarse, prior, encoder, decoder).  Returns (losses dict, outputs dict); .grad of each
    sub-network holds d L_k / d theta_k.  If ``optimizers`` (same keys) is given they are stepped."""
    mse97, lmk_loss, ce2d = MSELossFunc(), MSELoss_Landmark(), CrossEntropyLoss2d()
    _, coarse = nets["coarse"](lr_img)
    pf, lmk, par = nets["prior"](coarse)
    ef = nets["encoder"](coarse)
    sr = nets["decoder"](torch.cat((pf, ef), 1))
    pix = mse97(sr, hr_img)
    losses = {
        "coarse": 12.0 * mse97(coarse, hr_img),
        "encoder": 10.0 * pix,
        "prior": pix + lmk_loss(lmk, heatmap) + ce2d(par, parsing),
        "decoder": 10.0 * pix,
    }
    order = ("coarse", "encoder", "prior", "decoder")
    for i, k in enumerate(order):
        _pair_grads(losses[k], nets[k], retain=i + 1 < len(order))
    if optimizers is not None:
        for k in order:
            optimizers[k].step()
    outs = dict(sr=sr.detach(), coarse=coarse.detach(), landmark=lmk.detach(), parsing=par.detach())
    return {k: v.detach() for k, v in losses.items()}, outs


def fhn_perceptual_step(nets, backbone, lr_img, hr_img, heatmap, parsing, layer_list=("21", "22"), optimizers=None,
                        lam_feature=1.0, lam_landmark=1.0, lam_parsing=1.0):
    """SUPER_RESOLUTION/train_FHN.py:251-308: perceptual (IR-50 feature) losses.  ``nets`` = SR-variant generators
    dict(coarse, prior, encoder, decoder); ``backbone`` = frozen IR-50 in eval mode (NOT under no_grad: the input
    gradient flows through it back to the generator, :258-259).  Features are tapped with FeatureExtractor exactly as
    the reference does.  ``optimizers`` (optional): dict(coarse, prior, encdec)."""
    from .model.GroupDepthConv import FeatureExtractor
    fe, mse, ce2d = FeatureExtractor(), MSELoss(), CrossEntropyLoss2d()
    backbone.eval()

    def feats(img):
        d, _, _, _ = fe(backbone.input_layer(img), list(layer_list), backbone.body)
        return [d[k] for k in layer_list]

    coarse = nets["coarse"](lr_img)
    with torch.no_grad():
        f_hr = feats(hr_img)
    f_c = feats(coarse)
    l_coarse = lam_feature * sum(mse(a, b) for a, b in zip(f_hr, f_c))
    pf, lmk, par = nets["prior"](coarse)
    ef = nets["encoder"](coarse)
    sr = nets["decoder"](torch.cat((pf, ef), 1))
    # upstream Landmark_Loss raises (torch.pow without exponent); its evident intent mean((sum_c in - t)^2) is used
    l_prior = lam_landmark * _LMK1(lmk, heatmap) + lam_parsing * ce2d(par, parsing)
    f_sr = feats(sr)
    l_ed = lam_feature * sum(mse(a, b) for a, b in zip(f_hr, f_sr))
    _pair_grads(l_coarse, nets["coarse"], retain=True)
    _pair_grads(l_prior, nets["prior"], retain=True)
    params = [p for k in ("encoder", "decoder") for p in nets[k].parameters() if p.requires_grad]
    _assign_grads(params, torch.autograd.grad(l_ed, params, allow_unused=True))
    if optimizers is not None:
        for k in ("coarse", "prior", "encdec"):
            optimizers[k].step()
    losses = dict(coarse=l_coarse.detach(), prior=l_prior.detach(), encdec=l_ed.detach())
    return losses, dict(coarse=coarse.detach(), sr=sr.detach())


class _Landmark1(torch.nn.Module):
    def forward(self, input, target):
        return ops.landmark_loss(input, target, 1.0)


_LMK1 = _Landmark1()


def kd_step(teacher, student, assistant, x, student_optimizer=None, assistant_optimizer=None, taps=(2, 6, 20, 23)):
    """Residual knowledge distillation: student matches the frozen teacher's embedding; the assistant learns the
    residual (teacher - student) at the four stage taps and the embedding."""
    crit = MSELoss()
    teacher.eval()
    student.train()
    assistant.train()
    with torch.no_grad():
        t = teacher.forward_taps(x, taps) if hasattr(teacher, "forward_taps") else teacher(x)
    s = student(x)
    a = assistant(x)
    s_loss = crit(s[0], t[0])
    a_loss = crit(t[1] - s[1], a[1]) + crit(t[2] - s[2], a[2]) + crit(t[3] - s[3], a[3]) + crit(t[4] - s[4], a[4]) \
        + crit(t[0] - s[0], a[0])
    _pair_grads(s_loss, student, retain=True)
    _pair_grads(a_loss, assistant, retain=False)
    if student_optimizer is not None:
        student_optimizer.step()
    if assistant_optimizer is not None:
        assistant_optimizer.step()
    return (s_loss.detach(), a_loss.detach()), [v.detach() for v in s], [v.detach() for v in a], [v.detach() for v in t]


def teacher_step(model, x, target, optimizer=None, criterion=None):
    """output = model(x) (first element if the model returns the 5-tuple); loss = CE(output, target)."""
    criterion = criterion or CrossEntropyLoss()
    out = model(x)
    if isinstance(out, (tuple, list)):
        out = out[0]
    loss = criterion(out, target)
    if optimizer is not None:
        optimizer.zero_grad(set_to_none=True)
    loss.backward()
    if optimizer is not None:
        optimizer.step()
    return loss.detach(), out.detach()
