"""Training-step counterparts of the reference's loop bodies (the scripts themselves cannot run; SURVEY.md
section 0).  Pinned semantics: evaluate every forward once, compute each
(loss_k, theta_k) gradient at the PRE-step weights, then step every optimizer.

  fhn_step      Face_Hallucination_sub_Net.py:218-247 loss->optimizer map on the 112x112 composition of
                SUPER_RESOLUTION/train_FHN.py:274-279 (discriminator/MMD terms dropped: MMD is undefined upstream)
  fhn_step_fused  the same four gradients from ONE backward pass (the map is linear in the losses)
  gan_step      the same map on OverallNetwork_GAN at 224x224 with the discriminator / MMD terms kept
  c4_step       BASELINE configs[3]: FHN -> IR-SE-50 student + assistant vs frozen IR-SE-50 teacher, residual-KD losses
  kd_step       distill_main.py:59-74 (the second student_optimizer.step() at :74 is a reference bug; the
                assistant is stepped)
  verify_step   distill_main.py:111-138: pair verification of one batch (teacher alone, student + assistant summed)
  teacher_step  train_teacher_model.py:189-202
"""
from __future__ import annotations

import torch

from . import ops
from .loss.loss import CrossEntropyLoss, CrossEntropyLoss2d, MSELoss, MSELoss_Landmark, MSELossFunc


def _assign_grads(params, grads):
    for p, g in zip(params, grads):
        if getattr(p, "_xr_direct", False):
            # FlatParams: .grad is a view of the flat gradient buffer and stays one -- the HIP backward kernels accumulated
            # into it in place (autograd then returns None); a tensor autograd did return is added into the view
            if g is not None:
                p.grad.add_(g)
                p.__dict__["_xr_touched"] = True   # (autograd.grad fires no post-accumulate hook: the fused optimizers' skip mask)
            continue
        p.grad = g  # None stays None: stock optimizers then skip the parameter (incl. weight decay)


def _pair_grads(loss, params, retain=True):
    """d loss / d params at the current weights into ``.grad``.  ``needs_input_grad`` of the HIP Functions was fixed at
    forward time, so the backward nodes this traversal crosses would also produce gradients of parameters OUTSIDE ``params``
    (and, with FlatParams(direct=True), accumulate them in place): ops.grad_only restricts parameter-gradient work to
    ``params`` -- no wasted weight-gradient launches, no foreign accumulation."""
    if isinstance(params, torch.nn.Module):
        params = params.parameters()
    params = [p for p in params if p.requires_grad]
    with ops.grad_only(params):
        grads = torch.autograd.grad(loss, params, retain_graph=retain, allow_unused=True)
    _assign_grads(params, grads)


def _zero_grads(modules, optimizers=None):
    """Gradients of a step start from zero: fused flat optimizers zero their buffer (views stay), otherwise .grad = None."""
    if optimizers is not None:
        for o in (optimizers.values() if isinstance(optimizers, dict) else optimizers):
            if o is not None:
                o.zero_grad()
        return
    for m in modules:
        for p in m.parameters():
            if not getattr(p, "_xr_direct", False):
                p.grad = None


def _finish_reducers(reducers):
    """Data-parallel form of a step: every gradient is in its flat buffer (the caller's stream has joined the side /
    lockstep streams) -> launch the buckets that did not go out from the gradient hooks, wait for all of them.  ``reducers``:
    one parallel.BucketedAllReduce per flat buffer (iterable / dict / None)."""
    if reducers is None:
        return
    ops.join_side_stream()
    for r in (reducers.values() if isinstance(reducers, dict) else reducers):
        if r is not None:
            r.finish()


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


def fhn_step_fused(nets, lr_img, hr_img, heatmap, parsing, optimizers=None, reducers=None):
    """fhn_step's four (loss_k, theta_k) gradients from ONE backward pass.  The map is linear: the encoder and the decoder
    both take 10 * d pix, the prior takes 1 * d pix + d landmark + d parsing, the coarse net only d(12 * mse97(coarse, hr))
    -- so the prior / encoder see a detached coarse image, the pixel loss is back-propagated once with weight 10 and the
    prior's branch re-scales it by 0.1 (ops.grad_scale).  3x forward FLOPs (SURVEY 8d's count for C3) instead of the five
    partial traversals of the literal form; results agree with fhn_step to rounding (tests/test_gpu_models.py).
    ``reducers`` (data parallel, BASELINE configs[2]): parallel.BucketedAllReduce objects over the flat buffers of ``nets``;
    their buckets go out from the gradient hooks during backward and are waited for before the optimizers step."""
    mse97, lmk_loss, ce2d = MSELossFunc(), MSELoss_Landmark(), CrossEntropyLoss2d()
    order = ("coarse", "encoder", "prior", "decoder")
    _zero_grads([nets[k] for k in order], optimizers)
    _, coarse = nets["coarse"](lr_img)
    c_in = coarse.detach()
    pf, lmk, par = nets["prior"](c_in)
    ef = nets["encoder"](c_in)
    sr = nets["decoder"](torch.cat((ops.grad_scale(pf, 0.1), ef), 1))
    pix = mse97(sr, hr_img)
    l_coarse, l_lmk, l_par = 12.0 * mse97(coarse, hr_img), lmk_loss(lmk, heatmap), ce2d(par, parsing)
    (l_coarse + 10.0 * pix + l_lmk + l_par).backward()
    _finish_reducers(reducers)      # BASELINE configs[2]: gradient all-reduce between backward and the updates
    if optimizers is not None:
        for k in order:
            optimizers[k].step()
    pix_d = pix.detach()
    losses = {"coarse": l_coarse.detach(), "encoder": 10.0 * pix_d, "prior": pix_d + l_lmk.detach() + l_par.detach(),
              "decoder": 10.0 * pix_d}
    outs = dict(sr=sr.detach(), coarse=coarse.detach(), landmark=lmk.detach(), parsing=par.detach())
    return losses, outs


def gan_step(model, lr_img, hr_img, heatmap, parsing, optimizers=None, criterion_mmd=None):
    """Face_Hallucination_sub_Net.py:218-247 as written, on OverallNetwork_GAN (224x224 only, model/FSRnet.py:468) with the
    discriminator terms kept: L_disc = -MMD(emb1, emb2) -> discriminator; L_coarse = 12*mse97(coarse, hr) -> coarse;
    L_enc = 10*mse97(sr, hr) - L_disc -> encoder; L_prior = -L_disc + mse97(sr, hr) + lmk + CE -> prior;
    L_dec = 10*mse97(sr, hr) -> decoder.  ``criterion_mmd(emb1, emb2)`` is the reference's ``criterion_mmd`` (:216); its
    class is undefined upstream (:25), so the default is the build-defined Gaussian-kernel MMD (loss/loss.py:MMD; parity
    unpinned) -- with any pinned distance injected (tests: nn.MSELoss) the whole loss -> sub-network map is pinned by
    tests/golden/gan_step.npz.  ``optimizers``: dict(disc, coarse, encoder, prior, decoder)."""
    if criterion_mmd is None:
        from .loss.loss import MMD as criterion_mmd
    mse97, lmk_loss, ce2d = MSELossFunc(), MSELoss_Landmark(), CrossEntropyLoss2d()
    sr, coarse, lmk, par, e1, e2 = model(lr_img, hr_img)
    l_disc = -criterion_mmd(e1, e2)
    pix = mse97(sr, hr_img)
    losses = {
        "disc": l_disc,
        "coarse": 12.0 * mse97(coarse, hr_img),
        "encoder": 10.0 * pix - l_disc,
        "prior": -l_disc + pix + lmk_loss(lmk, heatmap) + ce2d(par, parsing),
        "decoder": 10.0 * pix,
    }
    subnets = {"disc": model._discriminator, "coarse": model._coarse_sr_network, "encoder": model._fine_sr_encoder,
               "prior": model._prior_estimation_network, "decoder": model._fine_sr_decoder}
    order = ("disc", "coarse", "encoder", "prior", "decoder")
    for i, k in enumerate(order):
        _pair_grads(losses[k], subnets[k], retain=i + 1 < len(order))
    if optimizers is not None:
        for k in order:
            optimizers[k].step()
    outs = dict(sr=sr.detach(), coarse=coarse.detach(), landmark=lmk.detach(), parsing=par.detach(), emb1=e1.detach(),
                emb2=e2.detach())
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
    _pair_grads(l_ed, [p for k in ("encoder", "decoder") for p in nets[k].parameters()], retain=False)
    if optimizers is not None:
        for k in ("coarse", "prior", "encdec"):
            optimizers[k].step()
    losses = dict(coarse=l_coarse.detach(), prior=l_prior.detach(), encdec=l_ed.detach())
    return losses, dict(coarse=coarse.detach(), sr=sr.detach())


class _Landmark1(torch.nn.Module):
    def forward(self, input, target):
        return ops.landmark_loss(input, target, 1.0)


_LMK1 = _Landmark1()


def kd_step(teacher, student, assistant, x, student_optimizer=None, assistant_optimizer=None, taps=(2, 6, 20, 23), reducers=None):
    """Residual knowledge distillation: student matches the frozen teacher's embedding; the assistant learns the
    residual (teacher - student) at the four stage taps and the embedding."""
    crit = MSELoss()
    teacher.eval()
    student.train()
    assistant.train()
    with torch.no_grad():
        t = teacher.forward_taps(x, taps) if hasattr(teacher, "forward_taps") else teacher(x)
    lock = False
    if ops._cfg["lockstep"] and x.is_cuda and not ops._graph["capturing"] and hasattr(student, "lockstep_plan") \
            and hasattr(assistant, "lockstep_plan") and type(student) is type(assistant):
        from . import lockstep
        plans = [student.lockstep_plan(), assistant.lockstep_plan()]
        lock = lockstep.plans_compatible(plans)
    if lock:
        # student and assistant are independent (both see x): stage by stage in lockstep on their own streams (xrface/lockstep.py)
        s, a = lockstep.run_lockstep((student, assistant), (x, x), plans)
    else:
        s = student(x)
        a = assistant(x)
    s_loss = crit(s[0], t[0])
    a_loss = crit(t[1] - s[1], a[1]) + crit(t[2] - s[2], a[2]) + crit(t[3] - s[3], a[3]) + crit(t[4] - s[4], a[4]) \
        + crit(t[0] - s[0], a[0])
    _pair_grads(s_loss, student, retain=True)
    _pair_grads(a_loss, assistant, retain=False)
    if lock:
        lockstep.join(x.device, 2)
    _finish_reducers(reducers)
    if student_optimizer is not None:
        student_optimizer.step()
    if assistant_optimizer is not None:
        assistant_optimizer.step()
    return (s_loss.detach(), a_loss.detach()), [v.detach() for v in s], [v.detach() for v in a], [v.detach() for v in t]


def c4_step(fhn, student, assistant, teacher, lr_img, hr_img, optimizers=None, taps=(2, 6, 20, 23), reducers=None):
    """BASELINE configs[3]: the two halves of the system composed into one training step.
    sr = FHN(lr) with the four generators composed as SUPER_RESOLUTION/train_FHN.py:274-279; the IR-SE-50 student and
    assistant both see ``sr``, the frozen eval-mode IR-SE-50 teacher sees ``hr`` (5-output form distill_main.py:59 unpacks,
    taps after body blocks 2/6/20/23); losses of distill_main.py:63-70.  Pinned (loss_k, theta_k) pairs at pre-step weights:
      student_loss   = MSE(s_out, t_out)                                      -> student + the four FHN generators
      assistant_loss = sum_k MSE(t_k - s_k, a_k) + MSE(t_out - s_out, a_out)  -> assistant
    The two parameter sets are disjoint and the assistant's targets / input are detached, so ONE backward pass of
    student_loss + assistant_loss yields exactly the two pair gradients (and is safe with FlatParams(direct=True)).
    ``fhn``: dict(coarse, prior, encoder, decoder); ``optimizers`` (optional): iterable / dict of optimizers, all stepped.
    ``reducers`` (data parallel, BASELINE configs[3]: "8 MI355X DDP"): one parallel.BucketedAllReduce per flat gradient buffer
    (FHN, student, assistant).  Buckets are launched from the gradient hooks on the stream of the chain that produced them (each
    launch first joins the weight-gradient side stream); after backward() the caller's stream joins the lockstep chains, the
    remaining buckets go out, and every all-reduce is waited for before the optimizers read the gradients.
    Returns ((student_loss, assistant_loss), outputs dict)."""
    crit = MSELoss()
    teacher.eval()
    student.train()
    assistant.train()
    _zero_grads([*fhn.values(), student, assistant], optimizers)
    lock = (ops._cfg["lockstep"] and hr_img.is_cuda and not ops._graph["capturing"] and hasattr(student, "body")
            and hasattr(assistant, "body") and len(student.body) == len(assistant.body))
    # (the frozen teacher's forward stays on the caller's stream: run beside the FHN forward on a third stream it cost +25 % --
    # its 160 KB-LDS tiles and the persistent one-workgroup-per-CU direct kernels evict each other from the CUs)
    with torch.no_grad():
        t = teacher.forward_taps(hr_img, taps)
    _, coarse = fhn["coarse"](lr_img)
    pf, _, _ = fhn["prior"](coarse)
    ef = fhn["encoder"](coarse)
    sr = fhn["decoder"](torch.cat((pf, ef), 1))
    if lock:
        # student and assistant are independent given sr: both advance in lockstep on their own streams (model_irse.py)
        from .model.model_irse import forward_taps_lockstep, lockstep_join
        s, a = forward_taps_lockstep((student, assistant), (sr, sr.detach()), taps)
    else:
        s = student.forward_taps(sr, taps)
        a = assistant.forward_taps(sr.detach(), taps)
    s_loss = crit(s[0], t[0])
    a_loss = crit((t[0] - s[0]).detach(), a[0])
    for k in range(1, 5):
        a_loss = a_loss + crit(ops.sub_detached(t[k], s[k]), a[k])
    (s_loss + a_loss).backward()
    if lock:
        lockstep_join(sr.device, 2)
    _finish_reducers(reducers)
    if optimizers is not None:
        for o in (optimizers.values() if isinstance(optimizers, dict) else optimizers):
            o.step()
    outs = dict(sr=sr.detach(), coarse=coarse.detach(), s=[v.detach() for v in s], a=[v.detach() for v in a],
                t=[v.detach() for v in t])
    return (s_loss.detach(), a_loss.detach()), outs


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


def verify_step(teacher, student, assistant, img1, img2, label, thresholds=None, nrof_folds=10, fold_id=None):
    """distill_main.py:111-138 for one batch of pairs: the teacher's embeddings of (img1, img2) go through calculate_roc; the
    student's and the assistant's embeddings are SUMMED (the assistant learned the teacher-minus-student residual) and go
    through it too.  Everything stays on the device (embeddings, distances, histogram, K-fold sweep); ``label``: same / different
    per pair.  ``fold_id`` pins the K-fold membership (the reference draws it from an unseeded KFold).
    Returns ((tpr, fpr, accuracy, best_thresholds) of the teacher, the same of student + assistant)."""
    import numpy as np
    from .utils.utils import calculate_roc
    thresholds = np.arange(0, 12000, 3) if thresholds is None else thresholds
    for m in (teacher, student, assistant):
        m.eval()
    first = lambda out: out[0] if isinstance(out, (tuple, list)) else out
    lab = np.asarray(label.cpu() if isinstance(label, torch.Tensor) else label)
    with torch.no_grad():
        t1, t2 = first(teacher(img1)), first(teacher(img2))
        t_res = calculate_roc(thresholds, t1, t2, lab, nrof_folds=nrof_folds, pca=0, fold_id=fold_id)
        s1 = first(student(img1)) + first(assistant(img1))
        s2 = first(student(img2)) + first(assistant(img2))
        s_res = calculate_roc(thresholds, s1, s2, lab, nrof_folds=nrof_folds, pca=0, fold_id=fold_id)
    return t_res, s_res
