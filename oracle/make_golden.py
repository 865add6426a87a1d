"""TEST INFRASTRUCTURE -- golden-fixture generator.  Runs ONLY in the build container.

Imports the reference's own pure-PyTorch modules from /root/reference on CPU, loads
deterministic weights (oracle/detgen.py) into them, runs them on seeded synthetic inputs,
and
  (1) asserts the CPU restatement in oracle/cpu_ref.py reproduces every output/gradient
      (tight fp32 tolerance) -- this is what *pins* the oracle, and
  (2) writes small fixtures (full tensors when small, digests otherwise) to tests/golden/.

The reference never travels to the GPU box; only these data files do.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
"""
from __future__ import annotations

import importlib
import importlib.util
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import cpu_ref as R  # noqa: E402
from oracle import detgen as G  # noqa: E402
from oracle.digest import digest  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
TOL = 2e-4  # restatement-vs-reference tolerance, relative to max-abs of the reference tensor


def ref_import(name):
    if REF not in sys.path:
        sys.path.insert(0, REF)
    return importlib.import_module(name)


def close(a, b, what, tol=TOL, floor=1e-12):
    a, b = a.detach().double(), b.detach().double()
    scale = max(float(b.abs().max()), floor)
    err = float((a - b).abs().max()) / scale
    assert err <= tol, f"{what}: restatement differs from reference by {err:.3e} (rel to max-abs)"
    return err


def pack(store, name, t):
    t = t.detach()
    if t.numel() <= 16384:
        store[name] = t.numpy().astype(np.float32 if t.dtype.is_floating_point else np.int64)
    else:
        store[name + "@digest"] = digest(t)


def load_det(module, seed=0):
    sd = G.det_state_dict(module.state_dict(), seed)
    module.load_state_dict(sd)
    return sd


def grads_ref(loss, module, retain=False):
    names, ps = zip(*module.named_parameters())
    gs = torch.autograd.grad(loss, ps, retain_graph=retain, allow_unused=True)
    return dict(zip(names, gs))


def compare_grads(gref, gmine, what, store=None, prefix="", keep=()):
    worst = 0.0
    # gradients that are mathematically zero (e.g. a conv bias feeding an InstanceNorm) are pure rounding
    # noise on both sides: compare them against a floor tied to the module's largest gradient instead
    gscale = max(float(g.abs().max()) for g in gref.values() if g is not None)
    for k, g in gref.items():
        if g is None:
            assert gmine[k] is None, f"{what}: {k} should receive no gradient"
            continue
        worst = max(worst, close(gmine[k], g, f"{what} grad {k}", 5e-4, floor=1e-2 * gscale))
    if store is not None:
        store[prefix + "none_grad_keys"] = np.array(sorted(k for k, g in gref.items() if g is None))
        for k in keep:
            pack(store, prefix + "grad/" + k, gref[k])
    return worst


def sec_fsrnet_root(keys, t0):
    # ------------------------------------------------------------------ FSRNet root (a1-a6, a14, a16)
    fsr = ref_import("model.FSRnet")
    loss_mod = ref_import("loss.loss")
    n = 2
    hr = G.synth_faces(n, 112, seed=1)
    lr = G.synth_lr_from_hr(hr)
    hm = G.synth_heatmap(n, 28, 97, 1.3, seed=2)
    par = G.synth_parsing(n, 28, 11, seed=2)

    nets = dict(coarse=fsr.Course_SR_Network(), encoder=fsr.Fine_SR_Encoder(),
                prior=fsr.Prior_Estimation_Network(), decoder=fsr.Fine_SR_Decoder())
    sds = {}
    for name, m in nets.items():
        sds[name] = load_det(m)
        keys["fsrnet_root." + name] = {k: list(v.shape) for k, v in m.state_dict().items()}
    st = {}
    # module-level forwards
    feat, coarse = nets["coarse"](lr)
    f2, c2 = R.coarse_sr(sds["coarse"], lr)
    close(f2, feat, "coarse feat"); close(c2, coarse, "coarse img")
    pack(st, "coarse/feat", feat); pack(st, "coarse/img", coarse)
    enc = nets["encoder"](hr)
    close(R.fine_encoder(sds["encoder"], hr), enc, "encoder"); pack(st, "encoder/out", enc)
    pf, lmk, prs = nets["prior"](hr)
    pf2, lmk2, prs2 = R.prior_net(sds["prior"], hr)
    close(pf2, pf, "prior feat"); close(lmk2, lmk, "prior lmk"); close(prs2, prs, "prior parsing")
    pack(st, "prior/feat", pf); pack(st, "prior/landmark", lmk); pack(st, "prior/parsing", prs)
    dec_in = torch.cat((pf, enc), 1)
    dec = nets["decoder"](dec_in)
    close(R.fine_decoder(sds["decoder"], dec_in), dec, "decoder"); pack(st, "decoder/out", dec)

    # config 1 step: 12 * mse97(coarse, hr) -> coarse net grads
    l_c1 = 12.0 * loss_mod.MSELossFunc()(coarse, hr)
    g_ref = grads_ref(l_c1, nets["coarse"], retain=True)
    l_mine, _, g_mine = R.coarse_step_grads(sds["coarse"], lr, hr)
    close(l_mine, l_c1, "c1 loss", 1e-5)
    compare_grads(g_ref, g_mine, "c1", st, "c1/", keep=("conv_input.weight", "conv_mid.weight", "bn_mid.weight",
                                                       "bn_mid.bias", "relu.weight", "residual.1.conv2.weight",
                                                       "residual.0.relu_out.weight", "residual.2.in1.weight"))
    st["c1/loss"] = np.float64(l_c1.item())

    # composed FHN step (train_FHN.py:274-279 composition + Face_Hallucination_sub_Net.py:225-247 loss map)
    for m in nets.values():
        m.zero_grad()
    _, coarse = nets["coarse"](lr)
    pf, lmk, prs = nets["prior"](coarse)
    ef = nets["encoder"](coarse)
    sr = nets["decoder"](torch.cat((pf, ef), 1))
    mse = loss_mod.MSELossFunc()
    pix = mse(sr, hr)
    losses = dict(coarse=12.0 * mse(coarse, hr), encoder=10.0 * pix,
                  prior=pix + loss_mod.MSELoss_Landmark()(lmk, hm) + loss_mod.CrossEntropyLoss2d()(prs, par),
                  decoder=10.0 * pix)
    l2, outs2, g2 = R.fhn_step_grads(sds, lr, hr, hm, par)
    close(outs2["sr"], sr, "fhn sr"); close(outs2["landmark"], lmk, "fhn lmk"); close(outs2["parsing"], prs, "fhn parsing")
    pack(st, "fhn/sr", sr); pack(st, "fhn/coarse", coarse); pack(st, "fhn/landmark", lmk); pack(st, "fhn/parsing", prs)
    keep = dict(coarse=("conv_input.weight", "residual.2.conv1.weight"),
                encoder=("conv_input.weight", "conv_end.weight", "relu.weight"),
                prior=("conv.weight", "fc.weight", "fc_landmark.bias", "hg.hg.0.3.1.conv2.weight", "hg.hg.1.0.0.relu.weight"),
                decoder=("conv_input.weight", "deconv.weight", "deconv.bias", "conv_out.weight", "bn_mid.weight"))
    for k in ("coarse", "encoder", "prior", "decoder"):
        close(l2[k], losses[k], f"fhn loss {k}", 1e-5)
        st[f"fhn/loss/{k}"] = np.float64(losses[k].item())
        compare_grads(grads_ref(losses[k], nets[k], retain=True), g2[k], f"fhn {k}", st, f"fhn/{k}/", keep[k])
    np.savez_compressed(os.path.join(OUT, "fsrnet_root.npz"), **st)
    print(f"[golden] fsrnet_root ok ({time.time() - t0:.1f}s)")



def sec_fsrnet_sr(keys, t0):
    # ------------------------------------------------------------------ FSRNet SR variant (a8, a9)
    loss_mod = ref_import("loss.loss")
    fsr_sr = ref_import("SUPER_RESOLUTION.model.FSRnet")
    st = {}
    hr1 = G.synth_faces(1, 112, seed=1, start=400)
    lr1 = G.synth_lr_from_hr(hr1)
    sr_nets = dict(coarse=fsr_sr.Coarse_SR_Network(), encoder=fsr_sr.Fine_SR_Encoder(),
                   prior=fsr_sr.Prior_Estimation_Network(), decoder=fsr_sr.Fine_SR_Decoder())
    sr_sds = {}
    for name, m in sr_nets.items():
        sr_sds[name] = load_det(m, 3)
        keys["fsrnet_sr." + name] = {k: list(v.shape) for k, v in m.state_dict().items()}
    c_img = sr_nets["coarse"](lr1)
    close(R.sr_coarse(sr_sds["coarse"], lr1), c_img, "sr coarse"); pack(st, "coarse/img", c_img)
    e_ft = sr_nets["encoder"](c_img)
    close(R.sr_encoder(sr_sds["encoder"], c_img), e_ft, "sr encoder"); pack(st, "encoder/out", e_ft)
    p_ft, p_lm, p_pa = sr_nets["prior"](c_img)
    m_ft, m_lm, m_pa = R.sr_prior(sr_sds["prior"], c_img)
    close(m_ft, p_ft, "sr prior feat", 5e-4); close(m_lm, p_lm, "sr prior lmk", 5e-4); close(m_pa, p_pa, "sr prior parsing", 5e-4)
    pack(st, "prior/feat", p_ft); pack(st, "prior/landmark", p_lm); pack(st, "prior/parsing", p_pa)
    d_img = sr_nets["decoder"](torch.cat((p_ft, e_ft), 1))
    close(R.sr_decoder(sr_sds["decoder"], torch.cat((p_ft, e_ft), 1)), d_img, "sr decoder", 5e-4); pack(st, "decoder/img", d_img)
    # one gradient check through the coarse generator: L = mse97(coarse, hr)
    l_sr = loss_mod.MSELossFunc()(c_img, hr1)
    g_ref = grads_ref(l_sr, sr_nets["coarse"], retain=True)
    sdg = R.with_grad(sr_sds["coarse"])
    g_mine = R.grads_of(R.mse97(R.sr_coarse(sdg, lr1), hr1), sdg)
    compare_grads(g_ref, g_mine, "sr coarse", st, "coarse/", keep=("model.1.weight", "model.13.weight", "model.18.relu.weight",
                                                                   "model.22.weight", "model.29.weight", "out.1.weight"))
    st["coarse/loss"] = np.float64(l_sr.item())
    np.savez_compressed(os.path.join(OUT, "fsrnet_sr.npz"), **st)
    print(f"[golden] fsrnet_sr ok ({time.time() - t0:.1f}s)")



def sec_irse(keys, t0):
    # ------------------------------------------------------------------ IR-50 / IR-SE-50 (a10, a11, a13, a19)
    irse = ref_import("SUPER_RESOLUTION.model.model_irse")
    gdc = ref_import("SUPER_RESOLUTION.model.GroupDepthConv")
    st = {}
    x = G.synth_faces(8, 112, seed=1, start=100)
    tgt = G.synth_labels(8, 512, seed=2)
    for tag, ctor, se in (("ir50", irse.IR_50, False), ("irse50", irse.IR_SE_50, True)):
        net = ctor([112, 112])
        sd = load_det(net)
        keys[tag] = {k: list(v.shape) for k, v in net.state_dict().items()}
        net.eval()
        with torch.no_grad():
            emb = net(x[:2])
            feats, _, last, _ = gdc.FeatureExtractor()(net.input_layer(x[:2]), ["2", "6", "20", "21", "22", "23"], net.body)
            e2, taps = R.ir_backbone(sd, x[:2], se=se, train=False, taps=(2, 6, 20, 21, 22, 23))
        close(e2, emb, f"{tag} eval emb")
        for i, k in enumerate(("2", "6", "20", "21", "22", "23")):
            close(taps[i], feats[k], f"{tag} tap {k}")
            pack(st, f"{tag}/eval/tap{k}", feats[k])
        pack(st, f"{tag}/eval/emb", emb)
        # train step: CE on the 512-d output (train_teacher_model.py:189-202); Dropout RNG pinned by p=0
        net.train()
        net.output_layer[1].p = 0.0
        out = net(x)
        loss = F.cross_entropy(out, tgt)
        g_ref = grads_ref(loss, net)
        l_m, e_m, g_m, stats = R.teacher_step_grads(sd, x, tgt, se=se, drop_mask=None)
        close(e_m, out, f"{tag} train emb"); close(l_m, loss, f"{tag} train loss", 1e-5)
        compare_grads(g_ref, g_m, f"{tag} train", st, f"{tag}/train/",
                      keep=("input_layer.0.weight", "input_layer.1.weight", "input_layer.2.weight",
                            "body.0.res_layer.1.weight", "body.3.shortcut_layer.0.weight", "body.3.shortcut_layer.1.bias",
                            "body.23.res_layer.4.weight", "output_layer.4.weight", "output_layer.3.bias")
                      + (("body.7.res_layer.5.fc1.weight", "body.7.res_layer.5.fc2.weight") if se else ()))
        pack(st, f"{tag}/train/emb", out)
        st[f"{tag}/train/loss"] = np.float64(loss.item())
        new_sd = net.state_dict()
        for k in ("input_layer.1.running_mean", "input_layer.1.running_var", "body.23.res_layer.4.running_var",
                  "output_layer.4.running_mean"):
            close(stats[k], new_sd[k], f"{tag} {k}")
            pack(st, f"{tag}/train/stats/{k}", new_sd[k])
        print(f"[golden] {tag} ok ({time.time() - t0:.1f}s)")
    np.savez_compressed(os.path.join(OUT, "irse.npz"), **st)



def sec_resnet_kd(keys, t0):
    # ------------------------------------------------------------------ ResNet-34 + KD step (a12, a18)
    irse = ref_import("SUPER_RESOLUTION.model.model_irse")
    gdc = ref_import("SUPER_RESOLUTION.model.GroupDepthConv")
    resnet = ref_import("model.resnet")
    st = {}
    x = G.synth_faces(8, 112, seed=1, start=200)
    teacher = irse.IR_50([112, 112]); t_sd = load_det(teacher, 0); teacher.eval()
    student = resnet.ResNet_34(); s_sd = load_det(student, 1)
    assistant = resnet.ResNet_34(); a_sd = load_det(assistant, 2)
    keys["resnet34"] = {k: list(v.shape) for k, v in student.state_dict().items()}
    student.eval()
    with torch.no_grad():
        ref_out = student(x[:2])
        mine = R.resnet34(s_sd, x[:2], train=False)
    for i, nm in enumerate(("emb", "x1", "x2", "x3", "x4")):
        close(mine[i], ref_out[i], f"resnet34 eval {nm}")
        pack(st, f"r34/eval/{nm}", ref_out[i])
    student.train(); assistant.train()
    with torch.no_grad():
        feats, _, last, _ = gdc.FeatureExtractor()(teacher.input_layer(x), ["2", "6", "20", "23"], teacher.body)
        t_emb = teacher.output_layer(last)
    t = (t_emb, feats["2"], feats["6"], feats["20"], feats["23"])
    s = student(x)
    a = assistant(x)
    crit = torch.nn.MSELoss()
    s_loss = crit(s[0], t[0].detach())
    a_loss = crit(t[1] - s[1], a[1]) + crit(t[2] - s[2], a[2]) + crit(t[3] - s[3], a[3]) + crit(t[4] - s[4], a[4]) \
        + crit(t[0] - s[0], a[0])
    gs_ref = grads_ref(s_loss, student, retain=True)
    ga_ref = grads_ref(a_loss, assistant, retain=True)
    (sl, al), gs, ga, _, s_m, a_m = R.kd_step_grads(t_sd, s_sd, a_sd, x, se=False)
    close(sl, s_loss, "kd student loss", 1e-5); close(al, a_loss, "kd assistant loss", 1e-5)
    compare_grads(gs_ref, gs, "kd student", st, "kd/student/", keep=("conv1.weight", "layer1.0.bn2.weight",
                                                                    "layer4.2.conv2.weight", "fc.bias", "bn_o2.weight"))
    compare_grads(ga_ref, ga, "kd assistant", st, "kd/assistant/", keep=("conv1.weight", "layer2.0.downsample.0.weight",
                                                                        "layer3.5.bn1.bias", "fc.bias"))
    st["kd/student_loss"] = np.float64(s_loss.item())
    st["kd/assistant_loss"] = np.float64(a_loss.item())
    pack(st, "kd/t_emb", t_emb); pack(st, "kd/s_emb", s[0]); pack(st, "kd/a_emb", a[0])
    np.savez_compressed(os.path.join(OUT, "resnet_kd.npz"), **st)
    print(f"[golden] resnet34 + kd ok ({time.time() - t0:.1f}s)")



def sec_losses_roc(keys, t0):
    # ------------------------------------------------------------------ losses (a14)
    loss_mod = ref_import("loss.loss")
    st = {}
    a_ = torch.from_numpy(G.normal("loss/a", 2 * 3 * 16 * 16).reshape(2, 3, 16, 16).astype(np.float32))
    b_ = torch.from_numpy(G.normal("loss/b", 2 * 3 * 16 * 16).reshape(2, 3, 16, 16).astype(np.float32))
    lm = torch.from_numpy(G.normal("loss/lm", 2 * 97 * 8 * 8).reshape(2, 97, 8, 8).astype(np.float32)) * 0.1
    lt = G.synth_heatmap(2, 8, 5, 1.3)
    pl = torch.from_numpy(G.normal("loss/pl", 2 * 11 * 8 * 8).reshape(2, 11, 8, 8).astype(np.float32))
    pt = G.synth_parsing(2, 8, 11)
    for nm, ref_l, mine_l, args in (("mse97", loss_mod.MSELossFunc(), R.mse97, (a_, b_)),
                                    ("landmark", loss_mod.MSELoss_Landmark(), R.landmark_loss, (lm, lt)),
                                    ("nll2d", loss_mod.CrossEntropyLoss2d(), R.nll2d, (pl, pt))):
        x0 = args[0].clone().requires_grad_(True)
        lv = ref_l(x0, args[1])
        g0, = torch.autograd.grad(lv, x0)
        x1 = args[0].clone().requires_grad_(True)
        lv1 = mine_l(x1, args[1])
        g1, = torch.autograd.grad(lv1, x1)
        close(lv1, lv, nm, 1e-6); close(g1, g0, nm + " grad", 1e-5)
        st[nm + "/in"] = args[0].numpy(); st[nm + "/target"] = args[1].numpy()
        st[nm + "/loss"] = np.float64(lv.item()); st[nm + "/grad"] = g0.numpy()
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **st)

    # ------------------------------------------------------------------ pair distance + ROC (a20)
    spec = importlib.util.spec_from_file_location("ref_utils_utils", os.path.join(REF, "utils", "utils.py"))
    uu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(uu)
    from sklearn.model_selection import KFold
    p = 600
    e1, e2, same = G.synth_pairs(p, 512, seed=0)
    e1 *= 1.0; e2 *= 1.0
    thresholds = np.arange(0, 12000, 3)
    np.random.seed(1234)
    tpr, fpr, acc, best = uu.calculate_roc(thresholds, e1, e2, same, nrof_folds=10, pca=0)
    np.random.seed(1234)
    folds = list(KFold(n_splits=10, shuffle=True).split(np.arange(p)))
    tpr2, fpr2, acc2, best2 = R.calculate_roc(thresholds, e1, e2, same, folds)
    assert np.array_equal(tpr, tpr2) and np.array_equal(fpr, fpr2) and acc == acc2 and np.array_equal(best, best2), \
        "ROC restatement differs from utils/utils.py:calculate_roc"
    fold_id = np.empty(p, dtype=np.int64)
    for f, (_, te) in enumerate(folds):
        fold_id[te] = f
    np.savez_compressed(os.path.join(OUT, "roc.npz"), tpr=tpr, fpr=fpr, acc=np.float64(acc), best=best,
                        fold_id=fold_id, dist=uu.np.sum(np.square(e1 - e2), 1), p=np.int64(p))
    print(f"[golden] losses + roc ok ({time.time() - t0:.1f}s)")



def _taps5(gdc, net, x, keys=("2", "6", "20", "23")):
    """(emb, t1..t4) of a reference IR backbone through the reference's own FeatureExtractor (GroupDepthConv.py:35-45)."""
    feats, _, last, _ = gdc.FeatureExtractor()(net.input_layer(x), list(keys), net.body)
    return (net.output_layer(last), *[feats[k] for k in keys])


def sec_c4(keys, t0):
    # ------------------------------------------------------------------ C4 composed step (BASELINE configs[3])
    fsr = ref_import("model.FSRnet")
    irse = ref_import("SUPER_RESOLUTION.model.model_irse")
    gdc = ref_import("SUPER_RESOLUTION.model.GroupDepthConv")
    st = {}
    n = 4
    hr = G.synth_faces(n, 112, seed=1, start=600)
    lr = G.synth_lr_from_hr(hr)
    nets = dict(coarse=fsr.Course_SR_Network(), encoder=fsr.Fine_SR_Encoder(),
                prior=fsr.Prior_Estimation_Network(), decoder=fsr.Fine_SR_Decoder())
    sds = {k: load_det(m, 5) for k, m in nets.items()}
    teacher = irse.IR_SE_50([112, 112]); t_sd = load_det(teacher, 0); teacher.eval()
    student = irse.IR_SE_50([112, 112]); s_sd = load_det(student, 1); student.train()
    assistant = irse.IR_SE_50([112, 112]); a_sd = load_det(assistant, 2); assistant.train()
    student.output_layer[1].p = 0.0       # Dropout RNG pinned off
    assistant.output_layer[1].p = 0.0
    # composition of SUPER_RESOLUTION/train_FHN.py:274-279 followed by distill_main.py:59-70
    _, coarse = nets["coarse"](lr)
    pf, lmk, prs = nets["prior"](coarse)
    ef = nets["encoder"](coarse)
    sr = nets["decoder"](torch.cat((pf, ef), 1))
    with torch.no_grad():
        t = _taps5(gdc, teacher, hr)
    s = _taps5(gdc, student, sr)
    a = _taps5(gdc, assistant, sr)
    crit = torch.nn.MSELoss()
    s_loss = crit(s[0], t[0].detach())
    a_loss = crit(t[1] - s[1], a[1]) + crit(t[2] - s[2], a[2]) + crit(t[3] - s[3], a[3]) + crit(t[4] - s[4], a[4]) \
        + crit(t[0] - s[0], a[0])
    (sl, al), outs, g, stats = R.c4_step_grads(sds, s_sd, a_sd, t_sd, lr, hr)
    close(outs["sr"], sr, "c4 sr"); close(outs["s_emb"], s[0], "c4 student emb"); close(outs["a_emb"], a[0], "c4 assistant emb")
    close(outs["t_emb"], t[0], "c4 teacher emb")
    for i in range(4):
        close(outs["s_taps"][i], s[1 + i], f"c4 student tap {i}"); close(outs["a_taps"][i], a[1 + i], f"c4 assistant tap {i}")
    close(sl, s_loss, "c4 student loss", 1e-5); close(al, a_loss, "c4 assistant loss", 1e-5)
    pack(st, "sr", sr); pack(st, "s_emb", s[0]); pack(st, "a_emb", a[0]); pack(st, "t_emb", t[0])
    pack(st, "s_tap3", s[4]); pack(st, "a_tap0", a[1])
    st["student_loss"] = np.float64(s_loss.item()); st["assistant_loss"] = np.float64(a_loss.item())
    keep = dict(student=("input_layer.0.weight", "body.0.res_layer.1.weight", "body.7.res_layer.5.fc1.weight",
                         "body.23.res_layer.4.weight", "output_layer.3.bias", "output_layer.4.weight"),
                assistant=("input_layer.0.weight", "body.3.shortcut_layer.0.weight", "body.20.res_layer.3.weight",
                           "output_layer.4.weight"),
                coarse=("conv_input.weight", "residual.2.conv1.weight", "conv_mid.weight"),
                encoder=("conv_input.weight", "conv_end.weight"),
                prior=("conv.weight", "hg.hg.1.0.0.conv1.weight"),
                decoder=("conv_input.weight", "deconv.weight", "conv_out.weight", "bn_mid.weight"))
    mods = dict(student=student, assistant=assistant, **nets)
    for k in ("student", "coarse", "prior", "encoder", "decoder"):
        compare_grads(grads_ref(s_loss, mods[k], retain=True), g[k], f"c4 {k}", st, f"{k}/", keep[k])
    compare_grads(grads_ref(a_loss, assistant, retain=True), g["assistant"], "c4 assistant", st, "assistant/", keep["assistant"])
    new_sd = student.state_dict()
    for k in ("input_layer.1.running_mean", "body.23.res_layer.4.running_var"):
        close(stats[0][k], new_sd[k], f"c4 student {k}")
        pack(st, f"student/stats/{k}", new_sd[k])
    np.savez_compressed(os.path.join(OUT, "c4.npz"), **st)
    print(f"[golden] c4 ok ({time.time() - t0:.1f}s)")


def sec_gan(keys, t0):
    # ------------------------------------------------------------------ OverallNetwork_GAN + Discriminator @224 (a7, f4)
    fsr = ref_import("model.FSRnet")
    st = {}
    n = 3
    hr = G.synth_faces(n, 224, seed=1, start=700)
    lr = G.synth_lr_from_hr(hr)
    net = fsr.OverallNetwork_GAN()
    sd = load_det(net, 7)
    keys["fsrnet_root.gan"] = {k: list(v.shape) for k, v in net.state_dict().items()}
    net.train()
    sr, coarse, lmk, prs, e1, e2 = net(lr, hr)
    m_sr, m_coarse, m_lmk, m_prs, m_e1, m_e2, stats = R.gan_forward(sd, lr, hr, train=True)
    for nm, a_, b_ in (("sr", m_sr, sr), ("coarse", m_coarse, coarse), ("landmark", m_lmk, lmk), ("parsing", m_prs, prs),
                       ("emb1", m_e1, e1), ("emb2", m_e2, e2)):
        close(a_, b_, "gan train " + nm, 5e-4)
        pack(st, "train/" + nm, b_)
    # one pinned adversarial-style pair: L_d = MSE(emb1, emb2) -> discriminator (MMD itself is undefined upstream)
    l_d = F.mse_loss(e1, e2)
    g_ref = grads_ref(l_d, net._discriminator, retain=True)
    sdg = R.with_grad(sd)
    o = R.gan_forward(sdg, lr, hr, train=True)
    g_m = R.grads_of(F.mse_loss(o[4], o[5]), {k[len("_discriminator."):]: v for k, v in sdg.items()
                                              if k.startswith("_discriminator.")})
    compare_grads(g_ref, g_m, "gan disc", st, "disc/", keep=("conv_input.weight", "bn_mid.weight", "fc.weight", "fc.bias",
                                                              "bn_end.weight"))
    st["disc/loss"] = np.float64(l_d.item())
    new_sd = net.state_dict()
    for k in ("_discriminator.bn_mid.running_mean", "_discriminator.bn_mid.running_var", "_discriminator.bn_end.running_var"):
        close(stats[k], new_sd[k], "gan " + k)
        pack(st, "train/stats/" + k, new_sd[k])
    st["train/nbt"] = np.int64(int(new_sd["_discriminator.bn_mid.num_batches_tracked"]))
    net.eval()
    net.load_state_dict(sd)
    with torch.no_grad():
        outs = net(lr[:2], hr[:2])
        mine = R.gan_forward(sd, lr[:2], hr[:2], train=False)
    for nm, a_, b_ in zip(("sr", "coarse", "landmark", "parsing", "emb1", "emb2"), mine[:6], outs):
        close(a_, b_, "gan eval " + nm, 5e-4)
        pack(st, "eval/" + nm, b_)
    np.savez_compressed(os.path.join(OUT, "gan224.npz"), **st)
    print(f"[golden] gan224 ok ({time.time() - t0:.1f}s)")


def sec_gan_step(keys, t0):
    # ------------------------------------------------------------------ adversarial loss -> optimizer map (a16 / f4)
    # Face_Hallucination_sub_Net.py:218-247 on the reference's own OverallNetwork_GAN @224; the undefined MMD import (:25) is
    # replaced by F.mse_loss(emb1, emb2) on BOTH sides -- the five (loss_k, theta_k) gradient sets are what is pinned.
    fsr = ref_import("model.FSRnet")
    loss_mod = ref_import("loss.loss")
    st = {}
    n = 3       # (train-mode BatchNorm1d over a batch of 2 is degenerate: outputs +-gamma)
    hr = G.synth_faces(n, 224, seed=1, start=900)
    lr = G.synth_lr_from_hr(hr)
    hm = G.synth_heatmap(n, 56, 97, 1.3, seed=5)
    par = G.synth_parsing(n, 56, 11, seed=5)
    net = fsr.OverallNetwork_GAN()
    sd = load_det(net, 9)
    net.train()
    sr, coarse, lmk, prs, e1, e2 = net(lr, hr)
    mse = loss_mod.MSELossFunc()
    gan_loss = -F.mse_loss(e1, e2)
    pix = mse(sr, hr)
    losses = dict(disc=gan_loss, coarse=12.0 * mse(coarse, hr), encoder=10.0 * pix - gan_loss,
                  prior=-gan_loss + pix + loss_mod.MSELoss_Landmark()(lmk, hm) + 1.0 * loss_mod.CrossEntropyLoss2d()(prs, par),
                  decoder=10.0 * pix)
    subs = dict(disc=net._discriminator, coarse=net._coarse_sr_network, encoder=net._fine_sr_encoder,
                prior=net._prior_estimation_network, decoder=net._fine_sr_decoder)
    l2, outs2, g2 = R.gan_step_grads(sd, lr, hr, hm, par)
    for nm, b_ in (("sr", sr), ("coarse", coarse), ("landmark", lmk), ("parsing", prs), ("emb1", e1), ("emb2", e2)):
        close(outs2[nm], b_, "gan_step " + nm, 5e-4)
        pack(st, "out/" + nm, b_)
    keep = dict(disc=("conv_input.weight", "bn_mid.weight", "fc.weight", "fc.bias", "bn_end.weight"),
                coarse=("conv_input.weight", "residual.2.conv1.weight", "conv_mid.weight"),
                encoder=("conv_input.weight", "conv_end.weight", "relu.weight", "residual.0.conv1.weight"),
                prior=("conv.weight", "fc.weight", "fc_landmark.bias", "hg.hg.0.3.1.conv2.weight", "residual.1.conv1.weight"),
                decoder=("conv_input.weight", "deconv.weight", "deconv.bias", "conv_out.weight", "bn_mid.weight"))
    for k in ("disc", "coarse", "encoder", "prior", "decoder"):
        close(l2[k], losses[k], f"gan_step loss {k}", 1e-5)
        st[f"loss/{k}"] = np.float64(losses[k].item())
        compare_grads(grads_ref(losses[k], subs[k], retain=True), g2[k], f"gan_step {k}", st, f"{k}/", keep[k])
    np.savez_compressed(os.path.join(OUT, "gan_step.npz"), **st)
    print(f"[golden] gan_step ok ({time.time() - t0:.1f}s)")


SECTIONS = {}   # filled below main's helpers (name -> function), in generation order


def _ref_methods(path, names):
    """The named methods of the reference's Dataset class as plain functions.  The module itself cannot be imported (torchvision /
    pandas loaders), but these methods are self-contained numpy code: their FunctionDef nodes are compiled unchanged."""
    import ast
    tree = ast.parse(open(path).read())
    fns = [n for c in ast.walk(tree) if isinstance(c, ast.ClassDef) for n in c.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {f.name for f in fns} == set(names), (path, names)
    ns = {"np": np}
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
    return type("RefMethods", (), {n: ns[n] for n in names})()


def sec_loader(keys, t0):
    """low-resolution synthesis (the PIL calls of FHN_loader.py:65-66, run on PIL itself) and heat-maps (the reference's
    own generate_hm / gaussian_k methods, FHN_loader.py:119-137 and helen_loader.py:124-143)."""
    import PIL
    from PIL import Image
    st = {}
    rng = np.random.RandomState(7)
    imgs = [(rng.rand(112, 112, 3) * 255).astype(np.uint8)]
    for k in (5, 11):   # smooth face-like fields
        lo = (rng.rand(k, k, 3) * 255).astype(np.uint8)
        imgs.append(np.array(Image.fromarray(lo).resize((112, 112), Image.BILINEAR)))
    hr = np.stack(imgs)
    scales = [8, 4, 2]
    lr = []
    for im, sc in zip(hr, scales):
        sr_img = Image.fromarray(im)
        lr_img = sr_img.resize((int(128 / sc), int(128 / sc))).resize((112, 112), Image.BICUBIC)   # FHN_loader.py:65-66 verbatim
        lr.append(np.array(lr_img))
        assert np.array_equal(lr[-1], R.lr_from_hr_u8(im, int(128 / sc))), "PIL restatement is not bit-exact"
        for low in (16, 32, 64):
            assert np.array_equal(np.array(sr_img.resize((low, low)).resize((112, 112), Image.BICUBIC)), R.lr_from_hr_u8(im, low))
    st["hr_u8"], st["lr_u8"], st["scale"] = hr, np.stack(lr), np.asarray(scales, np.int32)
    st["pil_version"] = np.asarray([int(v) for v in PIL.__version__.split(".")[:3]], np.int32)
    fhn = _ref_methods(os.path.join(REF, "SUPER_RESOLUTION", "FHN_loader.py"), ("generate_hm", "gaussian_k"))
    helen = _ref_methods(os.path.join(REF, "helen_loader.py"), ("generate_hm", "gaussian_k"))
    lm68 = rng.rand(2, 68, 2) * 130.0 - 9.0        # some landmarks fall outside the 112 x 112 map, as after rotate + crop
    lm194 = rng.rand(194, 2) * 112.0
    st["lm68"], st["lm194"] = lm68, lm194
    st["hm68"] = np.stack([fhn.generate_hm(height=112, width=112, landmark=l, s=2.0) for l in lm68])
    st["hm194"] = helen.generate_hm(height=112, width=112, landmark=lm194, s=1.3)
    for l, h in zip(lm68, st["hm68"]):
        assert np.array_equal(h, R.generate_hm(112, 112, l, 2.0)), "generate_hm restatement differs from the reference"
    assert np.array_equal(st["hm194"], R.generate_hm(112, 112, lm194, 1.3))
    assert st["hm68"].dtype == np.float32
    np.savez_compressed(os.path.join(OUT, "loader.npz"), **st)
    print(f"[golden] loader: PIL {PIL.__version__} resize chain bit-exact, heat-maps == reference methods  [{time.time() - t0:.1f}s]")



def main(argv):
    """python oracle/make_golden.py [section ...]   (no argument: every section)"""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    t0 = time.time()
    names = argv or list(SECTIONS)
    kpath = os.path.join(OUT, "state_dict_keys.json")
    keys = {}
    if argv and os.path.exists(kpath):      # partial run: keep the other sections' key dumps
        with open(kpath) as f:
            keys = json.load(f)
    for nm in names:
        SECTIONS[nm](keys, t0)
    with open(kpath, "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    print(f"[golden] wrote fixtures to {OUT} in {time.time() - t0:.1f}s")


SECTIONS.update(fsrnet_root=sec_fsrnet_root, fsrnet_sr=sec_fsrnet_sr, irse=sec_irse, resnet_kd=sec_resnet_kd,
                losses_roc=sec_losses_roc, c4=sec_c4, gan=sec_gan, gan_step=sec_gan_step, loader=sec_loader)

if __name__ == "__main__":
    main(sys.argv[1:])
