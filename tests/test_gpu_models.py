"""GPU: model-level parity of the HIP path against the CPU oracle (oracle/cpu_ref.py) and the golden
fixtures generated from the reference.  North-star tolerance: 1e-3 relative fp32 (BASELINE.json)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cpu_ref as R
from oracle import detgen as G
from tests.helpers import GOLD, check_against, grad_floor, load_gold, rel_err

pytestmark = pytest.mark.gpu
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


DEV = "cuda:0"
TOL = 1e-3       # forward outputs: SR images, heat-maps, embeddings, losses (north_star)
# Parameter gradients after 50-100 layers of backprop through small-batch train-mode Batch/InstanceNorm are
# ill-conditioned: the HIP path's own run-to-run spread (fp32 atomic summation order in the statistics, ~1e-7
# relative) already moves first-layer gradients by up to ~5e-3 of their max-abs, so they are held to 1e-2
# (3e-2 for the 2 x ResNet-34 + IR-50 KD chain); forward outputs and losses are held to the north-star 1e-3.
GRAD_TOL = 1e-2
KD_GRAD_TOL = 3e-2
# These flat caps are justified, not assumed: tests/golden/grad_spread.json (oracle/grad_conditioning.py, CPU only) holds the
# CPU oracle's OWN fp32-vs-fp64 error of the same gradients on the same inputs -- the distance of the fp32 fixtures from the exact
# answer.  spread_tol() bounds the HIP path's error by min(flat cap, K_SPREAD x the oracle's worst-tensor error): an fp32
# implementation is not asked to sit closer to the fixture than the fixture sits to the truth, and never gets more than the cap.
K_SPREAD = 2.0
with open(os.path.join(GOLD, "grad_spread.json")) as _f:
    GRAD_SPREAD = json.load(_f)


def spread_tol(case, net, cap):
    return min(cap, K_SPREAD * GRAD_SPREAD[case][net]["worst"])


def load_det(module, seed=0):
    sd = G.det_state_dict(module.state_dict(), seed)
    module.load_state_dict(sd)
    return module.to(DEV), sd


def grads_by_name(module):
    return {k: p.grad for k, p in module.named_parameters()}


def test_coarse_c1_step_matches_oracle_and_fixture():
    """BASELINE config 1: Course_SR_Network fwd + bwd of 12*mse97 (N=2 here, the fixture's batch)."""
    import xrface
    from xrface.loss.loss import MSELossFunc
    from xrface.model.FSRnet import Course_SR_Network
    xrface.set_compute_dtype(torch.float32)
    st = load_gold("fsrnet_root.npz")
    net, sd = load_det(Course_SR_Network())
    hr = G.synth_faces(2, 112, seed=1)
    lr = G.synth_lr_from_hr(hr)
    feat, img = net(lr.to(DEV))
    assert feat.shape == (2, 64, 112, 112) and img.shape == (2, 3, 112, 112)
    check_against(st, "coarse/feat", feat, TOL)
    check_against(st, "coarse/img", img, TOL)
    loss = 12.0 * MSELossFunc()(img, hr.to(DEV))
    loss.backward()
    assert abs(loss.item() - float(st["c1/loss"])) <= TOL * abs(float(st["c1/loss"]))
    g = grads_by_name(net)
    l_ref, _, g_ref = R.coarse_step_grads(sd, lr, hr)
    gscale = max(float(v.abs().max()) for v in g_ref.values() if v is not None)
    for k, gr in g_ref.items():
        if gr is None:
            assert g[k] is None, f"{k} must keep .grad None (stock optimizers skip it)"
            continue
        err = float((g[k].cpu().double() - gr.double()).abs().max()) / max(float(gr.abs().max()), 1e-2 * gscale)
        assert err < 2 * TOL, (k, err)
    for k in ("conv_input.weight", "conv_mid.weight", "bn_mid.weight", "relu.weight", "residual.1.conv2.weight",
              "residual.0.relu_out.weight", "residual.2.in1.weight"):
        check_against(st, "c1/grad/" + k, g[k], 2 * TOL)


def test_fhn_modules_forward():
    import xrface
    from xrface.model import FSRnet
    xrface.set_compute_dtype(torch.float32)
    st = load_gold("fsrnet_root.npz")
    hr = G.synth_faces(2, 112, seed=1).to(DEV)
    enc, _ = load_det(FSRnet.Fine_SR_Encoder())
    e = enc(hr)
    check_against(st, "encoder/out", e, TOL)
    prior, _ = load_det(FSRnet.Prior_Estimation_Network())
    pf, lmk, par = prior(hr)
    assert lmk.shape == (2, 97, 28, 28) and par.shape == (2, 11, 28, 28)
    check_against(st, "prior/feat", pf, TOL)
    check_against(st, "prior/landmark", lmk, TOL)
    check_against(st, "prior/parsing", par, TOL)
    dec, _ = load_det(FSRnet.Fine_SR_Decoder())
    out = dec(torch.cat((pf, e), 1))
    check_against(st, "decoder/out", out, TOL)


def test_fhn_step_per_pair_gradients():
    """One forward, per-(loss_k, theta_k) gradients at pre-step weights."""
    import xrface
    from xrface.steps import fhn_step
    from xrface.model import FSRnet
    xrface.set_compute_dtype(torch.float32)
    st = load_gold("fsrnet_root.npz")
    nets = {}
    for k, ctor in (("coarse", FSRnet.Course_SR_Network), ("encoder", FSRnet.Fine_SR_Encoder),
                    ("prior", FSRnet.Prior_Estimation_Network), ("decoder", FSRnet.Fine_SR_Decoder)):
        nets[k], _ = load_det(ctor())
    hr = G.synth_faces(2, 112, seed=1)
    lr = G.synth_lr_from_hr(hr)
    hm = G.synth_heatmap(2, 28, 97, 1.3, seed=2)
    par = G.synth_parsing(2, 28, 11, seed=2)
    losses, outs = fhn_step(nets, lr.to(DEV), hr.to(DEV), hm.to(DEV), par.to(DEV))
    check_against(st, "fhn/sr", outs["sr"], TOL)
    check_against(st, "fhn/landmark", outs["landmark"], TOL)
    for k in ("coarse", "encoder", "prior", "decoder"):
        ref = float(st[f"fhn/loss/{k}"])
        assert abs(losses[k].item() - ref) <= TOL * abs(ref), (k, losses[k].item(), ref)
        none_ref = set(st[f"fhn/{k}/none_grad_keys"].tolist())
        got_none = {n for n, p in nets[k].named_parameters() if p.grad is None}
        assert got_none == none_ref, (k, got_none ^ none_ref)
        for key in st.files:
            pre = f"fhn/{k}/grad/"
            if key.startswith(pre):
                name = key[len(pre):].replace("@digest", "")
                check_against(st, pre + name, dict(nets[k].named_parameters())[name].grad, GRAD_TOL,
                              floor=grad_floor(st, pre))


@pytest.mark.parametrize("tag,se", [("ir50", False), ("irse50", True)])
def test_ir_backbone_eval_and_train(tag, se):
    import xrface
    from xrface.loss.loss import CrossEntropyLoss
    from xrface.model import model_irse
    from xrface.model.GroupDepthConv import FeatureExtractor
    xrface.set_compute_dtype(torch.float32)
    st = load_gold("irse.npz")
    net, sd = load_det((model_irse.IR_SE_50 if se else model_irse.IR_50)([112, 112]))
    x = G.synth_faces(8, 112, seed=1, start=100)
    tgt = G.synth_labels(8, 512, seed=2)
    net.eval()
    with torch.no_grad():
        emb = net(x[:2].to(DEV))
        feats, _, last, _ = FeatureExtractor()(net.input_layer(x[:2].to(DEV)), ["2", "6", "20", "21", "22", "23"], net.body)
    check_against(st, f"{tag}/eval/emb", emb, TOL)
    e_rel = float((emb.cpu() - torch.from_numpy(st[f"{tag}/eval/emb"])).norm(dim=1).max() /
                  torch.from_numpy(st[f"{tag}/eval/emb"]).norm(dim=1).min())
    assert e_rel < TOL, f"embedding L2 error {e_rel:.2e}"
    for k in ("2", "6", "20", "21", "22", "23"):
        check_against(st, f"{tag}/eval/tap{k}", feats[k], TOL)
    # train step, Dropout pinned off (p = 0) exactly like the fixture
    # (through steps.teacher_step, the counterpart of train_teacher_model.py:189-202; no optimizer: gradients are inspected)
    from xrface.steps import teacher_step
    net.train()
    net.output_layer[1].p = 0.0
    loss, out = teacher_step(net, x.to(DEV), tgt.to(DEV), criterion=CrossEntropyLoss())
    check_against(st, f"{tag}/train/emb", out, TOL)
    assert abs(loss.item() - float(st[f"{tag}/train/loss"])) <= TOL * abs(float(st[f"{tag}/train/loss"]))
    g = grads_by_name(net)
    floor = grad_floor(st, f"{tag}/train/grad/")
    for key in st.files:
        pre = f"{tag}/train/grad/"
        if key.startswith(pre):
            name = key[len(pre):].replace("@digest", "")
            check_against(st, pre + name, g[name], GRAD_TOL, floor=floor)
    new_sd = net.state_dict()
    for k in ("input_layer.1.running_mean", "input_layer.1.running_var", "body.23.res_layer.4.running_var",
              "output_layer.4.running_mean"):
        check_against(st, f"{tag}/train/stats/{k}", new_sd[k], TOL)
    assert int(new_sd["input_layer.1.num_batches_tracked"]) == 1
    # the same step with a stock optimizer (train_teacher_model.py:200-202: zero_grad / backward / step): p <- p - lr * grad
    w_name = "body.23.res_layer.3.weight"
    w0, g0 = net.get_parameter(w_name).detach().clone(), g[w_name].clone()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    loss2, _ = teacher_step(net, x.to(DEV), tgt.to(DEV), optimizer=opt, criterion=CrossEntropyLoss())
    assert abs(loss2.item() - loss.item()) <= 1e-4 * abs(loss.item())
    step_err = float(((w0 - net.get_parameter(w_name).detach()) / 0.1 - g0).abs().max() / g0.abs().max())
    assert step_err < 5e-3, step_err


def test_irse50_train_step_at_a_ragged_medium_batch_matches_the_live_oracle():
    """The fixtures pin the IR-SE-50 step at N = 8; the full-size (N = 256) tests are property checks of the bf16 path against the
    fp32 path.  In between: N = 40 (not a multiple of any tile height: 40 x 12 544 .. 40 x 49 rows, ragged last tiles in every GEMM,
    two and a half 16-image groups in the per-image passes) against the CPU oracle run here on the same weights and inputs, in fp32
    AND in fp64 -- output, loss and BatchNorm running statistics against the fp32 oracle at the north-star 1e-3; every parameter
    gradient against the fp64 oracle, bounded as everywhere in this file by K_SPREAD x the fp32 oracle's OWN distance from fp64 on the
    same tensors (measured in the test: ~6e-3 of max-abs on conv1 of units 9 and 21) and never by more than KD_GRAD_TOL."""
    import xrface
    from oracle.grad_conditioning import _to, spread
    from xrface.loss.loss import CrossEntropyLoss
    from xrface.model import model_irse
    from xrface.steps import teacher_step
    xrface.set_compute_dtype(torch.float32)
    n = 40
    net, sd = load_det(model_irse.IR_SE_50([112, 112]), seed=3)
    x = G.synth_faces(n, 112, seed=5, start=300)
    tgt = G.synth_labels(n, 512, seed=6)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref_loss, ref_emb, g32, ref_stats = R.teacher_step_grads(sd, x, tgt, se=True)
    _, _, g64, _ = R.teacher_step_grads(_to(sd, torch.float64), x.double(), tgt, se=True)
    own = spread(g32, g64)                      # the fp32 oracle against fp64: what "exact" means for these gradients
    net.train()
    net.output_layer[1].p = 0.0
    # deterministic summation order: on these tensors the order of the fp32 atomics alone moves the HIP result between 5e-3 and 1e-2
    # of max-abs from run to run (conditioning ~1e5 x fp32 epsilon) -- the comparison should not depend on the scheduler
    xrface.set_deterministic(True)
    try:
        loss, out = teacher_step(net, x.to(DEV), tgt.to(DEV), criterion=CrossEntropyLoss())
        torch.cuda.synchronize()
    finally:
        xrface.set_deterministic(False)
    e_out = rel(out, ref_emb)
    assert e_out < TOL, f"train-mode output {e_out:.2e}"
    assert abs(loss.item() - float(ref_loss)) <= TOL * abs(float(ref_loss))
    g = {k: v.detach().float().cpu() for k, v in grads_by_name(net).items()}
    hip = spread(g, g64)
    tol = min(KD_GRAD_TOL, max(K_SPREAD * own["worst"], 2e-3))
    print(f"N={n}: output {e_out:.2e}; gradients vs fp64: HIP worst {hip['worst']:.2e} ({hip['worst_tensor']}), "
          f"fp32 oracle worst {own['worst']:.2e} ({own['worst_tensor']}), bound {tol:.2e}; cosine {hip['cosine']:.8f}")
    assert hip["worst"] <= tol, (hip["worst_tensor"], hip["worst"], own["worst"])
    assert hip["cosine"] > 1 - 1e-5
    new_sd = net.state_dict()
    for k, v in ref_stats.items():
        if k in new_sd and new_sd[k].dtype.is_floating_point:
            assert rel(new_sd[k], v) < TOL, k


def test_irse50_two_plane_fp32_mode_meets_the_embedding_tolerance():
    """north_star: "embeddings within 1e-3 of the CPU reference".  bf16 tensors miss it (7e-3, asserted < 1.1e-2 below); the
    three-plane fp32 mode holds it at ~1e-6 but costs six MFMAs per product.  XR_F32X2 (two planes, three MFMAs) must hold it too:
    eval-mode embedding relative L2 error < 1e-3 against the reference fixture, train-mode output / loss within 1e-3, gradients
    within the same bound as the three-plane mode."""
    import xrface
    from xrface.loss.loss import CrossEntropyLoss
    from xrface.model import model_irse
    from xrface.steps import teacher_step
    xrface.set_compute_dtype("fp32x2")
    st = load_gold("irse.npz")
    net, sd = load_det(model_irse.IR_SE_50([112, 112]))
    x = G.synth_faces(8, 112, seed=1, start=100)
    tgt = G.synth_labels(8, 512, seed=2)
    net.eval()
    with torch.no_grad():
        emb = net(x[:2].to(DEV))
    ref = torch.from_numpy(st["irse50/eval/emb"])
    e_rel = float((emb.float().cpu() - ref).norm(dim=1).max() / ref.norm(dim=1).min())
    print(f"fp32x2 embedding rel L2 error {e_rel:.3e}")
    assert e_rel < TOL, f"two-plane fp32 mode: embedding L2 error {e_rel:.2e}"
    check_against(st, "irse50/eval/emb", emb, TOL, what="fp32x2 irse50/eval/emb")
    net.train()
    net.output_layer[1].p = 0.0
    loss, out = teacher_step(net, x.to(DEV), tgt.to(DEV), criterion=CrossEntropyLoss())
    check_against(st, "irse50/train/emb", out, TOL, what="fp32x2 irse50/train/emb")
    assert abs(loss.item() - float(st["irse50/train/loss"])) <= TOL * abs(float(st["irse50/train/loss"]))
    g = grads_by_name(net)
    floor = grad_floor(st, "irse50/train/grad/")
    for key in st.files:
        pre = "irse50/train/grad/"
        if key.startswith(pre):
            name = key[len(pre):].replace("@digest", "")
            check_against(st, pre + name, g[name], GRAD_TOL, floor=floor, what="fp32x2 " + pre + name)


def test_ir_dropout_mask_matches_oracle():
    import xrface
    from xrface.model import model_irse
    xrface.set_compute_dtype(torch.float32)
    net, sd = load_det(model_irse.IR_50([112, 112]))
    x = G.synth_faces(4, 112, seed=1, start=300)
    mask_nchw = (torch.from_numpy(G.uniform01("mask", 4 * 512 * 49).reshape(4, 512, 7, 7)) < 0.5)
    net.train()
    net.output_layer[1].inject_mask = mask_nchw.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = net(x.to(DEV))
    ref, _ = R.ir_backbone(sd, x, se=False, train=True, drop_mask=mask_nchw.float())
    assert rel_err(out, ref) < TOL


def test_resnet34_and_kd_step():
    import xrface
    from xrface.model import model_irse, resnet
    from xrface.steps import kd_step
    xrface.set_compute_dtype(torch.float32)
    st = load_gold("resnet_kd.npz")
    x = G.synth_faces(8, 112, seed=1, start=200)
    teacher, _ = load_det(model_irse.IR_50([112, 112]), 0)
    student, _ = load_det(resnet.ResNet_34(), 1)
    assistant, _ = load_det(resnet.ResNet_34(), 2)
    student.eval()
    with torch.no_grad():
        out = student(x[:2].to(DEV))
    for i, nm in enumerate(("emb", "x1", "x2", "x3", "x4")):
        check_against(st, f"r34/eval/{nm}", out[i], TOL)
    (sl, al), s_out, a_out, t_out = kd_step(teacher, student, assistant, x.to(DEV))
    assert abs(sl.item() - float(st["kd/student_loss"])) <= TOL * abs(float(st["kd/student_loss"]))
    assert abs(al.item() - float(st["kd/assistant_loss"])) <= TOL * abs(float(st["kd/assistant_loss"]))
    check_against(st, "kd/t_emb", t_out[0], TOL)
    check_against(st, "kd/s_emb", s_out[0], TOL)
    for who, net in (("student", student), ("assistant", assistant)):
        g = grads_by_name(net)
        for key in st.files:
            pre = f"kd/{who}/grad/"
            if key.startswith(pre):
                name = key[len(pre):].replace("@digest", "")
                check_against(st, pre + name, g[name], spread_tol("kd", who, KD_GRAD_TOL), floor=grad_floor(st, pre))


def test_bf16_throughput_mode_embedding_error_is_reported():
    """bf16 mode is the throughput path; its embedding error versus the fp32 fixture is bounded loosely
    (bf16 has ~3 significant digits) and printed for DESIGN.md."""
    import xrface
    from xrface.model import model_irse
    st = load_gold("irse.npz")
    net, _ = load_det(model_irse.IR_SE_50([112, 112]))
    net.eval()
    x = G.synth_faces(2, 112, seed=1, start=100)
    xrface.set_compute_dtype(torch.bfloat16)
    try:
        with torch.no_grad():
            emb = net(x.to(DEV)).float()
    finally:
        xrface.set_compute_dtype(torch.float32)
    ref = torch.from_numpy(st["irse50/eval/emb"])
    err = float((emb.cpu() - ref).norm(dim=1).max() / ref.norm(dim=1).min())
    print(f"[bf16] IR-SE-50 embedding relative L2 error vs fp32 reference: {err:.3e}")
    # measured 7.3e-3 (bf16 operands, 8 significand bits, through 50 convolutions); bench.py reports the same quantity for the mode
    # it times (embedding_rel_l2_vs_cpu).  1.5 x the measurement, not an order of magnitude
    assert err < 1.1e-2


def test_c2_full_size_bf16_step_tracks_fp32_parity_mode():
    """BASELINE config 2 at its full size (IR-SE-50, batch 256, bf16 -- the benchmarked step, including the 8-wave kernels,
    the fused epilogues and the side stream): size-independent properties against the fp32 parity mode of the same network
    on the same batch (that mode is pinned to the reference fixtures at small N): the loss agrees to bf16 accuracy, every
    parameter group's gradient points the same way (cosine), and a repeated step reproduces the loss."""
    import copy
    import xrface
    from xrface import parallel
    from xrface.loss.loss import CrossEntropyLoss
    from xrface.model import model_irse
    torch.manual_seed(21)
    net32 = model_irse.IR_SE_50([112, 112]).to(DEV).train()
    for m in net32.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    net16 = copy.deepcopy(net32)
    g = torch.Generator(device=DEV); g.manual_seed(4)
    x = (torch.rand(256, 3, 112, 112, device=DEV, generator=g) * 2 - 1)
    y = torch.randint(0, 512, (256,), device=DEV, generator=g)
    crit = CrossEntropyLoss()
    out = {}
    try:
        for name, net, dtype in (("f32", net32, torch.float32), ("bf16", net16, torch.bfloat16)):
            xrface.set_compute_dtype(dtype)
            flat = parallel.FlatParams(net.parameters())
            flat.zero_grad()
            loss = crit(net(x), y)
            loss.backward()
            torch.cuda.synchronize()
            out[name] = (float(loss), flat.grad.clone(), flat)
            if name == "bf16":
                flat.zero_grad()
                loss2 = crit(net(x), y)
                loss2.backward()
                torch.cuda.synchronize()
                assert abs(float(loss2) - float(loss)) < 2e-3 * abs(float(loss))
                rep = float(torch.nn.functional.cosine_similarity(flat.grad, out[name][1], dim=0))
                print(f"[full size] repeated bf16 step: gradient cosine {rep:.5f}")
                assert rep > 0.995      # differences: fp32 atomics order + bf16 re-rounding only
    finally:
        xrface.set_compute_dtype(torch.float32)
    l32, g32, flat = out["f32"]
    l16, g16, _ = out["bf16"]
    assert abs(l16 - l32) < 2e-2 * abs(l32), (l16, l32)
    cos_all = float(torch.nn.functional.cosine_similarity(g16, g32, dim=0))
    assert cos_all > 0.98, cos_all
    worst = 1.0
    for p_, o in zip(flat.params, flat.offsets):
        n = p_.numel()
        if n < 4096:
            continue
        c = float(torch.nn.functional.cosine_similarity(g16[o:o + n], g32[o:o + n], dim=0))
        worst = min(worst, c)
    print(f"[full size] loss f32 {l32:.5f} bf16 {l16:.5f}; gradient cosine overall {cos_all:.4f}, worst large tensor {worst:.4f}")
    assert worst > 0.9


def test_c5_full_size_properties():
    """BASELINE config 5 at its full size (P = 1e6 pairs, 4000 thresholds, 10 folds): size-independent properties --
    the histogram accounts for every pair exactly once, TPR/FPR are monotone in the threshold, the end points are
    0 and 1, and a 20 000-pair sample of the distances matches numpy."""
    from xrface.utils.utils import calculate_roc, pair_dist, roc_histograms
    P = 1_000_000
    g = torch.Generator(device=DEV); g.manual_seed(0)
    e1 = torch.randn(P, 512, device=DEV, generator=g)
    same = torch.rand(P, device=DEV, generator=g) < 0.5
    e2 = torch.where(same[:, None], e1 + 0.5 * torch.randn(P, 512, device=DEV, generator=g),
                     torch.randn(P, 512, device=DEV, generator=g))
    dist = pair_dist(e1, e2)
    idx = torch.arange(0, P, 50, device=DEV)
    ref = np.sum(np.square(e1[idx].cpu().numpy() - e2[idx].cpu().numpy()), 1)
    assert np.abs(dist[idx].cpu().numpy() - ref).max() / ref.max() < 2e-6
    thresholds = np.arange(0, 12000, 3)
    fold = np.random.RandomState(0).randint(0, 10, P).astype(np.int32)
    hist = roc_histograms(dist, same.cpu().numpy(), fold, thresholds, 10).cpu().numpy()
    assert hist.sum() == P and (hist >= 0).all()
    assert hist[:, 1].sum() == int(same.sum()) and np.array_equal(hist.sum(axis=(1, 2)), np.bincount(fold, minlength=10))
    tpr, fpr, acc, best = calculate_roc(thresholds, e1, e2, same.cpu().numpy(), nrof_folds=10, fold_id=fold)
    assert (np.diff(tpr) >= 0).all() and (np.diff(fpr) >= 0).all()
    assert tpr[0] == 0.0 and fpr[0] == 0.0 and tpr[-1] == 1.0 and fpr[-1] == 1.0
    assert 0.99 < acc <= 1.0 and len(best) == 10


def test_fsrnet_sr_variant_forward_and_coarse_grads():
    """SR-variant generators (SUPER_RESOLUTION/model/FSRnet.py): ReflectionPad + conv, k3/s2 deconv, Tanh, 4 x depth-4
    bottleneck hourglass.  Forward outputs vs the reference fixture, and gradients of mse97(coarse, hr)."""
    import xrface
    from xrface.loss.loss import MSELossFunc
    from xrface.model import FSRnet_sr as M
    xrface.set_compute_dtype(torch.float32)
    st = load_gold("fsrnet_sr.npz")
    hr = G.synth_faces(1, 112, seed=1, start=400)
    lr = G.synth_lr_from_hr(hr)
    coarse, _ = load_det(M.Coarse_SR_Network(), 3)
    enc, _ = load_det(M.Fine_SR_Encoder(), 3)
    prior, _ = load_det(M.Prior_Estimation_Network(), 3)
    dec, _ = load_det(M.Fine_SR_Decoder(), 3)
    c = coarse(lr.to(DEV))
    assert c.shape == (1, 3, 112, 112)
    check_against(st, "coarse/img", c, TOL)
    e = enc(c.detach())
    check_against(st, "encoder/out", e, TOL)
    pf, lm, pa = prior(c.detach())
    assert lm.shape == (1, 68, 112, 112) and pa.shape == (1, 13, 112, 112)
    check_against(st, "prior/feat", pf, TOL)
    check_against(st, "prior/landmark", lm, TOL)
    check_against(st, "prior/parsing", pa, TOL)
    d = dec(torch.cat((pf, e), 1))
    check_against(st, "decoder/img", d, TOL)
    loss = MSELossFunc()(c, hr.to(DEV))
    loss.backward()
    assert abs(loss.item() - float(st["coarse/loss"])) <= TOL * abs(float(st["coarse/loss"]))
    g = grads_by_name(coarse)
    pre = "coarse/grad/"
    for key in st.files:
        if key.startswith(pre):
            name = key[len(pre):].replace("@digest", "")
            check_against(st, pre + name, g[name], GRAD_TOL, floor=grad_floor(st, pre))


def test_fhn_perceptual_step_matches_oracle():
    """SUPER_RESOLUTION/train_FHN.py:251-308 restated (row a17): SR-variant generators + frozen IR-50 feature losses;
    product step vs the CPU oracle composed from reference-pinned pieces (N = 2)."""
    import xrface
    from xrface.model import FSRnet_sr as M
    from xrface.model import model_irse
    from xrface.steps import fhn_perceptual_step
    xrface.set_compute_dtype(torch.float32)
    nets, sds = {}, {}
    for k, ctor in (("coarse", M.Coarse_SR_Network), ("encoder", M.Fine_SR_Encoder), ("prior", M.Prior_Estimation_Network),
                    ("decoder", M.Fine_SR_Decoder)):
        nets[k], sds[k] = load_det(ctor(), 3)
    bb, bb_sd = load_det(model_irse.IR_50([112, 112]), 0)
    for p_ in bb.parameters():
        p_.requires_grad_(False)
    # N = 2: the reference's CrossEntropyLoss2d squeezes ALL unit dims of the target (loss/loss.py:62), so N = 1 breaks it
    hr = G.synth_faces(2, 112, seed=1, start=500)
    lr = G.synth_lr_from_hr(hr)
    hm = G.synth_heatmap(2, 112, 68, 2.0, seed=2, start=500)
    par = G.synth_parsing(2, 112, 13, seed=2, start=500)
    losses, outs = fhn_perceptual_step(nets, bb, lr.to(DEV), hr.to(DEV), hm.to(DEV), par.to(DEV))
    l_ref, o_ref, g_ref = R.fhn_perceptual_grads(sds, bb_sd, lr, hr, hm, par)
    assert rel_err(outs["coarse"], o_ref["coarse"]) < TOL and rel_err(outs["sr"], o_ref["sr"]) < TOL
    for k in ("coarse", "prior", "encdec"):
        assert abs(losses[k].item() - l_ref[k].item()) <= 2 * TOL * abs(l_ref[k].item()), (k, losses[k].item(), l_ref[k].item())
    # The gradients of this step are ill-conditioned in fp32 (IR-50 + generator chains; the prior is 4 x depth-4 bottleneck
    # hourglasses: 475 convs, InstanceNorm over as few as 7 x 7 samples, ReLU / max-pool kinks).  So the experiment is run three
    # ways on the same weights and inputs -- CPU oracle fp64 (the reference point), CPU oracle fp32 (what the reference computes),
    # HIP fp32 mode -- and the HIP path's distance from the fp64 gradients is bounded by a multiple of the fp32 oracle's own
    # distance from them (per tensor, normalised like check_against; and for the whole gradient direction).
    to64 = lambda sd: {k_: (v.double() if v.dtype.is_floating_point else v) for k_, v in sd.items()}
    _, _, g64 = R.fhn_perceptual_grads({k: to64(v) for k, v in sds.items()}, to64(bb_sd), lr.double(), hr.double(), hm.double(), par)
    K_T, K_COS = 4.0, 16.0     # per-tensor error factor; (1 - cosine) is quadratic in the error, hence the square
    for k in ("coarse", "prior", "encoder", "decoder"):
        got = dict(nets[k].named_parameters())
        names = [n for n, g_ in g64[k].items() if g_ is not None]
        for name, gr in g_ref[k].items():
            assert (gr is None) == (got[name].grad is None), (k, name)
        scale = max(float(g64[k][n].abs().max()) for n in names)

        def worst_of(grads):
            w, wn = 0.0, None
            for n in names:
                den = max(float(g64[k][n].abs().max()), 1e-2 * scale)
                e = float((grads[n].double().cpu() - g64[k][n]).abs().max()) / den
                if e > w:
                    w, wn = e, n
            return w, wn

        def cos_of(grads):
            a = torch.cat([grads[n].double().cpu().flatten() for n in names])
            b = torch.cat([g64[k][n].flatten() for n in names])
            return float((a @ b) / (a.norm() * b.norm()))

        hip = {n: got[n].grad for n in names}
        w_hip, wn_hip = worst_of(hip)
        w_o32, wn_o32 = worst_of(g_ref[k])
        c_hip, c_o32 = cos_of(hip), cos_of(g_ref[k])
        print(f"[perceptual/{k}] vs fp64 oracle: HIP fp32 worst {w_hip:.2e} ({wn_hip}), CPU fp32 worst {w_o32:.2e} ({wn_o32}); "
              f"1 - cosine: HIP {1 - c_hip:.2e}, CPU fp32 {1 - c_o32:.2e}")
        assert w_hip <= max(K_T * w_o32, 2e-3), (k, wn_hip, w_hip, w_o32)
        assert 1.0 - c_hip <= max(K_COS * (1.0 - c_o32), 1e-6), (k, c_hip, c_o32)


# ---------------------------------------------------------------------------------------------------------------- C4
def _c4_nets(dtype_seed=5):
    from xrface.model import FSRnet, model_irse
    fhn, fhn_sd = {}, {}
    for k, ctor in (("coarse", FSRnet.Course_SR_Network), ("encoder", FSRnet.Fine_SR_Encoder),
                    ("prior", FSRnet.Prior_Estimation_Network), ("decoder", FSRnet.Fine_SR_Decoder)):
        fhn[k], fhn_sd[k] = load_det(ctor(), 5)
    teacher, t_sd = load_det(model_irse.IR_SE_50([112, 112]), 0)
    student, s_sd = load_det(model_irse.IR_SE_50([112, 112]), 1)
    assistant, a_sd = load_det(model_irse.IR_SE_50([112, 112]), 2)
    for m in (student, assistant):
        m.output_layer[1].p = 0.0      # Dropout RNG pinned off, as in the fixture
    for p_ in teacher.parameters():
        p_.requires_grad_(False)
    return fhn, student, assistant, teacher, (fhn_sd, s_sd, a_sd, t_sd)


def _check_prefixed_grads(st, prefix, module, tol):
    g = grads_by_name(module)
    pre = prefix + "grad/"
    floor = grad_floor(st, pre)
    for key in st.files:
        if key.startswith(pre):
            name = key[len(pre):].replace("@digest", "")
            check_against(st, pre + name, g[name], tol, floor=floor)
    none_ref = set(st[prefix + "none_grad_keys"].tolist())
    got_none = {n for n, p in module.named_parameters() if p.grad is None}
    assert got_none == none_ref, (prefix, got_none ^ none_ref)


def test_c4_composed_step_matches_reference_fixture():
    """BASELINE configs[3] (north-star headline workload): FHN -> IR-SE-50 student + assistant vs the frozen IR-SE-50 teacher,
    residual-KD losses.  fp32 parity mode, N = 4 (the fixture's batch, generated from the reference's own modules composed as
    SUPER_RESOLUTION/train_FHN.py:274-279 + distill_main.py:59-70): SR image, embeddings, taps and both losses within 1e-3;
    student / assistant / four-generator gradients and the .grad-is-None sets."""
    import xrface
    from xrface.steps import c4_step
    xrface.set_compute_dtype(torch.float32)
    st = load_gold("c4.npz")
    fhn, student, assistant, teacher, _ = _c4_nets()
    hr = G.synth_faces(4, 112, seed=1, start=600)
    lr = G.synth_lr_from_hr(hr)
    (sl, al), outs = c4_step(fhn, student, assistant, teacher, lr.to(DEV), hr.to(DEV))
    check_against(st, "sr", outs["sr"], TOL)
    check_against(st, "t_emb", outs["t"][0], TOL)
    check_against(st, "s_emb", outs["s"][0], TOL)
    check_against(st, "a_emb", outs["a"][0], TOL)
    check_against(st, "s_tap3", outs["s"][4], TOL)
    check_against(st, "a_tap0", outs["a"][1], TOL)
    for got, key in ((sl, "student_loss"), (al, "assistant_loss")):
        ref = float(st[key])
        assert abs(got.item() - ref) <= TOL * abs(ref), (key, got.item(), ref)
    _check_prefixed_grads(st, "student/", student, spread_tol("c4", "student", KD_GRAD_TOL))
    _check_prefixed_grads(st, "assistant/", assistant, spread_tol("c4", "assistant", KD_GRAD_TOL))
    for k in ("coarse", "prior", "encoder", "decoder"):
        _check_prefixed_grads(st, k + "/", fhn[k], spread_tol("c4", k, KD_GRAD_TOL))
    new_sd = student.state_dict()
    for k in ("input_layer.1.running_mean", "body.23.res_layer.4.running_var"):
        check_against(st, f"student/stats/{k}", new_sd[k], TOL)
    assert all(p_.grad is None for p_ in teacher.parameters())


def test_c4_flat_direct_mode_equals_plain_autograd():
    """The benchmarked form of the C4 step (FlatParams with in-kernel gradient accumulation + side-stream weight gradients +
    fused optimizers) produces the same gradients as the plain-autograd form checked against the fixture above, keeps the
    parameters that receive no gradient (prior heads, bn_end, residual_next ...) bit-identical through an optimizer step
    with weight decay, and moves the others."""
    import xrface
    from xrface import parallel
    from xrface.steps import c4_step
    xrface.set_compute_dtype(torch.float32)
    hr = G.synth_faces(4, 112, seed=1, start=600)
    lr = G.synth_lr_from_hr(hr)
    fhn, student, assistant, teacher, _ = _c4_nets()
    c4_step(fhn, student, assistant, teacher, lr.to(DEV), hr.to(DEV))
    ref = {}
    for tag, m in (("student", student), ("assistant", assistant), *fhn.items()):
        ref[tag] = {n: (None if p_.grad is None else p_.grad.clone()) for n, p_ in m.named_parameters()}
    fhn2, student2, assistant2, teacher2, _ = _c4_nets()
    fhn_params = [p_ for k in ("coarse", "prior", "encoder", "decoder") for p_ in fhn2[k].parameters()]
    flats = [parallel.FlatParams(fhn_params), parallel.FlatParams(student2.parameters()), parallel.FlatParams(assistant2.parameters())]
    opts = [parallel.FusedRMSprop(flats[0], lr=1e-4, alpha=0.99, weight_decay=1e-5),
            parallel.FusedSGD(flats[1], lr=1e-3, momentum=0.9, weight_decay=1e-4),
            parallel.FusedSGD(flats[2], lr=1e-3, momentum=0.9, weight_decay=1e-4)]
    before = [f.flat.clone() for f in flats]
    # gradients first (no optimizer step): compare with the plain-autograd run
    c4_step(fhn2, student2, assistant2, teacher2, lr.to(DEV), hr.to(DEV), optimizers=None)
    torch.cuda.synchronize()
    worst = 0.0
    for tag, m in (("student", student2), ("assistant", assistant2), *fhn2.items()):
        scale = max(float(g.abs().max()) for g in ref[tag].values() if g is not None)
        for n, p_ in m.named_parameters():
            g_ref = ref[tag][n]
            if g_ref is None:
                assert float(p_.grad.abs().max()) == 0.0 and not p_.__dict__.get("_xr_touched", False), (tag, n)
                continue
            assert p_.__dict__.get("_xr_touched", False), (tag, n)
            err = float((p_.grad - g_ref).abs().max()) / max(float(g_ref.abs().max()), 1e-2 * scale)
            worst = max(worst, err)
            # bar: the path's own run-to-run spread (fp32 atomics order x small-batch train-mode norms; measured up to 3.1e-2 on
            # single tensors of the N = 4 train-mode IR-SE-50)
            assert err < 6e-2, (tag, n, err)
    print(f"[c4 direct] worst gradient deviation from the plain-autograd form: {worst:.2e}")
    # now a full step with the optimizers: untouched parameters must not move (weight decay included)
    c4_step(fhn2, student2, assistant2, teacher2, lr.to(DEV), hr.to(DEV), optimizers=opts)
    torch.cuda.synchronize()
    for f, b in zip(flats, before):
        for p_, o in zip(f.params, f.offsets):
            seg_new, seg_old = f.flat[o:o + p_.numel()], b[o:o + p_.numel()]
            if p_.__dict__.get("_xr_touched", False):
                assert not torch.equal(seg_new, seg_old)
            else:
                assert torch.equal(seg_new, seg_old), "a parameter without gradient moved (weight decay on an unused parameter)"


def test_c4_full_size_bf16_properties():
    """C4 at BASELINE's per-GPU batch (256, bf16, every fused path and the side stream on): size-independent properties.
    (1) the step is repeatable: same losses, gradient cosine > 0.995 between two runs; (2) the bf16 losses agree with the fp32
    parity mode on a 64-image slice of the same batch (the parity mode is pinned to the reference fixture at N = 4);
    (3) three optimizer steps on a fixed batch reduce the student loss."""
    import xrface
    from xrface import parallel
    from xrface.model import FSRnet, model_irse
    from xrface.steps import c4_step
    torch.manual_seed(31)
    mk = lambda: {"coarse": FSRnet.Course_SR_Network().to(DEV), "prior": FSRnet.Prior_Estimation_Network().to(DEV),
                  "encoder": FSRnet.Fine_SR_Encoder().to(DEV), "decoder": FSRnet.Fine_SR_Decoder().to(DEV)}
    fhn = mk()
    student, assistant = model_irse.IR_SE_50([112, 112]).to(DEV), model_irse.IR_SE_50([112, 112]).to(DEV)
    teacher = model_irse.IR_SE_50([112, 112]).to(DEV).eval()
    for m in (student, assistant):
        m.output_layer[1].p = 0.0
    for p_ in teacher.parameters():
        p_.requires_grad_(False)
    g = torch.Generator(device=DEV); g.manual_seed(8)
    lo = torch.randn(256, 3, 14, 14, device=DEV, generator=g)
    hr = torch.nn.functional.interpolate(lo, size=(112, 112), mode="bilinear").clamp_(-1, 1).contiguous()
    lr = torch.nn.functional.interpolate(torch.nn.functional.avg_pool2d(hr, 7), size=(112, 112), mode="bilinear").contiguous()
    try:
        xrface.set_compute_dtype(torch.float32)
        (sl32, al32), _ = c4_step(fhn, student, assistant, teacher, lr[:64], hr[:64])
        for m in (*fhn.values(), student, assistant):
            for p_ in m.parameters():
                p_.grad = None
        xrface.set_compute_dtype(torch.bfloat16)
        (sl16, al16), _ = c4_step(fhn, student, assistant, teacher, lr[:64], hr[:64])
        for m in (*fhn.values(), student, assistant):
            for p_ in m.parameters():
                p_.grad = None
        print(f"[c4 N=64] student loss f32 {sl32.item():.5f} bf16 {sl16.item():.5f}; assistant f32 {al32.item():.5f} bf16 {al16.item():.5f}")
        assert abs(sl16.item() - sl32.item()) < 3e-2 * abs(sl32.item())
        assert abs(al16.item() - al32.item()) < 3e-2 * abs(al32.item())
        fhn_params = [p_ for k in ("coarse", "prior", "encoder", "decoder") for p_ in fhn[k].parameters()]
        flats = [parallel.FlatParams(fhn_params), parallel.FlatParams(student.parameters()), parallel.FlatParams(assistant.parameters())]
        (l1, a1), _ = c4_step(fhn, student, assistant, teacher, lr, hr)
        torch.cuda.synchronize()
        g1 = [f.grad.clone() for f in flats]
        for f in flats:
            f.zero_grad()
        (l2, a2), _ = c4_step(fhn, student, assistant, teacher, lr, hr)
        torch.cuda.synchronize()
        assert l1.item() == l1.item() and a1.item() == a1.item()
        assert abs(l2.item() - l1.item()) < 5e-3 * abs(l1.item()) and abs(a2.item() - a1.item()) < 5e-3 * abs(a1.item())
        cos = [float(torch.nn.functional.cosine_similarity(f.grad, ga, dim=0)) for f, ga in zip(flats, g1)]
        print(f"[c4 N=256 bf16] gradient cosine between two runs of the same step: FHN {cos[0]:.4f}, student {cos[1]:.4f}, assistant {cos[2]:.4f}")
        # assistant (MSE against five residual targets, a strong signal): atomics order + bf16 re-rounding only.  The student's
        # gradient comes from ONE 512-d embedding MSE through a random-initialised 50-layer train-mode-BatchNorm network
        # (measured 0.98), and the FHN gradient is what is left of it after ~100 more InstanceNorm layers (measured 0.87): weak
        # signals on which the same re-rounding noise weighs more
        assert cos[2] > 0.99 and cos[1] > 0.95 and cos[0] > 0.7, cos
        opts = [parallel.FusedRMSprop(flats[0], lr=1e-5), parallel.FusedSGD(flats[1], lr=0.05, momentum=0.9),
                parallel.FusedSGD(flats[2], lr=0.05, momentum=0.9)]
        hist = []
        for _ in range(4):
            (l_, _a), _ = c4_step(fhn, student, assistant, teacher, lr, hr, optimizers=opts)
            hist.append(l_.item())
        print(f"[c4 N=256 bf16] student loss over 4 steps on a fixed batch: {hist}")
        assert hist[-1] < hist[0]
    finally:
        xrface.set_compute_dtype(torch.float32)


def test_c4_full_size_deterministic_mode_repeats_bit_for_bit():
    """xrface.set_deterministic(True) (XR_DETERMINISTIC=1): two runs of the N = 256 bf16 C4 step on the same batch -- lockstep
    chains, weight-gradient side stream, FlatParams in-kernel accumulation and the fused optimizers on -- give BIT-IDENTICAL losses,
    gradients and post-step weights (every fp32 sum in a fixed order); and the mode agrees with the default mode to the default
    mode's own run-to-run spread.  Needed before anyone debugs replica divergence across GPUs."""
    import xrface
    from xrface import ops, parallel
    from xrface.model import FSRnet, model_irse
    from xrface.steps import c4_step
    xrface.set_compute_dtype(torch.bfloat16)
    g = torch.Generator(device=DEV); g.manual_seed(8)
    lo = torch.randn(256, 3, 14, 14, device=DEV, generator=g)
    hr = torch.nn.functional.interpolate(lo, size=(112, 112), mode="bilinear").clamp_(-1, 1).contiguous()
    lr = torch.nn.functional.interpolate(torch.nn.functional.avg_pool2d(hr, 7), size=(112, 112), mode="bilinear").contiguous()

    def run(det):
        torch.manual_seed(31)
        ops._drop_counter[0] = 0          # Dropout(0.5) stays ON: the counter-based mask stream restarts with the run
        fhn = {"coarse": FSRnet.Course_SR_Network().to(DEV), "prior": FSRnet.Prior_Estimation_Network().to(DEV),
               "encoder": FSRnet.Fine_SR_Encoder().to(DEV), "decoder": FSRnet.Fine_SR_Decoder().to(DEV)}
        student, assistant = model_irse.IR_SE_50([112, 112]).to(DEV), model_irse.IR_SE_50([112, 112]).to(DEV)
        teacher = model_irse.IR_SE_50([112, 112]).to(DEV).eval()
        for p_ in teacher.parameters():
            p_.requires_grad_(False)
        fhn_params = [p_ for k in ("coarse", "prior", "encoder", "decoder") for p_ in fhn[k].parameters()]
        flats = [parallel.FlatParams(fhn_params), parallel.FlatParams(student.parameters_in_execution_order()),
                 parallel.FlatParams(assistant.parameters_in_execution_order())]
        opts = [parallel.FusedRMSprop(f_, lr=1e-4, alpha=0.99, weight_decay=1e-5) for f_ in flats]
        xrface.set_deterministic(det)
        try:
            (l1, a1), _ = c4_step(fhn, student, assistant, teacher, lr, hr, optimizers=opts)
            g1 = [f_.grad.clone() for f_ in flats]
            (l2, a2), _ = c4_step(fhn, student, assistant, teacher, lr, hr, optimizers=opts)
            torch.cuda.synchronize()
        finally:
            xrface.set_deterministic(False)
        return (l1.item(), a1.item(), l2.item(), a2.item()), g1, [f_.flat.clone() for f_ in flats]

    try:
        la, ga, wa = run(True)
        lb, gb, wb = run(True)
        assert la == lb, (la, lb)
        for name, x1, x2 in zip(("fhn", "student", "assistant"), ga, gb):
            assert torch.equal(x1, x2), f"deterministic mode: {name} gradients of step 1 differ between two runs"
        for name, x1, x2 in zip(("fhn", "student", "assistant"), wa, wb):
            assert torch.equal(x1, x2), f"deterministic mode: {name} weights after two steps differ between two runs"
        assert all(v == v for v in la)
        ld, gd, _ = run(False)
        cos = [float(torch.nn.functional.cosine_similarity(x1, x2, dim=0)) for x1, x2 in zip(ga, gd)]
        print(f"[c4 N=256 bf16] deterministic vs default mode, step-1 gradient cosine: FHN {cos[0]:.4f}, student {cos[1]:.4f}, "
              f"assistant {cos[2]:.4f}; losses {la[:2]} vs {ld[:2]}")
        assert abs(la[0] - ld[0]) < 5e-3 * abs(ld[0]) and abs(la[1] - ld[1]) < 5e-3 * abs(ld[1])
        assert cos[2] > 0.99 and cos[1] > 0.95 and cos[0] > 0.7, cos
    finally:
        xrface.set_compute_dtype(torch.float32)


def test_c3_full_size_bf16_properties():
    """C3 at BASELINE's per-GPU batch (128, bf16: the chained residual trunks, the direct kernels with every epilogue, the
    row-walking weight gradients, the side stream): size-independent properties.  (1) on a 32-image slice the bf16 losses agree
    with the fp32 parity mode (pinned to the reference fixture at N = 2) and the four networks' gradients point the same way;
    (2) at N = 128 the step is repeatable; (3) four optimizer steps on a fixed batch reduce the summed loss."""
    import xrface
    from xrface import parallel
    from xrface.model import FSRnet
    from xrface.steps import fhn_step_fused
    torch.manual_seed(41)
    fhn = {"coarse": FSRnet.Course_SR_Network().to(DEV), "prior": FSRnet.Prior_Estimation_Network().to(DEV),
           "encoder": FSRnet.Fine_SR_Encoder().to(DEV), "decoder": FSRnet.Fine_SR_Decoder().to(DEV)}
    g = torch.Generator(device=DEV); g.manual_seed(9)
    n = 128
    lo = torch.randn(n, 3, 14, 14, device=DEV, generator=g)
    hr = torch.nn.functional.interpolate(lo, size=(112, 112), mode="bilinear").clamp_(-1, 1).contiguous()
    lr = torch.nn.functional.interpolate(torch.nn.functional.avg_pool2d(hr, 7), size=(112, 112), mode="bilinear").contiguous()
    hm = torch.rand(n, 28, 28, device=DEV, generator=g)
    par = torch.randint(0, 11, (n, 1, 28, 28), device=DEV, generator=g)
    order = ("coarse", "encoder", "prior", "decoder")
    flat_of = lambda: {k: torch.cat([p_.grad.detach().float().flatten() for p_ in fhn[k].parameters() if p_.grad is not None]) for k in order}
    clear = lambda: [setattr(p_, "grad", None) for k in order for p_ in fhn[k].parameters()]
    try:
        xrface.set_compute_dtype(torch.float32)
        l32, _ = fhn_step_fused(fhn, lr[:32], hr[:32], hm[:32], par[:32])
        g32 = flat_of()
        clear()
        xrface.set_compute_dtype(torch.bfloat16)
        l16, _ = fhn_step_fused(fhn, lr[:32], hr[:32], hm[:32], par[:32])
        g16 = flat_of()
        clear()
        for k in order:
            c = float(torch.nn.functional.cosine_similarity(g16[k], g32[k], dim=0))
            print(f"[c3 N=32] {k}: loss f32 {l32[k].item():.4f} bf16 {l16[k].item():.4f}; gradient cosine bf16 vs f32 {c:.4f}")
            assert abs(l16[k].item() - l32[k].item()) < 3e-2 * abs(l32[k].item()), k
            # the encoder / prior receive their gradient through the whole decoder (27 InstanceNorm layers downstream): bf16
            # re-rounding weighs more on them (measured 0.965 / see the printed values) than on the coarse net and the decoder
            assert c > (0.9 if k in ("encoder", "prior") else 0.97), (k, c)
        flats = {k: parallel.FlatParams(fhn[k].parameters()) for k in order}
        la, _ = fhn_step_fused(fhn, lr, hr, hm, par)
        torch.cuda.synchronize()
        ga = {k: flats[k].grad.clone() for k in order}
        for f in flats.values():
            f.zero_grad()
        lb, _ = fhn_step_fused(fhn, lr, hr, hm, par)
        torch.cuda.synchronize()
        for k in order:
            c = float(torch.nn.functional.cosine_similarity(flats[k].grad, ga[k], dim=0))
            assert abs(lb[k].item() - la[k].item()) < 5e-3 * abs(la[k].item()), k
            assert c > 0.99, (k, c)
        opts = {k: parallel.FusedRMSprop(flats[k], lr=2e-5, alpha=0.99, weight_decay=1e-5) for k in order}
        hist = []
        for _ in range(4):
            ls, _ = fhn_step_fused(fhn, lr, hr, hm, par, optimizers=opts)
            hist.append({k: v.item() for k, v in ls.items()})
        print(f"[c3 N=128 bf16] coarse / decoder loss over 4 steps on a fixed batch: {[round(h['coarse'], 2) for h in hist]} / "
              f"{[round(h['decoder'], 2) for h in hist]}")
        # the two pixel losses are the well-conditioned ones (the prior's landmark loss against a random heat-map is ~6e5 at
        # initialisation and oscillates under RMSprop at any step size that moves the others)
        assert all(v == v for h in hist for v in h.values())
        assert hist[-1]["coarse"] < hist[0]["coarse"] and hist[-1]["decoder"] < hist[0]["decoder"]
    finally:
        xrface.set_compute_dtype(torch.float32)


def test_fhn_step_fused_equals_literal_and_direct_mode():
    """fhn_step_fused (one backward pass) == fhn_step (one partial traversal per (loss_k, theta_k) pair), and both stay
    correct when the generators live in FlatParams(direct=True) buffers -- the literal form must then not let the traversal
    of a foreign sub-network accumulate into that sub-network's flat gradient (ops.grad_only)."""
    import xrface
    from xrface import parallel
    from xrface.model import FSRnet
    from xrface.steps import fhn_step, fhn_step_fused
    xrface.set_compute_dtype(torch.float32)
    st = load_gold("fsrnet_root.npz")
    hr = G.synth_faces(2, 112, seed=1)
    lr = G.synth_lr_from_hr(hr)
    hm = G.synth_heatmap(2, 28, 97, 1.3, seed=2)
    par = G.synth_parsing(2, 28, 11, seed=2)
    args = (lr.to(DEV), hr.to(DEV), hm.to(DEV), par.to(DEV))
    ctors = (("coarse", FSRnet.Course_SR_Network), ("encoder", FSRnet.Fine_SR_Encoder),
             ("prior", FSRnet.Prior_Estimation_Network), ("decoder", FSRnet.Fine_SR_Decoder))

    def check(nets, flat_mode, what):
        for k in ("coarse", "encoder", "prior", "decoder"):
            none_ref = set(st[f"fhn/{k}/none_grad_keys"].tolist())
            named = dict(nets[k].named_parameters())
            if flat_mode:
                for n in none_ref:
                    assert float(named[n].grad.abs().max()) == 0.0, (what, k, n)
            else:
                assert {n for n, p_ in named.items() if p_.grad is None} == none_ref, (what, k)
            pre = f"fhn/{k}/grad/"
            for key in st.files:
                if key.startswith(pre):
                    name = key[len(pre):].replace("@digest", "")
                    check_against(st, pre + name, named[name].grad, GRAD_TOL, what=f"{what} {pre}{name}", floor=grad_floor(st, pre))

    nets = {k: load_det(c())[0] for k, c in ctors}
    losses, outs = fhn_step_fused(nets, *args)
    check_against(st, "fhn/sr", outs["sr"], TOL)
    for k in ("coarse", "encoder", "prior", "decoder"):
        ref = float(st[f"fhn/loss/{k}"])
        assert abs(losses[k].item() - ref) <= TOL * abs(ref), (k, losses[k].item(), ref)
    check(nets, False, "fused")
    for step_fn, what in ((fhn_step, "literal+flat"), (fhn_step_fused, "fused+flat")):
        nets = {k: load_det(c())[0] for k, c in ctors}
        flats = {k: parallel.FlatParams(nets[k].parameters()) for k in nets}
        step_fn(nets, *args)
        torch.cuda.synchronize()
        check(nets, True, what)
        for k, f in flats.items():   # .grad must still be the flat views
            for p_, o in zip(f.params, f.offsets):
                assert p_.grad is not None and p_.grad.data_ptr() == f.grad.data_ptr() + 4 * o


def test_overall_network_gan_224_matches_reference_fixture():
    """Row a7 / f4: OverallNetwork_GAN + Discriminator (model/FSRnet.py:461-545) run only at 224x224 (Linear(64*56*56, 512),
    102.8 M parameters).  Train-mode forward (N = 3: BatchNorm batch statistics, bn_mid applied twice per call and the
    discriminator called twice per forward -> four running-statistics updates), gradient of MSE(emb1, emb2) with respect to
    the discriminator, eval-mode forward (N = 2)."""
    import xrface
    from xrface.loss.loss import MSELoss
    from xrface.model.FSRnet import OverallNetwork_GAN
    xrface.set_compute_dtype(torch.float32)
    st = load_gold("gan224.npz")
    net, _ = load_det(OverallNetwork_GAN(), 7)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    hr = G.synth_faces(3, 224, seed=1, start=700)
    lr = G.synth_lr_from_hr(hr)
    net.train()
    sr, coarse, lmk, par, e1, e2 = net(lr.to(DEV), hr.to(DEV))
    assert sr.shape == (3, 3, 224, 224) and lmk.shape == (3, 97, 56, 56) and par.shape == (3, 11, 56, 56) and e1.shape == (3, 512)
    for nm, t in (("sr", sr), ("coarse", coarse), ("landmark", lmk), ("parsing", par), ("emb1", e1), ("emb2", e2)):
        check_against(st, "train/" + nm, t, TOL)
    l_d = MSELoss()(e1, e2)
    assert abs(l_d.item() - float(st["disc/loss"])) <= TOL * abs(float(st["disc/loss"]))
    disc = net._discriminator
    params = [p_ for p_ in disc.parameters()]
    grads = torch.autograd.grad(l_d, params, allow_unused=True)
    for (n, _), g_ in zip(disc.named_parameters(), grads):
        disc.get_parameter(n).grad = g_
    _check_prefixed_grads(st, "disc/", disc, KD_GRAD_TOL)
    new_sd = net.state_dict()
    for k in ("_discriminator.bn_mid.running_mean", "_discriminator.bn_mid.running_var", "_discriminator.bn_end.running_var"):
        check_against(st, "train/stats/" + k, new_sd[k], TOL)
    assert int(new_sd["_discriminator.bn_mid.num_batches_tracked"]) == int(st["train/nbt"]) == 4
    net.load_state_dict(sd0)
    net.eval()
    with torch.no_grad():
        outs = net(lr[:2].to(DEV), hr[:2].to(DEV))
    for nm, t in zip(("sr", "coarse", "landmark", "parsing", "emb1", "emb2"), outs):
        check_against(st, "eval/" + nm, t, TOL)


def test_gan_step_adversarial_loss_map_matches_reference_fixture():
    """Row f4 / a16 with the discriminator terms kept: steps.gan_step (Face_Hallucination_sub_Net.py:218-247) on
    OverallNetwork_GAN at 224 x 224, N = 3, fp32 parity mode.  The reference's ``MMD`` import is undefined upstream, so the
    fixture (tests/golden/gan_step.npz, generated from the reference's own OverallNetwork_GAN + loss classes) and this test
    both inject nn.MSELoss-like ``criterion_mmd``: the five (loss_k, theta_k) pairs -- incl. the -L_disc terms that reach the
    encoder and the prior through BOTH discriminator calls -- are pinned: six outputs, five losses, every kept gradient and
    the .grad-is-None set of each sub-network."""
    import xrface
    from xrface.loss.loss import MSELoss
    from xrface.model.FSRnet import OverallNetwork_GAN
    from xrface.steps import gan_step
    xrface.set_compute_dtype(torch.float32)
    st = load_gold("gan_step.npz")
    net, _ = load_det(OverallNetwork_GAN(), 9)
    n = 3
    hr = G.synth_faces(n, 224, seed=1, start=900)
    lr = G.synth_lr_from_hr(hr)
    hm = G.synth_heatmap(n, 56, 97, 1.3, seed=5)
    par = G.synth_parsing(n, 56, 11, seed=5)
    net.train()
    losses, outs = gan_step(net, lr.to(DEV), hr.to(DEV), hm.to(DEV), par.to(DEV), criterion_mmd=MSELoss())
    for nm in ("sr", "coarse", "landmark", "parsing", "emb1", "emb2"):
        check_against(st, "out/" + nm, outs[nm], TOL)
    subs = {"disc": net._discriminator, "coarse": net._coarse_sr_network, "encoder": net._fine_sr_encoder,
            "prior": net._prior_estimation_network, "decoder": net._fine_sr_decoder}
    for k, m in subs.items():
        ref = float(st[f"loss/{k}"])
        assert abs(losses[k].item() - ref) <= TOL * abs(ref), (k, losses[k].item(), ref)
        _check_prefixed_grads(st, k + "/", m, GRAD_TOL)
    # with stock optimizers: the step moves every sub-network that received a gradient and leaves the .grad-None parameters alone
    opts = {k: torch.optim.RMSprop(m.parameters(), lr=5e-3, alpha=0.99, weight_decay=1e-5) for k, m in subs.items()}
    before = {k: {n_: p_.detach().clone() for n_, p_ in m.named_parameters()} for k, m in subs.items()}
    gan_step(net, lr.to(DEV), hr.to(DEV), hm.to(DEV), par.to(DEV), optimizers=opts, criterion_mmd=MSELoss())
    for k, m in subs.items():
        none_ref = set(st[k + "/none_grad_keys"].tolist())
        for n_, p_ in m.named_parameters():
            same = torch.equal(p_.detach(), before[k][n_])
            assert same == (n_ in none_ref), (k, n_, same)


def test_verify_step_matches_host_evaluation_of_the_same_embeddings():
    """steps.verify_step (distill_main.py:111-138: teacher alone, student + assistant summed, calculate_roc on each) against the
    oracle's calculate_roc fed with the very embeddings the HIP path produced: same tpr / fpr / accuracy / thresholds (the networks
    themselves are pinned by the KD fixture; this pins the wiring of the evaluation caller)."""
    import xrface
    from xrface.model import model_irse, resnet
    from xrface.steps import verify_step
    xrface.set_compute_dtype(torch.float32)
    teacher, _ = load_det(model_irse.IR_50([112, 112]), 0)
    student, _ = load_det(resnet.ResNet_34(), 1)
    assistant, _ = load_det(resnet.ResNet_34(), 2)
    P = 24
    img1 = G.synth_faces(P, 112, seed=1, start=900).to(DEV)
    img2 = torch.where((torch.arange(P) % 2 == 0)[:, None, None, None], G.synth_faces(P, 112, seed=1, start=900) * 0.97 + 0.01,
                       G.synth_faces(P, 112, seed=1, start=1300)).to(DEV)
    label = (torch.arange(P) % 2 == 0).numpy()
    fold = (np.arange(P) * 7 % 10).astype(np.int32)
    thresholds = np.arange(0, 40000, 10)
    (t_tpr, t_fpr, t_acc, t_best), (s_tpr, s_fpr, s_acc, s_best) = verify_step(teacher, student, assistant, img1, img2, label,
                                                                              thresholds=thresholds, nrof_folds=10, fold_id=fold)
    folds = [(np.where(fold != f)[0], np.where(fold == f)[0]) for f in range(10)]
    with torch.no_grad():
        for m in (teacher, student, assistant):
            m.eval()
        t1, t2 = teacher(img1), teacher(img2)
        s1, s2 = student(img1)[0] + assistant(img1)[0], student(img2)[0] + assistant(img2)[0]
    for got, (e1, e2) in (((t_tpr, t_fpr, t_acc, t_best), (t1, t2)), ((s_tpr, s_fpr, s_acc, s_best), (s1, s2))):
        r_tpr, r_fpr, r_acc, r_best = R.calculate_roc(thresholds, e1.float().cpu().numpy(), e2.float().cpu().numpy(), label, folds)
        # distances: HIP kernel vs numpy differ in the last bits -> a pair may cross one threshold of the grid
        assert np.abs(got[0] - r_tpr).max() <= 1.0 / (P // 2) + 1e-9 and np.abs(got[1] - r_fpr).max() <= 1.0 / (P // 2) + 1e-9
        assert abs(got[2] - r_acc) <= 0.5 / P * 10 + 1e-9 and np.abs(got[3] - r_best).max() <= 10 * 2
    assert 0.0 <= t_acc <= 1.0 and 0.0 <= s_acc <= 1.0
