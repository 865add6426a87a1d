"""The import-path shim (xrface/shim.py): the reference-side statements of the entry scripts
resolve to the HIP-path mirrors and keep working -- star import, constructors, ``.apply(weights_init)``, per-sub-network
stock optimizers, ``load_state_dict`` of a reference-shaped dict (CPU part); one stock-optimizer step through the shimmed
names on the device (``-m gpu`` part)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cross-resolution-face-recognition_amd")

# what a maintainer's script does after `import xrface.shim; xrface.shim.install()` -- the statements are the reference's
# (Face_Hallucination_sub_Net.py:19,25,99-124,133,368-380; SUPER_RESOLUTION/train_FHN.py:22-27,103-121; distill_main.py:14-20,
# 201-203,222-225; DISTILLATION/train_HRN.py:14) restated, run in a fresh interpreter so no other test's imports interfere
SCRIPT = textwrap.dedent('''
    import itertools, json, sys
    sys.path.insert(0, %(pkg)r); sys.path.insert(0, %(root)r)
    import torch, torch.nn as nn
    import xrface.shim
    xrface.shim.install(torch.float32)

    from model.FSRnet import *
    from loss.loss import MSELossFunc, MSELoss_Landmark, CrossEntropyLoss2d, MMD
    import model.model_irse as model_irse
    import model.resnet as ResNet
    from utils.utils import calculate_roc, AverageMeter, accuracy
    from SUPER_RESOLUTION.model.utils import init_log, AverageMeter as AM2
    from SUPER_RESOLUTION.loss.loss import Landmark_Loss, CrossEntropyLoss2d as CE2
    from SUPER_RESOLUTION.model.model_irse import IR_50
    from SUPER_RESOLUTION.model.FSRnet import Coarse_SR_Network, Fine_SR_Decoder, Fine_SR_Encoder, Prior_Estimation_Network
    from SUPER_RESOLUTION.model.GroupDepthConv import FeatureExtractor
    from DISTILLATION.model.model_irse import IR_50 as IR_50_d, IR_SE_50
    from DISTILLATION.model.utils import FeatureExtractor as FE2

    def weights_init(m):
        for each in m.modules():
            if isinstance(each, nn.Conv2d):
                nn.init.xavier_uniform_(each.weight.data)
                if each.bias is not None:
                    each.bias.data.zero_()
            elif isinstance(each, nn.BatchNorm2d):
                each.weight.data.fill_(1); each.bias.data.zero_()
            elif isinstance(each, nn.Linear):
                nn.init.xavier_uniform_(each.weight.data); each.bias.data.zero_()

    out = {}
''')

CPU_BODY = textwrap.dedent('''
    model = OverallNetwork_GAN()
    model.apply(weights_init)
    opts = [torch.optim.RMSprop(sub.parameters(), lr=5e-3, alpha=0.99, weight_decay=1e-5)
            for sub in (model._coarse_sr_network, model._prior_estimation_network, model._fine_sr_encoder,
                        model._fine_sr_decoder, model._discriminator)]
    out["n_opt_params"] = [sum(p.numel() for g in o.param_groups for p in g["params"]) for o in opts]
    ref_keys = json.load(open(%(keys)r))
    sd = {k: torch.zeros(shape, dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32)
          for k, shape in ref_keys["fsrnet_root.gan"].items()}
    model.load_state_dict(sd)                              # strict: keys AND shapes of the reference's own state_dict
    out["gan_keys"] = len(sd)
    coarse, prior, enc, dec = Coarse_SR_Network(), Prior_Estimation_Network(), Fine_SR_Encoder(), Fine_SR_Decoder()
    for m, tag in ((coarse, "coarse"), (prior, "prior"), (enc, "encoder"), (dec, "decoder")):
        m.apply(weights_init)
        m.load_state_dict({k: torch.zeros(s) for k, s in ref_keys["fsrnet_sr." + tag].items()})
    backbone = IR_50([112, 112])
    layer_list = list(backbone.body._modules.keys())[-3:-1]
    out["layer_list"] = layer_list
    o1 = torch.optim.Adam(coarse.parameters(), lr=1e-4, weight_decay=1e-5, betas=(0.5, 0.999))
    o2 = torch.optim.Adam(itertools.chain(enc.parameters(), dec.parameters()), lr=1e-4, weight_decay=1e-5, betas=(0.5, 0.999))
    sch = torch.optim.lr_scheduler.MultiStepLR(o1, [10, 20], gamma=0.1)
    teacher, student, assistant = model_irse.IR_50([112, 112]), ResNet.ResNet_34(), ResNet.ResNet_34()
    so = torch.optim.RMSprop(student.parameters(), lr=1e-4, weight_decay=1e-5)
    out["classes"] = [type(x).__module__ for x in (model, coarse, backbone, student, MSELossFunc(), Landmark_Loss(), FeatureExtractor())]
    out["mods"] = sorted(k for k in sys.modules if k.split(".")[0] in ("model", "loss", "utils", "SUPER_RESOLUTION", "DISTILLATION"))
    print("RESULT" + json.dumps(out))
''')

GPU_BODY = textwrap.dedent('''
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = Course_SR_Network().to(dev)
    net.apply(weights_init)
    opt = torch.optim.RMSprop(net.parameters(), lr=1e-3, alpha=0.99, weight_decay=1e-5)
    hr = torch.rand(2, 3, 112, 112, device=dev) * 2 - 1
    crit = MSELossFunc()
    losses = []
    w0 = net.conv_input.weight.detach().clone()
    unused0 = net.bn_end.weight.detach().clone()
    for _ in range(4):
        opt.zero_grad()
        feat, img = net(hr)
        loss = 12.0 * crit(img, hr)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    print("RESULT" + json.dumps({"losses": losses, "moved": float((net.conv_input.weight - w0).abs().max()),
                                 "unused_moved": float((net.bn_end.weight - unused0).abs().max()),
                                 "unused_grad_none": net.bn_end.weight.grad is None}))
''')


def _run_script(body):
    fmt = dict(pkg=PKG, root=ROOT, keys=os.path.join(ROOT, "tests", "golden", "state_dict_keys.json"))
    code = SCRIPT % fmt + (body % fmt if "%(" in body else body)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1]
    return json.loads(line[len("RESULT"):])


def test_reference_side_statements_run_through_the_shim():
    out = _run_script(CPU_BODY)
    # parameter counts of the five per-sub-network optimizers (its discriminator figure 103 096 322 also
    # counts the 1 154 BatchNorm buffer elements -- the strict load_state_dict in the script pins keys and shapes)
    assert out["n_opt_params"] == [226057, 5937900, 270665, 535689, 103095168]
    assert out["layer_list"] == ["21", "22"]
    assert all(m.startswith("xrface.") for m in out["classes"]), out["classes"]
    for name in ("model.FSRnet", "loss.loss", "utils.utils", "SUPER_RESOLUTION.model.FSRnet", "SUPER_RESOLUTION.loss.loss",
                 "DISTILLATION.model.model_irse"):
        assert name in out["mods"]


def test_shim_install_uninstall_restores_sys_modules():
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    import xrface.shim as shim
    roots = ("model", "loss", "utils", "SUPER_RESOLUTION", "DISTILLATION")
    before = {k for k in sys.modules if k.split(".")[0] in roots}
    shim.install()
    import SUPER_RESOLUTION.model.FSRnet as m
    assert m.__name__ == "xrface.model.FSRnet_sr"
    shim.uninstall()
    after = {k for k in sys.modules if k.split(".")[0] in roots}
    assert before == after


@pytest.mark.gpu
def test_stock_optimizer_step_through_the_shim():
    out = _run_script(GPU_BODY)
    assert out["moved"] > 0 and out["unused_moved"] == 0.0 and out["unused_grad_none"]
    assert out["losses"][-1] < out["losses"][0]
