"""Import-path shim: mounts the HIP-path mirrors under the module names the reference's entry scripts import, so their own
statements (``from model.FSRnet import *``, ``from SUPER_RESOLUTION.model.FSRnet import Coarse_SR_Network, ...``,
``from utils.utils import calculate_roc``) resolve to this package.

    import xrface.shim; xrface.shim.install(torch.bfloat16)      # before the script's own imports

Reference import sites covered:
  Face_Hallucination_sub_Net.py:19,25     model.FSRnet (star import), loss.loss (MSELossFunc, MSELoss_Landmark,
                                          CrossEntropyLoss2d, MMD)
  distill_main.py:14-15,20                model.model_irse, model.resnet, utils.utils (calculate_roc, AverageMeter, accuracy)
  SUPER_RESOLUTION/train_FHN.py:22-27     SUPER_RESOLUTION.model.{utils, model_irse, FSRnet, GroupDepthConv},
                                          SUPER_RESOLUTION.loss.loss (Landmark_Loss, CrossEntropyLoss2d)
  DISTILLATION/train_HRN.py:14-18         DISTILLATION.model.{model_irse, utils}
Data loaders, configs, MTCNN alignment and logging stay the reference's own modules (out of scope: SURVEY.md section 2).
"""
from __future__ import annotations

import importlib
import sys
import types

ALIASES = {
    "model.FSRnet": "xrface.model.FSRnet",
    "model.model_irse": "xrface.model.model_irse",
    "model.resnet": "xrface.model.resnet",
    "loss.loss": "xrface.loss.loss",
    "utils.utils": "xrface.utils.utils",
    "SUPER_RESOLUTION.model.FSRnet": "xrface.model.FSRnet_sr",
    "SUPER_RESOLUTION.model.model_irse": "xrface.model.model_irse",
    "SUPER_RESOLUTION.model.GroupDepthConv": "xrface.model.GroupDepthConv",
    "SUPER_RESOLUTION.model.utils": "xrface.model.utils",
    "SUPER_RESOLUTION.loss.loss": "xrface.loss.loss_sr",
    "DISTILLATION.model.model_irse": "xrface.model.model_irse",
    "DISTILLATION.model.utils": "xrface.model.utils",
}

_installed = {}


def install(compute_dtype=None, override_packages=False):
    """Register the aliases in ``sys.modules``.  Parent packages that are not importable (or all of them with
    ``override_packages``) are created as empty namespace stand-ins, so ``import model.FSRnet`` works even when the
    script's directory has no ``model/__init__.py`` of its own.  A parent package the script directory DOES provide is
    kept (its other sub-modules -- loaders, configs -- must stay reachable); only the aliased sub-modules are replaced."""
    import xrface
    if compute_dtype is not None:
        xrface.set_compute_dtype(compute_dtype)
    for alias, target in ALIASES.items():
        mod = importlib.import_module(target)
        parts = alias.split(".")
        for i in range(1, len(parts)):
            pname = ".".join(parts[:i])
            parent = sys.modules.get(pname)
            if parent is None and not override_packages:
                try:
                    parent = importlib.import_module(pname)
                except Exception:   # not importable here (missing / its __init__ needs absent third parties)
                    parent = None
            if parent is None or (override_packages and pname not in _installed and not getattr(parent, "_xr_shim", False)):
                parent = types.ModuleType(pname)
                parent.__path__ = []        # a package: sub-module imports consult sys.modules first
                parent._xr_shim = True
                _installed.setdefault(pname, sys.modules.get(pname))
                sys.modules[pname] = parent
            if i > 1:
                setattr(sys.modules[".".join(parts[:i - 1])], parts[i - 1], parent)
        _installed.setdefault(alias, sys.modules.get(alias))
        sys.modules[alias] = mod
        setattr(sys.modules[".".join(parts[:-1])], parts[-1], mod)
    return dict(ALIASES)


def uninstall():
    """Restore ``sys.modules`` to what it held before ``install``."""
    for name, old in list(_installed.items()):
        if old is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = old
    _installed.clear()
