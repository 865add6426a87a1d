"""Mirror of /root/reference SUPER_RESOLUTION/model/utils.py and DISTILLATION/model/utils.py: AverageMeter (:16-31 / :18-33),
FeatureExtractor (DISTILLATION/model/utils.py:36-52, a duplicate of GroupDepthConv.py:29-45) and the logging set-up helper."""
import logging

from ..utils.utils import AverageMeter  # noqa: F401
from .GroupDepthConv import FeatureExtractor  # noqa: F401


def init_log(output_dir, log="log.log"):
    """Counterpart of SUPER_RESOLUTION/model/utils.py:5-14 (host-side logging glue, kept only so that
    ``from SUPER_RESOLUTION.model.utils import init_log, AverageMeter`` resolves): messages of level CRITICAL are appended to
    ``<output_dir>/<log>`` and echoed on the console; returns the ``logging`` module like the reference does."""
    import os
    root = logging.getLogger()
    root.setLevel(logging.CRITICAL)
    fh = logging.FileHandler(os.path.join(output_dir, log), mode="a")
    fh.setFormatter(logging.Formatter("%(asctime)s %(message)s", "%Y%m%d-%H:%M:%S"))
    root.addHandler(fh)
    root.addHandler(logging.StreamHandler())
    return logging
