"""FeatureExtractor -- mirror of /root/reference SUPER_RESOLUTION/model/GroupDepthConv.py:29-45
(duplicate: DISTILLATION/model/utils.py:36-52)."""
import time

import torch.nn as nn


class FeatureExtractor(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, x, extracted_layers, submodule):
        """Run ``submodule``'s children in order; keep outputs whose string key is in ``extracted_layers``.
        Returns (outputs, times, last_x, last_time) like the reference."""
        outputs, times = {}, {}
        start = time.time()
        temp = 0.0
        for name, module in submodule._modules.items():
            x = module(x)
            temp = time.time() - start
            if name in extracted_layers:
                outputs[name] = x
                times[name] = temp
        return outputs, times, x, temp
