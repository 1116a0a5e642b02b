"""CustomOp: per-op platform dispatch, same contract as the reference's
python/sglang/srt/custom_op.py:12-92 (dispatch_forward picks forward_<platform> once at
construction; forward() calls it).  This package has one platform: forward_hip."""
from torch import nn


class CustomOp(nn.Module):
    def __init__(self):
        super().__init__()
        self._forward_method = self.dispatch_forward()

    def forward(self, *args, **kwargs):
        return self._forward_method(*args, **kwargs)

    def forward_native(self, *args, **kwargs):
        raise NotImplementedError

    def forward_hip(self, *args, **kwargs):
        raise NotImplementedError

    def forward_cuda(self, *args, **kwargs):
        # the reference routes forward_hip -> forward_cuda by default (custom_op.py:70-71);
        # here the HIP path is the primary one and "cuda" is an alias of it.
        return self.forward_hip(*args, **kwargs)

    def dispatch_forward(self):
        return self.forward_hip
