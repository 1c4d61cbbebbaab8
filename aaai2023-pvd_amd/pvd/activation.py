"""trunc_exp (reference: tools/activation.py:7-21): exp forward in float32, backward with the
exponent clamped to [-12, 12]."""
import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd


def make_trunc_exp(device_type="cuda"):
    class _TruncExp(Function):
        @staticmethod
        @custom_fwd(device_type=device_type, cast_inputs=torch.float32)
        def forward(ctx, x):
            ctx.save_for_backward(x)
            return torch.exp(x)

        @staticmethod
        @custom_bwd(device_type=device_type)
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            return g * torch.exp(x.clamp(-12, 12))

    return _TruncExp.apply
