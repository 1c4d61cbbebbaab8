"""``shencoder`` operator surface (reference: shencoder/sphere_harmonics.py), re-hosted on libpvd_hip.so."""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd


def make_sh_encode(backend, device_type="cuda"):
    class _SHEncode(Function):
        # reference: _sh_encoder, sphere_harmonics.py:15-62 (float32 forced)
        @staticmethod
        @custom_fwd(device_type=device_type, cast_inputs=torch.float32)
        def forward(ctx, inputs, degree, calc_grad_inputs=False):
            inputs = inputs.contiguous()
            B, input_dim = inputs.shape
            output_dim = degree ** 2
            outputs = torch.empty(B, output_dim, dtype=inputs.dtype, device=inputs.device)
            if calc_grad_inputs:
                dy_dx = torch.empty(B, input_dim * output_dim, dtype=inputs.dtype, device=inputs.device)
            else:
                dy_dx = torch.empty(1, dtype=inputs.dtype, device=inputs.device)
            backend.sh_encode_forward(inputs, outputs, B, input_dim, degree, calc_grad_inputs, dy_dx)
            ctx.save_for_backward(inputs, dy_dx)
            ctx.dims = [B, input_dim, degree]
            ctx.calc_grad_inputs = calc_grad_inputs
            return outputs

        @staticmethod
        @custom_bwd(device_type=device_type)
        def backward(ctx, grad):
            if not ctx.calc_grad_inputs:
                return None, None, None
            grad = grad.contiguous()
            inputs, dy_dx = ctx.saved_tensors
            B, input_dim, degree = ctx.dims
            grad_inputs = torch.zeros_like(inputs)
            backend.sh_encode_backward(grad, inputs, B, input_dim, degree, dy_dx, grad_inputs)
            return grad_inputs, None, None

    return _SHEncode.apply


class SHEncoderBase(nn.Module):
    """reference: SHEncoder, sphere_harmonics.py:67-95.  Subclasses bind ``_sh_encode``."""

    _sh_encode = None

    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert 0 < self.degree <= 8, "SH encoder only supports degree in [1, 8]"

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        # inputs [..., 3] in [-size, size] -> [..., degree^2]
        inputs = inputs / size
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        outputs = type(self)._sh_encode(inputs, self.degree, inputs.requires_grad)
        return outputs.reshape(prefix_shape + [self.output_dim])
