"""Bias-free Linear for the sigma / colour heads, with a weight gradient that is actually parallel.

The heads are tiny (<= 144 x 64 weights) but see ~1e5 samples per step, so the weight gradient
dW = g^T x is a GEMM with a minuscule output and a huge reduction dimension (K = M ~ 90k).  The
library GEMM picks a 16x16 tile and walks K serially in one or two workgroups (0.3-0.65 ms per layer
on MI355X, rocprofv3 r01); here K is split into S independent batches (one batched GEMM over
[S, out, M/S] x [S, M/S, in]) and the S partial products are summed in fp32.
Reference: the nn.Linear layers of sigma_net / color_net / basis_mat (distill_mutual/network.py:103-152)."""
import math

import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd


def _split(M):
    for s in (128, 64, 32, 16, 8):
        if M % s == 0 and M // s >= 64:
            return s
    return 1


def make_skinny_linear(device_type="cuda"):
    class _SkinnyLinear(Function):
        @staticmethod
        @custom_fwd(device_type=device_type, cast_inputs=torch.float16 if device_type == "cuda" else None)
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w.t()

        @staticmethod
        @custom_bwd(device_type=device_type)
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            gx = g @ w if ctx.needs_input_grad[0] else None
            gw = None
            if ctx.needs_input_grad[1]:
                M = x.shape[0]
                S = _split(M)
                if S == 1:
                    gw = g.t() @ x
                else:
                    part = torch.bmm(g.reshape(S, M // S, -1).transpose(1, 2), x.reshape(S, M // S, -1))
                    gw = part.sum(0, dtype=torch.float32).to(w.dtype)
            return gx, gw

    return _SkinnyLinear.apply


def make_skinny_linear_bias(device_type="cuda"):
    """nn.Linear WITH bias for the 256-wide layers of the NeRF MLP trunk (network.py:154-182): same story as above -- the
    library's weight-gradient GEMM [256 x M] . [M x 256] with M ~ 9e4 runs on 16 workgroups (212 us per layer on MI355X,
    rocprofv3 round 2: 1.7 ms of a 4.6 ms step) -- with the reduction dimension split into S independent batches, and the bias
    gradient reduced in the same two stages."""
    class _SkinnyLinearBias(Function):
        @staticmethod
        @custom_fwd(device_type=device_type, cast_inputs=torch.float16 if device_type == "cuda" else None)
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x, w)
            return torch.addmm(b, x, w.t())

        @staticmethod
        @custom_bwd(device_type=device_type)
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            gx = g @ w if ctx.needs_input_grad[0] else None
            gw = gb = None
            M = x.shape[0]
            S = _split(M)
            if ctx.needs_input_grad[1]:
                if S == 1:
                    gw = g.t() @ x
                else:
                    part = torch.bmm(g.reshape(S, M // S, -1).transpose(1, 2), x.reshape(S, M // S, -1))
                    gw = part.sum(0, dtype=torch.float32).to(w.dtype)
            if ctx.needs_input_grad[2]:
                gb = (g.reshape(S, M // S, -1).sum(1, dtype=torch.float32).sum(0) if S > 1 else g.sum(0, dtype=torch.float32)).to(g.dtype)
            return gx, gw, gb

    return _SkinnyLinearBias.apply
