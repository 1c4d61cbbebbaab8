"""Autograd wrappers of the fused Plenoxel lookup (reference formulation: network.py:311-322, 383-409)."""
import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd


def is_channels_last_3d(t):
    """[1,C,D,H,W] stored as [D][H][W][C]."""
    return t.dim() == 5 and t.shape[0] == 1 and t.permute(0, 2, 3, 4, 1).is_contiguous()


def to_channels_last_3d_param(t):
    """Same logical shape (state-dict compatible), channels-last storage."""
    _, C, D, H, W = t.shape
    out = torch.empty_strided(t.shape, (C * D * H * W, 1, H * W * C, W * C, C), dtype=t.dtype, device=t.device)
    out.copy_(t)
    return out


def _direct(p, vol):
    # the trainer's flat gradient bucket: a dense fp32 buffer with the parameter's own (channels-last) strides
    return p.is_leaf and p.grad is not None and p.grad.stride() == vol.stride() and p.grad.dtype == torch.float32


def make_plenoxel_ops(backend, device_type="cuda"):
    class _Features(Function):
        """(xyz [M,3] world, aabb [6] host floats, volume [1,C,D,H,W]) -> raw features [M,C] (compute_plenoxel_fea)."""

        @staticmethod
        @custom_fwd(device_type=device_type, cast_inputs=torch.float32)
        def forward(ctx, xyz, aabb_host, volume, degree):
            xyz = xyz.contiguous()
            vol = volume if is_channels_last_3d(volume) else to_channels_last_3d_param(volume.detach())
            feat = torch.empty(xyz.shape[0], vol.shape[1], dtype=torch.float32, device=xyz.device)
            backend.plenoxel_forward(xyz, None, aabb_host, vol, degree, 0.0, 0.0, feat, None, None, None, None)
            ctx.save_for_backward(xyz, vol)
            ctx.aabb_host, ctx.degree, ctx.leaf = aabb_host, degree, volume
            return feat

        @staticmethod
        @custom_bwd(device_type=device_type)
        def backward(ctx, g_feat):
            xyz, vol = ctx.saved_tensors
            p = ctx.leaf
            direct = _direct(p, vol)
            gv = p.grad if direct else torch.zeros_like(vol)  # zeros_like keeps the channels-last strides
            backend.plenoxel_backward(xyz, None, ctx.aabb_host, ctx.degree, 0.0, 0.0, None, None, g_feat.contiguous().float(), None, None,
                                      None, gv)
            return None, None, (None if direct else gv), None

    class _Head(Function):
        """(xyz, dirs, aabb, volume, degree, clip_min, clip_max) -> (sigma [M], rgb [M,3], sigma_l [M], h0_raw [M]):
        lookup + clamp + trunc_exp + SH colour + sigmoid in one kernel each way.  h0_raw carries no gradient."""

        @staticmethod
        @custom_fwd(device_type=device_type, cast_inputs=torch.float32)
        def forward(ctx, xyz, dirs, aabb_host, volume, degree, clip_min, clip_max):
            xyz, dirs = xyz.contiguous(), dirs.contiguous()
            vol = volume if is_channels_last_3d(volume) else to_channels_last_3d_param(volume.detach())
            M, dev = xyz.shape[0], xyz.device
            h0 = torch.empty(M, dtype=torch.float32, device=dev)
            sigma_l, sigma = torch.empty_like(h0), torch.empty_like(h0)
            rgb = torch.empty(M, 3, dtype=torch.float32, device=dev)
            backend.plenoxel_forward(xyz, dirs, aabb_host, vol, degree, clip_min, clip_max, None, h0, sigma_l, sigma, rgb)
            ctx.save_for_backward(xyz, dirs, vol, h0, rgb)
            ctx.cfg = (aabb_host, degree, clip_min, clip_max)
            ctx.leaf = volume
            ctx.mark_non_differentiable(h0)
            return sigma, rgb, sigma_l, h0

        @staticmethod
        @custom_bwd(device_type=device_type)
        def backward(ctx, g_sigma, g_rgb, g_sigma_l, _g_h0):
            xyz, dirs, vol, h0, rgb = ctx.saved_tensors
            aabb_host, degree, clip_min, clip_max = ctx.cfg
            p = ctx.leaf
            direct = _direct(p, vol)
            gv = p.grad if direct else torch.zeros_like(vol)
            c = lambda g: None if g is None else g.contiguous().float()
            backend.plenoxel_backward(xyz, dirs, aabb_host, degree, clip_min, clip_max, h0, rgb, None, c(g_sigma), c(g_sigma_l), c(g_rgb), gv)
            return None, None, None, (None if direct else gv), None, None, None

    return _Features.apply, _Head.apply
