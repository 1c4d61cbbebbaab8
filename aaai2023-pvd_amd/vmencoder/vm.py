"""Autograd wrapper of the fused VM lookup (reference formulation: network.py:216-309)."""
import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd


def is_channels_last(t):
    """[1,R,H,W] stored as [H][W][S >= R] (for lines [1,R,L,1]: [L][S]): channel stride 1, texel stride S, row stride W*S.
    S > R is the interleaved layout (`interleave_factors`): sigma and colour factors share one [H][W][64] buffer."""
    if t.dim() != 4 or t.shape[0] != 1 or (t.shape[1] > 1 and t.stride(1) != 1):
        return False
    s = t.stride(3) if t.shape[3] > 1 else t.stride(2)
    return s >= t.shape[1] and (t.shape[3] == 1 or t.shape[2] == 1 or t.stride(2) == t.shape[3] * s)


def interleave_factors(sigma, color):
    """Two channels-last factors over the same texels ([1,Rs,H,W] and [1,Rc,H,W]) re-stored in ONE [H][W][Rs+Rc] buffer;
    returns the two strided views (same logical shapes and values).  A tap of the lookup is then one contiguous
    (Rs+Rc)*4-byte access."""
    assert sigma.shape[2:] == color.shape[2:] and sigma.shape[0] == color.shape[0] == 1
    Rs, Rc, H, W = sigma.shape[1], color.shape[1], sigma.shape[2], sigma.shape[3]
    S = Rs + Rc
    buf = torch.empty(H * W * S, dtype=sigma.dtype, device=sigma.device)
    vs = torch.as_strided(buf, (1, Rs, H, W), (H * W * S, 1, W * S, S), 0)
    vc = torch.as_strided(buf, (1, Rc, H, W), (H * W * S, 1, W * S, S), Rs)
    vs.copy_(sigma)
    vc.copy_(color)
    return vs, vc


def to_channels_last_param(t):
    """Same logical shape (state-dict compatible), channels-last storage."""
    out = torch.empty_strided(t.shape, (t.shape[1] * t.shape[2] * t.shape[3], 1, t.shape[3] * t.shape[1], t.shape[1]),
                              dtype=t.dtype, device=t.device)
    out.copy_(t)
    return out


def make_vm_encode(backend, device_type="cuda"):
    class _VMEncode(Function):
        """(xyz [M,3], aabb [6], sigma_mat x3, sigma_vec x3, color_mat x3, color_vec x3)
        -> sigma_feat [M] (f32), color_prod [M,144] (f16 under autocast, else f32)."""

        @staticmethod
        @custom_fwd(device_type=device_type)
        def forward(ctx, xyz, aabb_host, *tables):
            # an optional trailing dict: the hand-over point of the VM head's weight-gradient reduction, which then rides on this
            # lookup's backward launch (fusedhead._VMHeadTrain.backward fills it, pvd_hip.vm_backward(head_dw=) consumes it)
            ctx.head_dw = None
            if tables and isinstance(tables[-1], dict):
                ctx.head_dw, tables = tables[-1], tables[:-1]
            xyz = xyz.contiguous().float()
            tabs = [t if is_channels_last(t) else to_channels_last_param(t.detach()) for t in tables]
            res = [0, 0, 0]
            # mat_i is [1,R,res[m1],res[m0]], vec_i [1,R,res[vec_id],1] (network.py:199-212)
            res[0], res[1] = tabs[0].shape[3], tabs[0].shape[2]
            res[2] = tabs[1].shape[2]
            M = xyz.shape[0]
            half = torch.is_autocast_enabled(device_type)
            sigma_feat = torch.empty(M, dtype=torch.float32, device=xyz.device)
            color_prod = torch.empty(M, 144, dtype=torch.float16 if half else torch.float32, device=xyz.device)
            # "pack" in the hand-over dict: (basis_mat.weight, color_net.0/1/2.weight, image) -- the VM head's packed weight image is
            # written by extra workgroups of this lookup's launch (pvd_hip.vm_forward(pack=)); "packed" tells the head it is there
            pack = ctx.head_dw.pop("pack", None) if ctx.head_dw is not None else None
            if pack is not None and M > 0:
                backend.vm_forward(xyz, aabb_host, tabs, res, sigma_feat, color_prod, pack=tuple(t.detach() for t in pack))
                ctx.head_dw["packed"] = pack[-1]
            else:
                backend.vm_forward(xyz, aabb_host, tabs, res, sigma_feat, color_prod)
            ctx.save_for_backward(xyz, *tabs)
            ctx.aabb_host, ctx.res = aabb_host, res
            ctx.leaves = tables  # the Parameter objects themselves (to reach their .grad buffers)
            return sigma_feat, color_prod

        @staticmethod
        @custom_bwd(device_type=device_type)
        def backward(ctx, g_sigma, g_prod):
            xyz, *tabs = ctx.saved_tensors
            # Fast path: every factor is a leaf that already owns a dense gradient buffer with the factor's
            # own (channels-last) strides -- the trainer's flat gradient bucket.  The kernel's atomics then
            # accumulate straight into it: no 69 MB of zero-fill + add per step.
            direct = all(p.is_leaf and p.grad is not None and p.grad.stride() == t.stride() and p.grad.dtype == torch.float32
                         for p, t in zip(ctx.leaves, tabs))
            hkw = {"head_dw": ctx.head_dw} if (ctx.head_dw is not None and ctx.head_dw.get("rider") is not None) else {}
            tail = (None,) if ctx.head_dw is not None else ()
            if direct:
                backend.vm_backward(xyz, ctx.aabb_host, tabs, ctx.res, g_sigma.contiguous().float(), g_prod.contiguous(),
                                    [p.grad for p in ctx.leaves], **hkw)
                return (None, None) + (None,) * len(tabs) + tail
            # (the factors' own strides; zeros_like would densify an interleaved view)
            grads = [torch.empty_strided(t.shape, t.stride(), dtype=torch.float32, device=t.device).zero_() for t in tabs]
            backend.vm_backward(xyz, ctx.aabb_host, tabs, ctx.res, g_sigma.contiguous().float(), g_prod.contiguous(), grads, **hkw)
            return (None, None, *grads) + tail

    return _VMEncode.apply
