"""Operator set used by the harness.  ``hip_ops()`` binds the product's HIP packages; the
test-suite builds the same namespace around the CPU oracle (tests/oracle_ops.py).  The product
never constructs anything but the HIP set."""
import os
import types

import torch


def _mse_loss():
    from .losses import mse_fused
    return mse_fused


def _distill_loss():
    from .losses import distill_loss_normL2
    return distill_loss_normL2


def _flat_adamw():
    from .flat_adamw import FlatAdamW
    return FlatAdamW


def hip_ops():
    import gridencoder
    import raymarching
    import fusedhead
    import pvd_hip
    import shencoder
    import vmencoder
    import plenoxel

    def get_rays_fused(pose, intrinsics, H, W, N, error_map=None, generator=None):
        """reference: get_rays (utils.py:324-404) for one pose: randint pixel ids (or the error-map draw, :357-381) + one HIP kernel."""
        dev = pose.device
        coarse = None
        if error_map is not None:
            from .scene import sample_pixels_by_error
            inds2, coarse = sample_pixels_by_error(error_map.reshape(1, -1).to(dev), N, H, W, generator)
            inds = inds2[0].contiguous()
        else:
            inds = torch.randint(0, H * W, size=[N], device=dev, generator=generator)  # may duplicate, like the reference
        rays_o = torch.empty(1, N, 3, device=dev)
        rays_d = torch.empty(1, N, 3, device=dev)
        fx, fy, cx, cy = intrinsics
        pvd_hip.get_rays(pose.reshape(4, 4).contiguous(), fx, fy, cx, cy, inds, W, N, rays_o, rays_d)
        out = {"rays_o": rays_o, "rays_d": rays_d, "inds": inds[None]}
        if coarse is not None:
            out["inds_coarse"] = coarse
        return out

    def make_batch(poses, state, seed, intrinsics, H, W, N, aabb, min_near):
        """One training batch in one launch (pvd_make_ray_batch): rays of N random pixels of poses[state[0]], a random
        background per ray and near/far; returns (rays_o [1,N,3], rays_d [1,N,3], bg [1,N,3], (nears, fars))."""
        dev = poses.device
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        rays_o, rays_d, bg, nears, fars = f(1, N, 3), f(1, N, 3), f(1, N, 3), f(N), f(N)
        fx, fy, cx, cy = intrinsics
        pvd_hip.make_ray_batch(poses, state, seed, fx, fy, cx, cy, H, W, N, aabb, min_near, None, rays_o, rays_d, bg, nears, fars)
        return rays_o, rays_d, bg, (nears, fars)

    def freq_encode(x, bands, include_input=True, out_dtype=torch.float32, row_stride=None):
        return pvd_hip.freq_encode(x, bands, include_input, out_dtype, row_stride)

    return types.SimpleNamespace(freq_encode=freq_encode, make_batch=make_batch, occupancy=pvd_hip.occupancy_backend, raymarching=raymarching, GridEncoder=gridencoder.GridEncoder, SHEncoder=shencoder.SHEncoder,
                                 vm_encode=vmencoder.vm_encode, vm_encode_infer=vmencoder.vm_encode_infer, plenoxel=plenoxel, get_rays=get_rays_fused, fused_head=fusedhead, distill_loss=_distill_loss(), mse_loss=_mse_loss(), flat_adamw=_flat_adamw(), device_type="cuda", name="hip")
