"""``gridencoder`` operator surface (reference: gridencoder/grid.py), re-hosted on libpvd_hip.so.

``make_grid_encode(backend)`` builds the autograd Function around a backend with the
reference's ``_gridencoder`` function pair; ``GridEncoderBase`` is the nn.Module with the
reference's constructor, parameter/buffer names (``embeddings``, ``offsets``) and forward.
"""
import math

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

_gridtype_to_id = {"hash": 0, "tiled": 1}


def make_grid_encode(backend, device_type="cuda"):
    class _GridEncode(Function):
        # reference: _grid_encode, grid.py:20-136
        @staticmethod
        @custom_fwd(device_type=device_type)
        def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                    align_corners=False):
            # inputs [B, D] float in [0, 1]; embeddings [sO, C]; offsets [L + 1] int32; returns [B, L * C]
            inputs = inputs.contiguous()
            B, D = inputs.shape
            L = offsets.shape[0] - 1
            C = embeddings.shape[1]
            S = float(np.log2(per_level_scale))  # the kernel applies exp2 (grid.py:45-47)
            H = base_resolution

            # autocast: half table only, positions stay float; odd C stays float (grid.py:49-52)
            if torch.is_autocast_enabled(device_type) and C % 2 == 0:
                embeddings = embeddings.to(torch.half)

            outputs = torch.empty(L, B, C, device=inputs.device, dtype=embeddings.dtype)  # level-major for the kernel
            if calc_grad_inputs:
                dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=embeddings.dtype)
            else:
                dy_dx = torch.empty(1, device=inputs.device, dtype=embeddings.dtype)

            backend.grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs, dy_dx,
                                        gridtype, align_corners)

            outputs = outputs.permute(1, 0, 2).reshape(B, L * C)
            ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
            ctx.dims = [B, D, C, L, S, H, gridtype]
            ctx.calc_grad_inputs = calc_grad_inputs
            ctx.align_corners = align_corners
            return outputs

        @staticmethod
        @custom_bwd(device_type=device_type)
        def backward(ctx, grad):
            inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
            B, D, C, L, S, H, gridtype = ctx.dims
            calc_grad_inputs = ctx.calc_grad_inputs

            grad = grad.view(B, L, C).permute(1, 0, 2).contiguous()  # [B, L*C] -> [L, B, C]
            grad_embeddings = torch.zeros_like(embeddings)
            if calc_grad_inputs:
                grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype)
            else:
                grad_inputs = torch.zeros(1, device=inputs.device, dtype=embeddings.dtype)

            backend.grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs,
                                         dy_dx, grad_inputs, gridtype, ctx.align_corners)

            if calc_grad_inputs:
                return grad_inputs.to(inputs.dtype), grad_embeddings, None, None, None, None, None, None
            return None, grad_embeddings, None, None, None, None, None, None

    return _GridEncode.apply


def level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners):
    """Row offsets of each level in the embedding table (reference: grid.py:176-190)."""
    max_params = 2 ** log2_hashmap_size
    offsets, offset = [], 0
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        side = resolution if align_corners else resolution + 1
        n = min(max_params, side ** input_dim)
        n = int(np.ceil(n / 8) * 8)  # multiple of 8 rows
        offsets.append(offset)
        offset += n
    offsets.append(offset)
    return offsets


class GridEncoderBase(nn.Module):
    """reference: GridEncoder, grid.py:142-232.  Subclasses bind ``_grid_encode``."""

    _grid_encode = None

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype="hash", align_corners=False):
        super().__init__()
        # the finest resolution, when given, overrides per_level_scale (grid.py:158-161)
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _gridtype_to_id[gridtype]
        self.align_corners = align_corners
        self.max_params = 2 ** log2_hashmap_size

        offsets = level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners)
        self.register_buffer("offsets", torch.from_numpy(np.array(offsets, dtype=np.int32)))
        self.n_params = offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(offsets[-1], level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)  # grid.py:200-202

    def __repr__(self):
        finest = int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> {finest} per_level_scale={self.per_level_scale:.4f} "
                f"params={tuple(self.embeddings.shape)} gridtype={self.gridtype} align_corners={self.align_corners}")

    def forward(self, inputs, bound=1):
        # inputs [..., input_dim] in [-bound, bound] -> [..., num_levels * level_dim]
        inputs = (inputs + bound) / (2 * bound)
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        outputs = type(self)._grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                                          inputs.requires_grad, self.gridtype_id, self.align_corners)
        return outputs.view(prefix_shape + [self.output_dim])
