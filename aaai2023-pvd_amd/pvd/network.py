"""Radiance-field models: the harness counterpart of the reference's ``NeRFNetwork``
(distill_mutual/network.py:13-683): one class, four representations (``hash`` INGP, ``mlp``
NeRF, ``vm`` TensoRF, ``tensors`` Plenoxels), identical parameter names and shapes so that
reference checkpoints line up (``encoder.embeddings``, ``sigma_net.N.weight``,
``color_net.N.weight``, ``sigma_mat.N`` ...).
"""
import ast
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .activation import make_trunc_exp
from .encoding import get_encoder
from .linear import make_skinny_linear, make_skinny_linear_bias
from .renderer import NeRFRenderer


def _channels_last(t):
    out = torch.empty_strided(t.shape, (t.shape[1] * t.shape[2] * t.shape[3], 1, t.shape[3] * t.shape[1], t.shape[1]), dtype=t.dtype,
                              device=t.device)
    return out.copy_(t)


def _channels_last_3d(t):
    _, C, D, H, W = t.shape
    return torch.empty_strided(t.shape, (C * D * H * W, 1, H * W * C, W * C, C), dtype=t.dtype, device=t.device).copy_(t)


class _L1MeanSum(torch.autograd.Function):
    _inv_cache = {}

    @staticmethod
    def forward(ctx, *tensors):
        ctx.save_for_backward(*tensors)
        ctx.leaves = tensors
        norms = torch._foreach_norm(list(tensors), 1)
        key = (tensors[0].device, tuple(t.numel() for t in tensors))
        inv = _L1MeanSum._inv_cache.get(key)
        if inv is None:  # created once, outside any graph capture (no H2D copy inside a captured step)
            inv = torch.tensor([1.0 / t.numel() for t in tensors], dtype=torch.float32, device=tensors[0].device)
            _L1MeanSum._inv_cache[key] = inv
        ctx.inv = inv
        return (torch.stack(norms).float() * inv).sum()

    @staticmethod
    def backward(ctx, g):
        tensors = ctx.saved_tensors
        signs = torch._foreach_sign(list(tensors))
        scales = (g.float() * ctx.inv).unbind(0)
        torch._foreach_mul_(signs, list(scales))
        # leaves that already own a gradient buffer with their own strides (the trainer's flat bucket):
        # one multi-tensor add instead of one AccumulateGrad launch per factor
        leaves = ctx.leaves
        if all(p.is_leaf and p.grad is not None and p.grad.stride() == s.stride() for p, s in zip(leaves, signs)):
            torch._foreach_add_([p.grad for p in leaves], signs)
            return (None,) * len(signs)
        return tuple(signs)


def _l1_mean_sum(*tensors):
    return _L1MeanSum.apply(*tensors)


def _mlp(dims):
    """bias-free Linear stack (network.py:103-152)."""
    return nn.ModuleList([nn.Linear(dims[i], dims[i + 1], bias=False) for i in range(len(dims) - 1)])


class NeRFNetwork(NeRFRenderer):
    def __init__(self, ops, encoding="hashgrid", encoding_dir="sphere_harmonics", num_layers=2, hidden_dim=64, geo_feat_dim=15,
                 num_layers_color=3, hidden_dim_color=64, bound=1, model_type="hash", args=None, is_teacher=False, **kwargs):
        super().__init__(ops, bound, **kwargs)
        assert model_type in ["hash", "mlp", "vm", "tensors"]
        self.is_teacher = is_teacher
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.geo_feat_dim = geo_feat_dim
        self.args = args
        self.opt = args
        self.model_type = model_type
        self.trunc_exp = make_trunc_exp(ops.device_type)
        self.linear = make_skinny_linear(ops.device_type)  # F.linear with a split-K weight gradient
        self.linear_bias = make_skinny_linear_bias(ops.device_type)  # the same for the biased layers of the NeRF MLP trunk
        self.plenoxel_degree = args.plenoxel_degree
        self.plenoxel_res = ast.literal_eval(args.plenoxel_res) if isinstance(args.plenoxel_res, str) else list(args.plenoxel_res)
        assert len(self.plenoxel_res) == 3

        # 14 levels, 2 features, finest resolution 2048*bound (network.py:47-51)
        self.in_dim = 14 * 2
        self.encoder = None
        if model_type == "hash":
            self.encoder, self.in_dim = get_encoder(ops, encoding, desired_resolution=2048 * bound, num_levels=14)
        elif model_type == "mlp":
            self.encoder_nerf_pe, self.in_dim_nerf = get_encoder(ops, "frequency", multires=args.PE)
            self.skips = args.skip
            self.nerf_layer_num = args.nerf_layer_num
            W = args.nerf_layer_wide
            layers = [nn.Linear(self.in_dim_nerf, W)]
            for i in range(self.nerf_layer_num - 2):
                layers.append(nn.Linear(W + self.in_dim_nerf if i == self.skips else W, W))  # skip feeds the NEXT layer
            layers.append(nn.Linear(W, self.in_dim))
            self.nerf_mlp = nn.ModuleList(layers)
        elif model_type == "vm":
            self.sigma_rank = [16] * 3
            self.color_rank = [48] * 3
            self.color_feat_dim = 15
            self.mat_ids = [[0, 1], [0, 2], [1, 2]]
            self.vec_ids = [2, 1, 0]
            self.resolution = [args.resolution0] * 3
            self.sigma_mat, self.sigma_vec = self.init_one_vm(self.sigma_rank, self.resolution)
            self.color_mat, self.color_vec = self.init_one_vm(self.color_rank, self.resolution)
            self.basis_mat = nn.Linear(sum(self.color_rank), self.color_feat_dim, bias=False)
        elif model_type == "tensors":
            s, fea_dim = 0.02, self.plenoxel_degree ** 2 * 3 + 1  # network.py:92-96
            self.tensor_volume = nn.ParameterList([nn.Parameter(_channels_last_3d(s * torch.randn((1, fea_dim, *self.plenoxel_res))))])

        if model_type in ("hash", "mlp"):
            dims = [self.in_dim] + [hidden_dim] * (num_layers - 1) + [1 + geo_feat_dim]
            self.sigma_net = _mlp(dims)

        self.num_layers_color = num_layers_color
        self.hidden_dim_color = hidden_dim_color
        if model_type == "tensors":
            self.encoder_dir, self.in_dim_dir = get_encoder(ops, "sphere_harmonics", degree=self.plenoxel_degree)
        else:
            self.encoder_dir, self.in_dim_dir = get_encoder(ops, encoding_dir, input_dim=3, multires=2)
            dims = [self.in_dim_dir + geo_feat_dim] + [hidden_dim] * (num_layers_color - 1) + [3]
            self.color_net = _mlp(dims)
        self.bg_net = None

        self.feature_sigma_color = None
        self.sigma_l = None
        self.color_l = None

    # ------------------------------------------------------------------ VM (TensoRF) tables
    def init_one_vm(self, n_component, resolution, scale=0.1):
        """plane [1,R,res,res] + line [1,R,res,1] factors per axis triple (network.py:193-214)."""
        mat, vec = [], []
        for i in range(3):
            m0, m1 = self.mat_ids[i]
            # same logical shapes as the reference, stored channels-last ([H][W][R]) for the fused lookup
            mat.append(nn.Parameter(_channels_last(scale * torch.randn((1, n_component[i], resolution[m1], resolution[m0])))))
            vec.append(nn.Parameter(_channels_last(scale * torch.randn((1, n_component[i], resolution[self.vec_ids[i]], 1)))))
        return nn.ParameterList(mat), nn.ParameterList(vec)

    def _vm_coords(self, x):
        mat_coord = torch.stack([x[..., self.mat_ids[i]] for i in range(3)]).detach().view(3, -1, 1, 2)
        vec = torch.stack([x[..., self.vec_ids[i]] for i in range(3)])
        vec_coord = torch.stack((torch.zeros_like(vec), vec), dim=-1).detach().view(3, -1, 1, 2)
        return mat_coord, vec_coord

    def get_sigma_feat(self, x):
        """sum_r plane_r(u,v) * line_r(w) over the 3 axis triples (network.py:216-262)."""
        N = x.shape[0]
        mat_coord, vec_coord = self._vm_coords(x)
        sigma_feat = torch.zeros([N], device=x.device)
        for i in range(3):
            m = F.grid_sample(self.sigma_mat[i], mat_coord[[i]], align_corners=True).view(-1, N)
            v = F.grid_sample(self.sigma_vec[i], vec_coord[[i]], align_corners=True).view(-1, N)
            sigma_feat = sigma_feat + torch.sum(m * v, dim=0)
        return sigma_feat

    def get_color_feat(self, x):
        """144 plane*line products -> basis_mat -> 15 colour features (network.py:264-309)."""
        N = x.shape[0]
        mat_coord, vec_coord = self._vm_coords(x)
        m = torch.cat([F.grid_sample(self.color_mat[i], mat_coord[[i]], align_corners=True).view(-1, N) for i in range(3)], dim=0)
        v = torch.cat([F.grid_sample(self.color_vec[i], vec_coord[[i]], align_corners=True).view(-1, N) for i in range(3)], dim=0)
        return self.basis_mat((m * v).T)

    def vm_features(self, x):
        """(sigma_feat [N], color_feat [N,15]) for world positions x.  HIP: one fused plane x line lookup
        (vmencoder) + basis_mat; oracle / reference formulation: the twelve grid_samples above."""
        vm_encode = getattr(self.ops, "vm_encode", None)
        if vm_encode is None:
            xn = self._unit_cube(x)
            return self.get_sigma_feat(xn), self.get_color_feat(xn)
        sigma_feat, prod = vm_encode(x, self._aabb(), *self.sigma_mat, *self.sigma_vec, *self.color_mat, *self.color_vec)
        return sigma_feat, self.linear(prod, self.basis_mat.weight)

    def _aabb(self):
        if not hasattr(self, "_aabb_host"):
            self._aabb_host = tuple(float(v) for v in self.aabb_train.tolist())  # one D2H copy, cached
        return self._aabb_host

    def density_loss(self):
        """L1 on the sigma factors: sum_i mean|sigma_mat_i| + mean|sigma_vec_i| (network.py:549-557), as two
        multi-tensor kernels forward and two backward instead of ~40 elementwise launches."""
        return _l1_mean_sum(*self.sigma_mat, *self.sigma_vec)

    # ------------------------------------------------------------------ Plenoxels / NeRF-MLP
    def compute_plenoxel_fea(self, x):
        """x already mapped to [-1,1]^3 -> [N, fea_dim]; the reference formulation (network.py:311-322)."""
        vol = self.tensor_volume[0]
        return F.grid_sample(vol, x.view(1, 1, -1, 1, 3), align_corners=True).view(-1, x.shape[0]).permute(1, 0)

    def _plenoxel_ops(self, x):
        px = getattr(self.ops, "plenoxel", None)
        # the editing demo rewrites the volume in place every call (network.py:313-316): keep that on the torch path
        return px if (px is not None and x.is_cuda and not self.args.enable_edit_plenoxel) else None

    def forward_nerf_mlp(self, x):
        fe = getattr(self.ops, "freq_encode", None)
        if (fe is not None and x.is_cuda and not torch.is_grad_enabled() and torch.is_autocast_enabled("cuda")
                and len(self.encoder_nerf_pe.freq_bands) <= 16 and self.skips + 1 < len(self.nerf_mlp) - 1):
            return self._forward_nerf_mlp_frozen(x, fe)
        x = self.encoder_nerf_pe(x)
        if x.is_cuda and self.in_dim_nerf % 8:
            return self._forward_nerf_mlp_aligned(x)
        pts = x
        last = len(self.nerf_mlp) - 1
        for i, layer in enumerate(self.nerf_mlp):
            x = layer(x)
            if i != last:
                x = F.relu(x, inplace=True)
            if i == self.skips:
                x = torch.cat([pts, x], -1)
        return x

    @torch.no_grad()
    def _forward_nerf_mlp_frozen(self, x, freq_encode):
        """forward_nerf_mlp of a model that is not being trained, under fp16 autocast: the positional encoding comes out of
        one kernel as f16 rows padded to a multiple of 8 columns; weights and biases are cast to f16 once (not per call and
        layer); Linear + ReLU is one library GEMM with a ReLU epilogue (torch._addmm_activation).  Same arithmetic as the
        autocast formulation -- f16 GEMMs with f32 accumulation, bias added before the rounding to f16, ReLU -- in 1 + 8
        launches instead of ~75 (rocprofv3, profiles/r02: 185 us of sin / cos launches, 17 us of ReLU and 9 us of casts
        per layer at 92 k rows)."""
        import pvd_hip
        enc = self.encoder_nerf_pe
        n_in = self.in_dim_nerf
        n_pad = (n_in + 7) // 8 * 8
        params = [p for layer in self.nerf_mlp for p in (layer.weight, layer.bias)]
        key = pvd_hip.weights_key(params)
        cache = getattr(self, "_nerf16_cache", None)
        if cache is None or cache[0] != key:
            ws = []
            last = len(self.nerf_mlp) - 1
            for i, layer in enumerate(self.nerf_mlp):
                w = layer.weight.detach().to(torch.float16)
                b = layer.bias.detach().to(torch.float16)
                if i == 0:
                    w = F.pad(w, (0, n_pad - n_in))
                elif i == self.skips + 1:
                    w = torch.cat([F.pad(w[:, :n_in], (0, n_pad - n_in)), w[:, n_in:]], dim=1)
                if i == last and w.shape[0] % 8:  # output rows of 28 halfs are not 16-byte multiples either
                    extra = (-w.shape[0]) % 8
                    w, b = F.pad(w, (0, 0, 0, extra)), F.pad(b, (0, extra))
                ws.append((w.contiguous(), b.contiguous()))
            cache = self._nerf16_cache = (key, ws)
        pts = freq_encode(x.reshape(-1, enc.input_dim).float().contiguous(), enc.freq_bands, enc.include_input, torch.float16, n_pad)
        h = pts
        last = len(self.nerf_mlp) - 1
        for i, (w, b) in enumerate(cache[1]):
            if i != last:
                h = torch._addmm_activation(b, h, w.t())  # relu(h W^T + b)
            else:
                h = torch.addmm(b, h, w.t())
            if i == self.skips:
                h = torch.cat([pts, h], -1)
        return h[:, :self.nerf_mlp[last].out_features]

    def _forward_nerf_mlp_aligned(self, pts):
        """forward_nerf_mlp with the 63-wide positional encoding padded to 64 by a zero column (and the matching zero weight
        column): the same sums, but the contraction lengths of the first layer (63) and of the skip layer (319) become 64 and
        320 -- the library GEMM takes a 5x slower kernel for rows that are not 16-byte multiples (233 vs 43 us per call at
        92 k rows on MI355X, rocprofv3 r02)."""
        n_in = self.in_dim_nerf
        pad = (-n_in) % 8
        ptsp = F.pad(pts, (0, pad))
        x = ptsp
        last = len(self.nerf_mlp) - 1
        for i, layer in enumerate(self.nerf_mlp):
            w = layer.weight
            if i == 0:
                w = F.pad(w, (0, pad))
            elif i == self.skips + 1:
                w = torch.cat([F.pad(w[:, :n_in], (0, pad)), w[:, n_in:]], dim=1)
            x = self.linear_bias(x, w, layer.bias) if x.dim() == 2 else F.linear(x, w, layer.bias)
            if i != last:
                x = F.relu(x, inplace=True)
            if i == self.skips:
                x = torch.cat([ptsp, x], -1)
        return x

    def _unit_cube(self, x):
        return 2 * (x - self.aabb_train[:3]) / (self.aabb_train[3:] - self.aabb_train[:3]) - 1

    def _in_stage1(self):
        return self.training and self.args.global_step < self.args.stage_iters["stage1"]

    def _color_head(self, enc_d, feat):
        h = torch.cat([enc_d, feat], dim=-1)
        for l in range(self.num_layers_color):
            h = self.linear(h, self.color_net[l].weight)
            if l != self.num_layers_color - 1:
                h = F.relu(h, inplace=True)
        return torch.sigmoid(h)

    # ------------------------------------------------------------------ forward / density
    def _fused_ok(self, x):
        fh = getattr(self.ops, "fused_head", None)
        return fh is not None and x.is_cuda and torch.is_autocast_enabled("cuda") and self.bg_net is None

    def forward(self, x, d):
        """x [N,3] in [-bound,bound], d [N,3] unit -> (sigma [N], rgb [N,3]); reference network.py:335-437.
        Side outputs kept for the distillation losses: feature_sigma_color, sigma_l, color_l."""
        a = self.args
        if self._fused_ok(x):
            fh = self.ops.fused_head
            out = None
            if self.model_type == "hash" and not torch.is_grad_enabled():
                out = fh.hash_head_infer(self, x, d)  # frozen teacher / inference: encoder + whole head, 2 launches
            elif self.model_type == "vm" and not torch.is_grad_enabled():
                sraw, prod = self.ops.vm_encode(x, self._aabb(), *self.sigma_mat, *self.sigma_vec, *self.color_mat, *self.color_vec)
                out = fh.vm_head_infer(self, sraw, prod, d)
            elif self.model_type == "vm" and hasattr(fh, "vm_head_train"):
                # a dict shared by the two autograd nodes: the head's weight-gradient reduction rides on the lookup's backward
                # launch (PVD_HEAD_DW_RIDE=0: a launch of its own)
                head_dw = {} if (torch.is_grad_enabled() and os.environ.get("PVD_HEAD_DW_RIDE", "1") != "0") else None
                if head_dw is not None and getattr(self, "_inf_check_in_backward", None) is not None:
                    # (trainer) the scaler's inf check of this model's gradients rides on the same launch: (flag, note)
                    head_dw["found_inf"] = self._inf_check_in_backward
                if head_dw is not None and hasattr(fh, "train_image_buffer") and fh.pack_rides_on_lookup() and "_train_image_ready" not in self.__dict__:
                    # ... and so does the head's packed f16 weight image, on the lookup's FORWARD launch (nothing between the update
                    # and the head's forward but the lookup: no pack launch, no wait for one packed on another stream)
                    head_dw["pack"] = (self.basis_mat.weight, self.color_net[0].weight, self.color_net[1].weight, self.color_net[2].weight,
                                       fh.train_image_buffer(self))
                sraw, prod = self.ops.vm_encode(x, self._aabb(), *self.sigma_mat, *self.sigma_vec, *self.color_mat, *self.color_vec,
                                                *(() if head_dw is None else (head_dw,)))
                if head_dw is not None:
                    head_dw.pop("pack", None)  # (a lookup that did not take it: the head packs for itself)
                    packed = head_dw.pop("packed", None)
                    if packed is not None:
                        self._train_image_ready = packed
                out = fh.vm_head_train(self, sraw, prod, d, head_dw=head_dw if prod.requires_grad else None)
            elif self.model_type == "hash" and hasattr(fh, "hash_head_train") and not x.requires_grad:
                out = fh.hash_head_train(self, x, d)  # teacher training / hash student
            elif (self.model_type == "mlp" and not torch.is_grad_enabled() and hasattr(fh, "features_head_infer")
                  and self.in_dim == 28 and self.sigma_net[0].weight.shape == (64, 28) and getattr(self.ops, "freq_encode", None) is not None):
                if hasattr(fh, "mlp_head_infer") and fh.mlp_supported(self) and len(self.encoder_nerf_pe.freq_bands) == 10 \
                        and os.environ.get("PVD_MLP_FUSED", "1") != "0":
                    out = fh.mlp_head_infer(self, x, d)  # frozen NeRF-MLP teacher: positional encoding, then trunk + head in one launch
                else:
                    out = fh.features_head_infer(self, self.forward_nerf_mlp(x), d)  # library GEMMs for the trunk, then the fused head
            if out is not None:
                sigma, color, feat = out[:3]
                self.feature_sigma_color = feat
                if self._in_stage1():
                    return None, None
                self.sigma_l = feat[..., 0]
                # a second handle on the same values: the fused backward adds the two gradients
                self.color_l = out[3] if len(out) > 3 else color
                return sigma, color
        if self.model_type == "vm":
            sigma_raw, color_raw = self.vm_features(x)
            sigma_feat = torch.clamp(sigma_raw, -100 if a.enable_edit_plenoxel else a.sigma_clip_min, a.sigma_clip_max)
            color_feat = torch.clamp(color_raw, a.sigma_clip_min, a.sigma_clip_max)
            self.feature_sigma_color = torch.cat([sigma_feat.unsqueeze(-1), color_feat], dim=-1)
            if self._in_stage1():
                return None, None
            self.sigma_l = sigma_feat
            sigma = self.trunc_exp(sigma_feat)
            color = self._color_head(self.encoder_dir(d), color_feat)
            self.color_l = color
            return sigma, color

        if self.model_type == "tensors":
            px = self._plenoxel_ops(x)
            if px is not None:  # lookup + clamp + trunc_exp + SH colour + sigmoid: one kernel each way
                sigma, color, self.sigma_l, _ = px.plenoxel_head(x, d, self._aabb(), self.tensor_volume[0], self.plenoxel_degree,
                                                                 a.sigma_clip_min, a.sigma_clip_max)
                self.feature_sigma_color = None
                self.color_l = color
                return sigma, color
            h = self.compute_plenoxel_fea(self._unit_cube(x))
            sigma = torch.clamp(h[..., 0], -100 if a.enable_edit_plenoxel else a.sigma_clip_min, a.sigma_clip_max)
            self.sigma_l = sigma
            sigma = self.trunc_exp(sigma)
            sh = h[..., 1:].view(-1, 3, self.plenoxel_degree ** 2)
            color = torch.sigmoid((sh * self.encoder_dir(d).unsqueeze(1)).sum(-1))
            self.feature_sigma_color = None
            self.color_l = color
            return sigma, color

        h = self.encoder(x, bound=self.bound) if self.model_type == "hash" else self.forward_nerf_mlp(x)
        for l in range(self.num_layers):
            h = self.linear(h, self.sigma_net[l].weight)
            if l != self.num_layers - 1:
                h = F.relu(h, inplace=True)
        # channel 0 is log-density, clamped; written in place like the reference (network.py:418-420)
        h[..., 0] = torch.clamp(h[..., 0].clone(), a.sigma_clip_min, a.sigma_clip_max)
        self.feature_sigma_color = h
        if self._in_stage1():
            return None, None
        self.sigma_l = h[..., 0]
        sigma = self.trunc_exp(h[..., 0])
        color = self._color_head(self.encoder_dir(d), h[..., 1:])
        self.color_l = color
        return sigma, color

    def supports_device_rows(self):
        """True if forward_rows() exists for this model here: hash and VM models on the HIP operator set under autocast."""
        return (self.model_type in ("hash", "vm") and getattr(self.ops, "fused_head", None) is not None and self.bg_net is None
                and (self.model_type != "vm" or getattr(self.ops, "vm_encode_infer", None) is not None)
                and not (self.model_type == "hash" and not getattr(self.ops.fused_head, "FUSED_LOOKUP", False)))

    @torch.no_grad()
    def forward_rows(self, x, d, rows_dev):
        """forward() of the inference rounds: (sigma, rgb) for the first `rows_dev` (DEVICE int32) rows of x / d; the other
        rows of the outputs are left unwritten.  No host-side knowledge of the row count is needed (renderer._run_rounds_device)."""
        fh = self.ops.fused_head
        if self.model_type == "hash":
            sigma, rgb, _ = fh.hash_head_infer(self, x, d, rows_dev=rows_dev)
        else:
            sraw, prod = self.ops.vm_encode_infer(x, self._aabb(), *self.sigma_mat, *self.sigma_vec, *self.color_mat, *self.color_vec,
                                                  rows_dev=rows_dev)
            sigma, rgb, _ = fh.vm_head_infer(self, sraw, prod, d, rows_dev=rows_dev)
        return sigma, rgb

    def density(self, x):
        """reference: network.py:439-494 (used by update_extra_state)."""
        a = self.args
        if self.model_type == "vm":
            s = torch.clamp(self.vm_features(x)[0], a.sigma_clip_min, a.sigma_clip_max)
            return {"sigma": self.trunc_exp(s)}
        if self.model_type == "tensors":
            px = self._plenoxel_ops(x)
            if px is not None and not torch.is_grad_enabled():
                _, _, _, h0 = px.plenoxel_head(x, torch.zeros_like(x), self._aabb(), self.tensor_volume[0], self.plenoxel_degree,
                                               a.sigma_clip_min, a.sigma_clip_max)
                return {"sigma": torch.exp(h0)}
            h = self.compute_plenoxel_fea(self._unit_cube(x))
            return {"sigma": self.trunc_exp(h[..., 0])}  # the reference's second, unclamped assignment wins (:481)
        if self.model_type == "hash" and not torch.is_grad_enabled() and self._fused_ok(x):
            # occupancy-grid maintenance queries ~1e6 densities per update: grid lookup + MFMA head, two launches
            sigma, _, feat = self.ops.fused_head.hash_head_infer(self, x, torch.zeros_like(x))
            return {"sigma": sigma, "geo_feat": torch.clamp(feat[..., 1:], a.sigma_clip_min, a.sigma_clip_max)}
        h = self.encoder(x, bound=self.bound) if self.model_type == "hash" else self.forward_nerf_mlp(x)
        for l in range(self.num_layers):
            h = self.linear(h, self.sigma_net[l].weight)
            if l != self.num_layers - 1:
                h = F.relu(h, inplace=True)
        h = torch.clamp(h, a.sigma_clip_min, a.sigma_clip_max)
        return {"sigma": self.trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        """Colours for the rows selected by `mask` (zeros elsewhere): the masked query of the fixed-step sampler
        (reference: color, network.py:513-546, which asserts out before doing this).  Models whose density() hands out
        geometry features (hash, mlp) run the colour head on them; the others evaluate the whole model on the rows."""
        out = torch.zeros(x.shape[0], 3, dtype=torch.float32, device=x.device)
        if mask is not None and not bool(mask.any()):
            return out
        sel = mask if mask is not None else torch.ones(x.shape[0], dtype=torch.bool, device=x.device)
        if geo_feat is not None and self.model_type in ("hash", "mlp"):
            rgb = self._color_head(self.encoder_dir(d[sel]), geo_feat[sel])
        else:
            rgb = self(x[sel], d[sel])[1]
        # differentiable scatter of the selected rows (the reference's `rgbs[mask] = h` on a fresh tensor)
        return out.index_put((sel.nonzero(as_tuple=True)[0],), rgb.to(out.dtype))

    def get_params(self, lr, lr2=1e-3):
        """optimizer groups (network.py:646-683)."""
        if self.model_type == "hash":
            return [{"params": self.encoder.parameters(), "lr": lr}, {"params": self.sigma_net.parameters(), "lr": lr},
                    {"params": self.color_net.parameters(), "lr": lr}]
        if self.model_type == "mlp":
            return [{"params": self.sigma_net.parameters(), "lr": lr}, {"params": self.color_net.parameters(), "lr": lr},
                    {"params": self.nerf_mlp.parameters(), "lr": lr}]
        if self.model_type == "vm":
            return [{"params": self.color_net.parameters(), "lr": lr2}, {"params": self.sigma_mat, "lr": lr},
                    {"params": self.sigma_vec, "lr": lr}, {"params": self.color_mat, "lr": lr},
                    {"params": self.color_vec, "lr": lr}, {"params": self.basis_mat.parameters(), "lr": lr2}]
        return [{"params": self.tensor_volume.parameters(), "lr": lr}]
