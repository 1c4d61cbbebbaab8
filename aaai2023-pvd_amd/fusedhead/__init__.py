"""``fusedhead`` -- the sigma / colour head of the hash and VM models as one MFMA kernel (libpvd_hip.so).

Reference formulation: NeRFNetwork.forward under autocast (distill_mutual/network.py:335-437): a chain of
bias-free half Linear layers, clamp, trunc_exp, SH degree 4, sigmoid.  Only valid under fp16 autocast (the
reference forces fp16 on, main_distill_mutual.py:251-254); callers fall back to the layer-by-layer torch
formulation otherwise."""
import os

import numpy as np
import torch

import pvd_hip

KIND_HASH, KIND_VM = 0, 1
FUSED_LOOKUP = True  # frozen hash model: lookup + head in one launch (tests switch it off to compare with the two launches)


def _outputs(M, dev):
    return (torch.empty(M, dtype=torch.float32, device=dev), torch.empty(M, 3, dtype=torch.float32, device=dev),
            torch.empty(M, 16, dtype=torch.float32, device=dev))


def _w(lin):
    w = lin.weight.detach()
    return w if w.is_contiguous() else w.contiguous()


def _cache_key(tensors):
    return pvd_hip.weights_key(tensors)


def _cached_image(model, kind, weights, params):
    """Packed f16 weight image of a model that is not being trained right now (frozen teacher / inference)."""
    key = _cache_key(params)
    cache = getattr(model, "_head_image_cache", None)
    if cache is None or cache[0] != key:
        w = [None if t is None else t.detach() for t in weights]
        cache = (key, pvd_hip.head_pack_weights(kind, *w))
        model._head_image_cache = cache
    return cache[1]


@torch.no_grad()
def hash_head_infer(model, x, d, rows_dev=None):
    """(sigma, rgb, feature_sigma_color) of a hash model for positions x, directions d -- no autograd.
    rows_dev: optional DEVICE int32 row count (inference rounds): only the first min(M, rows_dev) rows are computed."""
    enc = model.encoder
    M = x.shape[0]
    dev = x.device
    bound = model.bound
    xin = x.float().contiguous()  # GridEncoder.forward's mapping (x + bound) / (2 bound) (grid.py:211) happens in the kernel
    emb = enc.embeddings
    cache = getattr(model, "_emb_half_cache", None)
    key = _cache_key([emb])
    if cache is None or cache[0] != key:
        cache = (key, None, emb.detach().to(torch.float16))  # frozen teacher: cast once, not per step
        model._emb_half_cache = cache
    L = enc.offsets.shape[0] - 1
    C = emb.shape[1]
    assert L == 14 and C == 2 and enc.input_dim == 3, "fused head expects the 14-level, 2-feature hash grid"
    sigma, rgb, feat = _outputs(M, dev)
    a = model.args
    ws = [_w(model.sigma_net[0]), _w(model.sigma_net[1]), _w(model.color_net[0]), _w(model.color_net[1]), _w(model.color_net[2])]
    ps = [model.sigma_net[0].weight, model.sigma_net[1].weight, model.color_net[0].weight, model.color_net[1].weight, model.color_net[2].weight]
    image = _cached_image(model, KIND_HASH, ws, ps)
    S = float(np.log2(enc.per_level_scale))
    if FUSED_LOOKUP:  # lookup + head in one launch, no [L,M,C] intermediate (bit-identical outputs)
        pvd_hip.hash_head_forward_fused(xin, float(bound), float(2 * bound), cache[2], enc.offsets, S, enc.base_resolution, enc.gridtype_id,
                                        enc.align_corners, d.float().contiguous(), M, *ws, a.sigma_clip_min, a.sigma_clip_max, sigma, rgb, feat,
                                        image=image, rows_dev=rows_dev, span=model.__dict__.pop("_fused_span", None))
        return sigma, rgb, feat
    assert rows_dev is None, "a device-side row count needs the fused lookup + head launch"
    out = torch.empty(L, M, C, dtype=torch.float16, device=dev)
    pvd_hip.grid_encode_forward_affine(xin, float(bound), float(2 * bound), cache[2], enc.offsets, out, M, 3, C, L, S, enc.base_resolution,
                                       enc.gridtype_id, enc.align_corners)
    pvd_hip.head_forward(KIND_HASH, out, None, d.float().contiguous(), M, *ws, a.sigma_clip_min, a.sigma_clip_min, a.sigma_clip_max,
                         sigma, rgb, feat, image=image)
    return sigma, rgb, feat


@torch.no_grad()
def hash_infer_image(model, rays_o, rays_d, nears, fars, dt_gamma, max_steps):
    """(weights_sum, depth, image) of the eval branch's round loop (renderer.py:450-543) for a frozen hash model, as ONE persistent
    launch (pvd_infer_image_hash): rays [N,3], nears / fars [N]; the accumulators as the loop leaves them (before background compositing)."""
    enc = model.encoder
    dev = rays_o.device
    N = rays_o.shape[0]
    emb = enc.embeddings
    cache = getattr(model, "_emb_half_cache", None)
    key = _cache_key([emb])
    if cache is None or cache[0] != key:
        cache = (key, None, emb.detach().to(torch.float16))
        model._emb_half_cache = cache
    assert enc.offsets.shape[0] - 1 == 14 and emb.shape[1] == 2 and enc.input_dim == 3, "fused head expects the 14-level, 2-feature hash grid"
    a = model.args
    ws = [_w(model.sigma_net[0]), _w(model.sigma_net[1]), _w(model.color_net[0]), _w(model.color_net[1]), _w(model.color_net[2])]
    ps = [model.sigma_net[0].weight, model.sigma_net[1].weight, model.color_net[0].weight, model.color_net[1].weight, model.color_net[2].weight]
    image = _cached_image(model, KIND_HASH, ws, ps)
    f32 = dict(dtype=torch.float32, device=dev)
    weights_sum, depth, img = torch.zeros(N, **f32), torch.zeros(N, **f32), torch.zeros(N, 3, **f32)
    workspace = torch.empty(2 * N + 12, dtype=torch.int32, device=dev)
    model._last_infer_workspace = workspace  # [0] rays queued, [2 N + 2 .. 2 N + 6): local rounds / rows shaded / walk-only rounds / workgroups (tools/bench_render.py)
    pvd_hip.infer_image_hash(rays_o.float().contiguous(), rays_d.float().contiguous(), nears.float().contiguous(), fars.float().contiguous(),
                             model.density_bitfield, float(model.bound), float(dt_gamma), int(max_steps), int(model.cascade), int(model.grid_size),
                             float(model.density_scale), float(model.bound), float(2 * model.bound), cache[2], enc.offsets,
                             float(np.log2(enc.per_level_scale)), enc.base_resolution, enc.gridtype_id, enc.align_corners, *ws,
                             a.sigma_clip_min, a.sigma_clip_max, workspace, weights_sum, depth, img, image=image)
    return weights_sum, depth, img


def vm_infer_image(model, rays_o, rays_d, nears, fars, dt_gamma, max_steps):
    """(weights_sum, depth, image) of the eval branch's round loop (renderer.py:450-543) for a frozen VM model, as ONE persistent launch
    (pvd_infer_image_vm): rays [N,3], nears / fars [N]; the accumulators as the loop leaves them (before background compositing)."""
    from vmencoder.vm import is_channels_last
    dev = rays_o.device
    N = rays_o.shape[0]
    tables = [*model.sigma_mat, *model.sigma_vec, *model.color_mat, *model.color_vec]
    assert all(is_channels_last(t) for t in tables), "VM factors must be stored channels-last"
    res = [tables[0].shape[3], tables[0].shape[2], tables[1].shape[2]]  # mat_0 is [1,R,res[1],res[0]], mat_1 [1,R,res[2],res[0]]
    a = model.args
    smin = -100.0 if a.enable_edit_plenoxel else a.sigma_clip_min
    ws = [_w(model.basis_mat), None, _w(model.color_net[0]), _w(model.color_net[1]), _w(model.color_net[2])]
    image = _cached_image(model, KIND_VM, ws, [model.basis_mat.weight, model.color_net[0].weight, model.color_net[1].weight, model.color_net[2].weight])
    f32 = dict(dtype=torch.float32, device=dev)
    weights_sum, depth, img = torch.zeros(N, **f32), torch.zeros(N, **f32), torch.zeros(N, 3, **f32)
    workspace = torch.empty(2 * N + 12, dtype=torch.int32, device=dev)
    model._last_infer_workspace = workspace
    pvd_hip.infer_image_vm(rays_o.float().contiguous(), rays_d.float().contiguous(), nears.float().contiguous(), fars.float().contiguous(),
                           model.density_bitfield, float(model.bound), float(dt_gamma), int(max_steps), int(model.cascade), int(model.grid_size),
                           float(model.density_scale), model._aabb(), [t.detach() for t in tables], res, ws[0], ws[2], ws[3], ws[4],
                           smin, a.sigma_clip_min, a.sigma_clip_max, workspace, weights_sum, depth, img, image=image)
    return weights_sum, depth, img


def mlp_supported(model):
    """The layer structure pvd_mlp_head_forward_fused implements: 63 -> 256, hidden 256s with one skip concatenation, -> 28."""
    mlp = getattr(model, "nerf_mlp", None)
    if mlp is None or getattr(model, "in_dim_nerf", 0) != 63 or len(mlp) < 3:
        return False
    s = model.skips
    shapes = [tuple(l.weight.shape) for l in mlp]
    want = [(256, 63)] + [(256, 319 if i == s + 1 else 256) for i in range(1, len(mlp) - 1)] + [(28, 256)]
    return shapes == want and 0 <= s and s + 1 <= len(mlp) - 2 and all(l.bias is not None for l in mlp) \
        and tuple(model.sigma_net[0].weight.shape) == (64, 28)


@torch.no_grad()
def mlp_weight_stream(model):
    """The trunk's weights in the order and layout k_mlp_fwd_fused streams them through LDS (cached until the weights change):
    per layer, chunks of 64 output rows, each rows x (K + 8) halfs (input columns padded 63 -> 64 and permuted inside groups of 32, see below) followed by the rows' biases."""
    params = [p for layer in model.nerf_mlp for p in (layer.weight, layer.bias)]
    key = _cache_key(params)
    cache = getattr(model, "_mlp_stream_cache", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    import torch.nn.functional as F
    parts = []
    last = len(model.nerf_mlp) - 1
    for i, layer in enumerate(model.nerf_mlp):
        w, b = layer.weight.detach().to(torch.float16), layer.bias.detach().to(torch.float16)
        if i == 0:
            w = F.pad(w, (0, 1))                                           # [256, 64]
        elif i == model.skips + 1:
            w = torch.cat([F.pad(w[:, :63], (0, 1)), w[:, 63:]], dim=1)    # [256, 64 + 256]
        if i == last:
            w, b = F.pad(w, (0, 0, 0, 4)), F.pad(b, (0, 4))                # 28 -> 32 rows
        # within every group of 32 input columns (a pair of k-steps) store [hi][k-step][4]: the 8 halfs lane (row, hi) feeds to
        # one K = 32 MFMA become one 16-byte LDS read (logical column 32 p + 16 s + 4 hi + j -> position 32 p + 8 hi + 4 s + j)
        n, k = w.shape
        w = w.view(n, k // 32, 2, 4, 4).permute(0, 1, 3, 2, 4).reshape(n, k)
        w = F.pad(w, (0, 8))                                               # 8 halfs of row padding (LDS bank spread, kMlpPad)
        for c in range(0, w.shape[0], 64):
            parts += [w[c:c + 64].reshape(-1), b[c:c + 64]]
    stream = torch.cat(parts).contiguous()
    model._mlp_stream_cache = (key, stream)
    return stream


@torch.no_grad()
def mlp_head_infer(model, x, d):
    """(sigma, rgb, feature_sigma_color) of a frozen `mlp` model: positional encoding (one launch) + trunk and head (one launch)."""
    enc = model.encoder_nerf_pe
    M = x.shape[0]
    pts = pvd_hip.freq_encode(x.reshape(-1, 3).float().contiguous(), enc.freq_bands, enc.include_input, torch.float16, 64)
    sigma, rgb, feat = _outputs(M, x.device)
    a = model.args
    ws = [_w(model.sigma_net[0]), _w(model.sigma_net[1]), _w(model.color_net[0]), _w(model.color_net[1]), _w(model.color_net[2])]
    ps = [model.sigma_net[0].weight, model.sigma_net[1].weight, model.color_net[0].weight, model.color_net[1].weight, model.color_net[2].weight]
    n_before = model.skips
    n_after = len(model.nerf_mlp) - 3 - n_before
    pvd_hip.mlp_head_forward_fused(pts, mlp_weight_stream(model), n_before, n_after, d.float().contiguous(), M, *ws, a.sigma_clip_min,
                                   a.sigma_clip_max, sigma, rgb, feat, image=_cached_image(model, KIND_HASH, ws, ps))
    return sigma, rgb, feat


@torch.no_grad()
def features_head_infer(model, h, d):
    """sigma_net / color_net head of a frozen model on features that are already there -- the `mlp` model, whose 28 features
    come out of the NeRF MLP trunk instead of a hash grid (network.py:413-437): h [M,28] f16 (any row stride) is re-laid
    level-major [14][M][2], the layout pvd_head_forward reads, then one MFMA launch."""
    M = h.shape[0]
    assert h.shape[1] == 28 and model.sigma_net[0].weight.shape[1] == 28
    enc = torch.empty(14, M, 2, dtype=torch.float16, device=h.device)
    enc.copy_(h.reshape(M, 14, 2).permute(1, 0, 2))
    sigma, rgb, feat = _outputs(M, h.device)
    a = model.args
    ws = [_w(model.sigma_net[0]), _w(model.sigma_net[1]), _w(model.color_net[0]), _w(model.color_net[1]), _w(model.color_net[2])]
    ps = [model.sigma_net[0].weight, model.sigma_net[1].weight, model.color_net[0].weight, model.color_net[1].weight, model.color_net[2].weight]
    pvd_hip.head_forward(KIND_HASH, enc, None, d.float().contiguous(), M, *ws, a.sigma_clip_min, a.sigma_clip_min, a.sigma_clip_max,
                         sigma, rgb, feat, image=_cached_image(model, KIND_HASH, ws, ps))
    return sigma, rgb, feat


@torch.no_grad()
def vm_head_infer(model, sigma_raw, prod, d, rows_dev=None):
    M = prod.shape[0]
    sigma, rgb, feat = _outputs(M, prod.device)
    a = model.args
    smin = -100.0 if a.enable_edit_plenoxel else a.sigma_clip_min
    ws = [_w(model.basis_mat), None, _w(model.color_net[0]), _w(model.color_net[1]), _w(model.color_net[2])]
    pvd_hip.head_forward(KIND_VM, prod.contiguous(), sigma_raw.float().contiguous(), d.float().contiguous(), M, *ws,
                         smin, a.sigma_clip_min, a.sigma_clip_max, sigma, rgb, feat,
                         image=_cached_image(model, KIND_VM, ws, [model.basis_mat.weight, model.color_net[0].weight, model.color_net[1].weight,
                                                                  model.color_net[2].weight]), rows_dev=rows_dev)
    return sigma, rgb, feat


class _VMHeadTrain(torch.autograd.Function):
    """(sigma_raw [M], prod [M,144] f16, dirs, basis_mat.weight, color_net.{0,1,2}.weight) ->
    (sigma [M], rgb [M,3], feature_sigma_color [M,16]), all f32; one MFMA kernel each way."""

    @staticmethod
    def forward(ctx, sigma_raw, prod, dirs, Wb, Wc1, Wc2, Wc3, smin, fmin, cmax, image=None, head_dw=None):
        ctx.head_dw = head_dw  # (a dict shared with the lookup's autograd node: see backward)
        M = prod.shape[0]
        sigma_raw, prod, dirs = sigma_raw.float().contiguous(), prod.contiguous(), dirs.float().contiguous()
        sigma, rgb, feat = _outputs(M, prod.device)
        if image is None:  # (else: packed ahead of time by prepack_train_image)
            image = pvd_hip.head_pack_weights(KIND_VM, Wb.detach(), None, Wc1.detach(), Wc2.detach(), Wc3.detach())  # serves both passes
        pvd_hip.head_forward(KIND_VM, prod, sigma_raw, dirs, M, Wb.detach(), None, Wc1.detach(), Wc2.detach(), Wc3.detach(),
                             smin, fmin, cmax, sigma, rgb, feat, image=image)
        ctx.image = image
        ctx.save_for_backward(sigma_raw, prod, dirs, Wb, Wc1, Wc2, Wc3)
        ctx.clips = (smin, fmin, cmax)
        ctx.leaves = (Wb, Wc1, Wc2, Wc3)
        ctx.set_materialize_grads(False)
        # rgb feeds the compositing AND the colour term of the distillation objective: hand it out twice (same storage, two
        # autograd outputs) so that the two gradients arrive separately and the backward kernel adds them while loading,
        # instead of autograd launching an elementwise add in between
        rgb_l = torch.empty(0, dtype=rgb.dtype, device=rgb.device).set_(rgb.untyped_storage(), rgb.storage_offset(), rgb.shape, rgb.stride())
        return sigma, rgb, feat, rgb_l

    @staticmethod
    def backward(ctx, g_sigma, g_rgb, g_feat, g_rgb_l):
        sigma_raw, prod, dirs, Wb, Wc1, Wc2, Wc3 = ctx.saved_tensors
        M = prod.shape[0]
        dev = prod.device
        zeros = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        g_sigma = g_sigma.float().contiguous() if g_sigma is not None else zeros(M)
        if g_rgb is None:
            g_rgb, g_rgb_l = g_rgb_l, None
        g_rgb = g_rgb.float().contiguous() if g_rgb is not None else zeros(M, 3)
        g_rgb_l = g_rgb_l.float().contiguous() if g_rgb_l is not None else None
        g_feat = g_feat.float().contiguous() if g_feat is not None else zeros(M, 16)
        g_sraw = torch.empty(M, dtype=torch.float32, device=dev)
        g_prod = torch.empty(M, 144, dtype=torch.float16, device=dev)
        ws = torch.empty(pvd_hip.head_backward_workspace_floats(KIND_VM, M), dtype=torch.float32, device=dev)
        # accumulate straight into the leaves' gradient buffers when they exist (the trainer's flat bucket)
        direct = all(p.is_leaf and p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32 for p in ctx.leaves)
        grads = [p.grad for p in ctx.leaves] if direct else [torch.zeros_like(p, dtype=torch.float32) for p in ctx.leaves]
        # the reduction of the weight-gradient tiles rides on the lookup's backward launch (which autograd runs next, on this stream)
        # when the lookup's node shares a hand-over dict with this one, needs a gradient itself, and the weight gradients go
        # straight into their final buffers; otherwise it is a launch of its own, here
        defer = ctx.head_dw if (ctx.head_dw is not None and direct and ctx.needs_input_grad[1] and M > 0) else None
        pvd_hip.head_backward(KIND_VM, prod, sigma_raw, dirs, M, Wb.detach(), None, Wc1.detach(), Wc2.detach(), Wc3.detach(), *ctx.clips,
                              g_sigma, g_rgb, g_feat, g_sraw, g_prod, grads[0], None, grads[1], grads[2], grads[3], ws, image=ctx.image,
                              g_rgb2=g_rgb_l, **({"defer_reduce": defer} if defer is not None else {}))
        gw = (None, None, None, None) if direct else tuple(grads)
        return (g_sraw, g_prod, None) + gw + (None, None, None, None, None)


def vm_head_train(model, sigma_raw, prod, d, head_dw=None):
    """-> (sigma, rgb, feature_sigma_color, rgb_l): rgb_l is rgb again, for the colour term of the objective (see above)."""
    a = model.args
    smin = -100.0 if a.enable_edit_plenoxel else a.sigma_clip_min
    wait = model.__dict__.pop("_before_head", None)
    if wait is not None:
        wait()  # the weight image was packed on another stream (prepack_train_image): this stream waits for it here
    return _VMHeadTrain.apply(sigma_raw, prod, d, model.basis_mat.weight, model.color_net[0].weight, model.color_net[1].weight,
                              model.color_net[2].weight, smin, a.sigma_clip_min, a.sigma_clip_max, model.__dict__.pop("_train_image_ready", None),
                              head_dw)


def train_image_buffer(model):
    """the VM model's own buffer for the training head's packed weight image"""
    buf = getattr(model, "_train_image_buf", None)
    if buf is None:
        buf = model._train_image_buf = torch.empty(pvd_hip.head_image_halfs(KIND_VM), dtype=torch.float16, device=model.basis_mat.weight.device)
    return buf


def pack_rides_on_lookup():
    """The training head's weight image is packed by extra workgroups of the VM lookup's forward launch (network.py; round 6).
    PVD_HEAD_DW_RIDE=0 -- the switch of the head's riders on the lookup's launches -- makes it a launch again (tests)."""
    return os.environ.get("PVD_HEAD_DW_RIDE", "1") != "0"


def prepack_train_image(model):
    """Pack NOW, on the caller's stream, the weight image the next vm_head_train forward of `model` will use (the trainer
    issues this next to the marcher, on a parallel branch of the captured step).  The weights must not change in between.
    Nothing to do (False) when the pack rides on the lookup's launch."""
    if getattr(model, "model_type", None) != "vm" or pack_rides_on_lookup():
        return False
    buf = train_image_buffer(model)
    pvd_hip.head_pack_weights(KIND_VM, model.basis_mat.weight.detach(), None, model.color_net[0].weight.detach(),
                              model.color_net[1].weight.detach(), model.color_net[2].weight.detach(), image=buf)
    model._train_image_ready = buf
    return True


def _grid_dims(enc):
    L = enc.offsets.shape[0] - 1
    C = enc.embeddings.shape[1]
    assert L == 14 and C == 2 and enc.input_dim == 3, "fused head expects the 14-level, 2-feature hash grid"
    return L, C, float(np.log2(enc.per_level_scale))


_ZERO_CACHE = {}


def _zeros_cached(dev, *shape):
    """A zero-filled f32 buffer that is only ever READ (an absent upstream gradient handed to a kernel): allocated and filled once per
    shape instead of once per step.  While a graph is being recorded a fresh buffer is made (a cached one must not live in a graph's pool)."""
    if torch.cuda.is_current_stream_capturing():
        hit = _ZERO_CACHE.get((dev, shape))
        return hit if hit is not None else torch.zeros(*shape, dtype=torch.float32, device=dev)
    hit = _ZERO_CACHE.get((dev, shape))
    if hit is None:
        if len(_ZERO_CACHE) > 16:
            _ZERO_CACHE.clear()
        hit = _ZERO_CACHE[(dev, shape)] = torch.zeros(*shape, dtype=torch.float32, device=dev)
    return hit


class _HashHeadTrain(torch.autograd.Function):
    """(x01 [M,3] in [0,1], embeddings [rows,2] f32, dirs, sigma_net.{0,1}.weight, color_net.{0,1,2}.weight) ->
    (sigma, rgb, feature_sigma_color): grid lookup + MFMA head forward, MFMA head + grid scatter backward.
    Same dtype pipeline as the reference under autocast: table cast to half per call (grid.py:49-52), half
    gradient table from the atomics (grid.py:105-123), widened to f32 when it reaches the parameter."""

    @staticmethod
    def forward(ctx, x01, emb, dirs, Ws0, Ws1, Wc1, Wc2, Wc3, offsets, S, H, gridtype, align, smin, cmax, aff=None):
        """aff = (in_add, in_div): x01 holds positions in [-bound, bound] and both grid kernels map them while they read them
        (pvd_grid_encode_forward_affine / _backward_affine: GridEncoder.forward's two elementwise launches and their tensor are gone)."""
        M = x01.shape[0]
        dev = x01.device
        x01, dirs = x01.float().contiguous(), dirs.float().contiguous()
        emb16 = emb.detach().to(torch.float16)
        enc = torch.empty(14, M, 2, dtype=torch.float16, device=dev)
        image = None
        if aff is None:
            pvd_hip.grid_encode_forward(x01, emb16, offsets, enc, M, 3, 2, 14, S, H, False, enc, gridtype, align)
        elif M > 0 and pack_rides_on_lookup():
            # ... and the head's packed weight image is written by extra workgroups of the lookup's launch (no pack launch in between)
            image = torch.empty(pvd_hip.head_image_halfs(KIND_HASH), dtype=torch.float16, device=dev)
            pvd_hip.grid_encode_forward_affine_pack(x01, aff[0], aff[1], emb16, offsets, enc, M, 3, 2, 14, S, H, gridtype, align,
                                                    (Ws0.detach(), Ws1.detach(), Wc1.detach(), Wc2.detach(), Wc3.detach(), image))
        else:
            pvd_hip.grid_encode_forward_affine(x01, aff[0], aff[1], emb16, offsets, enc, M, 3, 2, 14, S, H, gridtype, align)
        ctx.aff = aff
        sigma, rgb, feat = _outputs(M, dev)
        if image is None:
            image = pvd_hip.head_pack_weights(KIND_HASH, Ws0.detach(), Ws1.detach(), Wc1.detach(), Wc2.detach(), Wc3.detach())
        pvd_hip.head_forward(KIND_HASH, enc, None, dirs, M, Ws0.detach(), Ws1.detach(), Wc1.detach(), Wc2.detach(), Wc3.detach(),
                             smin, smin, cmax, sigma, rgb, feat, image=image)
        ctx.image = image
        ctx.save_for_backward(x01, enc, dirs, offsets, Ws0, Ws1, Wc1, Wc2, Wc3)
        ctx.emb = emb
        ctx.grid = (S, H, gridtype, align)
        ctx.clips = (smin, smin, cmax)
        ctx.leaves = (Ws0, Ws1, Wc1, Wc2, Wc3)
        ctx.set_materialize_grads(False)
        return sigma, rgb, feat

    @staticmethod
    def backward(ctx, g_sigma, g_rgb, g_feat):
        x01, enc, dirs, offsets, Ws0, Ws1, Wc1, Wc2, Wc3 = ctx.saved_tensors
        emb = ctx.emb
        M = x01.shape[0]
        dev = x01.device
        zeros = lambda *s: _zeros_cached(dev, *s)  # (teacher training: no gradient reaches the feature rows)
        g_sigma = g_sigma.float().contiguous() if g_sigma is not None else zeros(M)
        g_rgb = g_rgb.float().contiguous() if g_rgb is not None else zeros(M, 3)
        g_feat = g_feat.float().contiguous() if g_feat is not None else None  # (NULL to the kernel: read as zeros)
        g_enc = torch.empty(14, M, 2, dtype=torch.float16, device=dev)
        ws = torch.empty(pvd_hip.head_backward_workspace_floats(KIND_HASH, M), dtype=torch.float32, device=dev)
        direct = all(p.is_leaf and p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32 for p in ctx.leaves)
        grads = [p.grad for p in ctx.leaves] if direct else [torch.zeros_like(p, dtype=torch.float32) for p in ctx.leaves]
        pvd_hip.head_backward(KIND_HASH, enc, None, dirs, M, Ws0.detach(), Ws1.detach(), Wc1.detach(), Wc2.detach(), Wc3.detach(),
                              *ctx.clips, g_sigma, g_rgb, g_feat, None, g_enc, *grads, ws, image=ctx.image)
        g_emb = None
        if ctx.needs_input_grad[1]:
            S, H, gridtype, align = ctx.grid
            g16 = torch.zeros(emb.shape, dtype=torch.float16, device=dev)
            dummy = g16[:1]
            if ctx.aff is None:
                pvd_hip.grid_encode_backward(g_enc, x01, g16, offsets, g16, M, 3, 2, 14, S, H, False, dummy, dummy, gridtype, align)
            else:
                pvd_hip.grid_encode_backward_affine(g_enc, x01, ctx.aff[0], ctx.aff[1], g16, offsets, g16, M, 3, 2, 14, S, H, gridtype, align)
            taker = getattr(emb, "_pvd_half_grad_taker", None)  # FlatAdamW: adds the f16 table inside its update kernel
            if emb.is_leaf and taker is not None and taker(emb, g16):
                pass
            elif emb.is_leaf and emb.grad is not None and emb.grad.dtype == torch.float32:
                emb.grad.add_(g16)  # widen + accumulate in one pass
            else:
                g_emb = g16.float()
        gw = (None,) * 5 if direct else tuple(grads)
        return (None, g_emb, None) + gw + (None,) * 8


def hash_head_train(model, x, d):
    enc = model.encoder
    _, _, S = _grid_dims(enc)
    a = model.args
    bound = model.bound
    # GridEncoder.forward's mapping (grid.py:211), x01 = (x + bound) / (2 bound): inside the two grid kernels, the same two operations
    return _HashHeadTrain.apply(x.float(), enc.embeddings, d, model.sigma_net[0].weight, model.sigma_net[1].weight, model.color_net[0].weight,
                                model.color_net[1].weight, model.color_net[2].weight, enc.offsets, S, enc.base_resolution, enc.gridtype_id,
                                enc.align_corners, a.sigma_clip_min, a.sigma_clip_max, (float(bound), float(2 * bound)))
