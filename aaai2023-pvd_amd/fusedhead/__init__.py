"""``fusedhead`` -- the sigma / colour head of the hash and VM models as one MFMA kernel (libpvd_hip.so).

Reference formulation: NeRFNetwork.forward under autocast (distill_mutual/network.py:335-437): a chain of
bias-free half Linear layers, clamp, trunc_exp, SH degree 4, sigmoid.  Only valid under fp16 autocast (the
reference forces fp16 on, main_distill_mutual.py:251-254); callers fall back to the layer-by-layer torch
formulation otherwise."""
import numpy as np
import torch

import pvd_hip

KIND_HASH, KIND_VM = 0, 1


def _outputs(M, dev):
    return (torch.empty(M, dtype=torch.float32, device=dev), torch.empty(M, 3, dtype=torch.float32, device=dev),
            torch.empty(M, 16, dtype=torch.float32, device=dev))


def _w(lin):
    w = lin.weight.detach()
    return w if w.is_contiguous() else w.contiguous()


@torch.no_grad()
def hash_head_infer(model, x, d):
    """(sigma, rgb, feature_sigma_color) of a hash model for positions x, directions d -- no autograd."""
    enc = model.encoder
    M = x.shape[0]
    dev = x.device
    bound = model.bound
    x01 = ((x.float() + bound) / (2 * bound)).contiguous()  # GridEncoder.forward's mapping (grid.py:211)
    emb = enc.embeddings
    cache = getattr(model, "_emb_half_cache", None)
    if cache is None or cache[0] != emb._version or cache[1] != emb.data_ptr():
        cache = (emb._version, emb.data_ptr(), emb.detach().to(torch.float16))  # frozen teacher: cast once, not per step
        model._emb_half_cache = cache
    L = enc.offsets.shape[0] - 1
    C = emb.shape[1]
    assert L == 14 and C == 2 and enc.input_dim == 3, "fused head expects the 14-level, 2-feature hash grid"
    out = torch.empty(L, M, C, dtype=torch.float16, device=dev)
    pvd_hip.grid_encode_forward(x01, cache[2], enc.offsets, out, M, 3, C, L, float(np.log2(enc.per_level_scale)), enc.base_resolution,
                                False, out, enc.gridtype_id, enc.align_corners)
    sigma, rgb, feat = _outputs(M, dev)
    a = model.args
    pvd_hip.head_forward(KIND_HASH, out, None, d.float().contiguous(), M, _w(model.sigma_net[0]), _w(model.sigma_net[1]),
                         _w(model.color_net[0]), _w(model.color_net[1]), _w(model.color_net[2]),
                         a.sigma_clip_min, a.sigma_clip_min, a.sigma_clip_max, sigma, rgb, feat)
    return sigma, rgb, feat


@torch.no_grad()
def vm_head_infer(model, sigma_raw, prod, d):
    M = prod.shape[0]
    sigma, rgb, feat = _outputs(M, prod.device)
    a = model.args
    smin = -100.0 if a.enable_edit_plenoxel else a.sigma_clip_min
    pvd_hip.head_forward(KIND_VM, prod.contiguous(), sigma_raw.float().contiguous(), d.float().contiguous(), M, _w(model.basis_mat), None,
                         _w(model.color_net[0]), _w(model.color_net[1]), _w(model.color_net[2]),
                         smin, a.sigma_clip_min, a.sigma_clip_max, sigma, rgb, feat)
    return sigma, rgb, feat
