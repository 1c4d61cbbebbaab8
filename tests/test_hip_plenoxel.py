"""GPU parity of the fused Plenoxel lookup + SH colour head (plenoxel package, pvd_plenoxel_*) against the
reference's own formulation -- a 3-D F.grid_sample(align_corners=True) over the [1,C,D,H,W] parameter, clamp,
trunc_exp, SH dot product, sigmoid (distill_mutual/network.py:311-322, 383-409) -- evaluated by PyTorch in
float32 on the same device (and float64 on the CPU as an absolute anchor).
Tolerance: fp32 with a different summation order; north_star's bar is 1e-4 on sigma / RGB."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

AABB = (-1.0, -1.0, -1.0, 1.0, 1.0, 1.0)


def _volume(degree, dims, seed=0, scale=0.5):
    import plenoxel
    g = torch.Generator(device="cuda").manual_seed(seed)
    C = 3 * degree * degree + 1
    v = torch.randn(1, C, *dims, device="cuda", generator=g) * scale
    return plenoxel.to_channels_last_3d_param(v)


def _points(M, seed, coherent):
    g = torch.Generator(device="cuda").manual_seed(seed)
    if coherent:  # samples along rays, 0.2 voxel apart at 128^3: what the marcher emits
        n_rays = M // 64
        o = torch.rand(n_rays, 1, 3, device="cuda", generator=g) * 1.6 - 0.8
        d = torch.randn(n_rays, 1, 3, device="cuda", generator=g)
        d = d / d.norm(dim=-1, keepdim=True)
        t = torch.arange(64, device="cuda").view(1, 64, 1) * 3.3829e-3
        x = (o + t * d).reshape(-1, 3)
        dirs = d.expand(n_rays, 64, 3).reshape(-1, 3).contiguous()
    else:
        x = torch.rand(M, 3, device="cuda", generator=g) * 2.4 - 1.2  # some outside: zero padding
        dirs = torch.randn(M, 3, device="cuda", generator=g)
        dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    x = x.contiguous()
    x[:5] = torch.tensor([[0, 0, 0], [1, 1, 1], [-1, -1, -1], [1, -1, 0.5], [0.999999, 0.3, -0.2]], device="cuda")
    x[5:40] = 0.0  # the marcher's padding rows
    return x[:M].contiguous(), dirs[:M].contiguous()


def _ref_features(vol, x):
    return F.grid_sample(vol, x.view(1, 1, -1, 1, 3), align_corners=True).view(-1, x.shape[0]).permute(1, 0)


def _ref_head(vol, x, d, degree, cmin, cmax):
    import shencoder
    from pvd.activation import make_trunc_exp
    h = _ref_features(vol, x)
    sigma_l = torch.clamp(h[..., 0], cmin, cmax)
    sigma = make_trunc_exp("cuda")(sigma_l)
    sh = h[..., 1:].view(-1, 3, degree * degree)
    enc = shencoder.SHEncoder(degree=degree)(d).unsqueeze(1)
    return sigma, torch.sigmoid((sh * enc).sum(-1)), sigma_l, h


@pytest.mark.parametrize("degree,dims,coherent", [(3, (128, 128, 128), True), (3, (128, 128, 128), False), (2, (20, 24, 28), False),
                                                   (1, (9, 7, 5), False), (3, (33, 17, 65), True)])
def test_plenoxel_features_match_grid_sample(degree, dims, coherent):
    import plenoxel
    vol = _volume(degree, dims).requires_grad_(True)
    M = 64 * 700 + 13 if not coherent else 64 * 700
    x, _ = _points(M, 1, coherent)
    f_h = plenoxel.plenoxel_features(x, AABB, vol, degree)
    f_r = _ref_features(vol, x)
    assert f_h.shape == f_r.shape == (M, 3 * degree * degree + 1)
    assert (f_h - f_r).abs().max().item() < 2e-5
    f64 = _ref_features(vol.detach().double().cpu().contiguous(), x[:3000].double().cpu())
    assert (f_h[:3000].detach().cpu().double() - f64).abs().max().item() < 2e-5
    g = torch.randn(f_r.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    (gr,) = torch.autograd.grad((f_r * g).sum(), vol)
    (gh,) = torch.autograd.grad((f_h * g).sum(), vol)
    assert gh.shape == gr.shape and gh.stride() == vol.stride()
    scale = gr.abs().max().item()
    assert (gh - gr).abs().max().item() <= 3e-5 * scale + 1e-6, ((gh - gr).abs().max().item(), scale)
    # conservation: a sample spreads its gradient with the weights of its in-bounds taps
    wsum = torch.ones(M, device="cuda")
    for a, size in enumerate((dims[2], dims[1], dims[0])):  # x -> W, y -> H, z -> D
        pos = (x[:, a] + 1) / 2 * (size - 1)
        i0 = torch.floor(pos)
        w1 = pos - i0
        in0 = ((i0 >= 0) & (i0 < size)).float()
        in1 = ((i0 + 1 >= 0) & (i0 + 1 < size)).float()
        wsum = wsum * (in0 * (1 - w1) + in1 * w1)
    expect = (g.sum(-1) * wsum).sum().item()
    assert abs(gh.sum().item() - expect) <= 1e-4 * g.abs().sum().item() / M ** 0.5 + 0.05, (gh.sum().item(), expect)


@pytest.mark.parametrize("degree,coherent", [(3, True), (3, False), (2, False), (1, True)])
def test_plenoxel_head_forward_backward(degree, coherent):
    import plenoxel
    dims = (64, 48, 56)
    vol = _volume(degree, dims, seed=3, scale=2.0).requires_grad_(True)  # scale 2: the clamp at [-2, 7] is active
    M = 64 * 500
    x, d = _points(M, 4, coherent)
    cmin, cmax = -2.0, 7.0
    s_h, c_h, sl_h, h0 = plenoxel.plenoxel_head(x, d, AABB, vol, degree, cmin, cmax)
    s_r, c_r, sl_r, h_r = _ref_head(vol, x, d, degree, cmin, cmax)
    assert (sl_h - sl_r).abs().max().item() < 2e-5
    assert (h0 - h_r[..., 0]).abs().max().item() < 2e-5
    assert ((s_h - s_r).abs() / (s_r.abs() + 1e-6)).max().item() < 5e-5
    assert (c_h - c_r).abs().max().item() < 2e-5
    frac_clamped = ((h_r[..., 0] < cmin) | (h_r[..., 0] > cmax)).float().mean().item()
    assert frac_clamped > 0.01  # the mask path is exercised
    g = torch.Generator(device="cuda").manual_seed(5)
    ws, wc, wl = torch.randn(M, device="cuda", generator=g) * 0.01, torch.randn(M, 3, device="cuda", generator=g), \
        torch.randn(M, device="cuda", generator=g)
    (gr,) = torch.autograd.grad((s_r * ws).sum() + (c_r * wc).sum() + (sl_r * wl).sum(), vol)
    (gh,) = torch.autograd.grad((s_h * ws).sum() + (c_h * wc).sum() + (sl_h * wl).sum(), vol)
    scale = gr.abs().max().item()
    assert torch.isfinite(gh).all()
    # samples whose h0 sits within rounding of a clip bound may take the other side of the mask; none here by construction
    assert (gh - gr).abs().max().item() <= 1e-4 * scale, ((gh - gr).abs().max().item(), scale)
    assert (gh - gr).abs().mean().item() <= 1e-5 * gr.abs().mean().item() + 1e-9


def test_plenoxel_backward_accumulates_into_existing_grad():
    import plenoxel
    vol = _volume(3, (32, 32, 32), seed=6).requires_grad_(True)
    x, d = _points(64 * 100, 7, True)
    outs = []
    for pre in (0.0, 1.0):
        vol.grad = torch.full_like(vol, pre)  # full_like keeps the channels-last strides -> direct accumulation
        assert vol.grad.stride() == vol.stride()
        s, c, sl, _ = plenoxel.plenoxel_head(x, d, AABB, vol, 3, -2.0, 7.0)
        (c.sum() + sl.sum()).backward()
        outs.append(vol.grad.clone())
    assert torch.allclose(outs[1] - 1.0, outs[0], rtol=1e-4, atol=1e-4)
    assert outs[0].abs().max().item() > 0


def test_plenoxel_model_fused_vs_torch_formulation():
    """NeRFNetwork('tensors'): fused forward / density / backward vs the same model on the torch path."""
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import make_model
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    opt = PVDConfig(model_type="tensors", plenoxel_res="[48,48,48]")
    hip = make_model(hip_ops(), opt, "tensors", False, dev)
    ref_ops = hip_ops()
    ref_ops.plenoxel = None
    ref = make_model(ref_ops, opt, "tensors", False, dev)
    ref.load_state_dict(hip.state_dict())
    assert hip.tensor_volume[0].shape == (1, 28, 48, 48, 48)
    x, d = _points(64 * 200, 8, True)
    with torch.autocast("cuda", dtype=torch.float16):
        s_h, c_h = hip(x, d)
        s_r, c_r = ref(x, d)
    assert s_h.dtype == torch.float32 and c_h.dtype == torch.float32
    assert ((s_h - s_r).abs() / (s_r.abs() + 1e-6)).max().item() < 5e-5 and (c_h - c_r).abs().max().item() < 2e-5
    assert (hip.sigma_l - ref.sigma_l).abs().max().item() < 2e-5
    with torch.no_grad():
        assert ((hip.density(x)["sigma"] - ref.density(x)["sigma"]).abs() / (ref.density(x)["sigma"] + 1e-6)).max().item() < 5e-5
    for m, (s, c) in ((hip, (s_h, c_h)), (ref, (s_r, c_r))):
        m.zero_grad(set_to_none=True)
        (s.sum() * 1e-2 + c.sum()).backward()
    a, b = hip.tensor_volume[0].grad, ref.tensor_volume[0].grad
    assert (a - b).abs().max().item() <= 1e-4 * b.abs().max().item()
