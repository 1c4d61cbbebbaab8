"""GPU parity of the fused VM plane x line lookup (vmencoder) against the reference's own
formulation -- twelve F.grid_sample(align_corners=True) calls + products (network.py:216-309) --
evaluated by PyTorch in float32 on the same device and in float64 on the CPU.
Tolerance: fp32 with a different summation order; north_star's bar is 1e-4 on sigma / RGB."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _models(res, seed=0):
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import make_model
    import types
    torch.manual_seed(seed)
    opt = PVDConfig(model_type="vm", resolution0=res)
    dev = torch.device("cuda:0")
    h_ops = hip_ops()
    h_ops.fused_head = None  # this file tests the lookup kernel, not the fused head
    hip = make_model(h_ops, opt, "vm", False, dev)
    ref_ops = hip_ops()
    ref_ops.vm_encode = None  # reference formulation (torch grid_sample) on the same weights
    ref_ops.fused_head = None
    ref = make_model(ref_ops, opt, "vm", False, dev)
    ref.load_state_dict(hip.state_dict())
    return hip, ref


@pytest.mark.parametrize("res", [300, 37])
def test_vm_forward_backward_match_grid_sample_formulation(res):
    hip, ref = _models(res)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    M = 50000
    x = (torch.rand(M, 3, device=dev, generator=g) * 2 - 1)
    x[:6] = torch.tensor([[0, 0, 0], [1, 1, 1], [-1, -1, -1], [1, -1, 0.5], [0.999999, 0.3, -0.2], [-0.5, 1.0, 1.0]], device=dev)
    x[6:40] = 0.0  # padding rows of the marcher: all the same texel
    s_h, c_h = hip.vm_features(x)
    s_r, c_r = ref.vm_features(x)
    assert s_h.dtype == torch.float32 and c_h.shape == (M, 15)
    assert (s_h - s_r).abs().max().item() < 2e-5
    assert (c_h - c_r).abs().max().item() < 2e-5
    # float64 reference on the CPU for an absolute anchor
    ref64 = ref.double().cpu()
    s64, c64 = ref64.vm_features(x[:4000].double().cpu())
    assert (s_h[:4000].cpu().double() - s64).abs().max().item() < 2e-5
    ref.float().to(dev)

    gs = torch.randn(M, device=dev, generator=g)
    gc = torch.randn(M, 15, device=dev, generator=g)
    for m, (s, c) in ((hip, (s_h, c_h)), (ref, (s_r, c_r))):
        m.zero_grad(set_to_none=True)
        ((s * gs).sum() + (c * gc).sum()).backward()
    for name in ("sigma_mat", "sigma_vec", "color_mat", "color_vec"):
        for i in range(3):
            a, b = getattr(hip, name)[i].grad, getattr(ref, name)[i].grad
            assert a.shape == b.shape
            scale = b.abs().max().item()
            assert (a - b).abs().max().item() <= 3e-5 * max(scale, 1.0), (name, i, (a - b).abs().max().item(), scale)
    assert torch.allclose(hip.basis_mat.weight.grad, ref.basis_mat.weight.grad, rtol=1e-3, atol=1e-3)


def test_vm_amp_half_products_and_density():
    hip, ref = _models(64, seed=2)
    dev = torch.device("cuda:0")
    x = torch.rand(20000, 3, device=dev) * 2 - 1
    with torch.autocast("cuda", dtype=torch.float16):
        s_h, c_h = hip.vm_features(x)
        s_r, c_r = ref.vm_features(x)
    assert c_h.dtype == torch.float16 and s_h.dtype == torch.float32
    assert (s_h - s_r).abs().max().item() < 2e-5
    assert (c_h.float() - c_r.float()).abs().max().item() < 4e-3  # half Linear in both
    d_h, d_r = hip.density(x)["sigma"], ref.density(x)["sigma"]
    assert torch.allclose(d_h, d_r, rtol=1e-4, atol=1e-5)


def test_vm_layout_is_channels_last_and_state_dict_compatible():
    hip, ref = _models(32)
    for name in ("sigma_mat", "color_mat"):
        p = getattr(hip, name)[0]
        assert p.shape[0] == 1 and p.permute(0, 2, 3, 1).is_contiguous()
    sd = {k: v.contiguous() for k, v in hip.state_dict().items()}  # a reference-style (channel-major) checkpoint
    hip.load_state_dict(sd)
    assert hip.sigma_mat[0].permute(0, 2, 3, 1).is_contiguous()  # load_state_dict copies in place: layout survives


def test_fused_get_rays_matches_torch_formulation():
    from pvd.ops import hip_ops
    from pvd.scene import BLENDER_INTRINSICS, get_rays, synthetic_poses
    dev = torch.device("cuda:0")
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(0))).to(dev)
    torch.manual_seed(0)
    r = hip_ops().get_rays(poses[3:4], BLENDER_INTRINSICS, 800, 800, 4096)
    ref = get_rays(poses[3:4], BLENDER_INTRINSICS, 800, 800, 4096, inds=r["inds"][0])
    assert torch.equal(r["rays_o"], ref["rays_o"].contiguous())
    assert (r["rays_d"] - ref["rays_d"]).abs().max().item() < 3e-7
    assert torch.allclose(r["rays_d"].norm(dim=-1), torch.ones(1, 4096, device=dev), atol=1e-6)


def test_fused_get_rays_with_an_error_map_draws_inside_the_weighted_cells():
    """--error_map on the HIP operator set (utils.py:357-381): cells by weight without replacement, a pixel inside each cell, the
    rays of exactly those pixels (the draw itself is pinned against the reference on the CPU: tests/test_golden.py)."""
    from pvd.ops import hip_ops
    from pvd.scene import BLENDER_INTRINSICS, get_rays, synthetic_poses
    dev = torch.device("cuda:0")
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(0))).to(dev)
    emap = torch.zeros(128 * 128, device=dev)
    emap[5000:9000] = torch.rand(4000, device=dev) + 0.1  # only these cells may be drawn
    g = torch.Generator(device=dev).manual_seed(4)
    r = hip_ops().get_rays(poses[3:4], BLENDER_INTRINSICS, 800, 800, 2048, error_map=emap, generator=g)
    coarse, inds = r["inds_coarse"][0], r["inds"][0]
    assert coarse.min().item() >= 5000 and coarse.max().item() < 9000 and coarse.unique().numel() == 2048
    px, py = inds // 800, inds % 800
    cx, cy = coarse // 128, coarse % 128
    assert ((px >= (cx * 6.25).long()) & (px <= ((cx + 1) * 6.25).long()) & (py >= (cy * 6.25).long()) & (py <= ((cy + 1) * 6.25).long())).all()
    ref = get_rays(poses[3:4], BLENDER_INTRINSICS, 800, 800, 2048, inds=inds)
    assert torch.equal(r["rays_o"], ref["rays_o"].contiguous()) and (r["rays_d"] - ref["rays_d"]).abs().max().item() < 3e-7


def test_vm_non_cubic_tables_and_incoherent_order():
    """The kernel's per-axis sampling state assumes axis a is always sampled at res[a] (planes and lines agree by
    construction, network.py:199-212); check it with three different resolutions, points outside the box (zero padding,
    generic window path) and a shuffled order (window jumps), against the grid_sample formulation."""
    import torch.nn.functional as F
    import vmencoder
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    res = [24, 31, 45]  # x, y, z
    mat_ids, vec_ids = [[0, 1], [0, 2], [1, 2]], [2, 1, 0]
    tabs = []
    for R in (16, 48):
        mats = [torch.randn(1, R, res[m1], res[m0], device=dev, generator=g) for m0, m1 in mat_ids]
        vecs = [torch.randn(1, R, res[v], 1, device=dev, generator=g) for v in vec_ids]
        tabs.append(([vmencoder.to_channels_last_param(t).requires_grad_(True) for t in mats],
                     [vmencoder.to_channels_last_param(t).requires_grad_(True) for t in vecs]))
    M = 64 * 300 + 11
    n_rays = M // 64 + 1
    o = torch.rand(n_rays, 1, 3, device=dev, generator=g) * 2.2 - 1.1
    d = torch.randn(n_rays, 1, 3, device=dev, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    x = (o + torch.arange(64, device=dev).view(1, 64, 1) * 0.01 * d).reshape(-1, 3)[:M].contiguous()  # some of it outside [-1,1]
    x = torch.cat([x[: M // 2], x[M // 2:][torch.randperm(M - M // 2, device=dev, generator=g)]]).contiguous()
    aabb = (-1.0, -1.0, -1.0, 1.0, 1.0, 1.0)
    (smat, svec), (cmat, cvec) = tabs
    sig_h, prod_h = vmencoder.vm_encode(x, aabb, *smat, *svec, *cmat, *cvec)

    def ref(mats, vecs):
        outs = []
        for i, (m0, m1) in enumerate(mat_ids):
            pc = torch.stack([x[:, m0], x[:, m1]], -1).view(1, -1, 1, 2)
            lc = torch.stack([torch.zeros_like(x[:, 0]), x[:, vec_ids[i]]], -1).view(1, -1, 1, 2)
            pv = F.grid_sample(mats[i], pc, align_corners=True).view(mats[i].shape[1], -1)
            lv = F.grid_sample(vecs[i], lc, align_corners=True).view(vecs[i].shape[1], -1)
            outs.append(pv * lv)
        return outs
    sig_r = torch.cat(ref(smat, svec), 0).sum(0)
    prod_r = torch.cat(ref(cmat, cvec), 0).T
    assert (sig_h - sig_r).abs().max().item() < 5e-5 and (prod_h.float() - prod_r).abs().max().item() < 5e-5
    gs = torch.randn(M, device=dev, generator=g)
    gp = torch.randn(M, 144, device=dev, generator=g)
    params = [*smat, *svec, *cmat, *cvec]
    gr = torch.autograd.grad((sig_r * gs).sum() + (prod_r * gp).sum(), params)
    gh = torch.autograd.grad((sig_h * gs).sum() + (prod_h.float() * gp).sum(), params)
    for a, b in zip(gh, gr):
        assert (a - b).abs().max().item() <= 5e-5 * b.abs().max().item() + 1e-6


@pytest.mark.parametrize("M", [1, 4097, 92928])
def test_head_weight_image_packed_on_the_lookup_launch_is_the_pack_kernels_image(M):
    """pvd_vm_forward_pack_rider (ABI 6): the VM head's f16 weight image written by extra workgroups of the lookup's forward launch --
    the same bits as pvd_head_pack_weights, and a lookup that is the plain launch's, for one row, a ragged and the metric's row count;
    then a training forward + backward of the model with the rider on (default) and off: identical outputs and gradients."""
    import os
    import pvd_hip
    from vmencoder.vm import to_channels_last_param
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    res = [300, 300, 300]
    mats = [to_channels_last_param(torch.randn(1, R, 300, 300, device=dev, generator=g) * 0.1) for R in (16, 16, 16)]
    vecs = [to_channels_last_param(torch.randn(1, R, 300, 1, device=dev, generator=g) * 0.1) for R in (16, 16, 16)]
    cmats = [to_channels_last_param(torch.randn(1, 48, 300, 300, device=dev, generator=g) * 0.1) for _ in range(3)]
    cvecs = [to_channels_last_param(torch.randn(1, 48, 300, 1, device=dev, generator=g) * 0.1) for _ in range(3)]
    tabs = mats + vecs + cmats + cvecs
    x = (torch.rand(M, 3, device=dev, generator=g) * 2 - 1).contiguous()
    aabb = [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]
    Wa1, Wc1, Wc2, Wc3 = (torch.randn(*s, device=dev, generator=g) for s in ((15, 144), (64, 31), (64, 64), (3, 64)))
    want = pvd_hip.head_pack_weights(1, Wa1, None, Wc1, Wc2, Wc3)
    s0, p0 = torch.empty(M, device=dev), torch.empty(M, 144, dtype=torch.float16, device=dev)
    pvd_hip.vm_forward(x, aabb, tabs, res, s0, p0)
    image = torch.full((pvd_hip.head_image_halfs(1),), float("nan"), dtype=torch.float16, device=dev)
    s1, p1 = torch.empty(M, device=dev), torch.empty(M, 144, dtype=torch.float16, device=dev)
    pvd_hip.vm_forward(x, aabb, tabs, res, s1, p1, pack=(Wa1, Wc1, Wc2, Wc3, image))
    assert torch.equal(image.view(torch.int16), want[:image.numel()].view(torch.int16))
    assert torch.equal(s0, s1) and torch.equal(p0.view(torch.int16), p1.view(torch.int16))
    with pytest.raises(pvd_hip.PvdHipError):
        pvd_hip.vm_forward(x[:0], aabb, tabs, res, s1[:0], p1[:0], pack=(Wa1, Wc1, Wc2, Wc3, image))
    if M != 4097:
        return
    # the model's training forward + backward, rider on / off
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import make_model
    outs = []
    for ride in ("1", "0"):
        os.environ["PVD_HEAD_DW_RIDE"] = ride
        try:
            torch.manual_seed(0)
            m = make_model(hip_ops(), PVDConfig(model_type="vm", stage_iters={"stage1": -1, "stage2": -1}), "vm", False, dev).train()
            m.args.global_step = 0
            d = torch.nn.functional.normalize(torch.randn(M, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(9)), dim=-1)
            with torch.autocast("cuda", dtype=torch.float16):
                sigma, rgb = m(x * 0.9, d)
                ((sigma * 1e-3).sum() + rgb.float().sum() + m.feature_sigma_color.sum()).backward()
            assert ("_train_image_buf" in m.__dict__) == (ride == "1")
            outs.append([sigma.detach().clone(), rgb.detach().float().clone()] + [p.grad.detach().float().clone() for p in m.parameters() if p.grad is not None])
        finally:
            os.environ.pop("PVD_HEAD_DW_RIDE", None)
    assert len(outs[0]) == len(outs[1]) > 4
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for a, b in zip(outs[0][2:], outs[1][2:]):
        assert (a - b).abs().max().item() <= 1e-5 * max(b.abs().max().item(), 1e-30)  # (table gradients: atomics' order)
