"""BASELINE.json configs[0]: `main_just_train_tea.py chair --model_type mlp`, 64x64 crops (4096 rays), WITHOUT cuda_ray --
the fixed-step pure-torch sampler NeRFRenderer.run (distill_mutual/renderer.py:139-317, defaults num_steps=512,
upsample_steps=0: main_just_train_tea.py:45-56).  A plumbing configuration: no occupancy grid, no native marcher; in the
reference its colour query asserts out (network.py:515-516), here it is the masked query.  CPU only (oracle SH encoder)."""
import numpy as np
import torch


def _teacher(num_rays=256, **kw):
    from oracle_ops import oracle_ops
    from pvd.config import PVDConfig
    from pvd.trainer import TeacherTrainer
    from pvd.workload import make_model
    opt = PVDConfig(model_type="mlp", teacher_type="mlp", cuda_ray=False, fp16=False, num_rays=num_rays, iters=100,
                    stage_iters={"stage1": -1, "stage2": -1}, **kw)
    torch.manual_seed(0)
    m = make_model(oracle_ops(), opt, "mlp", True, torch.device("cpu"), teacher_variant=True)
    assert not m.cuda_ray and not hasattr(m, "density_bitfield")
    return m, TeacherTrainer(opt, m, "cpu", fp16=False), opt


def _crop_batch(n_side, seed=0):
    """A square crop of n_side x n_side pixels of one 800x800 view (the reference's --num_rays 4096 = 64x64 patch)."""
    from pvd.scene import BLENDER_INTRINSICS, ChairScene, get_rays, synthetic_poses
    pose = torch.from_numpy(synthetic_poses(np.random.RandomState(seed))[5:6])
    jj, ii = torch.meshgrid(torch.arange(n_side) + 400 - n_side // 2, torch.arange(n_side) + 400 - n_side // 2, indexing="ij")
    inds = (jj * 800 + ii).reshape(-1)
    r = get_rays(pose, BLENDER_INTRINSICS, 800, 800, inds.numel(), inds=inds)
    return r["rays_o"], r["rays_d"], ChairScene()


def test_torch_near_far_is_the_kernels_slab_test():
    import oracle
    from pvd.renderer import near_far_from_aabb_torch
    rs = np.random.RandomState(2)
    o = rs.uniform(-3, 3, size=(5000, 3)).astype(np.float32)
    d = rs.standard_normal((5000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:40, 0] = 0.0   # axis-parallel rays: 1/0 = inf goes through the same min / max
    d[40:80, 2] = -0.0
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n_o, f_o = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    n_t, f_t = near_far_from_aabb_torch(torch.from_numpy(o), torch.from_numpy(d), torch.from_numpy(aabb), 0.2)
    assert np.array_equal(n_o, n_t.numpy()) and np.array_equal(f_o, f_t.numpy())
    assert (n_o == np.finfo(np.float32).max).sum() > 100  # misses are part of the sample


def test_sample_pdf_inverts_a_known_cdf():
    from pvd.renderer import sample_pdf
    bins = torch.linspace(2.0, 6.0, 5).repeat(3, 1)                     # 4 bins of width 1
    w = torch.tensor([[1.0, 0.0, 0.0, 3.0], [1, 1, 1, 1], [0, 2, 2, 0]])
    z = sample_pdf(bins, w, 4000, det=True)
    frac_last = ((z[0] >= 5.0).float().mean().item(), (z[1] >= 5.0).float().mean().item())
    assert abs(frac_last[0] - 0.75) < 2e-3 and abs(frac_last[1] - 0.25) < 2e-3
    assert z[2].min().item() >= 3.0 - 1e-3 and z[2].max().item() <= 5.0 + 1e-3
    assert (z[:, 1:] >= z[:, :-1]).all()  # deterministic quantiles come out sorted


def test_fixed_step_render_composites_like_the_definition():
    m, tr, opt = _teacher()
    rays_o, rays_d, scene = _crop_batch(8)
    m.eval()
    with torch.no_grad():
        out = m.render(rays_o, rays_d, staged=True, max_ray_batch=24, num_steps=48, upsample_steps=0, bg_color=1, perturb=False)
        whole = m.render(rays_o, rays_d, staged=False, num_steps=48, upsample_steps=0, bg_color=1, perturb=False)
    assert out["image"].shape == (1, 64, 3) and out["depth"].shape == (1, 64)
    assert torch.allclose(out["image"], whole["image"], atol=1e-6)  # ray batching does not change the picture
    assert (out["depth"] >= 0).all() and (out["depth"] <= 1).all() and torch.isfinite(out["image"]).all()
    # a second, independent evaluation of the same integral: per-ray python loop over the steps
    from pvd.renderer import near_far_from_aabb_torch
    o, d = rays_o.view(-1, 3), rays_d.view(-1, 3)
    nears, fars = near_far_from_aabb_torch(o, d, m.aabb_infer, m.min_near)
    img = torch.zeros(64, 3)
    with torch.no_grad():
        for n in range(0, 64, 9):
            z = nears[n] + (fars[n] - nears[n]) * torch.linspace(0, 1, 48)
            x = torch.min(torch.max(o[n] + d[n] * z[:, None], m.aabb_infer[:3]), m.aabb_infer[3:])
            sigma, rgb = m(x, d[n].expand(48, 3))
            T, acc = 1.0, torch.zeros(3)
            for k in range(48):
                dt = (z[k + 1] - z[k]) if k < 47 else (fars[n] - nears[n]) / 48
                a = 1 - torch.exp(-dt * sigma[k])
                if a * T > 1e-4:
                    acc = acc + a * T * rgb[k]
                T = T * (1 - a + 1e-15)
            wsum = 1 - T  # telescoping, up to the 1e-15 terms
            img[n] = acc + (1 - wsum) * 1.0
            assert torch.allclose(whole["image"][0, n], img[n], atol=2e-5), n


def test_mlp_teacher_trains_through_the_fixed_step_sampler():
    """A few optimisation steps of the mlp teacher on one 16x16 crop (the real configuration -- 64x64 crop, 512 steps, 8 x 256
    MLP -- is timed by tools/bench_cpu_plumbing.py; its number is BASELINE.md's first row)."""
    m, tr, opt = _teacher(num_steps=64, upsample_steps=16)
    rays_o, rays_d, scene = _crop_batch(16)
    # ground truth: the analytic scene through the same sampler (dense steps, no network)
    from pvd.renderer import near_far_from_aabb_torch
    o, d = rays_o.view(-1, 3), rays_d.view(-1, 3)
    nears, fars = near_far_from_aabb_torch(o, d, m.aabb_train, m.min_near)
    z = nears[:, None] + (fars - nears)[:, None] * torch.linspace(0, 1, 256)
    x = o[:, None] + d[:, None] * z[..., None]
    sig, col = scene.sigma(x), scene.color(x, d[:, None].expand_as(x))
    a = 1 - torch.exp(-sig * ((fars - nears) / 256)[:, None])
    w = a * torch.cumprod(torch.cat([torch.ones_like(a[:, :1]), 1 - a], -1), -1)[:, :-1]
    bg = torch.rand(1, 256, 3, generator=torch.Generator().manual_seed(1))
    gt = ((w[..., None] * col).sum(1) + (1 - w.sum(1))[:, None] * bg[0]).view(1, 256, 3)
    losses = []
    for _ in range(12):
        loss, pred = tr.train_step(rays_o, rays_d, gt, bg)
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < 0.6 * losses[0], losses
    assert pred.shape == (1, 256, 3)
    # every parameter group received a gradient through the masked colour query and the density path
    for n, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    assert any(p.grad.abs().max() > 0 for n, p in m.named_parameters() if n.startswith("color_net"))
    assert any(p.grad.abs().max() > 0 for n, p in m.named_parameters() if n.startswith("nerf_mlp"))
