"""GPU parity of the fused host-side pieces: compositing epilogue and the stage-3 distillation objective,
each against its reference formulation in plain PyTorch ops on the same device."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _samples(N=2048, seed=0):
    import raymarching
    from pvd.scene import BLENDER_INTRINSICS, ChairScene, get_rays, packbits_torch, synthetic_poses
    dev = torch.device("cuda:0")
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(seed))).to(dev)
    r = get_rays(poses[4:5], BLENDER_INTRINSICS, 800, 800, N, generator=torch.Generator(device=dev).manual_seed(seed))
    bits = packbits_torch(ChairScene().density_grid(128, 1.0, 1, device=dev), 10.0)
    o, d = r["rays_o"].reshape(-1, 3).contiguous(), r["rays_d"].reshape(-1, 3).contiguous()
    nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.2)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, None, -1, True, 128, True)
    return xyzs, deltas, rays, nears, fars


@pytest.mark.parametrize("bg_kind", ["tensor", "scalar"])
def test_composite_bg_matches_composition(bg_kind):
    import raymarching
    dev = torch.device("cuda:0")
    xyzs, deltas, rays, nears, fars = _samples()
    M, N = xyzs.shape[0], rays.shape[0]
    g = torch.Generator(device=dev).manual_seed(1)
    sig0 = torch.exp(torch.rand(M, device=dev, generator=g) * 6 - 2)
    rgb0 = torch.rand(M, 3, device=dev, generator=g)
    bg = torch.rand(1, N, 3, device=dev, generator=g) if bg_kind == "tensor" else 1
    w_img = torch.randn(N, 3, device=dev, generator=g)
    w_ws = torch.randn(N, device=dev, generator=g)
    res = []
    for fused in (True, False):
        sig, rgb = sig0.clone().requires_grad_(True), rgb0.clone().requires_grad_(True)
        if fused:
            ws, depth, img = raymarching.composite_rays_train_bg(sig, rgb, deltas, rays, bg, nears, fars, 1e-6)
        else:
            ws, depth, img = raymarching.composite_rays_train(sig, rgb, deltas, rays)
            img = img + (1 - ws).unsqueeze(-1) * (bg.reshape(-1, 3) if torch.is_tensor(bg) else bg)
            depth = torch.clamp(depth - nears, min=0) / (fars - nears + 1e-6)
        ((img * w_img).sum() + (ws * w_ws).sum()).backward()
        res.append((ws.detach(), depth.detach(), img.detach(), sig.grad, rgb.grad))
    a, b = res
    assert torch.equal(a[0], b[0])
    assert torch.allclose(a[1], b[1], atol=1e-6) and torch.allclose(a[2], b[2], atol=1e-6)
    assert torch.allclose(a[4], b[4], atol=1e-6)
    assert (a[3] - b[3]).abs().max() <= 2e-5 * b[3].abs().max()


def test_fused_distill_loss_matches_torch_norms():
    from pvd.losses import distill_loss_normL2
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(2)
    N, M = 4096, 16 * 5000 + 128
    img_t = torch.rand(1, N, 3, device=dev, generator=g)
    fea_t = torch.randn(M, 16, device=dev, generator=g)
    col_t = torch.rand(M, 3, device=dev, generator=g)
    rates = torch.tensor([1.0, 0.002, 0.003, 0.004], device=dev)
    up = 65536.0  # GradScaler-style upstream factor
    res = []
    for fused in (True, False):
        img_s = (img_t + 0.1 * torch.randn(1, N, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(3))).requires_grad_(True)
        fea_s = (fea_t + 0.2 * torch.randn(M, 16, device=dev, generator=torch.Generator(device=dev).manual_seed(4))).requires_grad_(True)
        col_s = (col_t + 0.05 * torch.randn(M, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(5))).requires_grad_(True)
        if fused:
            loss, norms = distill_loss_normL2(img_s, img_t, fea_s, fea_t, col_s, col_t, rates, None)
        else:
            norms = torch.stack([torch.norm(img_t - img_s), torch.norm(fea_s - fea_t), torch.norm(fea_s[:, 0] - fea_t[:, 0]), torch.norm(col_s - col_t)])
            loss = (norms * rates).sum()
        (loss * up).backward()
        res.append((loss.detach(), norms.detach(), img_s.grad, fea_s.grad, col_s.grad))
    a, b = res
    assert torch.allclose(a[0], b[0], rtol=1e-5) and torch.allclose(a[1], b[1], rtol=1e-5)
    for x, y in zip(a[2:], b[2:]):
        assert (x - y).abs().max() <= 1e-5 * y.abs().max()
    # identical inputs: zero loss and zero (not NaN) gradients, like torch.norm's subgradient
    z = fea_t.clone().requires_grad_(True)
    loss, _ = distill_loss_normL2(img_t.clone().requires_grad_(True), img_t, z, fea_t, col_t.clone().requires_grad_(True), col_t, rates, None)
    loss.backward()
    assert float(loss) == 0.0 and torch.count_nonzero(z.grad) == 0


def test_fused_objective_without_a_feature_vector_matches_torch_norms():
    """fea_width = 1 (the Plenoxel student has no feature_sigma_color): the rows hold sigma_l alone -- rgb + sigma + colour terms,
    no feature term, as utils.py:1109-1176 does when the model has no features."""
    from pvd.losses import distill_loss_normL2
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    N, M = 4096, 91003
    img_t, sig_t, col_t = torch.rand(1, N, 3, device=dev, generator=g), torch.randn(M, device=dev, generator=g) * 3, torch.rand(M, 3, device=dev, generator=g)
    rates = torch.tensor([1.0, 0.002, 0.003, 0.004], device=dev)
    res = []
    for fused in (True, False):
        img_s = (img_t + 0.1 * torch.randn(1, N, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(3))).requires_grad_(True)
        sig_s = (sig_t + 0.2 * torch.randn(M, device=dev, generator=torch.Generator(device=dev).manual_seed(4))).requires_grad_(True)
        col_s = (col_t + 0.05 * torch.randn(M, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(5))).requires_grad_(True)
        if fused:
            loss, norms = distill_loss_normL2(img_s, img_t, sig_s.unsqueeze(-1), sig_t.unsqueeze(-1), col_s, col_t, rates.clone(), None, fea_decay=0.995)
            assert float(norms[1]) == 0.0
        else:
            norms = torch.stack([torch.norm(img_t - img_s), torch.zeros((), device=dev), torch.norm(sig_s - sig_t), torch.norm(col_s - col_t)])
            loss = (norms * rates).sum()
        (loss * 1024.0).backward()
        res.append((loss.detach(), norms.detach(), img_s.grad, sig_s.grad, col_s.grad))
    a, b = res
    assert torch.allclose(a[0], b[0], rtol=1e-5) and torch.allclose(a[1], b[1], rtol=1e-5)
    for x, y in zip(a[2:], b[2:]):
        assert x.shape == y.shape and (x - y).abs().max() <= 1e-5 * y.abs().max()


def test_objective_finished_inside_the_backward_launch_is_bit_identical():
    """defer=True (pvd_distill_loss_backward: loss / norms / coefficients finished by every workgroup of the backward launch)
    vs the three-launch form: loss, norms, the decayed feature rate and all three gradients bit for bit, with a parameter-only
    extra term and the 0.995 decay, over two consecutive steps (the rate carries over)."""
    from pvd.losses import distill_loss_normL2
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(12)
    N, M = 4096, 92928
    img_t = torch.rand(1, N, 3, device=dev, generator=g)
    fea_t = torch.randn(M, 16, device=dev, generator=g)
    col_t = torch.rand(M, 3, device=dev, generator=g)
    extra = torch.rand(1024, device=dev, generator=g) * 1e-3
    up = torch.tensor(65536.0, device=dev)
    res = []
    for defer in (False, True):
        rates = torch.tensor([1.0, 0.002, 0.003, 0.004], device=dev)
        out = []
        for step in range(2):
            img_s = (img_t + 0.1 * torch.randn(1, N, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(3 + step))).requires_grad_(True)
            fea_s = (fea_t + 0.2 * torch.randn(M, 16, device=dev, generator=torch.Generator(device=dev).manual_seed(4 + step))).requires_grad_(True)
            col_s = (col_t + 0.05 * torch.randn(M, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(5 + step))).requires_grad_(True)
            loss, norms = distill_loss_normL2(img_s, img_t, fea_s, fea_t, col_s, col_t, rates, None, fea_decay=0.995, extra=extra, defer=defer)
            loss.backward(gradient=up)
            out.append((loss.detach().clone(), norms.clone(), rates.clone(), img_s.grad, fea_s.grad, col_s.grad))
        res.append(out)
    for a, b in zip(*res):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert float(res[1][1][2][1]) == float(torch.tensor(0.002) * 0.995 * 0.995)
    # without autograd (an evaluation pass) defer is ignored: the value is there when the call returns
    with torch.no_grad():
        rates = torch.tensor([1.0, 0.002, 0.003, 0.004], device=dev)
        loss, _ = distill_loss_normL2(img_t * 0.9, img_t, fea_t * 1.1, fea_t, col_t, col_t, rates, None, defer=True)
        ref = 1.0 * torch.norm(img_t * 0.1) + 0.002 * torch.norm(fea_t * 0.1) + 0.003 * torch.norm(fea_t[:, 0] * 0.1)
        assert abs(float(loss) - float(ref)) <= 1e-4 * float(ref)


def test_flat_adamw_matches_torch_fused_adamw():
    """Same parameters / gradients through torch.optim.AdamW(fused) and FlatAdamW, with GradScaler-style
    grad_scale, a skipped (found_inf) step and two learning-rate groups, over several steps."""
    from pvd.flat_adamw import FlatAdamW
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    shapes = [(1, 16, 30, 30), (64, 31), (3, 64), (1, 48, 30, 1), (15, 144), (7,)]
    def make():
        ps = []
        gg = torch.Generator(device=dev).manual_seed(1)
        for i, s in enumerate(shapes):
            t = torch.randn(*s, device=dev, generator=gg)
            if len(s) == 4:
                t = t.contiguous(memory_format=torch.channels_last) if i == 0 else t
            ps.append(torch.nn.Parameter(t))
        return ps
    pa, pb = make(), make()
    groups = lambda ps: [{"params": ps[:3], "lr": 1e-2}, {"params": ps[3:], "lr": 1e-3}]
    ref = torch.optim.AdamW(groups(pa), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
    mine = FlatAdamW(groups(pb), betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-2)
    for a, b in zip(pa, pb):
        assert torch.equal(a, b) and a.stride() == b.stride()
    scale = torch.tensor(1024.0, device=dev)
    for it in range(6):
        inf = torch.tensor(1.0 if it == 3 else 0.0, device=dev)
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, device=dev, generator=g) * 1024.0
            a.grad = gr.clone().contiguous(memory_format=torch.channels_last) if a.dim() == 4 and a.stride() != a.contiguous().stride() else gr.clone()
            mine.zero_grad() if False else None
            b.grad.copy_(gr)
        ref.grad_scale, ref.found_inf = scale, inf
        mine.grad_scale, mine.found_inf = scale, inf
        ref.step(); mine.step()
        if it == 4:  # schedulers fill tensor lrs in place
            for grp in mine.param_groups:
                grp["lr"].mul_(0.5)
            for grp in ref.param_groups:
                grp["lr"] = grp["lr"] * 0.5 if torch.is_tensor(grp["lr"]) else grp["lr"] * 0.5
    assert float(mine.step_count) == 5.0  # one step skipped
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (a - b).abs().max()


def test_flat_adamw_device_schedule_l1_and_finite_check():
    """FlatAdamW with the cosine / exponential schedule evaluated inside the update kernel and the L1
    regulariser folded into it == torch AdamW + torch LR scheduler + autograd of weight * sum mean|t|."""
    import math
    import pvd_hip
    from pvd.flat_adamw import DeviceSchedule, FlatAdamW, FlatGradScaler
    dev = torch.device("cuda:0")
    for kind in ("cosine", "exp"):
        def make():
            gg = torch.Generator(device=dev).manual_seed(3)
            return [torch.nn.Parameter(torch.randn(*s, device=dev, generator=gg)) for s in [(64, 32), (1, 16, 24, 24), (1, 16, 24, 1), (3, 64)]]
        pa, pb = make(), make()
        groups = lambda ps: [{"params": ps[:1], "lr": 1e-2}, {"params": ps[1:3], "lr": 2e-2}, {"params": ps[3:], "lr": 1e-3}]
        ga = groups(pa)
        for grp in ga:
            grp["lr"] = torch.tensor(grp["lr"], device=dev)
        ref = torch.optim.AdamW(ga, betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
        mine = FlatAdamW(groups(pb), betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-2)
        T, w = 40, 3e-2
        if kind == "cosine":
            sched_ref = torch.optim.lr_scheduler.CosineAnnealingLR(ref, T_max=T, eta_min=5e-5)
            sched = DeviceSchedule(mine, "cosine", T, 5e-5)
        else:
            sched_ref = torch.optim.lr_scheduler.LambdaLR(ref, lambda it: 0.1 ** min(it / T, 1))
            sched = DeviceSchedule(mine, "exp", T, 0.1)
        mine.set_l1([pb[1], pb[2]], w)
        g = torch.Generator(device=dev).manual_seed(4)
        for it in range(50):
            l1_ref = w * (pa[1].abs().mean() + pa[2].abs().mean())
            assert abs(float(mine.l1_value()) - float(l1_ref)) <= 1e-5 * abs(float(l1_ref))
            # the partial sums kept current by the update kernel itself (one entry per workgroup) describe the same value
            assert abs(float(mine.l1_partials().sum()) - float(l1_ref)) <= 2e-5 * abs(float(l1_ref))
            for a, b in zip(pa, pb):
                gr = torch.randn(a.shape, device=dev, generator=g)
                a.grad = gr.clone()
                b.grad.copy_(gr)
            for a, gl in zip((pa[1], pa[2]), torch.autograd.grad(l1_ref, (pa[1], pa[2]))):
                a.grad += gl
            ref.step(); mine.step()
            sched_ref.step(); sched.step()
            lr_ref = [float(grp["lr"]) for grp in ref.param_groups]
            # the kernel wrote the rate it used for THIS step; torch's scheduler already holds the next one
            assert it == 0 or all(abs(a - b) <= 2e-6 * b for a, b in zip(prev_lr_mine_next, mine.lr_dev.tolist()))
            prev_lr_mine_next = lr_ref
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=1e-4, atol=2e-6), (kind, (a - b).abs().max())

    # finite check: read-only, sets (never clears) the flag
    flag = torch.zeros(1, device=dev)
    buf = torch.randn(4 * 1000, device=dev)
    pvd_hip.check_finite(buf, flag)
    assert float(flag) == 0.0
    for bad in (float("inf"), float("-inf"), float("nan")):
        flag.zero_()
        b2 = buf.clone(); b2[1237] = bad
        pvd_hip.check_finite(b2, flag)
        assert float(flag) == 1.0
        pvd_hip.check_finite(buf, flag)
        assert float(flag) == 1.0
    # FlatGradScaler: a scaled step with an inf gradient is skipped and the scale backs off
    ps = [torch.nn.Parameter(torch.ones(8, device=dev))]
    opt = FlatAdamW([{"params": ps, "lr": 1e-2}], betas=(0.9, 0.99), eps=1e-15, weight_decay=0.0)
    scaler = FlatGradScaler("cuda", init_scale=1024.0)
    for it, gval in enumerate((1.0, float("inf"), 1.0)):
        opt.zero_grad()
        ps[0].grad.fill_(gval * 1024.0)
        before = ps[0].detach().clone()
        scaler.scale(torch.zeros((), device=dev))  # lazily creates the device-side scale, as scale(loss) does in a step
        scaler.step(opt)
        scaler.update()
        changed = not torch.equal(before, ps[0].detach())
        assert changed == (gval == 1.0)
    assert float(scaler.get_scale()) == 512.0 and float(opt.step_count) == 2.0


def test_make_ray_batch_matches_get_rays_and_near_far():
    """pvd_make_ray_batch == get_rays on the pixel ids it drew + near_far_from_aabb, cycles through the poses, draws
    different pixels every call and uniform backgrounds."""
    import pvd_hip
    import raymarching
    from pvd.scene import BLENDER_INTRINSICS, synthetic_poses
    dev = torch.device("cuda:0")
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(0))).to(dev).float().contiguous()
    P, N, H, W = poses.shape[0], 4096, 800, 800
    state = torch.zeros(3, dtype=torch.int64, device=dev)
    aabb = torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev)
    fx, fy, cx, cy = BLENDER_INTRINSICS
    seen = []
    for it in range(P + 2):
        f = lambda *s: torch.empty(*s, device=dev)
        inds = torch.empty(N, dtype=torch.int64, device=dev)
        ro, rd, bg, nears, fars = f(N, 3), f(N, 3), f(N, 3), f(N), f(N)
        pvd_hip.make_ray_batch(poses, state, 1234, fx, fy, cx, cy, H, W, N, aabb, 0.2, inds, ro, rd, bg, nears, fars)
        assert state.tolist() == [(it + 1) % P, it + 1, 0]
        assert int(inds.min()) >= 0 and int(inds.max()) < H * W
        ro2, rd2 = torch.empty(N, 3, device=dev), torch.empty(N, 3, device=dev)
        pvd_hip.get_rays(poses[it % P].contiguous(), fx, fy, cx, cy, inds, W, N, ro2, rd2)
        assert torch.equal(ro, ro2) and torch.equal(rd, rd2)
        n2, f2 = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
        assert torch.equal(nears, n2) and torch.equal(fars, f2)
        assert 0.0 <= float(bg.min()) and float(bg.max()) < 1.0 and abs(float(bg.mean()) - 0.5) < 0.02
        seen.append(inds.clone())
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[0], seen[P])  # new pixels every batch
    cover = torch.bincount(torch.cat(seen) // (H * W // 16), minlength=16).float()
    assert (cover / cover.sum() - 1 / 16).abs().max() < 0.01  # uniform over the image


@pytest.mark.parametrize("kind", ["vm", "tensors"])
def test_gradient_outside_the_footprint_mask_is_zero(kind):
    """The compact ray-DP exchange (pvd/dp_compact.py) leaves out every table row no occupied cell can touch: after real
    HIP training steps those gradient entries must be exactly zero, and the rest must not be."""
    from pvd.config import PVDConfig
    from pvd.dp_compact import GradCompactor
    from pvd.ops import hip_ops
    from pvd.workload import DistillWorkload
    dev = torch.device("cuda:0")
    opt = PVDConfig(num_rays=4096, model_type=kind, resolution0=128, plenoxel_res="[64,64,64]", iters=100)
    w = DistillWorkload(hip_ops(), dev, opt, teacher_pretrain_steps=0, seed=0)
    tr = w.trainer
    c = GradCompactor(w.stu, tr.flat.params, tr.optimizer.offsets, dev)
    assert 0.02 < c.fraction < 0.6, c.fraction
    touched = torch.zeros(tr.flat.flat.numel(), dtype=torch.bool, device=dev)
    for _ in range(6):  # different poses
        tr.flat.zero_()
        with torch.autocast("cuda", dtype=torch.float16):
            loss, *_ = tr.compute_loss(*w.next_batch())
        tr._backward(loss)
        touched |= tr.flat.flat != 0
    outside = torch.ones_like(touched)
    outside[c.idx] = False
    assert outside.any() and not (touched & outside).any(), int((touched & outside).sum())
    assert int(touched.sum()) > 0.2 * c.idx.numel()  # the mask is not vacuously large


def test_segments_scatter_and_check_in_one_pass():
    """pvd_segments_op(4): the exchanged rows are put back AND looked at (ray-DP: the scaler's check rides on the scatter of the
    summed gradient).  Same result in `flat` as op 2; the flag as op 3 on the rows moved -- float4 and scalar segments."""
    import pvd_hip
    from pvd.dp_compact import segments_of
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(2)
    idx = torch.cat([torch.arange(8, 8 + 4096 * 3), torch.arange(20001, 20001 + 37), torch.arange(30000, 30000 + 5000, 1)]).to(dev)
    segs = segments_of(idx)
    buf = torch.randn(idx.numel(), generator=g).to(dev)
    flat_a, flat_b = torch.zeros(40000, device=dev), torch.zeros(40000, device=dev)
    flag = torch.zeros(1, device=dev)
    pvd_hip.segments_op(pvd_hip.SEG_SCATTER, flat_a, segs, buf=buf)
    pvd_hip.segments_op(pvd_hip.SEG_SCATTER_CHECK, flat_b, segs, buf=buf, found_inf=flag)
    assert torch.equal(flat_a, flat_b) and torch.equal(flat_b[idx], buf) and float(flag) == 0.0
    for pos in (0, 4095, 4096 * 3, 4096 * 3 + 36, idx.numel() - 1):
        for bad in (float("inf"), float("nan")):
            b2 = buf.clone(); b2[pos] = bad
            flag.zero_()
            pvd_hip.segments_op(pvd_hip.SEG_SCATTER_CHECK, flat_b, segs, buf=b2, found_inf=flag)
            assert float(flag) == 1.0, (pos, bad)
    with pytest.raises(pvd_hip.PvdHipError):
        pvd_hip.segments_op(pvd_hip.SEG_SCATTER_CHECK, flat_b, segs, buf=buf)


def test_mixed_inf_check_covers_exactly_the_fp32_buffer_minus_the_range_plus_the_half_buffer():
    """pvd_check_finite_mixed (FlatAdamW.check_finite when a parameter's gradient came in half precision): one launch over the fp32
    gradient outside that parameter's range and over the half buffer.  Every position of both is seen; the skipped range is not."""
    import pvd_hip
    dev = torch.device("cuda:0")
    n, b, e = 40000, 12000, 12000 + 20480
    g = torch.randn(n, device=dev)
    g16 = torch.randn(20480, device=dev).half()
    flag = torch.zeros(1, device=dev)
    pvd_hip.check_finite_mixed(g, b, e, g16, flag)
    assert float(flag) == 0.0
    for pos in (b, b + 7777, e - 1):  # inside the skipped range: somebody else's (the half buffer stands for it)
        g2 = g.clone(); g2[pos] = float("nan")
        pvd_hip.check_finite_mixed(g2, b, e, g16, flag)
        assert float(flag) == 0.0, pos
    for pos in (0, 3, b - 1, e, e + 5, n - 1):
        for bad in (float("inf"), float("-inf"), float("nan")):
            g2 = g.clone(); g2[pos] = bad
            flag.zero_()
            pvd_hip.check_finite_mixed(g2, b, e, g16, flag)
            assert float(flag) == 1.0, (pos, bad)
    for pos in (0, 1, 4095, 20479):
        h2 = g16.clone(); h2[pos] = float("inf")
        flag.zero_()
        pvd_hip.check_finite_mixed(g, b, e, h2, flag)
        assert float(flag) == 1.0, pos
    flag.zero_()
    pvd_hip.check_finite_mixed(g, 0, 0, g16, flag)      # empty range
    pvd_hip.check_finite_mixed(g, 0, n, g16, flag)      # the whole fp32 buffer skipped
    assert float(flag) == 0.0
    with pytest.raises(pvd_hip.PvdHipError):
        pvd_hip.check_finite_mixed(g, 2, 8, g16, flag)  # ranges are whole groups of four


def test_inf_check_rides_on_the_vm_scatter_launch(monkeypatch):
    """pvd_head_dw_rider.found_inf: the scaler's inf check of the VM student's gradients is done by the launch that completes them (the
    table scatter looks at every incoming gradient value it reads, the riding weight-gradient reduction at every sum it adds), and
    FlatGradScaler.step launches no check of its own.  Same protocol as the separate check (PVD_INF_CHECK_RIDE=0): a clean step
    leaves the flag alone, an overflowing one raises it in the backward, is skipped, halves the scale and clears the flag; a single
    nan in the upstream gradient of one sample is seen; eager and replayed steps train alike."""
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import DistillWorkload
    dev = torch.device("cuda:0")
    finals = {}
    for ride in ("1", "0"):
        monkeypatch.setenv("PVD_INF_CHECK_RIDE", ride)
        opt = PVDConfig(num_rays=1024, resolution0=64, iters=200)
        opt.stage_iters = {"stage1": -1, "stage2": -1}
        w = DistillWorkload(hip_ops(), dev, opt, teacher_pretrain_steps=0, seed=0)
        tr, o = w.trainer, w.trainer.optimizer
        assert (getattr(w.stu, "_inf_check_in_backward", None) is not None) == (ride == "1")
        calls = []
        orig = o.check_finite
        o.check_finite = lambda flag, orig=orig, calls=calls: (calls.append(1), orig(flag))[1]
        for _ in range(3):
            w.step()
        flag = o.inf_flag()
        assert (len(calls) == 0) == (ride == "1") and float(flag) == 0.0 and float(o.step_count) == 3.0

        def backward_only(scale=None, poison=None):
            if scale is not None:
                tr.scaler._scale.fill_(scale)
            tr._zero_grads()
            with torch.autocast("cuda", dtype=torch.float16):
                loss, *_ = tr.compute_loss(*w.next_batch())
            if poison is not None:  # one nan in the gradient flowing back into ONE sample's colour products
                h = w.stu.feature_sigma_color.register_hook(lambda g: g.index_put((torch.tensor([poison], device=dev), torch.tensor([3], device=dev)),
                                                                                   torch.tensor([float("nan")], device=dev, dtype=g.dtype)))
            tr._backward(loss)
            if poison is not None:
                h.remove()

        # an overflowing backward (every f16 gradient value inf / nan)
        p0, s0 = o.flat_p.clone(), float(tr.scaler._scale)
        backward_only(scale=2.0 ** 40)
        if ride == "1":
            assert float(flag) == 1.0 and o._checked_by_backward  # raised by the backward's own launch
        tr._exchange(); tr._optimize(); tr.scheduler.step(); tr.global_step += 1
        assert torch.equal(o.flat_p, p0) and float(tr.scaler._scale) == 2.0 ** 39 and float(flag) == 0.0 and float(o.step_count) == 3.0
        tr.scaler._scale.fill_(s0)
        # one poisoned value
        backward_only(poison=517)
        tr._exchange(); tr._optimize(); tr.scheduler.step(); tr.global_step += 1
        assert torch.equal(o.flat_p, p0) and float(tr.scaler._scale) == s0 / 2 and float(flag) == 0.0
        tr.scaler._scale.fill_(s0)
        # a gradient poisoned AFTER the backward, before the next zero_grad, is somebody else's to find: the separate check still exists
        tr._zero_grads()
        o.flat_g[o.touched.idx[10] if o.touched is not None else 10] = float("inf")
        tr._optimize()
        assert torch.equal(o.flat_p, p0) and len(calls) >= 1
        tr.scaler._scale.fill_(s0)
        # ... and training goes on, eagerly and replayed
        w.step()
        w.enable_graph(steps_per_graph=2)
        losses = [float(w.step()[0]) for _ in range(6)]
        assert all(l == l for l in losses) and float(flag) == 0.0 and not torch.equal(o.flat_p, p0)
        finals[ride] = (losses, float(tr.scaler._scale))
    assert finals["1"][1] == finals["0"][1]
    assert all(abs(a - b) <= 3e-2 * abs(b) for a, b in zip(finals["1"][0], finals["0"][0])), finals


def test_flat_adamw_half_gradient_equals_widen_and_add():
    """FlatAdamW.accept_half_grad: an f16 gradient added inside the update kernel == adding it into the fp32 gradient
    first (what autograd does for the hash table under autocast, grid.py:105-136), incl. the inf check of a scaled step."""
    import pvd_hip
    from pvd.flat_adamw import FlatAdamW, FlatGradScaler
    dev = torch.device("cuda:0")
    def make():
        gg = torch.Generator(device=dev).manual_seed(5)
        return [torch.nn.Parameter(torch.randn(64, 32, device=dev, generator=gg)), torch.nn.Parameter(torch.randn(4096, 2, device=dev, generator=gg))]
    pa, pb = make(), make()
    oa = FlatAdamW([{"params": pa, "lr": 1e-2}], betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-2)
    ob = FlatAdamW([{"params": pb, "lr": 1e-2}], betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-2)
    g = torch.Generator(device=dev).manual_seed(6)
    for it in range(5):
        oa.zero_grad(); ob.zero_grad()
        g32 = [torch.randn(p.shape, device=dev, generator=g) for p in pa]
        g16 = (torch.randn(4096, 2, device=dev, generator=g) * 4).half()
        for p, q, gr in zip(pa, pb, g32):
            p.grad.copy_(gr); q.grad.copy_(gr)
        pa[1].grad.add_(g16)
        assert ob.accept_half_grad(pb[1], g16) and not ob.accept_half_grad(pb[1], g16)  # one pending gradient at a time
        oa.step(); ob.step()
        assert ob._half_grad is None
    for p, q in zip(pa, pb):
        assert torch.allclose(p, q, rtol=1e-6, atol=1e-7)
    flag = torch.zeros(1, device=dev)
    h = torch.randn(4096, device=dev).half()
    pvd_hip.check_finite_f16(h, flag)
    assert float(flag) == 0.0
    h[1001] = float("inf")
    pvd_hip.check_finite_f16(h, flag)
    assert float(flag) == 1.0
    # a scaled step whose only non-finite value sits in the half gradient is skipped
    scaler = FlatGradScaler("cuda", init_scale=16.0)
    scaler.scale(torch.zeros((), device=dev))
    ob.zero_grad()
    before = pb[1].detach().clone()
    assert ob.accept_half_grad(pb[1], h[: 4096 * 2 // 2].repeat(2).reshape(4096, 2).contiguous())
    scaler.step(ob); scaler.update()
    assert torch.equal(before, pb[1].detach()) and float(scaler.get_scale()) == 8.0
    # a SECOND backward before the step: the optimizer already holds a half gradient, refuses the new one, and the caller
    # adds it into the fp32 range the mixed check skips -- an inf in there must still skip the step (full check for this step)
    ob.zero_grad()
    fin = (torch.randn(4096, 2, device=dev) * 0.1).half()
    assert ob.accept_half_grad(pb[1], fin) and not ob._half_range_dirty
    bad = fin.clone()
    bad[77, 1] = float("inf")
    assert not ob.accept_half_grad(pb[1], bad) and ob._half_range_dirty
    pb[1].grad.add_(bad)  # (what fusedhead's backward does when the taker refuses)
    before = pb[1].detach().clone()
    scaler.step(ob); scaler.update()
    assert torch.equal(before, pb[1].detach()) and float(scaler.get_scale()) == 4.0 and not ob._half_range_dirty
    # ... and with two finite gradients the step goes through and applies both
    ob.zero_grad()
    assert ob.accept_half_grad(pb[1], fin) and not ob.accept_half_grad(pb[1], fin)
    pb[1].grad.add_(fin)
    scaler.step(ob); scaler.update()
    assert not torch.equal(before, pb[1].detach()) and float(scaler.get_scale()) == 4.0


def test_segments_op_matches_index_ops():
    """pvd_segments_op (zero / gather / scatter / inf check over a run table) against torch indexing with the same set:
    ragged runs, runs longer than one table entry, unaligned starts and lengths."""
    import pvd_hip
    from pvd.dp_compact import segments_of
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    n = 300000
    keep = torch.zeros(n, dtype=torch.bool)
    for start, length in [(0, 5), (17, 1), (64, 4096 * 2 + 13), (20000, 48 * 300), (90001, 7), (120000, 16 * 1000), (n - 9, 9)]:
        keep[start:start + length] = True
    keep |= torch.rand(n, generator=g) < 0.01  # isolated elements and chance merges
    idx = keep.nonzero().squeeze(-1).to(dev)
    segs = segments_of(idx)
    assert int(segs[:, 2].sum()) == idx.numel() and int(segs[:, 2].max()) <= 4096
    flat = torch.randn(n, generator=g).to(dev)
    buf = torch.empty(idx.numel(), device=dev)
    pvd_hip.segments_op(pvd_hip.SEG_GATHER, flat, segs, buf=buf)
    assert torch.equal(buf, flat[idx])
    new = torch.randn(idx.numel(), generator=g).to(dev)
    want = flat.clone()
    want[idx] = new
    pvd_hip.segments_op(pvd_hip.SEG_SCATTER, flat, segs, buf=new)
    assert torch.equal(flat, want)
    flag = torch.zeros(1, device=dev)
    pvd_hip.segments_op(pvd_hip.SEG_CHECK, flat, segs, found_inf=flag)
    assert float(flag) == 0.0
    outside = (~keep).nonzero().squeeze(-1).to(dev)
    flat[outside[5]] = float("inf")  # outside the set: not looked at
    pvd_hip.segments_op(pvd_hip.SEG_CHECK, flat, segs, found_inf=flag)
    assert float(flag) == 0.0
    for bad in (float("nan"), float("-inf")):
        f2 = flat.clone()
        f2[idx[idx.numel() // 2 + 3]] = bad
        flag.zero_()
        pvd_hip.segments_op(pvd_hip.SEG_CHECK, f2, segs, found_inf=flag)
        assert float(flag) == 1.0
    want = flat.clone()
    want[idx] = 0
    pvd_hip.segments_op(pvd_hip.SEG_ZERO, flat, segs)
    assert torch.equal(torch.nan_to_num(flat, posinf=7.0), torch.nan_to_num(want, posinf=7.0))
    empty = torch.zeros(0, 3, dtype=torch.int32, device=dev)
    pvd_hip.segments_op(pvd_hip.SEG_ZERO, flat, empty)  # no-op
    with pytest.raises(pvd_hip.PvdHipError):
        pvd_hip.segments_op(pvd_hip.SEG_GATHER, flat, segs)  # no compact buffer


def test_touched_set_training_equals_dense_zero_and_check(monkeypatch):
    """zero_grad / inf check restricted to the rows a sample can touch (FlatAdamW.set_touched) is the same training as
    clearing and checking the whole buffer: identical gradients before the update, step by step."""
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import DistillWorkload
    dev = torch.device("cuda:0")
    grads = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PVD_TOUCHED_SET", mode)
        opt = PVDConfig(num_rays=2048, resolution0=96, iters=200)
        w = DistillWorkload(hip_ops(), dev, opt, teacher_pretrain_steps=0, seed=0)
        tr = w.trainer
        torch.cuda.manual_seed(5)
        out = []
        for it in range(5):
            batch = w.next_batch()
            tr._zero_grads()
            assert (tr.optimizer.touched is not None) == (mode == "1")
            with torch.autocast("cuda", dtype=torch.float16):
                loss, *_ = tr.compute_loss(*batch)
            tr._backward(loss)
            out.append(tr.flat.flat.clone())
            tr._exchange()
            tr._optimize()
            tr.scheduler.step()
            tr.global_step += 1
        grads[mode] = out
        if mode == "1":
            assert tr.optimizer._outside_is_zero
            # a poisoned gradient inside the set is seen by the restricted check and skips the step
            p0 = tr.optimizer.flat_p.clone()
            tr._zero_grads()
            tr.flat.flat[tr.optimizer.touched.idx[10]] = float("inf")
            tr._optimize()
            assert torch.equal(tr.optimizer.flat_p, p0)
    for a, b in zip(grads["1"], grads["0"]):
        # float atomics in the backward: same values up to summation order
        assert (a - b).abs().max() <= 2e-3 * b.abs().max() + 1e-12


@pytest.mark.parametrize("budget", ["slack", "half", "trimmed"])
def test_composite_bg_backward_fresh_gradients(budget):
    """packed_rays=True: the fused compositing backward takes UNINITIALISED gradient buffers and clears the slots no ray
    owns itself (tail beyond the last ray, slots of dropped rays) -- same gradients as the zero-filled call."""
    import pvd_hip
    import raymarching
    from pvd.scene import BLENDER_INTRINSICS, ChairScene, get_rays, packbits_torch, synthetic_poses
    dev = torch.device("cuda:0")
    N = 1500
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(2))).to(dev)
    r = get_rays(poses[3:4], BLENDER_INTRINSICS, 800, 800, N, generator=torch.Generator(device=dev).manual_seed(2))
    bits = packbits_torch(ChairScene().density_grid(128, 1.0, 1, device=dev), 10.0)
    o, d = r["rays_o"].reshape(-1, 3).contiguous(), r["rays_d"].reshape(-1, 3).contiguous()
    nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    raymarching.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, counter, -1, True, 128, True)
    total = int(counter[0])
    if budget == "trimmed":
        xyzs, dirs, deltas, rays = raymarching.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, None, -1, True, 128, True)
    else:
        mc = total // 2 if budget == "half" else total * 5 // 4
        xyzs, dirs, deltas, rays = raymarching.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, None, mc, True, 128, False)
    M = xyzs.shape[0]
    rr = rays.cpu()
    dropped = (rr[:, 2] > 0) & (rr[:, 1] + rr[:, 2] >= M)
    assert bool(dropped.any()) == (budget == "half")
    g = torch.Generator(device=dev).manual_seed(1)
    sig0 = torch.exp(torch.rand(M, device=dev, generator=g) * 6 - 2)
    rgb0 = torch.rand(M, 3, device=dev, generator=g)
    bg = torch.rand(1, N, 3, device=dev, generator=g)
    w_img = torch.randn(N, 3, device=dev, generator=g)
    # poison the caching allocator's free blocks so that "uninitialised" is not accidentally zero
    junk = [torch.full((M * 4,), float("nan"), device=dev) for _ in range(4)]
    del junk
    res = []
    calls = []
    real = pvd_hip.raymarching_backend.composite_rays_train_bg_backward
    pvd_hip.raymarching_backend.composite_rays_train_bg_backward = lambda *a, **k: (calls.append(k.get("fresh", False)), real(*a, **k))[1]
    try:
        for packed in (True, False):
            sig, rgb = sig0.clone().requires_grad_(True), rgb0.clone().requires_grad_(True)
            ws, depth, img = raymarching.composite_rays_train_bg(sig, rgb, deltas, rays, bg, nears, fars, 1e-6, packed)
            (img * w_img).sum().backward()
            res.append((sig.grad.clone(), rgb.grad.clone()))
    finally:
        pvd_hip.raymarching_backend.composite_rays_train_bg_backward = real
    assert calls == [True, False]  # the wrapper really handed over uninitialised buffers the first time
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.isfinite(res[0][0]).all() and torch.isfinite(res[0][1]).all()
    # and straight through the binding with NaN-filled buffers
    gs, gr = torch.full((M,), float("nan"), device=dev), torch.full((M, 3), float("nan"), device=dev)
    ws, depth, img = raymarching.composite_rays_train_bg(sig0, rgb0, deltas, rays, bg, nears, fars, 1e-6)
    pvd_hip.composite_rays_train_bg_backward(None, w_img, sig0, rgb0, deltas, rays, ws, img, M, N, bg.reshape(-1, 3).contiguous(), 0.0, gs, gr,
                                             fresh=True)
    assert torch.equal(gs, res[1][0]) and torch.equal(gr, res[1][1])


def test_adamw_cold_groups_are_bit_identical_to_the_dense_update():
    """pvd_adamw_extras.cold_bits: groups of 4 parameters whose gradient and moments are zero for good get the weight decay
    alone, without g / m / v being read or written -- the very bits the dense kernel produces (g = m = v = 0 makes the Adam
    term exactly zero).  Six steps with a schedule, a GradScaler scale, L1 on a warm range and an inf step in between."""
    import pvd_hip
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    n = 128 * 1000 + 64
    n4 = n // 4
    cold4 = torch.rand(n4, device=dev, generator=g) < 0.7
    cold4[: 4096 // 4] = False  # the L1 range is warm
    cold_el = cold4.repeat_interleave(4)
    words = (n + 127) // 128
    bits = torch.zeros(words * 32, dtype=torch.int64, device=dev)
    bits[:n4] = cold4.to(torch.int64)
    packed = (bits.view(words, 32) << torch.arange(32, device=dev)).sum(1)
    packed = torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32).contiguous()
    p0 = torch.randn(n, device=dev, generator=g)
    p0[cold_el.nonzero()[:50].squeeze(-1)] = 0.0  # zeros and negative zeros keep their sign bit
    p0[cold_el.nonzero()[50:60].squeeze(-1)] = -0.0
    grads = []
    for s in range(6):
        gr = torch.randn(n, device=dev, generator=g) * 64.0
        gr[cold_el] = 0.0
        if s == 3:
            gr[(~cold_el).nonzero()[7]] = float("inf")
        grads.append(gr)
    res, res_two = [], []
    seg_ends = [n // 2 // 4 * 4, n]
    a_el = torch.zeros(n, dtype=torch.bool, device=dev)
    a_el[30000:] = True
    grads_two = [torch.where(a_el, torch.zeros_like(gr), gr) for gr in grads]  # (the inf of step 3 sits in a B group: element < 30000)
    assert not torch.isfinite(grads_two[3]).all()
    # (part A's moments are still decaying: rows that used to be reachable -- so the recorded step count and rates matter)
    m_init, v_init = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    decaying = a_el & ~cold_el
    m_init[decaying] = torch.randn(int(decaying.sum()), device=dev, generator=g) * 0.1
    v_init[decaying] = torch.rand(int(decaying.sum()), device=dev, generator=g) * 0.01
    if True:  # the single dense launch on those gradients: what both two-part runs must reproduce bit for bit
        p, m, v = p0.clone(), m_init.clone(), v_init.clone()
        lr = torch.tensor([1e-2, 3e-3], device=dev)
        base = lr.clone()
        step, sched = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        scale, tracker, flag = torch.tensor([64.0], device=dev), torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, device=dev)
        for gr in grads_two:
            pvd_hip.check_finite(gr, flag)
            pvd_hip.adamw_step(p, gr, m, v, seg_ends, lr, 0.9, 0.99, 1e-15, 0.01, step, scale, flag, schedule=(1, 100.0, 5e-5, base, sched),
                               l1_ranges=[(0, 4096, 1e-3)], amp_update=(scale, tracker, 2.0, 0.5, 2000))
        res_two.append((p, m, v, step.clone(), scale.clone()))
    # dense; cold groups decayed on every step; cold groups' decay DEFERRED (lazy log) and replayed in one go at the end, and
    # in two goes (a flush after the third step)
    warm_list = (~cold4).nonzero().squeeze(1).to(torch.int32).contiguous()
    for use_cold, lazy_flush_at in ((False, None), (True, None), (True, (6,)), (True, (3, 6)), (True, (2, 6, "list"))):
        p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        m_probe = m.clone()
        lr = torch.tensor([1e-2, 3e-3], device=dev)
        base = lr.clone()
        step, sched = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        scale, tracker, flag = torch.tensor([64.0], device=dev), torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, device=dev)
        lazy = (torch.zeros(16, 2, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)) if lazy_flush_at else None
        if lazy_flush_at and "list" in lazy_flush_at:  # the update walks the list of warm groups instead of testing every group's bit
            lazy = lazy + (warm_list,)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        replayed = []
        for k, gr in enumerate(grads):
            pvd_hip.check_finite(gr, flag)
            pvd_hip.adamw_step(p, gr, m, v, seg_ends, lr, 0.9, 0.99, 1e-15, 0.01, step, scale, flag,
                               schedule=(1, 100.0, 5e-5, base, sched), l1_ranges=[(0, 4096, 1e-3)],
                               amp_update=(scale, tracker, 2.0, 0.5, 2000), cold_bits=packed if use_cold else None, lazy=lazy)
            if lazy is not None and k + 1 in lazy_flush_at:
                if k + 1 == 6 and len(lazy_flush_at) == 1:
                    assert torch.equal(p[cold_el], p0[cold_el])  # untouched until the flush
                pvd_hip.adamw_lazy_flush(p, seg_ends, packed, lazy[0], lazy[1], 0.01, status)
                replayed.append(int(status[0]))
                assert int(lazy[1][0]) == 0
        if lazy is not None:
            assert sum(replayed) == 5  # the skipped (inf) step logs nothing
        res.append((p, m, v, step.clone(), scale.clone()))
    # the TWO-PART update (pvd_adamw_extras.snapshot / replay): the warm groups split into B (may hold a gradient) and A (gradient
    # structurally zero: here the warm groups past the first 30 000 elements, whose gradients are zeroed for every run of this
    # comparison below); B runs with the tail and records the step's scalars, A runs LATER -- here after the next step's
    # gradient check has already raised the live inf flag / the tail has moved on -- from the record
    for part_a_late in (False, True):
        p, m, v = p0.clone(), m_init.clone(), v_init.clone()
        lr = torch.tensor([1e-2, 3e-3], device=dev)
        base = lr.clone()
        step, sched = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        scale, tracker, flag = torch.tensor([64.0], device=dev), torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, device=dev)
        log, count = torch.zeros(16, 2, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        snap = torch.zeros(4 + 2, device=dev)
        is_a = torch.zeros(n4, dtype=torch.bool, device=dev)
        is_a[30000 // 4:] = True
        warm_b = ((~cold4) & ~is_a).nonzero().squeeze(1).to(torch.int32).contiguous()
        warm_a = ((~cold4) & is_a).nonzero().squeeze(1).to(torch.int32).contiguous()
        assert warm_a.numel() > 1000 and warm_b.numel() > 1000
        owed = False

        def part_a():
            pvd_hip.adamw_step(p, gr_prev, m, v, seg_ends, lr, 0.9, 0.99, 1e-15, 0.01, step, snap[2:3], None, l1_ranges=[(0, 4096, 1e-3)],
                               cold_bits=packed, lazy=(log, count, warm_a), replay=snap)
        for k, gr in enumerate(grads_two):
            pvd_hip.check_finite(gr, flag)
            if owed and part_a_late:
                part_a()  # the previous step's part A, with the NEXT step's inf flag already live: it must use the record
            pvd_hip.adamw_step(p, gr, m, v, seg_ends, lr, 0.9, 0.99, 1e-15, 0.01, step, scale, flag, schedule=(1, 100.0, 5e-5, base, sched),
                               l1_ranges=[(0, 4096, 1e-3)], amp_update=(scale, tracker, 2.0, 0.5, 2000), cold_bits=packed,
                               lazy=(log, count, warm_b), snapshot=snap)
            gr_prev, owed = gr, True
            if not part_a_late:
                part_a()
        if part_a_late:
            part_a()
        pvd_hip.adamw_lazy_flush(p, seg_ends, packed, log, count, 0.01, status)
        res_two.append((p, m, v, step.clone(), scale.clone()))
    # fewer launches (pvd_adamw_extras.arrivals / zero_grad_after): the tail's work done inside the update kernel by the last
    # workgroup to arrive, and every gradient group zeroed once it has been read (also on the skipped step) -- same bits, and
    # the gradient buffer comes back clean wherever the update walks (cold groups hold no gradient to begin with)
    for use_cold in (False, True):
        p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        lr = torch.tensor([1e-2, 3e-3], device=dev)
        base = lr.clone()
        step, sched = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        scale, tracker, flag = torch.tensor([64.0], device=dev), torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, device=dev)
        arrivals = torch.zeros(65 * 32, dtype=torch.int32, device=dev)
        log, count = torch.zeros(16, 2, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        for k, gr in enumerate(grads):
            gk = gr.clone()
            pvd_hip.check_finite(gk, flag)
            pvd_hip.adamw_step(p, gk, m, v, seg_ends, lr, 0.9, 0.99, 1e-15, 0.01, step, scale, flag, schedule=(1, 100.0, 5e-5, base, sched),
                               l1_ranges=[(0, 4096, 1e-3)], amp_update=(scale, tracker, 2.0, 0.5, 2000),
                               cold_bits=packed if use_cold else None, lazy=(log, count, warm_list) if use_cold else None,
                               zero_after=True, arrivals=arrivals)
            assert not arrivals.any() and float(flag[0]) == 0.0  # counters back at zero, inf flag cleared by the in-kernel tail
            assert not gk[: n4 * 4].any(), k  # (also after the skipped step k = 3)
        if use_cold:
            pvd_hip.adamw_lazy_flush(p, seg_ends, packed, log, count, 0.01, status)
        res.append((p, m, v, step.clone(), scale.clone()))
    # one launch over the list [B run | A run] with warm_zero_grad_from = |B|: the A run's gradient is structurally zero, so it is
    # neither read nor zeroed -- poisoned with NaN here to prove it; same bits as the dense launch on the zero gradients
    if True:
        p, m, v = p0.clone(), m_init.clone(), v_init.clone()
        lr = torch.tensor([1e-2, 3e-3], device=dev)
        base = lr.clone()
        step, sched = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        scale, tracker, flag = torch.tensor([64.0], device=dev), torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, device=dev)
        log, count = torch.zeros(16, 2, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        is_a = torch.zeros(n4, dtype=torch.bool, device=dev)
        is_a[30000 // 4:] = True
        wb = ((~cold4) & ~is_a).nonzero().squeeze(1).to(torch.int32)
        wa = ((~cold4) & is_a).nonzero().squeeze(1).to(torch.int32)
        both = torch.cat([wb, wa]).contiguous()
        a_warm_el = (wa.long()[:, None] * 4 + torch.arange(4, device=dev)).reshape(-1)
        for gr in grads_two:
            gk = gr.clone()
            pvd_hip.check_finite(gk, flag)
            gk[a_warm_el] = float("nan")
            pvd_hip.adamw_step(p, gk, m, v, seg_ends, lr, 0.9, 0.99, 1e-15, 0.01, step, scale, flag, schedule=(1, 100.0, 5e-5, base, sched),
                               l1_ranges=[(0, 4096, 1e-3)], amp_update=(scale, tracker, 2.0, 0.5, 2000), cold_bits=packed,
                               lazy=(log, count, both, int(wb.numel())), zero_after=True)
            assert torch.isnan(gk[a_warm_el]).all() and not gk[(wb.long()[:, None] * 4 + torch.arange(4, device=dev)).reshape(-1)].any()
        pvd_hip.adamw_lazy_flush(p, seg_ends, packed, log, count, 0.01, status)
        res_two.append((p, m, v, step.clone(), scale.clone()))
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b)
    for other in res_two[1:]:
        for a, b in zip(res_two[0], other):
            assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b)
    p, m, v = res[1][:3]
    assert float(res[1][3]) == 5.0  # one of the six steps was skipped
    assert not torch.equal(p, p0) and (m[cold_el] == 0).all() and (v[cold_el] == 0).all() and (m[~cold_el] != 0).any()
    assert (p[cold_el] != p0[cold_el]).any()  # the cold groups did get their weight decay


def test_trainer_cold_bitmap_covers_only_unreachable_unregularised_groups():
    """FlatAdamW._build_cold_bits in the real trainer: cold groups are disjoint from the touched set and from the L1 ranges,
    stay at g = m = v = 0 over training steps, still receive their weight decay, and make up most of the colour planes."""
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import DistillWorkload
    dev = torch.device("cuda:0")
    opt = PVDConfig(num_rays=2048, resolution0=96, iters=200)
    w = DistillWorkload(hip_ops(), dev, opt, teacher_pretrain_steps=0, seed=0)
    tr, o = w.trainer, w.trainer.optimizer
    p0 = o.flat_p.clone()
    for _ in range(4):
        tr.train_step(*w.next_batch())
    bits = o._cold_bits
    assert bits is not None and 0.3 < o.cold_fraction < 0.95, o.cold_fraction
    if o._lazy is not None:  # the cold groups' decay is deferred: nothing has touched them yet
        n0 = o.flat_p.numel() // 4
        w0 = bits.to(torch.int64) & 0xFFFFFFFF
        c0 = ((w0[:, None] >> torch.arange(32, device=dev)) & 1).reshape(-1)[:n0].bool().repeat_interleave(4)
        assert torch.equal(o.flat_p[:c0.numel()][c0], p0[:c0.numel()][c0]) and o._lazy_logged >= 1
        o.flush()
        assert o._lazy_logged == 0 and int(o._lazy[1][0]) == 0
    n4 = o.flat_p.numel() // 4
    words = bits.to(torch.int64) & 0xFFFFFFFF
    cold4 = ((words[:, None] >> torch.arange(32, device=dev)) & 1).reshape(-1)[:n4].bool()
    cold = cold4.repeat_interleave(4)
    assert not cold[o.touched.idx].any()
    for b, e, _ in o._l1:
        assert not cold[b:e].any()
    n = cold.numel()
    assert (o.flat_g[:n][cold] == 0).all() and (o.flat_m[:n][cold] == 0).all() and (o.flat_v[:n][cold] == 0).all()
    live = p0[:n][cold] != 0
    assert ((o.flat_p[:n][cold] != p0[:n][cold]) == live).all()  # decayed (AdamW's default weight decay), zeros stay zeros
    ratio = (o.flat_p[:n][cold][live] / p0[:n][cold][live]).double()
    assert (ratio < 1).all() and (ratio > 0.99).all() and float(ratio.max() - ratio.min()) < 1e-5  # one common decay factor per lr group


def test_whole_table_readers_see_the_deferred_decay():
    """FlatAdamW defers the weight decay of table rows no training sample touches.  Readers that go OUTSIDE the training
    samples -- an evaluation render, checkpoint loading, update_extra_state's density sweep (DistillTrainer.train_step with
    update_stu_extra) -- must find those rows decayed: after each of them nothing is pending, and the rows that were cold hold,
    bit for bit, what the per-step decay (PVD_ADAMW_LAZY=0) leaves after the same sequence of calls."""
    import os
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import DistillWorkload
    from pvd.checkpoint import _load_model, checkpoint_dict
    dev = torch.device("cuda:0")
    runs = {}
    for lazy in (True, False):
        old = os.environ.get("PVD_ADAMW_LAZY")
        os.environ["PVD_ADAMW_LAZY"] = "1" if lazy else "0"
        try:
            torch.manual_seed(0)
            opt = PVDConfig(num_rays=1024, resolution0=96, iters=200)
            w = DistillWorkload(hip_ops(), dev, opt, teacher_pretrain_steps=0, seed=0)
            tr, o = w.trainer, w.trainer.optimizer
            pending = lambda: o._lazy_logged if lazy else 1
            for _ in range(3):
                tr.train_step(*w.next_batch())
            assert pending() >= 1
            w.stu.eval()  # reader 1: an evaluation render (marches on the model's own grid)
            with torch.no_grad():
                b = w.next_batch()
                w.stu.render(b[0][:, :64].contiguous(), b[1][:, :64].contiguous(), bg_color=1, perturb=False)
            w.stu.train()
            assert o._lazy_logged == 0
            tr.train_step(*w.next_batch())
            ck = checkpoint_dict(w.stu)  # (state_dict flushes too)
            tr.train_step(*w.next_batch())
            assert pending() >= 1
            _load_model(w.stu, ck)  # reader 2: pending decays are applied BEFORE the rows are overwritten, not after
            assert o._lazy_logged == 0
            for k, v in w.stu.state_dict().items():
                assert torch.equal(v.float().cpu(), ck["model"][k].float().cpu()), k
            tr.train_step(*w.next_batch())
            assert pending() >= 1
            n4 = o.flat_p.numel() // 4
            words = o._cold_bits.to(torch.int64) & 0xFFFFFFFF
            cold = ((words[:, None] >> torch.arange(32, device=dev)) & 1).reshape(-1)[:n4].bool().repeat_interleave(4)
            with torch.autocast("cuda", dtype=torch.float16):
                w.stu.update_extra_state()  # reader 3: the sweep over the whole grid
            assert o._lazy_logged == 0 and (not lazy or int(o._lazy[1][0]) == 0)
            runs[lazy] = (cold, o.flat_p[:cold.numel()][cold].clone())
        finally:
            if old is None:
                os.environ.pop("PVD_ADAMW_LAZY", None)
            else:
                os.environ["PVD_ADAMW_LAZY"] = old
    (ca, pa), (cb, pb) = runs[True], runs[False]
    assert ca.any() and torch.equal(ca, cb)  # the same rows are cold either way (the touched set comes from the frozen grid)
    assert torch.equal(pa, pb)  # the decays deferred and flushed == applied step by step


@pytest.mark.parametrize("n_rays", [1001, 4096])
def test_objective_riding_on_the_compositing_launches_equals_the_separate_launches(n_rays):
    """pvd_composite_objective_forward / _backward (ObjectiveRide): the student's compositing launch also forms the partial sums
    of the four squared norms, the compositing backward launch forms the image gradient from the coefficients and writes the
    feature / colour gradients -- against composite_rays_train_bg + distill_loss_normL2 as separate launches: images and every
    gradient bit-identical (same per-element arithmetic), the loss to summation order."""
    import raymarching
    from pvd.losses import ObjectiveRide, distill_loss_normL2
    from pvd.scene import BLENDER_INTRINSICS, ChairScene, get_rays, packbits_torch, synthetic_poses
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(0))).to(dev)
    bits = packbits_torch(ChairScene(thicken=0.08).density_grid(128, 1.0, 1, device=dev), 10.0)
    r = get_rays(poses[1:2], BLENDER_INTRINSICS, 800, 800, n_rays, generator=torch.Generator(device=dev).manual_seed(2))
    o, d = r["rays_o"].reshape(-1, 3).contiguous(), r["rays_d"].reshape(-1, 3).contiguous()
    nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.2)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, None, -1, True, 128, True)
    M, N = xyzs.shape[0], n_rays
    sig0 = torch.rand(M, device=dev, generator=g) * 20
    rgb0 = torch.rand(M, 3, device=dev, generator=g)
    fea0 = torch.randn(M, 16, device=dev, generator=g)
    col0 = torch.rand(M, 3, device=dev, generator=g)
    fea_t, col_t = torch.randn(M, 16, device=dev, generator=g), torch.rand(M, 3, device=dev, generator=g)
    img_t = torch.rand(1, N, 3, device=dev, generator=g)
    bg = torch.rand(1, N, 3, device=dev, generator=g)
    rates = torch.tensor([1.0, 0.002, 0.002, 0.002], device=dev)
    outs = []
    extra = torch.rand(8192, device=dev, generator=g) * 1e-3  # partial sums of a parameter-only term (the L1 regulariser's value)
    # separate launches; riding with k_loss_final between the passes; riding with the objective finished by the backward launch
    class OneRankWorld:  # ray-DP's interface as _DistillNormL2 uses it; a world of one rank sums nothing
        enabled, calls = True, []

        def all_reduce_sum_(self, t, overlap=None):
            self.calls.append(int(t.numel()))
            return t
    # ... and (round 6) riding under ray-DP: a partial-sum count that depends on N alone, the partials all-reduced between the launches
    for riding in (None, "final", "finish", "dp"):
        sig, rgb, fea, col = [t.clone().requires_grad_(True) for t in (sig0, rgb0, fea0, col0)]
        r4 = rates.clone()
        ride = None
        if riding:
            ride = ObjectiveRide(img_t, fea_t, col_t, rates_decay=r4 if riding in ("finish", "dp") else None, fea_decay=0.995,
                                 fixed_parts=riding == "dp")
            assert ride.with_student(fea, col)
        ws, depth, img = raymarching.composite_rays_train_bg(sig, rgb, deltas, rays, bg, nears, fars, 1e-6, True, **({"objective": ride} if riding else {}))
        assert (ride is not None and ride.S is not None and ride.nparts >= 2) == bool(riding)
        loss, norms = distill_loss_normL2(img.view(1, N, 3), img_t, fea, fea_t, col, col_t, r4, OneRankWorld() if riding == "dp" else None,
                                          fea_decay=0.995, extra=extra, **({"ride": ride} if riding else {}))
        assert (ride is not None and ride.finish is not None) == (riding in ("finish", "dp"))
        if riding == "dp":  # ONE collective, over the partials: (rays / 4 workgroups + 256) float4 entries whatever the sample count
            assert OneRankWorld.calls == [4 * ((N + 3) // 4 + 256)] and ride.nparts == (N + 3) // 4 + 256
            OneRankWorld.calls.clear()
        (loss * 3.0).backward()  # (finish: loss / norms are filled in by the backward launch)
        outs.append((img.detach(), ws.detach(), depth.detach(), float(loss), norms.clone(), sig.grad, rgb.grad, fea.grad, col.grad, r4.clone()))
    a = outs[0]
    assert a[9][1] == rates[1] * 0.995 and a[9][0] == rates[0]  # the per-step decay of the feature rate
    for b in outs[1:]:
        for k in (0, 1, 2, 9):
            assert torch.equal(a[k], b[k])
        assert abs(a[3] - b[3]) <= 2e-6 * abs(a[3]) and torch.allclose(a[4], b[4], rtol=2e-6)
        # the coefficients are functions of the (differently ordered) sums: gradients agree to their rounding, element for element
        for k in (5, 6, 7, 8):
            assert a[k] is not None and b[k] is not None and torch.isfinite(b[k]).all()
            assert torch.allclose(a[k], b[k], rtol=5e-6, atol=0.0), k
    for k in (3, 4, 5, 6, 7, 8):  # the two riding forms reduce the same partial sums in the same order: identical
        x, y = outs[1][k], outs[2][k]
        assert (x == y) if isinstance(x, float) else torch.equal(x, y), k
    assert float(a[5].abs().max()) > 0 and float(a[7].abs().max()) > 0


@pytest.mark.parametrize("n_rays", [1, 4096, 5000])
def test_fused_mse_matches_the_library_formulation(n_rays):
    """pvd_mse_forward (ABI 6): the teacher's objective mean((pred - gt)^2) and its gradient in one launch, against the reference's
    formulation -- MSELoss(reduction='none'), .mean(-1), .mean() (just_train_tea/utils.py:573-581) -- value and gradient, with an
    upstream gradient (the loss scale)."""
    from pvd.losses import mse_fused
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(n_rays)
    pred = torch.rand(1, n_rays, 3, device=dev, generator=g, requires_grad=True)
    gt = torch.rand(1, n_rays, 3, device=dev, generator=g)
    ref_in = pred.detach().clone().requires_grad_(True)
    ref = torch.nn.MSELoss(reduction="none")(ref_in, gt).mean(-1).mean()
    ref.backward(gradient=torch.tensor(65536.0, device=dev))
    out = mse_fused(pred, gt)
    out.backward(gradient=torch.tensor(65536.0, device=dev))
    assert out.shape == ref.shape and abs(float(out) - float(ref)) <= 2e-6 * abs(float(ref))
    assert (pred.grad - ref_in.grad).abs().max().item() <= 2e-6 * ref_in.grad.abs().max().item()
