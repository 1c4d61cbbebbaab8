"""f2 of SURVEY section 8: the inference renderer whose round state (alive count, n_step, rows) stays on the device
(pvd_infer_*, NeRFRenderer._run_rounds_device) against the reference-shaped loop with one `alive_counter.item()` per round
(distill_mutual/renderer.py:450-543; raymarching.cu:704-948): same rays, same weights -> the same image per ray."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(kind):
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.scene import ChairScene
    from pvd.workload import install_occupancy, make_model
    torch.manual_seed(3)
    opt = PVDConfig(model_type=kind, resolution0=64)
    opt.stage_iters = {"stage1": -1, "stage2": -1}
    m = make_model(hip_ops(), opt, kind, False, torch.device(DEV))
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "embeddings" in n:
                p.uniform_(-0.4, 0.4)
            elif p.dim() == 2:
                p.mul_(2.0)
            elif p.dim() == 4:
                p.mul_(2.5)
            elif p.dim() == 5:  # the Plenoxel volume starts at 0.02 randn: densities worth compositing
                p.mul_(150.0)
    install_occupancy(m, ChairScene(), opt)
    return m.eval()


def _rays(n, full_image=False):
    from pvd.scene import BLENDER_INTRINSICS, get_rays, synthetic_poses
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(2))).to(DEV)
    if full_image:  # every pixel of a 200 x 200 view (same field of view): most rays miss the object, some cross all of it
        r = get_rays(poses[9][None], (277.775, 277.775, 100.0, 100.0), 200, 200, -1)
    else:
        r = get_rays(poses[9][None], BLENDER_INTRINSICS, 800, 800, n, generator=torch.Generator(device=DEV).manual_seed(8))
    return r["rays_o"], r["rays_d"]


@pytest.mark.parametrize("kind", ["hash", "vm"])
@pytest.mark.parametrize("full_image", [False, True])
def test_device_rounds_match_the_host_synchronised_loop(kind, full_image, monkeypatch):
    m = _model(kind)
    o, d = _rays(3000, full_image)
    outs = []
    monkeypatch.setenv("PVD_INFER_PERSISTENT", "0")  # (the round loops; the one-launch render of a hash model: below)
    for on_device in ("0", "1"):
        monkeypatch.setenv("PVD_INFER_DEVICE_ROUNDS", on_device)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            out = m.render(o, d, staged=False, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)
        outs.append((out["image"].float(), out["depth"].float()))
    (img0, dep0), (img1, dep1) = outs
    assert m._last_rounds > 3  # the device loop ran (several rounds, no per-round read-back)
    assert torch.isfinite(img1).all() and img0.std().item() > 0.02
    assert (img0 - img1).abs().max().item() <= 1e-5
    # rays that miss the box have near = far = FLT_MAX and a 0 / 0 depth in the reference's normalisation (renderer.py:541)
    assert torch.equal(torch.isnan(dep0), torch.isnan(dep1))
    assert (torch.nan_to_num(dep0) - torch.nan_to_num(dep1)).abs().max().item() <= 1e-5


def test_device_rounds_respect_max_steps_and_empty_images():
    """`while step < max_steps` (renderer.py:483): with a small budget every ray stops after the same number of steps as in
    the host loop; rays that miss the box never enter a round."""
    m = _model("hash")
    o, d = _rays(2048)
    import os
    res = []
    for on_device in ("0", "1"):
        os.environ["PVD_INFER_DEVICE_ROUNDS"] = on_device
        os.environ["PVD_INFER_PERSISTENT"] = "0"
        try:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                a = m.render(o, d, staged=False, bg_color=1, perturb=False, dt_gamma=0, max_steps=24)["image"].float()
                b = m.render(o + 100.0, d, staged=False, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)["image"].float()
        finally:
            os.environ.pop("PVD_INFER_DEVICE_ROUNDS", None)
            os.environ.pop("PVD_INFER_PERSISTENT", None)
        res.append((a, b))
    assert (res[0][0] - res[1][0]).abs().max().item() <= 1e-5
    assert torch.equal(res[1][1], torch.ones_like(res[1][1])) and torch.equal(res[0][1], res[1][1])  # all background


@pytest.mark.parametrize("kind", ["hash", "vm", "tensors"])
@pytest.mark.parametrize("full_image", [False, True])
def test_persistent_hash_render_is_the_round_loops_image(kind, full_image, monkeypatch):
    """pvd_infer_image_hash / pvd_infer_image_vm / pvd_infer_image_plenoxel (ONE persistent launch: ray slots in registers, samples in LDS, the alive queue in
    device memory) against the round loop with the reference's per-round read-back: a ray's samples and sums depend on nothing but the
    ray, so every pixel, every depth and every accumulated weight must be the round loop's -- bit for bit."""
    m = _model(kind)
    o, d = _rays(5000, full_image)
    outs = []
    for persistent in ("0", "1"):
        monkeypatch.setenv("PVD_INFER_PERSISTENT", persistent)
        monkeypatch.setenv("PVD_INFER_DEVICE_ROUNDS", "1" if persistent == "1" else "0")
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            out = m.render(o, d, staged=False, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)
        outs.append((out["image"].float(), out["depth"].float()))
    (img0, dep0), (img1, dep1) = outs
    assert img0.std().item() > 0.02 and torch.isfinite(img1).all()
    assert torch.equal(img0, img1), (img0 - img1).abs().max().item()
    assert torch.equal(torch.isnan(dep0), torch.isnan(dep1)) and torch.equal(torch.nan_to_num(dep0), torch.nan_to_num(dep1))
    # all background when nothing enters the box; a second call on the same model reuses its caches
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        b = m.render(o + 100.0, d, staged=False, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)["image"].float()
        again = m.render(o, d, staged=False, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)["image"].float()
    assert torch.equal(b, torch.ones_like(b)) and torch.equal(again, img1)


def test_persistent_hash_render_with_two_cascades_and_a_growing_step(monkeypatch):
    """bound 2 (two cascades of the occupancy grid) and dt_gamma = 1/256 (configs[4]'s marcher) through the persistent launch"""
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.scene import ChairScene
    from pvd.workload import install_occupancy, make_model
    torch.manual_seed(4)
    opt = PVDConfig(model_type="hash", bound=2.0, dt_gamma=1.0 / 256)
    opt.stage_iters = {"stage1": -1, "stage2": -1}
    m = make_model(hip_ops(), opt, "hash", False, torch.device(DEV))
    with torch.no_grad():
        m.encoder.embeddings.uniform_(-0.4, 0.4)
    install_occupancy(m, ChairScene(scale=1.9), opt)
    m.eval()
    o, d = _rays(4096)
    o = o * 1.9
    outs = []
    for persistent in ("0", "1"):
        monkeypatch.setenv("PVD_INFER_PERSISTENT", persistent)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            outs.append(m.render(o, d, staged=False, bg_color=1, perturb=False, dt_gamma=1.0 / 256, max_steps=1024)["image"].float())
    assert outs[0].std().item() > 0.02 and torch.equal(outs[0], outs[1]), (outs[0] - outs[1]).abs().max().item()


@pytest.mark.parametrize("kind,n_rays,shuffle", [("hash", 1, 7919), ("hash", 63, 7919), ("hash", 2 * 7919, 7919), ("hash", 777, 1),
                                                 ("vm", 1, 7919), ("vm", 63, 7919), ("vm", 2 * 7919, 7919), ("vm", 777, 1),
                                                 ("tensors", 1, 7919), ("tensors", 63, 7919), ("tensors", 2 * 7919, 7919), ("tensors", 777, 1)])
def test_persistent_hash_render_renders_every_ray_once(kind, n_rays, shuffle, monkeypatch):
    """The queue hands rays out through a multiplicative shuffle: it must be a permutation for ANY number of rays (also a multiple of
    the multiplier), and workgroups with fewer rays than slots must terminate."""
    m = _model(kind)
    o, d = _rays(n_rays)
    monkeypatch.setenv("PVD_INFER_SHUFFLE", str(shuffle))
    outs = []
    for persistent in ("0", "1"):
        monkeypatch.setenv("PVD_INFER_PERSISTENT", persistent)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            outs.append(m.render(o, d, staged=False, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)["image"].float())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("kind", ["hash", "vm", "tensors"])
def test_persistent_hash_render_of_a_whole_400x400_view(kind):
    """160 000 rays in one launch; the ones that meet an occupied cell are queued and rendered by several hundred workgroups"""
    from pvd.scene import get_rays, synthetic_poses
    import os
    m = _model(kind)
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(2))).to(DEV)
    r = get_rays(poses[40][None], (555.55, 555.55, 200.0, 200.0), 400, 400, -1)
    outs = []
    for persistent in ("0", "1"):
        os.environ["PVD_INFER_PERSISTENT"] = persistent
        try:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                out = m.render(r["rays_o"], r["rays_d"], staged=False, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)
        finally:
            os.environ.pop("PVD_INFER_PERSISTENT", None)
        outs.append((out["image"].float(), torch.nan_to_num(out["depth"].float())))
    assert outs[0][0].std().item() > 0.02
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    st = m._last_infer_workspace[-10:-6].tolist()
    assert st[3] >= 100 and st[1] > 100000 and int(m._last_infer_workspace[0]) > 64 * 100  # workgroups that took rays; rows shaded; rays queued


@pytest.mark.parametrize("kind", ["hash", "vm", "tensors"])
def test_persistent_render_with_a_small_step_budget_differs_from_the_rounds_only_as_documented(kind, monkeypatch):
    """with max_steps small enough for rays to hit it the two loops stop differently BY DESIGN (include/pvd_hip.h): the
    reference's round loop stops ALL rays once the rounds' n_step add up to max_steps (and overshoots by up to n_step - 1), the
    persistent render stops a ray after ITS OWN max_steps samples.  Rays that end before either cap must agree bit for bit; every
    other pixel of the persistent render must equal a round-loop render whose budget no ray reaches, composited over the first
    max_steps samples -- checked here through monotonicity: with a larger cap the persistent pixel moves towards the uncapped one."""
    m = _model(kind)
    o, d = _rays(3000)
    def render(persistent, max_steps):
        monkeypatch.setenv("PVD_INFER_PERSISTENT", persistent)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return m.render(o, d, staged=False, bg_color=1, perturb=False, dt_gamma=0, max_steps=max_steps)["image"].float()
    full_r, full_p = render("0", 1024), render("1", 1024)
    assert torch.equal(full_r, full_p)
    cap_r, cap_p = render("0", 24), render("1", 24)
    done_early = (cap_r == full_r).all(-1) & (cap_p == full_p).all(-1)  # rays neither cap touched
    assert done_early.float().mean().item() > 0.3 and (~done_early).any()
    assert torch.equal(cap_r[done_early], cap_p[done_early])
    # a capped ray has composited a prefix of its samples: its accumulated opacity is below the uncapped one's (white background:
    # the pixel is at least as bright in every channel where the object is darker than white) -- for both loops
    for cap in (cap_r, cap_p):
        assert ((cap - full_r)[~done_early].abs().max().item()) > 0
    assert torch.isfinite(cap_p).all()


def test_persistent_plenoxel_render_without_autocast_and_with_degree_2(monkeypatch):
    """The Plenoxel model is fp32 with or without autocast (network.py:383-409), so its persistent render does not depend on it; SH degree 2
    (C = 13 channels: the half-wave's idle lanes) through the same launch."""
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.scene import ChairScene
    from pvd.workload import install_occupancy, make_model
    o, d = _rays(3000)
    for degree in (3, 2):
        torch.manual_seed(5)
        opt = PVDConfig(model_type="tensors", plenoxel_degree=degree, plenoxel_res="[48,64,80]")
        opt.stage_iters = {"stage1": -1, "stage2": -1}
        m = make_model(hip_ops(), opt, "tensors", False, torch.device(DEV))
        with torch.no_grad():
            m.tensor_volume[0].mul_(150.0)
        install_occupancy(m, ChairScene(), opt)
        m.eval()
        assert m.tensor_volume[0].shape[1] == 3 * degree * degree + 1
        outs = []
        for persistent in ("0", "1"):
            monkeypatch.setenv("PVD_INFER_PERSISTENT", persistent)
            with torch.no_grad():
                outs.append(m.render(o, d, staged=False, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)["image"].float())
        assert outs[0].std().item() > 0.02 and torch.equal(outs[0], outs[1]), (degree, (outs[0] - outs[1]).abs().max().item())
