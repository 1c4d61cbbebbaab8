"""The golden fixtures (tests/golden/reference_python.npz: inputs + expected outputs produced by the REFERENCE's own
Python -- its GridEncoder / SHEncoder / composite_rays_train wrappers and get_rays -- see tests/golden/make_golden.py)
fed to the PRODUCT operators on the GPU, through the public packages and the C ABI.  The CPU consumers of the same
fixtures are in tests/test_golden.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_python.npz"), allow_pickle=False)
DEV = "cuda:0"


def _t(name, **kw):
    return torch.from_numpy(G[name]).to(DEV, **kw)


def test_grid_encoder_on_the_reference_wrapper_fixture():
    """gw_*: reference GridEncoder wrapper forward / backward (fp32 table).  Forward is bit-exact; the backward's
    atomics only change the summation order."""
    from gridencoder import GridEncoder
    e = GridEncoder(**eval(str(G["gw_cfg"]))).to(DEV)
    e.embeddings.data.copy_(_t("gw_emb"))
    y = e(_t("gw_x"), bound=1)
    assert np.array_equal(y.detach().cpu().numpy(), G["gw_y"])
    y.backward(_t("gw_g"))
    ref = G["gw_gemb"]
    err = np.abs(e.embeddings.grad.cpu().numpy() - ref).max()
    assert err <= 2e-5 * np.abs(ref).max(), err
    assert np.array_equal(e(_t("gw_xb"), bound=2).detach().cpu().numpy(), G["gw_yb"])


def test_grid_encoder_half_table_under_autocast_on_the_fixture():
    """Same fixture under fp16 autocast (table cast to half per call, grid.py:49-52): compared with the fp32 answer at
    half precision -- the bit-exact half comparison against the oracle is tests/test_hip_parity.py."""
    from gridencoder import GridEncoder
    e = GridEncoder(**eval(str(G["gw_cfg"]))).to(DEV)
    e.embeddings.data.copy_(_t("gw_emb"))
    with torch.autocast("cuda", dtype=torch.float16):
        y = e(_t("gw_x"), bound=1)
    assert y.dtype == torch.float16
    ref = G["gw_y"]
    assert np.abs(y.detach().float().cpu().numpy() - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max())


def test_sh_encoder_on_the_reference_wrapper_fixture():
    from shencoder import SHEncoder
    sh = SHEncoder(input_dim=3, degree=4)
    d = _t("sh_d").requires_grad_(True)
    y = sh(d)
    assert np.abs(y.detach().cpu().numpy() - G["sh_y"]).max() <= 3e-6
    y.backward(_t("sh_g"))
    assert np.abs(d.grad.cpu().numpy() - G["sh_gd"]).max() <= 5e-5 * max(1.0, np.abs(G["sh_gd"]).max())


def test_get_rays_kernel_on_the_reference_fixture():
    """rays_* / rays_n_*: the reference's get_rays (utils.py:324-404), all pixels of a 40x48 image and 64 drawn pixels."""
    import pvd_hip
    poses = _t("pose_ngp").float().contiguous()
    H, W = 40, 48
    for p in range(poses.shape[0]):
        o = torch.empty(H * W, 3, device=DEV)
        d = torch.empty(H * W, 3, device=DEV)
        pvd_hip.get_rays(poses[p].contiguous(), 1111.1, 1111.1, 24.0, 20.0, None, W, H * W, o, d)
        assert np.array_equal(o.cpu().numpy(), G["rays_o"][p])
        np.testing.assert_allclose(d.cpu().numpy(), G["rays_d"][p], atol=1e-7)
    inds = torch.from_numpy(G["rays_n_inds"][0]).to(DEV)
    o = torch.empty(64, 3, device=DEV)
    d = torch.empty(64, 3, device=DEV)
    pvd_hip.get_rays(poses[0].contiguous(), 1111.1, 1111.1, 400.0, 400.0, inds, 800, 64, o, d)
    assert np.array_equal(o.cpu().numpy(), G["rays_n_o"][0])
    np.testing.assert_allclose(d.cpu().numpy(), G["rays_n_d"][0], atol=1e-7)


def test_composite_rays_train_on_the_reference_wrapper_fixture():
    """comp_*: the reference's composite_rays_train autograd wrapper (raymarching.py:292-357), forward + backward:
    shuffled ray rows, an empty ray, a ray longer than a wavefront, an overflowing ray, ignored depth gradient."""
    import raymarching
    sig = _t("comp_sigmas").requires_grad_(True)
    rgb = _t("comp_rgbs").requires_grad_(True)
    ws, dep, img = raymarching.composite_rays_train(sig, rgb, _t("comp_deltas"), _t("comp_rays"))
    for got, name in ((ws, "comp_ws"), (dep, "comp_depth"), (img, "comp_image")):
        ref = G[name]
        assert np.abs(got.detach().cpu().numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), name
    torch.autograd.backward([ws, dep, img], [_t("comp_g_ws"), _t("comp_g_depth"), _t("comp_g_image")])
    for got, name in ((sig.grad, "comp_g_sigmas"), (rgb.grad, "comp_g_rgbs")):
        ref = G[name]
        assert np.abs(got.cpu().numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), name
    # nothing leaks into the slack after the last ray or into the overflowing ray's rows
    M_used = int(G["comp_rays"][:, 2].sum()) - 5
    assert (sig.grad[M_used:] == 0).all() and (rgb.grad[M_used:] == 0).all()


def test_polar_from_ray_matches_oracle():
    """pvd_polar_from_ray (raymarching.cu:164-200) vs the oracle, through the package and the C ABI."""
    import oracle
    import raymarching
    rs = np.random.RandomState(3)
    o = (rs.uniform(-1, 1, size=(4096, 3)) * 0.5).astype(np.float32)
    d = rs.standard_normal((4096, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[0] = (0, 1, 0); d[1] = (1, 0, 0); d[2] = (0, 0, 1); d[3] = (0, -1, 0)
    o[:4] = 0
    for radius in (2.0, 3.2):
        got = raymarching.polar_from_ray(torch.from_numpy(o).to(DEV), torch.from_numpy(d).to(DEV), radius).cpu().numpy()
        ref = oracle.polar_from_ray(o, d, radius)
        assert got.shape == (4096, 2)
        # device atan2f / sqrtf vs libm: a few ulp of pi-normalised angles
        assert np.abs(got - ref).max() <= 2e-6, np.abs(got - ref).max()
    assert np.allclose(got[:4], [[-1, 0], [0, 0], [0, 0.5], [1, 0]], atol=1e-6)


@pytest.mark.parametrize("mt", ["vm", "mlp", "hash"])
def test_reference_checkpoint_on_the_hip_path(mt, tmp_path):
    """A reference-format .pth holding the state-dict of the REFERENCE's NeRFNetwork (refnet_* fixtures), loaded with the
    reference's rules into the HIP-backed model: forward, density and the whole parameter gradient must reproduce what the
    reference's own torch code computed (fp32: the fused VM lookup / grid encoder / SH kernels + torch linears)."""
    from pvd.checkpoint import load_teacher_checkpoint
    from pvd.ops import hip_ops
    from test_checkpoint_provider import build, check_against_reference, reference_checkpoint
    path = str(tmp_path / ("ref_%s.pth" % mt))
    reference_checkpoint(mt, path)
    net = build(hip_ops(), mt, DEV)
    assert load_teacher_checkpoint(net, path) == ([], [])
    if mt == "vm":
        assert net.sigma_mat[0].stride(1) == 1 and net.sigma_mat[0].is_cuda
    check_against_reference(net, mt, DEV, fwd_tol=2e-5, grad_tol=2e-4)


def test_vm_checkpoint_round_trip_renders_bit_identically():
    """Save a trained-looking VM model in the reference's (channel-major) file layout, load it into a fresh model
    (channels-last storage): the two render the same image bit for bit, training branch and inference rounds."""
    import io

    from pvd.checkpoint import checkpoint_dict, _load_model
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.scene import BLENDER_INTRINSICS, ChairScene, get_rays, synthetic_poses
    from pvd.workload import install_occupancy, make_model
    opt = PVDConfig(model_type="vm", resolution0=64, fp16=True)
    opt.stage_iters = {"stage1": -1, "stage2": -1}
    torch.manual_seed(4)
    src = make_model(hip_ops(), opt, "vm", False, torch.device(DEV))
    install_occupancy(src, ChairScene(), opt)
    buf = io.BytesIO()
    torch.save(checkpoint_dict(src, epoch=1, global_step=10), buf)
    buf.seek(0)
    ckpt = torch.load(buf, map_location=DEV, weights_only=False)
    assert ckpt["model"]["sigma_mat.0"].is_contiguous() and ckpt["resolution"] == [64, 64, 64]
    torch.manual_seed(5)
    dst = make_model(hip_ops(), opt, "vm", False, torch.device(DEV))
    assert _load_model(dst, ckpt) == ([], [])
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(1))).to(DEV)
    r = get_rays(poses[7][None], BLENDER_INTRINSICS, 800, 800, 2048, generator=torch.Generator(device=DEV).manual_seed(2))
    for training in (True, False):
        imgs = []
        for m in (src, dst):
            m.train(training)
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                out = m.render(r["rays_o"], r["rays_d"], staged=False, bg_color=1, perturb=False, force_all_rays=True, dt_gamma=0, max_steps=1024)
            imgs.append(out["image"])
        assert torch.equal(imgs[0], imgs[1]) and imgs[0].float().std().item() > 0.01
