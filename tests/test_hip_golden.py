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
