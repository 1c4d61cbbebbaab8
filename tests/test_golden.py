"""Fixtures generated from the reference's own Python (tests/golden/make_golden.py) vs this repo's
host-side restatements.  CPU only."""
import os
import types

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_python.npz"), allow_pickle=False)


def test_grid_offsets_tables():
    from oracle_ops import OracleGridEncoder
    cfgs = [eval(c) for c in G["grid_cfgs"]]
    for i, c in enumerate(cfgs):
        e = OracleGridEncoder(**c)
        assert np.array_equal(e.offsets.numpy(), G["grid%d_offsets" % i]), c
        assert float(e.per_level_scale) == float(G["grid%d_pls" % i])
        assert tuple(e.embeddings.shape) == tuple(G["grid%d_shape" % i])
    # the table the whole benchmark runs on
    assert G["grid0_offsets"].tolist()[:7] == [0, 4920, 20552, 63432, 196088, 585112, 1109400] and G["grid0_offsets"][-1] == 5303704


def test_grid_wrapper_logic_matches_reference_wrapper():
    from oracle_ops import OracleGridEncoder
    e = OracleGridEncoder(**eval(str(G["gw_cfg"])))
    e.embeddings.data.copy_(torch.from_numpy(G["gw_emb"]))
    y = e(torch.from_numpy(G["gw_x"]), bound=1)
    assert np.array_equal(y.detach().numpy(), G["gw_y"])
    y.backward(torch.from_numpy(G["gw_g"]))
    assert np.array_equal(e.embeddings.grad.numpy(), G["gw_gemb"])
    assert np.array_equal(e(torch.from_numpy(G["gw_xb"]), bound=2).detach().numpy(), G["gw_yb"])


def test_sh_wrapper_logic_matches_reference_wrapper():
    from oracle_ops import OracleSHEncoder
    sh = OracleSHEncoder(input_dim=3, degree=4)
    d = torch.from_numpy(G["sh_d"]).requires_grad_(True)
    y = sh(d)
    assert np.array_equal(y.detach().numpy(), G["sh_y"])
    y.backward(torch.from_numpy(G["sh_g"]))
    assert np.array_equal(d.grad.numpy(), G["sh_gd"])


def test_trunc_exp():
    from pvd.activation import make_trunc_exp
    te = make_trunc_exp("cpu")
    x = torch.from_numpy(G["te_x"]).requires_grad_(True)
    y = te(x)
    assert np.array_equal(y.detach().numpy(), G["te_y"])
    y.backward(torch.from_numpy(G["te_g"]))
    assert np.array_equal(x.grad.numpy(), G["te_gx"])


@pytest.mark.parametrize("mr", [10, 2, 6])
def test_freq_encoder(mr):
    from pvd.encoding import FreqEncoder
    fe = FreqEncoder(input_dim=3, max_freq_log2=mr - 1, N_freqs=mr)
    assert fe.output_dim == int(G["freq%d_dim" % mr])
    assert np.array_equal(fe(torch.from_numpy(G["freq%d_x" % mr])).numpy(), G["freq%d_y" % mr])


@pytest.mark.parametrize("mt", ["hash", "mlp", "vm"])
def test_network_state_dict_layout(mt):
    """Reference checkpoints must load: same keys, shapes and dtypes (SURVEY.md section 5 checkpoint row)."""
    from oracle_ops import oracle_ops
    from pvd.config import PVDConfig
    from pvd.network import NeRFNetwork
    opt = PVDConfig(model_type=mt)
    net = NeRFNetwork(oracle_ops(), model_type=mt, args=opt, bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10,
                      bg_radius=-1, grid_size=128)
    sd = net.state_dict()
    keys = sorted(sd.keys())
    assert keys == list(G["net_%s_keys" % mt])
    assert [repr(tuple(sd[k].shape)) for k in keys] == list(G["net_%s_shapes" % mt])
    assert [str(sd[k].dtype) for k in keys] == list(G["net_%s_dtypes" % mt])


def test_cameras_and_rays():
    from pvd.scene import get_rays, nerf_matrix_to_ngp, pose_spherical
    sph = np.stack([pose_spherical(th, ph, r) for th, ph, r in ((30.0, -20.0, 4.0), (-170.0, -5.0, 4.0), (0.0, -89.0, 3.0))])
    np.testing.assert_allclose(sph, G["pose_sph"], atol=1e-6)
    ngp = np.stack([nerf_matrix_to_ngp(p, 0.8) for p in G["pose_sph"]])
    assert np.array_equal(ngp, G["pose_ngp"])
    poses = torch.from_numpy(G["pose_ngp"])
    r = get_rays(poses, (1111.1, 1111.1, 24.0, 20.0), 40, 48, -1)
    np.testing.assert_allclose(r["rays_d"].numpy(), G["rays_d"], atol=1e-7)
    assert np.array_equal(r["rays_o"].numpy(), G["rays_o"])
    r = get_rays(poses[:1], (1111.1, 1111.1, 400.0, 400.0), 800, 800, 64, inds=torch.from_numpy(G["rays_n_inds"][0]))
    np.testing.assert_allclose(r["rays_d"].numpy(), G["rays_n_d"], atol=1e-7)
    # and the un-seeded draw is the same torch.randint call
    torch.manual_seed(3)
    r2 = get_rays(poses[:1], (1111.1, 1111.1, 400.0, 400.0), 800, 800, 64)
    assert np.array_equal(r2["inds"].numpy(), G["rays_n_inds"])


def test_get_rays_error_map_sampling_is_the_references_draw():
    """--error_map (utils.py:357-381): the reference's own get_rays with a [2, 128*128] error map, seeded; the same seed here must
    give the same cells, the same pixels and the same rays, and no draw may land in a zero-weight cell."""
    from pvd.scene import get_rays, update_error_map
    poses = torch.from_numpy(G["pose_ngp"])[:2]
    emap = torch.from_numpy(G["rays_e_map"])
    torch.manual_seed(11)
    assert torch.equal(torch.rand(2, 128 * 128) ** 4 * (emap > 0), emap)  # (the fixture's map, and the generator state behind it)
    r = get_rays(poses, (1111.1, 1111.1, 400.0, 400.0), 800, 800, 96, error_map=emap)
    assert np.array_equal(r["inds_coarse"].numpy(), G["rays_e_inds_coarse"]) and np.array_equal(r["inds"].numpy(), G["rays_e_inds"])
    np.testing.assert_allclose(r["rays_d"].numpy(), G["rays_e_d"], atol=1e-7)
    assert np.array_equal(r["rays_o"].numpy(), G["rays_e_o"])
    assert (G["rays_e_inds_coarse"][0] >= 4000).all()  # cells 0..3999 of image 0 carry weight 0
    # a drawn pixel lies inside its cell (800 / 128 = 6.25 pixels per cell side)
    px, py = r["inds"] // 800, r["inds"] % 800
    cx, cy = r["inds_coarse"] // 128, r["inds_coarse"] % 128
    assert ((px >= (cx * 6.25).long()) & (px <= ((cx + 1) * 6.25).long())).all() and ((py >= (cy * 6.25).long()) & (py <= ((cy + 1) * 6.25).long())).all()
    # the EMA update of the sampled cells (utils.py:1120-1129)
    err = torch.rand(2, 96)
    before = emap.clone()
    update_error_map(emap, r["inds_coarse"], err)
    want = 0.1 * before.gather(1, r["inds_coarse"]) + 0.9 * err
    assert torch.equal(emap.gather(1, r["inds_coarse"]), want)
    mask = torch.ones_like(emap, dtype=torch.bool).scatter_(1, r["inds_coarse"], False)
    assert torch.equal(emap[mask], before[mask])


def test_composite_wrapper_logic_matches_reference_wrapper():
    """comp_*: the reference's composite_rays_train autograd wrapper (raymarching/raymarching.py:292-357) run on the
    oracle backend vs this repo's wrapper on the same backend: allocation, saved tensors, ignored depth gradient."""
    from oracle_ops import oracle_ops
    rm = oracle_ops().raymarching
    sig = torch.from_numpy(G["comp_sigmas"]).requires_grad_(True)
    rgb = torch.from_numpy(G["comp_rgbs"]).requires_grad_(True)
    ws, dep, img = rm.composite_rays_train(sig, rgb, torch.from_numpy(G["comp_deltas"]), torch.from_numpy(G["comp_rays"]))
    assert np.array_equal(ws.detach().numpy(), G["comp_ws"])
    assert np.array_equal(dep.detach().numpy(), G["comp_depth"])
    assert np.array_equal(img.detach().numpy(), G["comp_image"])
    torch.autograd.backward([ws, dep, img], [torch.from_numpy(G["comp_g_ws"]), torch.from_numpy(G["comp_g_depth"]),
                                             torch.from_numpy(G["comp_g_image"])])
    assert np.array_equal(sig.grad.numpy(), G["comp_g_sigmas"])
    assert np.array_equal(rgb.grad.numpy(), G["comp_g_rgbs"])


# ------------------------------------------------------------------ random distillation cameras (--data_type synthetic | tank | llff)
def test_rand_pose_generators_reproduce_the_reference():
    """tests/golden/reference_poses.npz: the reference's get_rand_poses (distill_mutual/utils.py:100-197) run with a seeded
    np.random (tests/golden/make_golden_poses.py); a RandomState with the same seed draws the same stream."""
    from pvd.scene import rand_poses
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_poses.npz"))
    for seed in (0, 7):
        for kind in ("synthetic", "tank", "llff"):
            orig = ref["llff_original_seed%d" % seed] if kind == "llff" else None
            got = rand_poses(kind, np.random.RandomState(seed), original_poses=orig)
            want = ref["%s_seed%d" % (kind, seed)]
            assert got.shape == want.shape and got.dtype == np.float32, (kind, got.shape, want.shape)
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-6, err_msg=kind)
    with pytest.raises(ValueError):
        rand_poses("blender", np.random.RandomState(0))
