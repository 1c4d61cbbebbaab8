"""f4 of SURVEY section 8: reference-format checkpoints with the reference's loading rules (pvd/checkpoint.py) and the
Blender-format scene reader (pvd/provider.py).  CPU: the oracle operator set.  The `refnet_*` fixtures are what the
REFERENCE's own NeRFNetwork computed here (tests/golden/make_golden.py): state-dict, inputs, forward outputs, parameter
gradients -- so loading them and reproducing the numbers pins this repo's networks against the reference's torch code."""
import json
import os
import types

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_python.npz"), allow_pickle=False)


def small_opt(model_type):
    from pvd.config import PVDConfig
    return PVDConfig(model_type=model_type, teacher_type=model_type, PE=6, skip=2, nerf_layer_num=5, nerf_layer_wide=32, resolution0=12,
                     grid_size=16, density_thresh=10.0, fp16=False, stage_iters={"stage1": 2000, "stage2": 5000}, global_step=10 ** 6)


def reference_checkpoint(mt, path):
    """A .pth file exactly as the reference's Trainer.save_checkpoint lays it out (utils.py:1405-1447), holding the state-dict
    the reference's NeRFNetwork had when it produced the refnet_* numbers."""
    pre = "refnet_%s__" % mt
    model = {}
    for k in [str(k) for k in G[pre + "keys"]]:
        if "embeddings" in k:
            torch.manual_seed(777)  # 42 MB table: regenerated, see make_golden.py
            n_rows = int(G["grid0_offsets"][-1])
            model[k] = (torch.rand(n_rows, 2) - 0.5) * 0.6
        else:
            model[k] = torch.from_numpy(G[pre + "sd__" + k])
    ckpt = {"epoch": 3, "global_step": 1234, "stats": {"loss": [], "valid_loss": [], "results": [], "checkpoints": [], "best_result": None},
            "mean_count": 4321, "mean_density": 0.25, "model": model}
    if mt == "vm":
        ckpt["resolution"] = [12, 12, 12]
    torch.save(ckpt, path)
    return ckpt


def build(ops, mt, device="cpu"):
    from pvd.workload import make_model
    torch.manual_seed(1)
    return make_model(ops, small_opt(mt), mt, False, torch.device(device))


def check_against_reference(net, mt, device, fwd_tol, grad_tol):
    pre = "refnet_%s__" % mt
    net.train()
    net.args.global_step = 10 ** 6
    x, d = torch.from_numpy(G["refnet_x"]).to(device), torch.from_numpy(G["refnet_d"]).to(device)
    sigma, color = net(x, d)
    fea = net.feature_sigma_color
    for got, name in ((sigma, "sigma"), (color, "color"), (fea, "feature_sigma_color")):
        ref = G[pre + name]
        err = np.abs(got.detach().float().cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
        assert err <= fwd_tol, (mt, name, err)
    loss = (sigma * torch.from_numpy(G[pre + "g_sigma"]).to(device)).sum() + (color * torch.from_numpy(G[pre + "g_color"]).to(device)).sum() \
        + (fea * torch.from_numpy(G[pre + "g_fea"]).to(device)).sum()
    loss.backward()
    for n, p in net.named_parameters():
        if "embeddings" in n:
            rows = torch.from_numpy(G[pre + "grad_rows__" + n])
            ref = G[pre + "grad_vals__" + n]
            got = p.grad.detach().float().cpu()
            assert np.abs(got[rows].numpy() - ref).max() <= grad_tol * np.abs(ref).max(), (mt, n)
            mask = torch.ones(got.shape[0], dtype=torch.bool)
            mask[rows] = False
            assert got[mask].abs().max().item() <= grad_tol * np.abs(ref).max()
            continue
        ref = G[pre + "grad__" + n]
        got = p.grad.detach().float().cpu().numpy()
        assert got.shape == ref.shape, n
        assert np.abs(got - ref).max() <= grad_tol * max(np.abs(ref).max(), 1e-12), (mt, n, np.abs(got - ref).max(), np.abs(ref).max())
    with torch.no_grad():
        dens = net.density(x)["sigma"].float().cpu().numpy()
    ref = G[pre + "density_sigma"].reshape(-1)
    assert np.abs(dens.reshape(-1) - ref).max() <= fwd_tol * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("mt", ["vm", "mlp", "hash"])
def test_reference_checkpoint_loads_and_reproduces_the_reference_network(mt, tmp_path):
    from oracle_ops import oracle_ops
    from pvd.checkpoint import load_teacher_checkpoint
    path = str(tmp_path / ("ref_%s.pth" % mt))
    reference_checkpoint(mt, path)
    net = build(oracle_ops(), mt)
    epoch0 = net.occ_epoch
    missing, unexpected = load_teacher_checkpoint(net, path)
    assert missing == [] and unexpected == []
    assert net.mean_count == 4321 and net.mean_density == 0.25 and net.occ_epoch > epoch0
    if mt == "vm":
        assert net.sigma_mat[0].stride(1) == 1  # still channels-last after loading a channel-major file
    check_against_reference(net, mt, "cpu", fwd_tol=2e-6, grad_tol=2e-5)


def test_checkpoint_round_trip_and_student_inherits_from_the_teacher_file(tmp_path):
    """save -> load is the identity (and the file holds channel-major tables, as the reference's would); a VM student with no
    checkpoint of its own starts from the TEACHER's file (utils.py:1529-1537): occupancy buffers, step counter, aabb and the
    colour head carry over, its own tables stay, the teacher-only tensors are reported as unexpected."""
    from oracle_ops import oracle_ops
    from pvd.checkpoint import load_student_checkpoint, load_teacher_checkpoint, save_checkpoint
    ops = oracle_ops()
    tea = build(ops, "hash")
    with torch.no_grad():
        tea.density_grid.uniform_(0, 20)
        tea.density_bitfield.copy_(torch.randint(0, 256, tea.density_bitfield.shape, dtype=torch.uint8))
        tea.step_counter.copy_(torch.arange(32, dtype=torch.int32).view(16, 2))
    tea.mean_count, tea.mean_density = 777, 1.5
    p_tea = save_checkpoint(str(tmp_path / "tea.pth"), tea, epoch=7, global_step=99)
    raw = torch.load(p_tea, weights_only=False)
    assert sorted(raw.keys()) == ["epoch", "global_step", "mean_count", "mean_density", "model", "stats"]
    assert raw["epoch"] == 7 and raw["global_step"] == 99 and sorted(raw["model"].keys()) == sorted(tea.state_dict().keys())
    tea2 = build(ops, "hash")
    assert load_teacher_checkpoint(tea2, p_tea) == ([], [])
    for (k, a), (_, b) in zip(tea.state_dict().items(), tea2.state_dict().items()):
        assert torch.equal(a, b), k
    assert tea2.mean_count == 777 and tea2.mean_density == 1.5

    stu = build(ops, "vm")
    own = stu.sigma_mat[1].detach().clone()
    missing, unexpected = load_student_checkpoint(stu, p_tea)
    assert sorted(unexpected) == ["encoder.embeddings", "encoder.offsets", "sigma_net.0.weight", "sigma_net.1.weight"]
    assert all(k.split(".")[0] in ("sigma_mat", "sigma_vec", "color_mat", "color_vec", "basis_mat") for k in missing) and len(missing) == 13
    for k in ("density_grid", "density_bitfield", "step_counter", "aabb_train", "aabb_infer", "color_net.0.weight", "color_net.2.weight"):
        assert torch.equal(stu.state_dict()[k], tea.state_dict()[k]), k
    assert torch.equal(stu.sigma_mat[1], own) and stu.mean_count == 777

    # a VM file: channel-major on disk, channels-last in the model, resolution restored through the resampler
    p_stu = save_checkpoint(str(tmp_path / "stu.pth"), stu)
    raw = torch.load(p_stu, weights_only=False)
    assert raw["resolution"] == [12, 12, 12] and raw["model"]["color_mat.0"].is_contiguous()
    from pvd.config import PVDConfig
    from pvd.workload import make_model
    big = make_model(ops, PVDConfig(**{**small_opt("vm").__dict__, "resolution0": 20}), "vm", False, torch.device("cpu"))
    assert big.sigma_mat[0].shape[-1] == 20
    load_student_checkpoint(big, p_tea, ckpt_student=p_stu)
    assert big.resolution == [12, 12, 12] and big.sigma_mat[0].shape == stu.sigma_mat[0].shape and big.sigma_mat[0].stride(1) == 1
    x, d = torch.from_numpy(G["refnet_x"]), torch.from_numpy(G["refnet_d"])
    stu.eval(); big.eval()
    with torch.no_grad():
        (s1, c1), (s2, c2) = stu(x, d), big(x, d)
    assert torch.equal(s1, s2) and torch.equal(c1, c2)


def _write_scene(root, n_frames=3, H=12, W=16, alpha=True, with_hw=False):
    from PIL import Image
    from pvd.scene import pose_spherical
    rs = np.random.RandomState(0)
    frames, images, mats = [], [], []
    for k in range(n_frames):
        img = rs.randint(0, 256, size=(H, W, 4 if alpha else 3)).astype(np.uint8)
        Image.fromarray(img, "RGBA" if alpha else "RGB").save(os.path.join(root, "r_%d.png" % k))
        m = pose_spherical(30.0 * k - 40, -20.0 - 5 * k, 4.0)
        frames.append({"file_path": "./r_%d" % k, "transform_matrix": m.tolist()})
        images.append(img)
        mats.append(m)
    t = {"camera_angle_x": 0.6911112070083618, "frames": frames + [{"file_path": "./missing", "transform_matrix": np.eye(4).tolist()}]}
    if with_hw:
        t.update(h=H, w=W)
    for split in ("train", "val"):
        with open(os.path.join(root, "transforms_%s.json" % split), "w") as f:
            json.dump(t, f)
    return np.stack(images), np.stack(mats)


@pytest.mark.parametrize("alpha", [True, False])
def test_blender_scene_reader(tmp_path, alpha):
    from pvd.provider import BlenderScene, training_target
    from pvd.scene import get_rays, nerf_matrix_to_ngp
    root = str(tmp_path)
    images, mats = _write_scene(root, alpha=alpha)
    sc = BlenderScene(root, "train", scale=0.8, num_rays=40)
    assert len(sc) == 3 and (sc.H, sc.W) == (12, 16)  # the frame whose file does not exist is skipped (provider.py:200-201)
    assert np.array_equal(sc.images.numpy(), images.astype(np.float32) / 255)
    assert np.array_equal(sc.poses.numpy(), np.stack([nerf_matrix_to_ngp(m, 0.8) for m in mats]))
    f = 16 / (2 * np.tan(0.6911112070083618 / 2))
    np.testing.assert_allclose(sc.intrinsics, [f, f, 6.0, 8.0])  # cx = H/2, cy = W/2: the reference's defaults (:273-274)
    g = torch.Generator().manual_seed(5)
    b = sc.batch([1], generator=g)
    assert b["rays_o"].shape == (1, 40, 3) and b["images"].shape == (1, 40, images.shape[-1])
    flat = torch.from_numpy(images[1].astype(np.float32) / 255).view(-1, images.shape[-1])
    assert torch.equal(b["images"][0], flat[b["inds"][0]])
    r = get_rays(sc.poses[1:2], tuple(sc.intrinsics), 12, 16, 40, inds=b["inds"][0])
    assert torch.equal(r["rays_d"], b["rays_d"])
    gt, bg = training_target(b["images"], generator=torch.Generator().manual_seed(1))
    if alpha:
        a = b["images"][..., 3:]
        assert torch.allclose(gt, b["images"][..., :3] * a + bg * (1 - a)) and bg.shape == (1, 40, 3)
        # with a background model: bg_color = 1, and the RGBA target is STILL blended by alpha (over white)
        gt1, bg1 = training_target(b["images"], bg_radius=32.0)
        assert bg1 == 1 and torch.allclose(gt1, b["images"][..., :3] * a + (1 - a))
    else:
        assert bg == 1 and torch.equal(gt, b["images"])
    # evaluation split: whole images, all rays in pixel order; trainval = both files
    val = BlenderScene(root, "val", scale=0.8)
    bv = val.batch([2])
    assert bv["rays_o"].shape == (1, 12 * 16, 3) and bv["images"].shape == (1, 12, 16, images.shape[-1])
    assert len(BlenderScene(root, "trainval")) == 6 and len(BlenderScene(root, "all")) == 6
    # downscale: area-averaged images, intrinsics from the new size
    half = BlenderScene(root, "train", downscale=2)
    assert (half.H, half.W) == (6, 8)
    blocks = (images.astype(np.float32) / 255).reshape(3, 6, 2, 8, 2, -1).mean(axis=(2, 4))
    assert np.abs(half.images.numpy() - blocks).max() <= 1.01 / 255  # 8-bit result: the rounding of the average


def test_error_map_sampling_loop_feeds_back(tmp_path):
    """--error_map end to end on the provider side (the update used to be unwired): batches carry `index` / `inds_coarse`,
    BlenderScene.update_error puts the EMA rows BACK into the frame's map (utils.py:1120-1129), untouched frames and cells keep
    their weights, and the next draw of that frame follows the updated map."""
    from pvd.provider import BlenderScene
    root = str(tmp_path)
    _write_scene(root, alpha=True, H=256, W=256)
    sc = BlenderScene(root, "train", scale=0.8, num_rays=64, error_map=True)
    assert sc.error_map.shape == (3, 128 * 128) and float(sc.error_map.min()) == 1.0
    g = torch.Generator().manual_seed(3)
    b = sc.batch([1], generator=g)
    assert b["index"] == [1] and b["inds_coarse"].shape == (1, 64)
    err = torch.zeros(1, 64)  # "these rays are already perfect": their cells drop to 0.1, everything else keeps weight 1
    sc.update_error(b, err)
    row = sc.error_map[1]
    assert torch.allclose(row[b["inds_coarse"][0]], torch.full((64,), 0.1)) and int((row != 1.0).sum()) == len(set(b["inds_coarse"][0].tolist()))
    assert float(sc.error_map[0].min()) == 1.0 and float(sc.error_map[2].min()) == 1.0
    # a very wrong batch on the same frame: its cells now weigh 0.1 * old + 0.9 * 50
    b2 = sc.batch([1], generator=g)
    sc.update_error(b2, torch.full((1, 64), 50.0))
    hot = set(b2["inds_coarse"][0].tolist())
    assert all(float(sc.error_map[1, c]) > 40.0 for c in hot)
    # ... and the next draw of that frame concentrates on them (64 draws without replacement from weights ~45 vs ~1 over 16 k cells)
    b3 = sc.batch([1], generator=g)
    assert len(hot & set(b3["inds_coarse"][0].tolist())) >= 8
