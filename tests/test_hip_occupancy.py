"""Device-side occupancy-grid maintenance (csrc/occupancy.hip, pvd_occ_*) against the torch formulation of
NeRFRenderer.update_extra_state (distill_mutual/renderer.py:647-775): cell selection and jitter as distributions (the
reference draws from torch's generator), the running-maximum update, mean / threshold and packbits exactly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")
H = 64
H3 = H ** 3


def _grid(seed=0, frac=0.07):
    g = torch.Generator(device=dev).manual_seed(seed)
    grid = torch.rand(H3, device=dev, generator=g)
    grid = torch.where(grid < frac, grid * 10 + 0.5, torch.zeros_like(grid))
    grid[::97] = -1.0  # "untrained" cells (mark_untrained_grid)
    return grid


def _scratch():
    return (torch.empty(H3, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))


def test_full_sweep_enumerates_every_cell_with_jitter_inside_the_cell():
    import pvd_hip
    grid = _grid()
    idx = torch.empty(H3, dtype=torch.int32, device=dev)
    xyz = torch.empty(H3, 3, device=dev)
    bound_c = 2.0
    pvd_hip.occ_sample(grid, H, 0, 0, True, bound_c, 1, None, None, idx, xyz)
    assert torch.equal(idx, torch.arange(H3, dtype=torch.int32, device=dev))
    hgs = bound_c / H
    i = idx.long()
    from pvd.dp_compact import _gather3
    c = torch.stack([_gather3(i), _gather3(i >> 1), _gather3(i >> 2)], dim=1).float()
    centre = (2 * c / (H - 1) - 1) * (bound_c - hgs)
    d = (xyz - centre).abs()
    assert d.max().item() <= hgs * (1 + 1e-5) and d.mean().item() > 0.4 * hgs  # uniform jitter of +- half a cell
    assert (xyz.abs() <= bound_c).all()


def test_partial_update_draws_uniform_cells_and_occupied_cells():
    import pvd_hip
    grid = _grid(1)
    n = H3 // 4
    idx = torch.empty(2 * n, dtype=torch.int32, device=dev)
    xyz = torch.empty(2 * n, 3, device=dev)
    lst, cnt = _scratch()
    pvd_hip.occ_sample(grid, H, n, n, False, 1.0, 7, lst, cnt, idx, xyz)
    occupied = (grid > 0)
    assert int(cnt) == int(occupied.sum())
    assert torch.equal(torch.sort(lst[: int(cnt)].long())[0], torch.nonzero(occupied).squeeze(-1))
    uni, occ = idx[:n].long(), idx[n:].long()
    assert int(uni.min()) >= 0 and int(uni.max()) < H3
    hist = torch.bincount(uni // (H3 // 64), minlength=64).float()
    assert (hist / hist.sum() - 1 / 64).abs().max().item() < 0.004  # uniform over the grid
    assert occupied[occ].all()  # second half: occupied cells only
    hits = torch.bincount(occ, minlength=H3)[occupied].float()
    assert (hits > 0).float().mean().item() > 0.9 and hits.mean().item() == pytest.approx(n / int(cnt), rel=1e-6)
    # another seed draws other cells; an empty grid yields no occupied queries
    idx2 = torch.empty_like(idx)
    pvd_hip.occ_sample(grid, H, n, n, False, 1.0, 8, lst, cnt, idx2, xyz)
    assert not torch.equal(idx, idx2)
    pvd_hip.occ_sample(torch.zeros_like(grid), H, n, n, False, 1.0, 9, lst, cnt, idx2, xyz)
    assert int(cnt) == 0 and (idx2[n:] == -1).all() and (idx2[:n] >= 0).all()


def test_update_and_finish_match_the_torch_formulation():
    import pvd_hip
    import raymarching
    grid = _grid(2)
    g = torch.Generator(device=dev).manual_seed(3)
    n = 50000
    idx = torch.randperm(H3, device=dev, generator=g)[:n].to(torch.int32)  # unique: duplicate order is unspecified in the reference too
    idx[:10] = -1
    sig = torch.rand(n, device=dev, generator=g) * 3
    ref = grid.clone()
    tmp = -torch.ones_like(ref)
    ok = idx >= 0
    tmp[idx[ok].long()] = sig[ok] * 0.5
    valid = (ref >= 0) & (tmp >= 0)
    ref[valid] = torch.maximum(ref[valid] * 0.95, tmp[valid])
    got = grid.clone()
    pvd_hip.occ_update(got, torch.empty(H3, device=dev), idx, sig, H, 0.5, 0.95)
    assert torch.equal(got, ref)
    assert (got[::97] == -1).all()  # untrained cells are never touched

    two = torch.cat([got, _grid(4)])  # two cascades
    mt = torch.zeros(2, device=dev)
    bits = torch.empty(two.numel() // 8, dtype=torch.uint8, device=dev)
    pvd_hip.occ_finish(two, 0.3, mt, torch.empty(1024, device=dev), bits)
    mean = two.clamp(min=0).mean()
    assert abs(float(mt[0]) - float(mean)) <= 1e-5 * float(mean)
    assert float(mt[1]) == min(float(mt[0]), np.float32(0.3))
    ref_bits = raymarching.packbits(two.view(2, -1), float(mt[1]))
    assert torch.equal(bits, ref_bits)


def test_update_extra_state_device_path_tracks_the_torch_path():
    """The same density field (the analytic chair: 50 inside, 0 outside), 20 occupancy updates through the device path and
    through the torch path: different random samples, so compare what matters -- the set of occupied cells (they can only
    disagree in cells the surface cuts) -- and that the device path reproduces the scene's true occupancy."""
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.scene import ChairScene
    from pvd.workload import make_model
    opt = PVDConfig(num_rays=2048)
    scene = ChairScene(thicken=0.08)
    sets, means = [], []
    for device_path in (True, False):
        ops = hip_ops()
        if not device_path:
            ops.occupancy = None
        m = make_model(ops, opt, "hash", True, dev)
        m.density = lambda x: {"sigma": scene.sigma(x)}
        for _ in range(20):
            with torch.no_grad():
                m.update_extra_state()
        thresh = min(float(m.mean_density), m.density_thresh)
        sets.append(m.density_grid.view(-1) > thresh)
        means.append(float(m.mean_density))
        bits = m.density_bitfield.clone()
        from pvd.scene import packbits_torch
        assert torch.equal(bits, packbits_torch(m.density_grid, thresh))  # the bitfield is the thresholded grid
        assert m.iter_density == 20
    a, b = sets
    inter, union = (a & b).sum().item(), (a | b).sum().item()
    assert union > 50000 and inter / union > 0.97, (inter, union)
    assert abs(means[0] - means[1]) <= 0.03 * means[1], means
    truth = scene.density_grid(128, 1.0, 1, device=dev).view(-1) > 10.0  # cell-centre occupancy of the analytic scene
    assert (a & truth).sum().item() / truth.sum().item() > 0.98  # nothing solid is missed


def test_mark_untrained_grid_on_the_device_marks_the_cells_the_cpu_run_marks():
    """mark_untrained_grid (reference: renderer.py:561-645) is torch code on top of pvd_morton3D: the HIP-backed model on the
    GPU must mark exactly the cells the oracle-backed model marks on the CPU (the frustum test is a handful of fp32 compares:
    cells whose camera-space margin is below 1e-5 of the cell size may fall either way), two cascades, and the marked cells must
    survive a device-side update_extra_state untouched."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_ops import oracle_ops
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.scene import BLENDER_INTRINSICS, synthetic_poses
    from pvd.workload import make_model
    opt = PVDConfig(model_type="hash", bound=2.0, fp16=True)
    poses = synthetic_poses(np.random.RandomState(3))[:5]
    grids = []
    for ops, d in ((hip_ops(), dev), (oracle_ops(), torch.device("cpu"))):
        torch.manual_seed(0)
        m = make_model(ops, opt, "hash", True, d)
        assert m.cascade == 2
        m.density_grid.zero_()
        e0 = m.occ_epoch
        m.mark_untrained_grid(poses, BLENDER_INTRINSICS)
        assert m.occ_epoch > e0
        grids.append((m, m.density_grid.detach().cpu()))
    g_hip, g_cpu = grids[0][1], grids[1][1]
    marked = (g_cpu < 0)
    assert 0.02 < marked.float().mean().item() < 0.9
    assert ((g_hip < 0) != marked).float().mean().item() < 1e-5
    m = grids[0][0]
    before = m.density_grid.clone()
    with torch.autocast("cuda", dtype=torch.float16):
        m.update_extra_state()
    assert torch.equal(m.density_grid[before < 0], before[before < 0])
    assert (m.density_grid[before >= 0] >= 0).all()


@pytest.mark.parametrize("bound", [1, 2])
def test_device_path_replays_the_reference_run_cell_for_cell(bound):
    """The REFERENCE's own update_extra_state run (tests/golden/reference_step.npz: hash model, 16^3 grid, one and two cascades, full
    sweeps and partial updates, CPU generator seeded per call -- the fixture tests/test_golden_occupancy.py pins the torch path with)
    replayed through the device path: `occ_replay` makes it consume torch's CPU generator in the reference's order
    (pvd_occ_sample_replay) and resolve duplicate cells like the reference's sequential assignment (pvd_occ_update_ordered).
    Every cell of the grid then carries the reference's value (up to the fp32 rounding of the density network on another
    device), the same cells are untouched, and the bitfield differs in at most a few threshold-marginal cells."""
    import os
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import make_model
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_step.npz"), allow_pickle=False)
    pre = "occ_b%d__" % bound
    opt = PVDConfig(model_type="hash", teacher_type="hash", bound=float(bound), PE=6, skip=2, nerf_layer_num=5, nerf_layer_wide=32,
                    resolution0=12, plenoxel_res="[12,12,12]", grid_size=int(G["grid_size"]), density_thresh=10.0, fp16=False)
    torch.manual_seed(0)
    net = make_model(hip_ops(), opt, "hash", True, dev)
    sd = {}
    for k in [str(k) for k in G[pre + "keys"]]:
        if "embeddings" in k:
            torch.manual_seed(777)
            sd[k] = (torch.rand(net.state_dict()[k].shape) - 0.5) * 0.6
        else:
            sd[k] = torch.from_numpy(G[pre + "sd__" + k])
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    net.occ_replay = True
    with torch.no_grad():
        net.density_grid.copy_(torch.from_numpy(G[pre + "marked"]).to(dev))  # (mark_untrained_grid on the device: its own test above)
    n_partial = 0
    for i in G[pre + "calls"]:
        c = pre + "u%d__" % int(i)
        net.iter_density = int(G[c + "iter_density"])
        n_partial += net.iter_density >= 16
        counts = G[c + "counts"]
        if len(counts):
            net.step_counter.zero_()
            net.step_counter[:len(counts)] = torch.from_numpy(counts).to(dev)
            net.local_step = len(counts)
        torch.manual_seed(int(G[c + "seed"]))
        with torch.no_grad():
            net.update_extra_state()
        ref = G[c + "grid"]
        got = net.density_grid.cpu().numpy()
        np.testing.assert_array_equal(got < 0, ref < 0)
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=1e-5)
        assert float(net.mean_density) == pytest.approx(float(G[c + "mean_density"]), rel=1e-4)
        flips = np.unpackbits(net.density_bitfield.cpu().numpy() ^ G[c + "bitfield"]).sum()
        assert flips <= 4, flips
        assert int(net.mean_count) == int(G[c + "mean_count"])
        assert net.local_step == 0 and net.iter_density == int(G[c + "iter_density"]) + 1
    assert n_partial >= 1  # the uniform + occupied-cell branch (duplicates, ordered scatter) was replayed too
