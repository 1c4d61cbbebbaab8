"""Host-side logic of the product (autograd Functions, renderer, network, trainer) driven through the
CPU oracle backend.  CPU only; the same code paths run on the HIP backend in test_hip_*.py."""
import numpy as np
import pytest
import torch

from oracle_ops import oracle_ops
from pvd.config import PVDConfig
from pvd.scene import BLENDER_INTRINSICS, ChairScene, get_rays, packbits_torch, synthetic_poses
from pvd.workload import DistillWorkload, install_occupancy, make_model

OPS = oracle_ops()
RM = OPS.raymarching


def _rays(n, seed=0):
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(seed)))
    r = get_rays(poses[7][None], BLENDER_INTRINSICS, 800, 800, n, generator=torch.Generator().manual_seed(seed))
    grid = ChairScene().density_grid(128, 1.0, 1)
    return r["rays_o"], r["rays_d"], packbits_torch(grid, 10.0)


def test_march_rays_train_sizing_rules():
    o, d, bits = _rays(512)
    aabb = torch.tensor([-1, -1, -1, 1, 1, 1.0])
    nears, fars = RM.near_far_from_aabb(o, d, aabb, 0.2)
    assert nears.shape == (512,)
    # mean_count <= 0: trimmed to the real count rounded UP to `align`, +align when already aligned (raymarching.py:276-284)
    counter = torch.zeros(2, dtype=torch.int32)
    xyzs, dirs, deltas, rays = RM.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, counter, -1, False, 128, False, 0, 1024)
    m = int(counter[0])
    assert xyzs.shape[0] == m + 128 - m % 128 and dirs.shape == xyzs.shape and deltas.shape == (xyzs.shape[0], 2)
    assert rays.shape == (512, 3) and rays.dtype == torch.int32
    # mean_count > 0: M = mean_count rounded up by align, no trimming, overflowing rays dropped
    mc = m // 2
    counter.zero_()
    x2, _, _, r2 = RM.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, counter, mc, False, 128, False, 0, 1024)
    assert x2.shape[0] == mc + 128 - mc % 128
    assert torch.equal(r2, rays)
    # force_all_rays ignores mean_count
    counter.zero_()
    x3, _, _, _ = RM.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, counter, mc, False, 128, True, 0, 1024)
    assert x3.shape[0] == xyzs.shape[0]
    # step_counter=None allocates its own
    x4, _, _, _ = RM.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars)
    assert x4.shape[0] == m  # align = -1: no rounding


def test_composite_autograd_matches_torch_restatement():
    o, d, bits = _rays(256, 1)
    aabb = torch.tensor([-1, -1, -1, 1, 1, 1.0])
    nears, fars = RM.near_far_from_aabb(o, d, aabb, 0.2)
    xyzs, dirs, deltas, rays = RM.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, None, -1, True, 128, True)
    M = xyzs.shape[0]
    g = torch.Generator().manual_seed(0)
    sig = torch.exp(torch.rand(M, generator=g) * 6 - 2).requires_grad_(True)
    rgb = torch.rand(M, 3, generator=g).requires_grad_(True)
    ws, depth, img = RM.composite_rays_train(sig, rgb, deltas, rays)
    tgt = torch.rand(256, 3, generator=g)
    loss = ((img + (1 - ws)[:, None] * 0.3 - tgt) ** 2).sum() + depth.sum() * 0.0
    loss.backward()
    s2, r2 = sig.detach().double().requires_grad_(True), rgb.detach().double().requires_grad_(True)
    ws_l, img_l = [], []
    for n in range(256):
        a, c = int(rays[n, 1]), int(rays[n, 2])
        al = 1 - torch.exp(-s2[a:a + c] * deltas[a:a + c, 0].double())
        T = torch.cumprod(torch.cat([torch.ones(1, dtype=torch.float64), 1 - al]), 0)[:-1]
        ws_l.append((al * T).sum()); img_l.append(((al * T)[:, None] * r2[a:a + c]).sum(0))
    ws_t, img_t = torch.stack(ws_l), torch.stack(img_l)
    ((img_t + (1 - ws_t)[:, None] * 0.3 - tgt.double()) ** 2).sum().backward()
    assert torch.allclose(img.double(), img_t, atol=2e-6)
    assert (sig.grad.double() - s2.grad).abs().max() <= 2e-5 * s2.grad.abs().max()
    assert torch.allclose(rgb.grad.double(), r2.grad, atol=1e-5)


def _hash_model(training=True, seed=0):
    torch.manual_seed(seed)
    opt = PVDConfig(model_type="hash", num_rays=512)
    opt.stage_iters = {"stage1": -1, "stage2": -1}
    m = make_model(OPS, opt, "hash", False, "cpu")
    install_occupancy(m, ChairScene(), opt)
    m.encoder.embeddings.data.uniform_(-0.5, 0.5)  # something to see
    return m.train(training), opt


def test_run_cuda_train_and_inference_branches_agree():
    m, opt = _hash_model()
    o, d, _ = _rays(512, 2)
    with torch.no_grad():
        tr = m.render(o, d, staged=False, bg_color=1, perturb=False, force_all_rays=True, max_steps=1024)
        m.eval()
        ev = m.render(o, d, staged=False, bg_color=1, perturb=False, max_steps=1024)
    assert tr["image"].shape == (1, 512, 3) and ev["depth"].shape == (1, 512)
    # same samples, same network; inference stops a ray once T < 1e-4 (raymarching.cu:886) -> <= 1e-4 apart
    assert (tr["image"] - ev["image"]).abs().max() < 2e-4
    assert "inherited_params" in tr and len(tr["inherited_params"]) == 4 and tr["rays"].shape == (512, 3)


def test_update_extra_state_and_mark_untrained_grid():
    torch.manual_seed(0)
    m, opt = _hash_model()
    scene = ChairScene()
    m.density = lambda x: {"sigma": scene.sigma(x)}  # analytic field instead of the network
    m.density_grid.zero_(); m.density_bitfield.zero_(); m.iter_density = 0
    m.step_counter[:4, 0] = torch.tensor([100, 200, 300, 400], dtype=torch.int32); m.local_step = 4
    m.update_extra_state()
    assert m.mean_count == 250 and m.local_step == 0 and m.iter_density == 1
    occ = m.density_grid[0] > 0
    ref = scene.density_grid(128, 1.0, 1)[0] > 0
    assert (occ & ~ref).sum() == 0 and occ.sum() > 0.5 * ref.sum()  # jittered point samples: subset of the conservative grid
    thresh = min(m.mean_density, opt.density_thresh)
    assert torch.equal(m.density_bitfield, packbits_torch(m.density_grid, thresh))
    g1 = m.density_grid.clone()
    m.iter_density = 16  # partial branch
    m.update_extra_state()
    assert (m.density_grid >= g1 * 0.95 - 1e-6).all()  # EMA-max never drops a cell by more than the decay
    # cells outside every camera frustum are marked -1 and stay untouched
    poses = synthetic_poses(np.random.RandomState(0))[:4]
    m.density_grid.zero_()
    m.mark_untrained_grid(poses, BLENDER_INTRINSICS)
    frac = (m.density_grid < 0).float().mean().item()
    assert 0.0 < frac < 0.9
    before = m.density_grid.clone()
    m.iter_density = 0
    m.update_extra_state()
    assert torch.equal(m.density_grid[before < 0], before[before < 0])


@pytest.mark.parametrize("student", ["vm", "hash", "mlp", "tensors"])
def test_distillation_stages_and_loss_decrease(student):
    opt = PVDConfig(num_rays=256, resolution0=32, iters=50, fp16=False, model_type=student, plenoxel_res="[32,32,32]",
                    nerf_layer_wide=32, nerf_layer_num=4, skip=1)
    opt.stage_iters = {"stage1": 2 if student != "tensors" else -1, "stage2": 4}
    w = DistillWorkload(OPS, "cpu", opt, teacher_pretrain_steps=3, start_stage="stage1")
    w.trainer.global_step = 0
    kinds, losses = [], []
    for it in range(7):
        loss, info, ps, pt = w.step()
        kinds.append(tuple(sorted(info)))
        losses.append(float(loss))
        assert np.isfinite(losses[-1])
        if "rgb" in info:
            assert ps.shape == (1, 256, 3) and pt.shape == (1, 256, 3)
    if student != "tensors":
        assert kinds[:2] == [("fea",), ("fea",)]
    assert kinds[2:4] == [("color", "sigma")] * 2 if student != "tensors" else True
    assert all(k == ("rgb",) for k in kinds[4:])
    # the student inherits occupancy + same-shaped weights from the teacher (utils.py:1536-1545)
    assert torch.equal(w.stu.density_bitfield, w.tea.density_bitfield)
    if student in ("vm", "hash", "mlp"):
        assert all(torch.equal(a, b) for a, b in zip(w.tea.color_net.parameters(), w.stu.color_net.parameters())) is False  # trained since
    # teacher never changes
    t0 = [p.clone() for p in w.tea.parameters()]
    w.step()
    assert all(torch.equal(a, b) for a, b in zip(t0, w.tea.parameters()))


def test_stage3_loss_decreases_for_vm_student():
    opt = PVDConfig(num_rays=512, resolution0=48, iters=200, fp16=False, model_type="vm")
    w = DistillWorkload(OPS, "cpu", opt, teacher_pretrain_steps=20)
    first = np.mean([float(w.step()[1]["rgb"]) for _ in range(3)])
    for _ in range(25):
        w.step()
    last = np.mean([float(w.step()[1]["rgb"]) for _ in range(3)])
    assert last < 0.8 * first, (first, last)


def test_teacher_training_improves_psnr():
    opt = PVDConfig(num_rays=512, iters=100, fp16=False)
    w0 = DistillWorkload(OPS, "cpu", opt, teacher_pretrain_steps=2)
    w1 = DistillWorkload(OPS, "cpu", opt, teacher_pretrain_steps=40)
    assert w1.teacher_psnr > w0.teacher_psnr + 0.5


def test_flat_gradient_views_survive_training():
    opt = PVDConfig(num_rays=128, resolution0=16, iters=10, fp16=False)
    w = DistillWorkload(OPS, "cpu", opt, teacher_pretrain_steps=0)
    w.step(); w.step()
    flat = w.trainer.flat
    off = 0
    for p in flat.params:
        assert p.grad.data_ptr() == flat.flat[off:off + 1].data_ptr()
        off += p.numel()
    assert flat.flat.abs().sum() > 0


@pytest.mark.parametrize("bound,cascade", [(1.0, 1), (2.0, 2)])
def test_dp_compact_footprint_mask_covers_every_sample_footprint(bound, cascade):
    """pvd/dp_compact.py: the footprint mask must contain every texel a linear-interpolation footprint of ANY point inside
    an occupied cell can touch -- brute force with random points in random occupied cells, two cascades, non-cubic table."""
    from pvd.dp_compact import footprint_mask, occupied_cells
    from pvd.scene import ChairScene, packbits_torch
    g = torch.Generator().manual_seed(0)
    grid = ChairScene(thicken=0.08).density_grid(128, bound, cascade)
    bits = packbits_torch(grid, 10.0)
    boxes = occupied_cells(bits, cascade, 128, bound)
    assert len(boxes) == cascade
    aabb = [-bound] * 3 + [bound] * 3
    sizes, axes = [37, 300], [2, 0]  # table W axis samples world z, H axis samples world x
    mask = footprint_mask(boxes, aabb, sizes, axes)
    assert mask.shape == (300, 37) and 0.01 < mask.float().mean().item() < 0.9
    for lo, hi in boxes:
        pick = torch.randint(0, lo.shape[0], (20000,), generator=g)
        p = lo[pick] + (hi[pick] - lo[pick]) * torch.rand(20000, 3, generator=g, dtype=torch.float64)
        p = p.clamp(-bound, bound)
        idx = []
        for k in range(2):
            a, n = axes[k], sizes[k]
            u = ((2 * (p[:, a] - aabb[a]) / (aabb[a + 3] - aabb[a]) - 1) + 1) / 2 * (n - 1)
            idx.append(torch.floor(u).long())
        for dx in (0, 1):
            for dy in (0, 1):
                x, y = idx[0] + dx, idx[1] + dy
                ok = (x >= 0) & (x < sizes[0]) & (y >= 0) & (y < sizes[1])
                assert mask[y[ok], x[ok]].all()
    # 3-D (Plenoxel volume)
    m3 = footprint_mask(boxes, aabb, [16, 20, 24], [0, 1, 2])
    assert m3.shape == (24, 20, 16) and m3.any() and not m3.all()


def test_dp_compact_scatter_can_look_at_what_it_puts_back():
    """GradCompactor.scatter(found_inf=...): the CPU twin of pvd_segments_op(4) -- the exchanged rows go back into the flat buffer and
    an inf / nan among them raises the flag (left alone otherwise: the caller's flag is sticky)."""
    from pvd.dp_compact import GradCompactor
    c = GradCompactor.__new__(GradCompactor)
    c.idx = torch.tensor([3, 4, 5, 9, 20, 21], dtype=torch.int64)
    flat = torch.zeros(32)
    buf = torch.arange(1.0, 7.0)
    flag = torch.zeros(1)
    c.scatter(flat, buf, found_inf=flag)
    assert torch.equal(flat[c.idx], buf) and float(flat.sum()) == float(buf.sum()) and float(flag) == 0.0
    bad = buf.clone()
    bad[4] = float("nan")
    c.scatter(flat, bad, found_inf=flag)
    assert float(flag) == 1.0 and torch.isnan(flat[20])
    c.scatter(flat, buf, found_inf=flag)  # a clean exchange does not clear it
    assert float(flag) == 1.0 and torch.equal(flat[c.idx], buf)
    c.scatter(flat, bad)  # no flag: plain scatter
    assert torch.isnan(flat[20])


def test_dp_compact_run_table_covers_the_index_set_exactly():
    """segments_of(): the run table the HIP kernels walk lists exactly the index set, in compact order, in pieces of at most
    seg_max elements."""
    from pvd.dp_compact import segments_of
    g = torch.Generator().manual_seed(0)
    for n, p, seg_max in [(5000, 0.5, 64), (20000, 0.97, 256), (3000, 0.02, 4096), (10, 1.1, 4)]:
        keep = torch.rand(n, generator=g) < p
        idx = keep.nonzero().squeeze(-1)
        segs = segments_of(idx, seg_max)
        assert segs.dtype == torch.int32 and segs.shape[1] == 3
        assert int(segs[:, 2].max()) <= seg_max and int(segs[:, 2].min()) >= 1
        rebuilt = torch.cat([torch.arange(s, s + l) for s, d, l in segs.tolist()])
        assert torch.equal(rebuilt, idx)
        assert segs[:, 1].tolist() == (torch.cumsum(segs[:, 2], 0) - segs[:, 2]).tolist()  # dst = exclusive prefix sum
        # maximal runs: consecutive pieces either continue a run cut at seg_max or are separated by a gap
        for (s0, d0, l0), (s1, d1, l1) in zip(segs.tolist()[:-1], segs.tolist()[1:]):
            assert s1 > s0 + l0 or (s1 == s0 + l0 and l0 == seg_max)
    assert segments_of(torch.zeros(0, dtype=torch.int64)).shape == (0, 3)


def test_mlp_weight_stream_layout_matches_the_kernels_chunk_order():
    """fusedhead.mlp_weight_stream (host side of pvd_mlp_head_forward_fused): per layer, chunks of 64 output rows, each
    rows x (K + 8) halfs -- the 63 positional-encoding columns padded to 64, columns permuted inside groups of 32 so that a
    lane's operand of one K = 32 MFMA is contiguous -- followed by the rows' biases; the last layer is one chunk of 32 rows."""
    import fusedhead
    opt = PVDConfig(model_type="mlp", fp16=False)
    opt.stage_iters = {"stage1": -1, "stage2": -1}
    torch.manual_seed(0)
    from pvd.workload import make_model
    m = make_model(OPS, opt, "mlp", True, "cpu")
    assert fusedhead.mlp_supported(m)
    s = fusedhead.mlp_weight_stream(m)
    nb, na = m.skips, len(m.nerf_mlp) - 3 - m.skips
    need = 4 * (64 * 72 + 64) + (nb + na) * 4 * (64 * 264 + 64) + 4 * (64 * 328 + 64) + (32 * 264 + 32)
    assert s.dtype == torch.float16 and s.numel() == need

    def logical(blk, K):  # undo the permutation: stored position 32 p + 8 h + 4 s + j holds logical column 32 p + 16 s + 4 h + j
        n = blk.shape[0]
        return blk[:, :K].reshape(n, K // 32, 4, 2, 4).permute(0, 1, 3, 2, 4).reshape(n, K)

    # first layer, chunk 1 (rows 64..127): [64, 72] then 64 biases
    off = 64 * 72 + 64
    blk = s[off:off + 64 * 72].view(64, 72)
    w0 = m.nerf_mlp[0].weight.detach().half()
    lg = logical(blk, 64)
    assert torch.equal(lg[:, :63], w0[64:128]) and not lg[:, 63].any() and not blk[:, 64:].any()
    assert torch.equal(s[off + 64 * 72:off + 64 * 72 + 64], m.nerf_mlp[0].bias.detach().half()[64:128])
    # the skip layer's first chunk: logical columns [pts 63 | 0 | x 256], then 8 zeros
    off = 4 * (64 * 72 + 64) + nb * 4 * (64 * 264 + 64)
    blk = s[off:off + 64 * 328].view(64, 328)
    ws = m.nerf_mlp[m.skips + 1].weight.detach().half()
    lg = logical(blk, 320)
    assert torch.equal(lg[:, :63], ws[:64, :63]) and not lg[:, 63].any() and torch.equal(lg[:, 64:], ws[:64, 63:]) and not blk[:, 320:].any()
    # the last layer: 32 rows, rows 28.. zero
    off = need - (32 * 264 + 32)
    blk = s[off:off + 32 * 264].view(32, 264)
    assert torch.equal(logical(blk, 256)[:28], m.nerf_mlp[-1].weight.detach().half()) and not blk[28:].any()
    assert torch.equal(s[-32:-4], m.nerf_mlp[-1].bias.detach().half()) and not s[-4:].any()
    # a model with another width is not taken by the fused kernel
    opt2 = PVDConfig(model_type="mlp", nerf_layer_wide=32, nerf_layer_num=4, skip=1, fp16=False)
    opt2.stage_iters = opt.stage_iters
    assert not fusedhead.mlp_supported(make_model(OPS, opt2, "mlp", True, "cpu"))


def test_carried_prefix_keeps_views_and_copies_storage_by_storage():
    """The static home of a prefix across graph replays (pvd/trainer.py CarriedPrefix): views of one buffer stay views of one
    buffer, a store() brings every value across, aliases inside the structure stay aliases."""
    from pvd.trainer import CarriedPrefix

    def prefix(seed):
        g = torch.Generator().manual_seed(seed)
        feat = torch.rand(7, 16, generator=g)
        xyz = torch.rand(7, 3, generator=g)
        inh = [xyz, torch.rand(7, 3, generator=g), torch.rand(7, 2, generator=g), torch.arange(8, dtype=torch.int32).view(4, 2) + seed]
        return dict(rays_o=torch.rand(4, 3, generator=g), bg=1, inh=inh, nf=(torch.rand(4, generator=g), torch.rand(4, generator=g)),
                    out_tea=dict(image=torch.rand(4, 3, generator=g), depth=None, inherited_params=inh),
                    tea_attrs=dict(feature_sigma_color=feat, sigma_l=feat[..., 0], color_l=feat[:, 1:4]))

    a, b = prefix(1), prefix(2)
    c = CarriedPrefix(a)
    home = c.pre
    assert home["bg"] == 1 and home["out_tea"]["depth"] is None
    assert torch.equal(home["tea_attrs"]["feature_sigma_color"], a["tea_attrs"]["feature_sigma_color"])
    assert home["tea_attrs"]["feature_sigma_color"].data_ptr() != a["tea_attrs"]["feature_sigma_color"].data_ptr()
    # the view relation survives
    assert home["tea_attrs"]["sigma_l"].data_ptr() == home["tea_attrs"]["feature_sigma_color"].data_ptr()
    assert home["tea_attrs"]["sigma_l"].stride() == (16,)
    assert home["out_tea"]["inherited_params"][0].data_ptr() == home["inh"][0].data_ptr()
    ptrs = [t.data_ptr() for t in (home["inh"][0], home["tea_attrs"]["feature_sigma_color"], home["out_tea"]["image"])]
    c.store(b)
    assert ptrs == [t.data_ptr() for t in (c.pre["inh"][0], c.pre["tea_attrs"]["feature_sigma_color"], c.pre["out_tea"]["image"])]
    for key in ("feature_sigma_color", "sigma_l", "color_l"):
        assert torch.equal(c.pre["tea_attrs"][key], b["tea_attrs"][key])
    for i in range(4):
        assert torch.equal(c.pre["inh"][i], b["inh"][i])
    assert torch.equal(c.pre["nf"][1], b["nf"][1]) and torch.equal(c.pre["out_tea"]["image"], b["out_tea"]["image"])
    bad = prefix(3)
    bad["inh"][0] = torch.rand(9, 3)
    with pytest.raises(AssertionError):
        c.store(bad)


def test_every_environment_switch_is_registered():
    """aaai2023-pvd_amd/pvd/knobs.py is the one place a reader finds the PVD_* switches: every name the product sources (Python and
    the C side's getenv) or bench.py read must be in it, and it must not list names nothing reads."""
    import os
    import re
    from pvd import knobs
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    root = os.path.join(REPO, "aaai2023-pvd_amd")
    read = set()
    for dirpath, _, files in os.walk(root):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py") and not path.endswith(os.path.join("pvd", "knobs.py")):
                src = open(path).read()
                read |= set(re.findall(r'environ(?:\.get|\.setdefault)?[\(\[]\s*"(PVD_[A-Z0-9_]+)"', src))
            elif f.endswith((".hip", ".h")):
                read |= set(re.findall(r'getenv\("(PVD_[A-Z0-9_]+)"\)', open(path).read()))
    read |= set(re.findall(r'environ(?:\.get|\.setdefault)?[\(\[]\s*"(PVD_[A-Z0-9_]+)"', open(os.path.join(REPO, "bench.py")).read()))
    assert read - set(knobs.ALL) == set(), "unregistered switches: %s" % sorted(read - set(knobs.ALL))
    assert set(knobs.ALL) - read == set(), "registered but read by nothing: %s" % sorted(set(knobs.ALL) - read)
    assert len(knobs.PRODUCTION) <= 8


@pytest.mark.parametrize("chunks", [1, 2, 3, 8])
def test_exchange_layout_covers_every_touched_group_once(chunks):
    """pvd/dp_compact.py: ExchangeLayout (the ray-DP exchange buffer of round 6) on a synthetic touched set -- every group of four lands at
    its list position inside its chunk, no run crosses a chunk boundary or reaches a flag group, the chunks' row ranges tile the list."""
    import types
    from pvd.dp_compact import ExchangeLayout, segments_of
    g = torch.Generator().manual_seed(chunks)
    n_groups_total = 5000
    keep = torch.rand(n_groups_total, generator=g) < 0.3
    keep[100:1400] = True  # a long run (cut into several table entries)
    groups = keep.nonzero().squeeze(1)
    idx = (groups[:, None] * 4 + torch.arange(4)).reshape(-1)
    c = types.SimpleNamespace(idx=idx, segs=segments_of(idx))
    L = ExchangeLayout(c, groups.to(torch.int32), chunks, torch.device("cpu"), with_params=chunks > 1)
    G = groups.numel()
    assert L.n_groups == G and L.chunk == 4 * L.gpc + 4 and L.slot == 4 * L.gpc and L.xbuf.numel() == chunks * L.chunk
    assert (L.pbuf is not None) == (chunks > 1)
    # gather by the table (what k_segments<5> does) == list order inside the chunked layout
    flat = torch.arange(4 * n_groups_total, dtype=torch.float32) + 1.0
    buf = torch.zeros_like(L.xbuf)
    covered = torch.zeros(L.xbuf.numel(), dtype=torch.int32)
    for start, dst, length in L.segs.tolist():
        buf[dst:dst + length] = flat[start:start + length]
        covered[dst:dst + length] += 1
        assert dst // L.chunk == (dst + length - 1) // L.chunk and (dst + length - 1) % L.chunk < L.slot  # inside one chunk's data
    assert int(covered.max()) == 1 and int(covered.sum()) == 4 * G
    rows = [L.rows_of(r) for r in range(chunks)]
    assert rows[0][0] == 0 and rows[-1][1] == G and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    for r, (lo, hi) in enumerate(rows):
        want = flat[idx[4 * lo:4 * hi]]
        assert torch.equal(L.chunk_of(buf, r)[:4 * (hi - lo)], want)
        assert float(L.chunk_of(buf, r)[4 * (hi - lo):].abs().max()) == 0.0  # padding and the flag group stay untouched
