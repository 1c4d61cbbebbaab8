"""Ray data parallelism on 2 CPU processes (gloo): the N > 1 path of bench.py / DistillTrainer.

What must hold for "shard the rays, all-reduce the flat gradient bucket (SUM)" to be the same
optimisation as one process on all rays:
  * norm-type losses are global norms (sum of squares all-reduced before the sqrt) with the right
    per-shard gradient; mean-type losses are global means; parameter-only terms are not multiplied by G;
  * after the all-reduce every rank holds the full-batch gradient, and replicas stay bit-identical.
Runs the real trainer on the CPU oracle operators (no GPU here); RCCL replaces gloo on the GPU box.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup_paths():
    for p in (REPO, os.path.join(REPO, "aaai2023-pvd_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)


def _make(opt_kw, dp=None):
    from oracle_ops import oracle_ops
    from pvd.config import PVDConfig
    from pvd.workload import DistillWorkload
    opt = PVDConfig(**opt_kw)
    torch.manual_seed(0)
    return DistillWorkload(oracle_ops(), "cpu", opt, teacher_pretrain_steps=0, seed=0, dp=dp)


OPT = dict(num_rays=256, resolution0=24, iters=50, fp16=False, model_type="vm",
           loss_rate_fea_sc=0.0, loss_rate_color=0.0, loss_rate_sigma=0.0)  # rgb norm + L1 reg: independent of row padding


def _worker(rank, world, port, out_path, opt_kw=None, expect_compact=False, skew_rank=None, stu_scale=None):
    OPT = opt_kw or globals()["OPT"]
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from pvd.trainer import RayDP
    dp = RayDP()
    assert dp.enabled and dp.world_size == world and dp.rank == rank

    # --- collective helpers against closed forms
    g = torch.Generator().manual_seed(5)
    full = torch.randn(world * 100, 7, generator=g)
    mine = full[rank * 100:(rank + 1) * 100].clone().requires_grad_(True)
    n = dp.global_norm_l2(mine)
    assert torch.allclose(n, full.norm(), rtol=1e-6)
    n.backward()
    assert torch.allclose(mine.grad, (full / full.norm())[rank * 100:(rank + 1) * 100], rtol=1e-5, atol=1e-7)
    assert torch.allclose(dp.global_mean(mine.detach()), full.mean(), rtol=1e-5, atol=1e-7)
    assert torch.allclose(dp.global_norm_l1(mine.detach()), full.abs().sum(), rtol=1e-5)

    # --- the real trainer: every rank renders its half of the rays
    w = _make(OPT, dp=dp)
    if stu_scale is not None:  # (a hash student takes over EVERY tensor of a hash teacher: make it a different model)
        with torch.no_grad():
            for p in w.stu.parameters():
                p.mul_(stu_scale)
    if skew_rank is not None and rank == skew_rank:
        # a replica whose occupancy grid differs (must never happen; if it does the ranks must notice instead of hanging in
        # a collective with different buffer sizes): one more occupied byte -> a different footprint mask on this rank only
        bf = w.stu.density_bitfield
        empty = (bf == 0).nonzero().squeeze(-1)
        bf[empty[len(empty) // 2]] = 0xFF
    base = _make(OPT)  # only for the shared batch (same seed on every rank)
    rays_o, rays_d, bg = base.next_batch()
    half = OPT["num_rays"] // world
    sl = slice(rank * half, (rank + 1) * half)
    p0 = [p.detach().clone() for p in w.stu.parameters()]
    loss, info, _, _ = w.trainer.train_step(rays_o[:, sl].contiguous(), rays_d[:, sl].contiguous(), bg[:, sl].contiguous())
    flat = w.trainer.flat.flat.clone()
    c = w.trainer._grad_compactor()
    assert (c is not None) == expect_compact
    if expect_compact:
        assert 0.0 < c.fraction < 0.7
    if skew_rank is not None:
        assert w.trainer._compactor is not None and not w.trainer._compactor.agreed  # noticed on every rank
        dist.barrier()
        dist.destroy_process_group()
        return
    # replicas identical after the step
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], t) for t in gathered)
    params = torch.cat([p.detach().reshape(-1) for p in w.stu.parameters()])
    gp = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(gp, params)
    assert all(torch.equal(gp[0], t) for t in gp)
    if rank == 0:
        torch.save({"loss": float(loss), "flat": flat, "rgb": float(info["rgb"])}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ray_dp_two_ranks_equals_single_process(tmp_path):
    _setup_paths()
    port = _free_port()
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    dp_res = torch.load(out)

    # single process, same two shards (the marcher's perturbation is a function of the ray's index INSIDE a
    # launch, raymarching.cu:352, so the shards -- not the concatenated batch -- are the comparable unit):
    # one autograd graph over both shards with the loss written on the concatenation
    w = _make(OPT)
    base = _make(OPT)
    rays_o, rays_d, bg = base.next_batch()
    tr, stu, tea = w.trainer, w.stu, w.tea
    tr.opt.global_step = tr.global_step
    tr.flat.zero_()
    diffs = []
    half = OPT["num_rays"] // 2
    for r in range(2):
        sl = slice(r * half, (r + 1) * half)
        o, d, b = rays_o[:, sl].contiguous(), rays_d[:, sl].contiguous(), bg[:, sl].contiguous()
        out_s = stu.render(o, d, staged=False, bg_color=b, perturb=True, force_all_rays=False, dt_gamma=0, max_steps=1024)
        with torch.no_grad():
            out_t = tea.render(o, d, staged=False, bg_color=b, perturb=True, force_all_rays=False,
                               inherited_params=out_s["inherited_params"], dt_gamma=0, max_steps=1024)
        diffs.append(out_t["image"] - out_s["image"])
    l_rgb = torch.norm(torch.cat(diffs, dim=1))
    loss = l_rgb * tr.opt.loss_rate_rgb + stu.density_loss() * tr.opt.l1_reg_weight
    loss.backward()
    flat = tr.flat.flat
    assert abs(float(l_rgb.detach()) - dp_res["rgb"]) <= 1e-5 * abs(dp_res["rgb"]), (float(l_rgb), dp_res["rgb"])
    # a rank's reported loss carries 1/G of the parameter-only L1 term (its gradients are summed over ranks)
    l1 = float(stu.density_loss().detach()) * tr.opt.l1_reg_weight
    assert abs((float(loss.detach()) - l1 / 2) - dp_res["loss"]) <= 1e-5 * abs(dp_res["loss"])
    scale = flat.abs().max().item()
    assert scale > 0
    assert (flat - dp_res["flat"]).abs().max().item() <= 2e-5 * scale, ((flat - dp_res["flat"]).abs().max().item(), scale)


OPT_COMPACT = dict(OPT, l1_reg_weight=0.0)  # no dense L1 gradient -> only table rows under occupied cells are exchanged


@pytest.mark.timeout(600)
def test_ray_dp_compact_exchange_is_exact(tmp_path):
    """pvd/dp_compact.py: all-reducing only the rows of the VM planes that occupied cells can touch gives the same
    full gradient as one process on both shards -- i.e. everything outside the footprint mask is exactly zero."""
    _setup_paths()
    port = _free_port()
    out = str(tmp_path / "dpc.pt")
    mp.spawn(_worker, args=(2, port, out, OPT_COMPACT, True), nprocs=2, join=True)
    dp_res = torch.load(out)
    w = _make(OPT_COMPACT)
    base = _make(OPT_COMPACT)
    rays_o, rays_d, bg = base.next_batch()
    tr, stu, tea = w.trainer, w.stu, w.tea
    tr.opt.global_step = tr.global_step
    tr.flat.zero_()
    diffs = []
    half = OPT_COMPACT["num_rays"] // 2
    for r in range(2):
        sl = slice(r * half, (r + 1) * half)
        o, d, b = rays_o[:, sl].contiguous(), rays_d[:, sl].contiguous(), bg[:, sl].contiguous()
        out_s = stu.render(o, d, staged=False, bg_color=b, perturb=True, force_all_rays=False, dt_gamma=0, max_steps=1024)
        with torch.no_grad():
            out_t = tea.render(o, d, staged=False, bg_color=b, perturb=True, force_all_rays=False,
                               inherited_params=out_s["inherited_params"], dt_gamma=0, max_steps=1024)
        diffs.append(out_t["image"] - out_s["image"])
    (torch.norm(torch.cat(diffs, dim=1)) * tr.opt.loss_rate_rgb).backward()
    flat = tr.flat.flat
    scale = flat.abs().max().item()
    assert scale > 0
    assert (flat - dp_res["flat"]).abs().max().item() <= 2e-5 * scale
    # and the mask really leaves something out
    from pvd.dp_compact import GradCompactor
    offs, acc = [], 0
    for p in tr.flat.params:
        offs.append(acc); acc += p.numel()
    c = GradCompactor(stu, tr.flat.params, offs, torch.device("cpu"))
    outside = torch.ones_like(flat, dtype=torch.bool)
    outside[c.idx] = False
    assert outside.any() and flat[outside].abs().max().item() == 0.0


@pytest.mark.timeout(600)
def test_ray_dp_ranks_with_different_masks_fall_back_to_the_dense_exchange(tmp_path):
    """The compact exchange needs the same row set on every rank.  If a replica's occupancy grid differs, the one-off
    agreement check (size + checksum, all-reduced) makes ALL ranks use the dense all-reduce -- no mismatched collective."""
    _setup_paths()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path / "skew.pt"), OPT_COMPACT, False, 1), nprocs=2, join=True)


@pytest.mark.timeout(600)
def test_ray_dp_with_the_teacher_marching_first(tmp_path):
    """render_stu_first = False (utils.py:1020-1043, renderer.py:392-411): the TEACHER marches on its occupancy grid and the student
    inherits the samples; the compact exchange then takes its footprint from the teacher's grid.  Two ranks against one process."""
    _setup_paths()
    opt = dict(OPT_COMPACT, render_stu_first=False)
    out = str(tmp_path / "dpt.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, opt, True), nprocs=2, join=True)
    dp_res = torch.load(out)
    w = _make(opt)
    base = _make(opt)
    rays_o, rays_d, bg = base.next_batch()
    tr, stu, tea = w.trainer, w.stu, w.tea
    tr.opt.global_step = tr.global_step
    tr.flat.zero_()
    diffs = []
    half = opt["num_rays"] // 2
    for r in range(2):
        sl = slice(r * half, (r + 1) * half)
        o, d, b = rays_o[:, sl].contiguous(), rays_d[:, sl].contiguous(), bg[:, sl].contiguous()
        with torch.no_grad():
            out_t = tea.render(o, d, staged=False, bg_color=b, perturb=True, force_all_rays=False, dt_gamma=0, max_steps=1024)
        out_s = stu.render(o, d, staged=False, bg_color=b, perturb=True, force_all_rays=False,
                           inherited_params=out_t["inherited_params"], dt_gamma=0, max_steps=1024)
        diffs.append(out_t["image"] - out_s["image"])
    l_rgb = torch.norm(torch.cat(diffs, dim=1))
    (l_rgb * tr.opt.loss_rate_rgb).backward()
    assert abs(float(l_rgb.detach()) - dp_res["rgb"]) <= 1e-5 * abs(dp_res["rgb"]), (float(l_rgb), dp_res["rgb"])
    flat = tr.flat.flat
    scale = flat.abs().max().item()
    assert scale > 0
    assert (flat - dp_res["flat"]).abs().max().item() <= 2e-5 * scale, ((flat - dp_res["flat"]).abs().max().item(), scale)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("wire", ["f16", "bf16"])
def test_ray_dp_sixteen_bit_wire_is_an_opt_in_approximation(tmp_path, wire, monkeypatch):
    """PVD_DP_WIRE=f16 | bf16: the (compact) gradient crosses the links in 16 bits.  Replicas still end bit-identical (every rank
    receives the same sum) and the gradient is the fp32 exchange's to the wire format's rounding; off by default."""
    _setup_paths()
    out32, out16 = str(tmp_path / "w32.pt"), str(tmp_path / "w16.pt")
    mp.spawn(_worker, args=(2, _free_port(), out32, OPT_COMPACT, True), nprocs=2, join=True)
    monkeypatch.setenv("PVD_DP_WIRE", wire)
    mp.spawn(_worker, args=(2, _free_port(), out16, OPT_COMPACT, True), nprocs=2, join=True)  # (asserts identical replicas itself)
    a, b = torch.load(out32)["flat"], torch.load(out16)["flat"]
    scale = a.abs().max().item()
    err = (a - b).abs().max().item() / scale
    assert 0 < err <= (2e-3 if wire == "f16" else 1.6e-2), err  # f16: 11 bits, bf16: 8 bits of significand, two roundings


def _sharded_worker(rank, world, port):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PVD_DP_EXCHANGE="sharded")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from pvd.trainer import RayDP
    dp = RayDP()
    for n in (65536, 70001, 3 * 65536 + 5):  # divisible by the world size and not (the padded tail)
        g = torch.Generator().manual_seed(100 + n)
        parts = [torch.randn(n, generator=g) for _ in range(world)]
        mine = parts[rank].clone()
        dp.all_reduce_sum_(mine)
        want = parts[0].clone()
        for k in range(1, world):
            want.add_(parts[k])
        # the library sums a chunk in whatever order its ring visits the ranks -- but ONE rank forms each element's sum and every
        # rank receives that sum: equal to the rank-order sum to fp32 rounding (bit for bit with two ranks), identical on all ranks
        assert torch.allclose(mine, want, rtol=0, atol=1e-6 * float(want.abs().max())), (n, (mine - want).abs().max())
        if world == 2:
            assert torch.equal(mine, want)
        everyone = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        assert all(torch.equal(everyone[0], e) for e in everyone)
    small = torch.full((10,), float(rank + 1))
    dp.all_reduce_sum_(small)  # (below the size bar: the plain all-reduce)
    assert torch.equal(small, torch.full((10,), float(sum(range(1, world + 1)))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_exchange_sums_each_element_on_one_rank(world):
    """PVD_DP_EXCHANGE=sharded (opt-in) on a path without the flat optimizer's sharded update: the standard pair reduce_scatter_tensor +
    all_gather_into_tensor (round 6: replaces round 4's all-to-all two-shot, which RCCL could not capture).  Every element is summed by
    exactly one rank, so every rank ends with the same bits (the rank-order sum to rounding; exactly it with two ranks), with and
    without a padded tail; short tensors keep the plain all-reduce."""
    _setup_paths()
    mp.spawn(_sharded_worker, args=(world, _free_port()), nprocs=world, join=True)


@pytest.mark.timeout(600)
def test_ray_dp_step_with_the_sharded_exchange(tmp_path, monkeypatch):
    """The trainer's step under PVD_DP_EXCHANGE=sharded: replicas bit-identical (asserted inside the worker), the gradient equal to
    the all-reduce's up to the order of the fp32 sum over two ranks (a two-term sum commutes: equal bits here)."""
    _setup_paths()
    ref, two = str(tmp_path / "ar.pt"), str(tmp_path / "two.pt")
    mp.spawn(_worker, args=(2, _free_port(), ref, OPT_COMPACT, True), nprocs=2, join=True)
    monkeypatch.setenv("PVD_DP_EXCHANGE", "sharded")
    monkeypatch.setenv("PVD_DP_SHARDED_MIN", "64")  # (the toy model's compact gradient is short: take the path anyway; the loss sums stay all-reduces)
    mp.spawn(_worker, args=(2, _free_port(), two, OPT_COMPACT, True), nprocs=2, join=True)
    a, b = torch.load(ref), torch.load(two)
    assert torch.equal(a["flat"], b["flat"]) and a["loss"] == b["loss"]


# ---- evaluation split over the ranks, occupancy state agreed after an update

def _thicken(model):
    """an untrained VM student is empty space (sigma features of +-0.1): scale its sigma factors so that the picture shows something"""
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("sigma_"):
                p.mul_(8.0)


def _eval_worker(rank, world, port, out_path, res):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from pvd.scene import BLENDER_INTRINSICS, get_rays, synthetic_poses
    from pvd.trainer import RayDP
    dp = RayDP()
    w = _make(OPT, dp=dp)
    _thicken(w.stu)
    pose = torch.from_numpy(synthetic_poses(np.random.RandomState(7))[:1])
    r = get_rays(pose, tuple(v * res / 800.0 for v in BLENDER_INTRINSICS), res, res, -1)
    N = res * res
    bg = torch.rand(1, N, 3, generator=torch.Generator().manual_seed(3))  # a per-ray background travels with its rays
    w.stu.eval()
    with torch.no_grad():
        out = dp.render_sharded(w.stu, r["rays_o"], r["rays_d"], bg_color=bg, perturb=False, max_steps=1024)
    assert out["image"].shape == (1, N, 3) and out["depth"].shape == (1, N)
    both = torch.cat([out["image"].reshape(-1), out["depth"].reshape(-1)])
    everyone = [torch.empty_like(both) for _ in range(world)]
    dist.all_gather(everyone, both)
    # the whole image, the same bits, on every rank (bits: the depth of a ray that misses the box is NaN, as in the reference --
    # (inf - inf) / 0 in run_cuda's normalisation, renderer.py:539)
    assert all(torch.equal(everyone[0].view(torch.int32), e.view(torch.int32)) for e in everyone)
    if rank == 0:
        torch.save({"image": out["image"], "depth": out["depth"]}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_evaluation_render_split_over_the_ranks_is_the_one_rank_image(tmp_path, world):
    """RayDP.render_sharded (the reference all-gathers per-rank predictions, distill_mutual/utils.py:1243-1258): bands of an
    image rendered by 2 / 3 ranks and all-gathered equal the image one process renders -- 25 x 25 pixels, so the last band is padded
    (625 = 2 x 313 - 1 = 3 x 209 - 2) and the padding must not show."""
    _setup_paths()
    res = 25
    out = str(tmp_path / "eval.pt")
    mp.spawn(_eval_worker, args=(world, _free_port(), out, res), nprocs=world, join=True)
    got = torch.load(out)
    from pvd.scene import BLENDER_INTRINSICS, get_rays, synthetic_poses
    w = _make(OPT)
    _thicken(w.stu)
    pose = torch.from_numpy(synthetic_poses(np.random.RandomState(7))[:1])
    r = get_rays(pose, tuple(v * res / 800.0 for v in BLENDER_INTRINSICS), res, res, -1)
    bg = torch.rand(1, res * res, 3, generator=torch.Generator().manual_seed(3))
    w.stu.eval()
    with torch.no_grad():
        want = w.stu.render(r["rays_o"], r["rays_d"], staged=True, bg_color=bg, perturb=False, max_steps=1024)
    assert float(torch.nan_to_num(want["depth"], nan=0.0).max()) > 0 and float((want["image"] - bg).abs().max()) > 0.05  # the object is in the picture
    # (CPU: the library's matrix products may block a band differently from the whole image -- rounding, not bits; the GPU form of
    # this test asserts equality)
    assert (got["image"] - want["image"].float()).abs().max().item() <= 1e-5
    assert torch.equal(torch.isnan(got["depth"]), torch.isnan(want["depth"]))  # (rays that miss the box)
    assert (torch.nan_to_num(got["depth"], nan=0.0) - torch.nan_to_num(want["depth"].float(), nan=0.0)).abs().max().item() <= 1e-5


TEA_OPT = dict(num_rays=256, resolution0=24, iters=50, fp16=False, model_type="vm", teacher_type="vm", update_extra_interval=16,
               stage_iters={"stage1": -1, "stage2": -1})


def _teacher_and_batch(dp=None):
    from oracle_ops import oracle_ops
    from pvd.config import PVDConfig
    from pvd.scene import BLENDER_INTRINSICS, ChairScene, get_rays, synthetic_poses
    from pvd.trainer import TeacherTrainer
    from pvd.workload import AnalyticTarget, install_occupancy, make_model
    opt = PVDConfig(**TEA_OPT)
    torch.manual_seed(0)
    m = make_model(oracle_ops(), opt, "vm", True, torch.device("cpu"), teacher_variant=True)
    scene = ChairScene()
    install_occupancy(m, scene, opt)
    m.mean_count = 0  # (first block: the marcher sizes its buffers exactly)
    tr = TeacherTrainer(opt, m, "cpu", fp16=False, dp=dp)
    gen = torch.Generator().manual_seed(11)
    pose = torch.from_numpy(synthetic_poses(np.random.RandomState(0))[:1])
    r = get_rays(pose, BLENDER_INTRINSICS, 800, 800, opt.num_rays, generator=gen)
    bg = torch.rand(1, opt.num_rays, 3, generator=gen)
    gt = AnalyticTarget(oracle_ops(), scene, m)(r["rays_o"], r["rays_d"], bg)
    return m, tr, (r["rays_o"], r["rays_d"], gt, bg)


def _teacher_worker(rank, world, port, out_path):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from pvd.trainer import RayDP
    dp = RayDP()
    m, tr, (o, d, gt, bg) = _teacher_and_batch(dp)
    half = o.shape[1] // world
    sl = slice(rank * half, (rank + 1) * half)
    torch.manual_seed(100 + rank)  # the occupancy update's random cells and jitter differ from rank to rank ...
    assert tr.global_step % tr.opt.update_extra_interval == 0
    loss, _ = tr.train_step(o[:, sl].contiguous(), d[:, sl].contiguous(), gt[:, sl].contiguous(), bg[:, sl].contiguous())
    state = torch.cat([m.density_grid.reshape(-1), m.density_bitfield.reshape(-1).float(),
                       torch.tensor([float(m.mean_density), float(m.iter_density)])])
    everyone = [torch.empty_like(state) for _ in range(world)]
    dist.all_gather(everyone, state)
    assert all(torch.equal(everyone[0], e) for e in everyone)  # ... and every rank marches on rank 0's grid all the same
    params = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    gp = [torch.empty_like(params) for _ in range(world)]
    dist.all_gather(gp, params)
    assert all(torch.equal(gp[0], t) for t in gp)
    if rank == 0:
        torch.save({"loss": float(loss), "flat": tr.flat.flat.clone(), "grid": m.density_grid.clone(), "bits": m.density_bitfield.clone()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_teacher_step_on_two_ranks_agrees_on_the_occupancy_grid_and_equals_one_process(tmp_path):
    """Teacher training under ray-DP (occupancy state): the step that updates the occupancy grid, on two ranks whose
    random streams differ.  After it every rank holds rank 0's grid / bitfield / running mean (RayDP.sync_occupancy), the replicas'
    parameters are identical, and loss and gradient are those of one process that ran rank 0's update and rendered both shards."""
    _setup_paths()
    out = str(tmp_path / "tea.pt")
    mp.spawn(_teacher_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    m, tr, (o, d, gt, bg) = _teacher_and_batch()
    torch.manual_seed(100)  # rank 0's stream
    m.update_extra_state()
    assert torch.equal(m.density_grid, got["grid"]) and torch.equal(m.density_bitfield, got["bits"])
    tr.flat.zero_()
    half = o.shape[1] // 2
    errs = []
    for r in range(2):
        sl = slice(r * half, (r + 1) * half)
        out_r = m.render(o[:, sl].contiguous(), d[:, sl].contiguous(), staged=False, bg_color=bg[:, sl].contiguous(), perturb=True,
                         force_all_rays=False, dt_gamma=0, max_steps=1024)
        errs.append((out_r["image"].float() - gt[:, sl].float()) ** 2)
    mse = torch.cat(errs, dim=1).mean()
    l1 = tr._l1_term() if (tr.opt.l1_reg_weight > 0.0) else torch.zeros(())
    (mse + l1).backward()
    # a rank's reported loss carries 1/G of the parameter-only L1 term (its gradients are summed over ranks)
    assert abs((float(mse.detach()) + float(l1.detach()) / 2) - got["loss"]) <= 1e-5 * abs(got["loss"]), (float(mse), float(l1), got["loss"])
    flat = tr.flat.flat
    scale = flat.abs().max().item()
    assert scale > 0
    assert (flat - got["flat"]).abs().max().item() <= 2e-5 * scale, ((flat - got["flat"]).abs().max().item(), scale)


OPT_HASH = dict(num_rays=256, iters=50, fp16=False, model_type="hash", loss_rate_fea_sc=0.0, loss_rate_color=0.0, loss_rate_sigma=0.0, l1_reg_weight=0.0)


@pytest.mark.timeout(900)
def test_ray_dp_hash_student_two_ranks_equals_single_process(tmp_path):
    """BASELINE configs[4]'s student (hash table + sigma / colour MLPs) under ray-DP: two ranks on their halves of the rays against one
    process on both shards -- the global rgb norm, and the gradient of EVERY parameter (the table's scatter-add result included)."""
    _setup_paths()
    out = str(tmp_path / "dph.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, OPT_HASH, False, None, 1.5), nprocs=2, join=True)
    dp_res = torch.load(out)
    w = _make(OPT_HASH)
    with torch.no_grad():
        for p in w.stu.parameters():
            p.mul_(1.5)
    base = _make(OPT_HASH)
    rays_o, rays_d, bg = base.next_batch()
    tr, stu, tea = w.trainer, w.stu, w.tea
    tr.opt.global_step = tr.global_step
    tr.flat.zero_()
    diffs = []
    half = OPT_HASH["num_rays"] // 2
    for r in range(2):
        sl = slice(r * half, (r + 1) * half)
        o, d, b = rays_o[:, sl].contiguous(), rays_d[:, sl].contiguous(), bg[:, sl].contiguous()
        out_s = stu.render(o, d, staged=False, bg_color=b, perturb=True, force_all_rays=False, dt_gamma=0, max_steps=1024)
        with torch.no_grad():
            out_t = tea.render(o, d, staged=False, bg_color=b, perturb=True, force_all_rays=False,
                               inherited_params=out_s["inherited_params"], dt_gamma=0, max_steps=1024)
        diffs.append(out_t["image"] - out_s["image"])
    l_rgb = torch.norm(torch.cat(diffs, dim=1))
    (l_rgb * tr.opt.loss_rate_rgb).backward()
    assert abs(float(l_rgb.detach()) - dp_res["rgb"]) <= 1e-5 * abs(dp_res["rgb"]), (float(l_rgb), dp_res["rgb"])
    flat = tr.flat.flat
    scale = flat.abs().max().item()
    assert scale > 0
    assert (flat - dp_res["flat"]).abs().max().item() <= 2e-5 * scale, ((flat - dp_res["flat"]).abs().max().item(), scale)
    n_table = stu.encoder.embeddings.numel()
    assert n_table > 0 and float(dp_res["flat"].abs().max()) > 0


OPT_TENSORS = dict(num_rays=256, iters=50, fp16=False, model_type="tensors", plenoxel_res="[32,32,32]", loss_rate_fea_sc=0.0, loss_rate_color=0.0,
                   loss_rate_sigma=0.0, l1_reg_weight=0.0)


@pytest.mark.timeout(900)
def test_ray_dp_plenoxel_student_two_ranks_equals_single_process(tmp_path):
    """BASELINE configs[3]'s student (the Plenoxel volume) under ray-DP with the compact exchange (only voxels under occupied cells cross
    the links): two ranks on their halves of the rays against one process on both shards."""
    _setup_paths()
    out = str(tmp_path / "dpt.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, OPT_TENSORS, True), nprocs=2, join=True)
    dp_res = torch.load(out)
    w = _make(OPT_TENSORS)
    base = _make(OPT_TENSORS)
    rays_o, rays_d, bg = base.next_batch()
    tr, stu, tea = w.trainer, w.stu, w.tea
    tr.opt.global_step = tr.global_step
    tr.flat.zero_()
    diffs = []
    half = OPT_TENSORS["num_rays"] // 2
    for r in range(2):
        sl = slice(r * half, (r + 1) * half)
        o, d, b = rays_o[:, sl].contiguous(), rays_d[:, sl].contiguous(), bg[:, sl].contiguous()
        out_s = stu.render(o, d, staged=False, bg_color=b, perturb=True, force_all_rays=False, dt_gamma=0, max_steps=1024)
        with torch.no_grad():
            out_t = tea.render(o, d, staged=False, bg_color=b, perturb=True, force_all_rays=False,
                               inherited_params=out_s["inherited_params"], dt_gamma=0, max_steps=1024)
        diffs.append(out_t["image"] - out_s["image"])
    l_rgb = torch.norm(torch.cat(diffs, dim=1))
    (l_rgb * tr.opt.loss_rate_rgb).backward()
    assert abs(float(l_rgb.detach()) - dp_res["rgb"]) <= 1e-5 * abs(dp_res["rgb"]), (float(l_rgb), dp_res["rgb"])
    flat = tr.flat.flat
    scale = flat.abs().max().item()
    assert scale > 0
    assert (flat - dp_res["flat"]).abs().max().item() <= 2e-5 * scale, ((flat - dp_res["flat"]).abs().max().item(), scale)
