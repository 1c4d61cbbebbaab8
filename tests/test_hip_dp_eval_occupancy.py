"""SURVEY 8(e)'s two non-gradient bullets on the GPU, two ranks (gloo, sharing cuda:0):

  * evaluation: an image rendered as bands by the ranks and all-gathered (RayDP.render_sharded; the reference all-gathers per-rank
    predictions, distill_mutual/utils.py:1243-1258) has the BITS of the one-rank render -- through the persistent inference launch
    of the hash teacher and of the VM student, with a band boundary inside an image row and a padded last band;
  * occupancy state: `update_extra_state` (device-side: csrc/occupancy.hip) on both ranks, bit-identical replicas, the same call
    count: the grids may STILL differ (the list of occupied cells is compacted with atomics, its order decides which occupied
    cells are re-queried) -- RayDP.sync_occupancy leaves rank 0's grid, bitfield and running mean everywhere, moves the occupancy
    epoch only on a rank whose grid changed, and a second call finds nothing to do."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _paths():
    for p in (REPO, os.path.join(REPO, "aaai2023-pvd_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)


def _workload(dp=None, pretrain=30):
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import DistillWorkload
    dev = torch.device("cuda:0")
    return DistillWorkload(hip_ops(), dev, PVDConfig(num_rays=1024, iters=300, resolution0=64), teacher_pretrain_steps=pretrain, seed=0, dp=dp)


def _image_rays(res, dev):
    from pvd.scene import BLENDER_INTRINSICS, get_rays, synthetic_poses
    pose = torch.from_numpy(synthetic_poses(np.random.RandomState(7))[:1]).to(dev)
    return get_rays(pose, tuple(v * res / 800.0 for v in BLENDER_INTRINSICS), res, res, -1)


def _bits(t):
    return t.float().contiguous().view(torch.int32)


def _worker(rank, world, port, out_path, res):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    _paths()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    import pvd_hip
    from pvd.trainer import RayDP
    dp = RayDP()
    w = _workload(dp)
    for m in (w.tea, w.stu):  # replicas start bit-identical (teacher pre-training uses float atomics), as bench.py does
        for t in list(m.parameters()) + list(m.buffers()):
            d = t.data
            if not d.is_contiguous():
                d = d.permute(0, 2, 3, 1) if d.dim() == 4 else d.permute(0, 2, 3, 4, 1)
            dp.broadcast_(d, src=0)
        pvd_hip.note_weights_changed(list(m.parameters()))
        m.eval()
    r = _image_rays(res, dev)
    res_out = {}
    with torch.no_grad():
        for name, m in (("tea", w.tea), ("stu", w.stu)):
            with torch.autocast("cuda", dtype=torch.float16):
                out = dp.render_sharded(m, r["rays_o"], r["rays_d"], bg_color=1, perturb=False, max_steps=1024)
                one = m.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False, max_steps=1024)
            assert out["image"].shape == (1, res * res, 3) and out["depth"].shape == (1, res * res)
            # the bands, gathered, are the image this very rank renders alone
            assert torch.equal(_bits(out["image"]), _bits(one["image"])), name
            assert torch.equal(_bits(out["depth"]), _bits(one["depth"])), name
            res_out[name] = (out["image"].cpu(), out["depth"].cpu())
            both = torch.cat([_bits(out["image"]).reshape(-1), _bits(out["depth"]).reshape(-1)]).cpu()
            ev = [torch.empty_like(both) for _ in range(world)]
            dist.all_gather(ev, both)
            assert all(torch.equal(ev[0], e) for e in ev), name

    # ---- occupancy: the device-side update on both ranks, then the agreement
    tea = w.tea
    tea.train()
    tea.iter_density = 20  # (a partial update: random cells + occupied cells)
    with torch.autocast("cuda", dtype=torch.float16):
        tea.update_extra_state()
    def everyone_equal():
        state = torch.cat([tea.density_grid.reshape(-1).float(), tea.density_bitfield.reshape(-1).float(),
                           torch.tensor([float(tea.mean_density), float(tea.iter_density)], device=dev)]).cpu()
        ev = [torch.empty_like(state) for _ in range(world)]
        dist.all_gather(ev, state)
        return all(torch.equal(ev[0], e) for e in ev)
    epoch = tea.occ_epoch
    agreed_by_itself = everyone_equal()
    same = dp.sync_occupancy(tea)
    assert rank != 0 or same  # rank 0 keeps its own
    assert tea.occ_epoch == epoch + (0 if same else 1)
    assert everyone_equal()
    assert agreed_by_itself or rank == 0 or not same  # (if the updates disagreed, rank 1 was the one that changed)
    epoch = tea.occ_epoch
    assert dp.sync_occupancy(tea) and tea.occ_epoch == epoch  # nothing left to do
    if rank == 1:  # a replica that drifted: put right by the broadcast, and it knows
        tea.density_grid[0, 12345] += 1.0
        tea.density_bitfield[77] ^= 0x5A
    same = dp.sync_occupancy(tea)
    assert same == (rank == 0) and tea.occ_epoch == epoch + (0 if rank == 0 else 1)
    assert everyone_equal()
    res_out["updates_agreed_by_themselves"] = bool(agreed_by_itself)
    if rank == 0:
        torch.save(res_out, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(1800)
def test_sharded_evaluation_has_the_bits_of_the_one_rank_render_and_replicas_agree_on_the_grid(tmp_path):
    res = 75  # 5625 rays = 2 x 2813 - 1: the band boundary falls inside an image row and the last band is padded
    out = str(tmp_path / "eval.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, res), nprocs=2, join=True)
    got = torch.load(out)
    print("device-side occupancy updates of two identical replicas agreed by themselves:", got["updates_agreed_by_themselves"])
    for name in ("tea", "stu"):
        img, depth = got[name]
        assert torch.isfinite(img).all()
    # the trained teacher shows the chair: not a blank picture
    assert float((got["tea"][0] - 1.0).abs().max()) > 0.2 and float(torch.nan_to_num(got["tea"][1], nan=0.0).max()) > 0.1


def _hash_student_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    _paths()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    import pvd_hip
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.trainer import RayDP
    from pvd.workload import DistillWorkload
    dp = RayDP()
    w = DistillWorkload(hip_ops(), dev, PVDConfig(num_rays=1024, iters=300, model_type="hash"), teacher_pretrain_steps=20, seed=0, dp=dp)
    for m in (w.tea, w.stu):
        for t in list(m.parameters()) + list(m.buffers()):
            dp.broadcast_(t.data, src=0)
        pvd_hip.note_weights_changed(list(m.parameters()))
    with torch.no_grad():  # a hash student takes over EVERY tensor of a hash teacher: make it a different model, or there is nothing to learn
        for p in w.stu.parameters():
            p.mul_(1.5)
    pvd_hip.note_weights_changed(list(w.stu.parameters()))
    p0 = torch.cat([p.detach().reshape(-1).float() for p in w.stu.parameters()]).clone()
    losses = []
    for _ in range(4):  # eager steps: every rank its own rays, the table's gradient leaves the backward in half precision
        loss, info, _, _ = w.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    tr = w.trainer
    if getattr(tr, "flat_opt", False):
        tr.optimizer.flush()
    params = torch.cat([p.detach().reshape(-1).float() for p in w.stu.parameters()]).cpu()
    both = [torch.empty_like(params) for _ in range(world)]
    dist.all_gather(both, params)
    assert all(torch.equal(both[0], t) for t in both), "replicas of the hash student differ after four ray-DP steps"
    assert all(l == l and 1e-6 < abs(l) < 1e6 for l in losses), losses  # (a real objective: gradients flowed)
    moved = (params - p0.cpu()).abs()
    table = w.stu.encoder.embeddings.detach().float().cpu().reshape(-1)
    assert float(moved.max()) > 0 and float((table - p0.cpu()[:table.numel()]).abs().max()) >= 0  # (the step did update something)
    # ---- the exchange itself, deterministically: a synthetic half-precision table gradient per rank (no atomics), the heads' fp32 gradients = rank + 1
    emb = w.stu.encoder.embeddings
    o = tr.optimizer
    tr._zero_grads()
    tables = [(torch.randn(emb.shape, generator=torch.Generator(device=dev).manual_seed(100 + r), device=dev) * 0.01).half() for r in range(world)]
    mine = tables[rank].clone()
    assert o.accept_half_grad(emb, mine), "the hash table's gradient is taken in half precision under ray-DP too (PVD_DP_HASH_WIRE=f16)"
    lo, hi, _ = o._half_grad
    assert hi - lo == emb.numel()
    o.flat_g[:lo] = float(rank + 1)
    o.flat_g[hi:] = float(rank + 1)
    tr._exchange()
    torch.cuda.synchronize()
    want = tables[0].clone()
    for r in range(1, world):
        want += tables[r]  # half + half, rounded once: what a half-precision all-reduce of two ranks leaves
    assert torch.equal(mine, want), "the half table was not summed in half precision over the ranks"
    assert float(o.flat_g[lo:hi].abs().max()) == 0.0, "the table's fp32 range stays zero_grad's zeros (nothing is widened into it)"
    heads = torch.cat([o.flat_g[:lo], o.flat_g[hi:]])
    assert heads.numel() > 0 and bool((heads == float(sum(range(1, world + 1)))).all()), "the heads' fp32 gradients are summed next to it"
    o._half_grad = None
    tr._zero_grads()
    if rank == 0:
        torch.save({"losses": losses, "moved": float(moved.max()), "moved_rows": int((moved > 0).sum())}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(1800)
def test_hash_student_replicas_stay_identical_under_ray_dp(tmp_path):
    """BASELINE configs[4] (hash -> hash on 8 GPUs): the hash table's gradient leaves the scatter in HALF precision and crosses the
    links as it is (21 instead of 42 MB; the reference's arithmetic for this gradient is ONE half table every sample adds into), the
    heads' fp32 gradients next to it (pvd/trainer.py: _exchange).  Two ranks with different rays: after four steps the replicas'
    parameters are the same bits -- every rank applied the SUMMED gradient, the table's included; then the exchange alone on
    synthetic gradients: the half tables' sum in half precision, the fp32 table range untouched, the heads summed."""
    out = str(tmp_path / "hash_dp.pt")
    mp.spawn(_hash_student_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["moved"] > 0 and got["moved_rows"] > 1000, got
