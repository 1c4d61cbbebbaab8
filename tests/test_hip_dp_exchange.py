"""Ray-DP's gradient exchange in its round-6 forms against the classic sequence, DETERMINISTICALLY: two ranks (gloo, sharing
cuda:0) put seeded synthetic gradients on the touched rows -- no backward, hence no float atomics -- and run the trainer's own
exchange + update in each form:

  classic   : gather -> all-reduce -> scatter + inf check -> AdamW                                  (rounds 1-5)
  allreduce : gather that zeroes and checks, flag word in the buffer -> all-reduce -> AdamW part B reads the buffer
  sharded   : ... -> reduce_scatter -> AdamW on this rank's rows -> all_gather of the updated rows -> scatter into the parameters

A two-term fp32 sum commutes, so all three must leave the SAME BITS in parameters, moments, step count, learning rates and loss
scale -- on both ranks; an inf planted on ONE rank must skip the step on both (and halve the scale) in all three."""
import os
import socket
import sys

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


def _worker(rank, world, port, mode, out_path, student="vm"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PVD_DP_EXCHANGE=mode)
    for p in (REPO, os.path.join(REPO, "aaai2023-pvd_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.trainer import RayDP
    from pvd.workload import DistillWorkload
    dp = RayDP()
    kw = dict(resolution0=64) if student == "vm" else dict(model_type="tensors", plenoxel_res="[64,64,64]")
    w = DistillWorkload(hip_ops(), dev, PVDConfig(num_rays=1024, iters=300, **kw), teacher_pretrain_steps=0, seed=0, dp=dp)
    tr = w.trainer
    o = tr.optimizer
    tr.scaler.scale(torch.zeros((), device=dev))  # (creates the device-side loss scale)
    if student == "vm":
        tr._l1_term(partials_only=True)           # the L1 regulariser folded into the update: rows that are warm without a gradient
    tr._zero_grads()                              # builds the compactor / touched set, one full clear
    c = tr._grad_compactor()
    assert c is not None and c is o.touched and c.fraction < 0.7
    if student == "vm":
        assert o.begin_two_part(defer=False)      # the update in its two-part form, eagerly (part A right behind part B)
    else:  # the Plenoxel student has no deferred rows: ONE launch whose warm list is the touched set (built by a first, gradient-free step)
        o.step()
        assert not o.begin_two_part(defer=False) and o._warm_A.numel() == 0 and o._warm_B.numel() * 4 == c.idx.numel()
    assert (tr._exchange_mode(c) is None) == (mode == "classic") and (mode == "classic" or tr._exchange_mode(c) == mode)
    scales, skipped = [], []
    for k in range(6):
        tr._zero_grads()
        assert float(o.flat_g.abs().max()) == 0.0, "zero_grad left something behind (step %d)" % k
        g = torch.Generator(device=dev).manual_seed(1000 * k + rank)
        vals = torch.randn(c.idx.numel(), generator=g, device=dev) * (float(tr.scaler.get_scale()) * 1e-3)
        o.flat_g[c.idx] = vals
        if k == 3 and rank == 1:
            o.flat_g[c.idx[5]] = float("inf")
        before = o.flat_p.clone()
        tr._exchange()
        tr._optimize()
        tr.scheduler.step()
        skipped.append(bool(torch.equal(before[c.idx], o.flat_p[c.idx])))
        scales.append(float(tr.scaler.get_scale()))
    assert skipped == [False, False, False, True, False, False], skipped
    assert scales[3] == scales[2] * 0.5, scales
    o.end_two_part()
    tr.sync_sharded_state()
    o.flush()
    torch.cuda.synchronize()
    state = {"p": o.flat_p.cpu(), "m": o.flat_m.cpu(), "v": o.flat_v.cpu(), "step": o.step_count.cpu(), "lr": o.lr_dev.cpu(), "scales": scales}
    for name in ("p", "m", "v", "step", "lr"):  # replicas identical
        both = [torch.zeros_like(state[name]) for _ in range(world)]
        dist.all_gather(both, state[name])
        assert all(torch.equal(both[0], t) for t in both), "replicas differ in %s (%s)" % (name, mode)
    if rank == 0:
        torch.save(state, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("student", ["vm", "tensors"])
def test_exchange_forms_leave_the_same_bits(tmp_path, student):
    res = {}
    for mode in ("classic", "allreduce", "sharded"):
        out = str(tmp_path / ("x_%s.pt" % mode))
        mp.spawn(_worker, args=(2, _free_port(), mode, out, student), nprocs=2, join=True)
        res[mode] = torch.load(out)
    ref = res["classic"]
    assert float(ref["step"]) == (5.0 if student == "vm" else 6.0)  # six steps, one skipped (+ the Plenoxel run's gradient-free first step)
    for mode in ("allreduce", "sharded"):
        for name in ("p", "m", "v", "step", "lr"):
            assert torch.equal(ref[name], res[mode][name]), "%s: %s differs from the classic sequence (max abs %g)" % (
                mode, name, float((ref[name] - res[mode][name]).abs().max()))
        assert ref["scales"] == res[mode]["scales"]
