#!/usr/bin/env python3
"""k_grid_bwd_lps2 on the inputs of a REAL teacher training step (captured after N eager steps of bench.py's teacher workload) against
the same launch with the gradient values, the sample order or the positions replaced: what makes the in-training launch 250-280 us
when tools/bench_grid_bwd.py's launch of the same size takes ~150?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "aaai2023-pvd_amd")]
import torch

import pvd_hip
from pvd.config import PVDConfig
from pvd.ops import hip_ops
from pvd.scene import BLENDER_INTRINSICS, get_rays
from pvd.trainer import TeacherTrainer
from pvd.workload import DistillWorkload, measure_mean_count

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 336
opt = PVDConfig()
wl = DistillWorkload(hip_ops(), dev, opt, teacher_pretrain_steps=0)
topt = PVDConfig(**{**opt.__dict__, "model_type": opt.teacher_type, "iters": 30000, "stage_iters": {"stage1": -1, "stage2": -1}})
tea = wl.tea
tea.teacher_variant = True
tea.requires_grad_(True).train()
tea.args = tea.opt = topt
tr = TeacherTrainer(topt, tea, dev, fp16=True)
tea.mean_count = measure_mean_count(tea, wl.poses, opt, generator=wl.gen)
batches = []
for it in range(8):
    r = get_rays(wl.poses[it % len(wl.poses)][None], BLENDER_INTRINSICS, 800, 800, opt.num_rays, generator=wl.gen)
    bg = torch.rand(1, opt.num_rays, 3, device=dev, generator=wl.gen)
    batches.append((r["rays_o"], r["rays_d"], wl.target(r["rays_o"], r["rays_d"], bg), bg))
captured = {}
orig = pvd_hip.grid_encode_backward


def spy(grad, inputs, emb, offsets, ge, B, *rest):
    captured["args"] = (grad.clone(), inputs.clone(), emb, offsets, B, rest)
    return orig(grad, inputs, emb, offsets, ge, B, *rest)


for it in range(steps):
    if it == steps - 1:
        pvd_hip.grid_encode_backward = spy
        import fusedhead
        fusedhead.pvd_hip.grid_encode_backward = spy
    tr.train_step(*batches[it % 8])
pvd_hip.grid_encode_backward = orig
torch.cuda.synchronize()
grad, x, emb, offsets, B, rest = captured["args"]
print("captured after %d steps: B = %d rows, mean_count %d; grad dtype %s, |grad| max %.3e, zero rows %.1f %%, positions at exactly 0 or 1: %.1f %%"
      % (steps, B, int(tea.mean_count), grad.dtype, float(grad.abs().max()), 100 * float((grad.abs().amax((0, 2)) == 0).float().mean()),
         100 * float(((x == 0) | (x == 1)).all(1).float().mean())))
g32 = grad.float()
print("grad magnitude percentiles (non-zero entries): ", [float(v) for v in torch.quantile(g32[g32 != 0].abs()[:4000000], torch.tensor([0.01, 0.5, 0.99], device=dev))])
print("f16 subnormal entries (0 < |g| < 6.1e-5): %.1f %%" % (100 * float(((g32.abs() > 0) & (g32.abs() < 6.1e-5)).float().mean())))


def timeit(g, xx, iters=10):
    ge = torch.zeros(emb.shape, dtype=g.dtype, device=dev)
    run = lambda: orig(g, xx, ge, offsets, ge, B, *rest)
    run()
    with pvd_hip.KernelTimer({"pvd_grid_encode_backward"}) as kt:
        for _ in range(iters):
            run()
    return kt.mean_ms("pvd_grid_encode_backward") * 1e3


print("as captured                                   : %7.1f us" % timeit(grad, x))
print("same positions, gradient = randn * 1e-3       : %7.1f us" % timeit((torch.randn_like(g32) * 1e-3).to(grad.dtype), x))
print("same positions, gradient = 1.0                : %7.1f us" % timeit(torch.ones_like(grad), x))
live = grad.abs().amax((0, 2)) != 0
print("same positions, subnormals flushed to zero    : %7.1f us" % timeit(torch.where(g32.abs() < 6.1e-5, torch.zeros_like(g32), g32).to(grad.dtype), x))
perm = torch.randperm(B, device=dev)
print("rows shuffled (same gradient per row)         : %7.1f us" % timeit(grad[:, perm].contiguous(), x[perm].contiguous()))
xr = torch.rand_like(x)
print("uniform random positions, captured gradient   : %7.1f us" % timeit(grad, xr))
n = int(live.sum())
print("rows with a non-zero gradient: %d of %d" % (n, B))
