#!/usr/bin/env python3
"""Which line of the harness issues which small launch of a teacher-training step (configs[1])?  A few eager steps under torch.profiler
with Python stacks; prints every device kernel of one step with the innermost repo frame that issued it."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd")]
import torch
from torch.profiler import ProfilerActivity, profile

from pvd.config import PVDConfig
from pvd.ops import hip_ops
from pvd.scene import BLENDER_INTRINSICS, get_rays
from pvd.trainer import TeacherTrainer
from pvd.workload import DistillWorkload, measure_mean_count

kind = sys.argv[1] if len(sys.argv) > 1 else "hash"
dev = torch.device("cuda:0")
opt = PVDConfig(model_type="vm", teacher_type=kind, fp16=True)
w = DistillWorkload(hip_ops(), dev, opt, teacher_pretrain_steps=0, seed=0)
topt = PVDConfig(**{**opt.__dict__, "model_type": kind, "iters": 3000, "update_extra_interval": 10 ** 9, "stage_iters": {"stage1": -1, "stage2": -1}})
w.tea.teacher_variant = True
w.tea.args = w.tea.opt = topt
for p in w.tea.parameters():
    p.requires_grad = True
tr = TeacherTrainer(topt, w.tea, dev, fp16=True)
w.tea.mean_count = measure_mean_count(w.tea, w.poses, opt, generator=w.gen)


def step(it):
    r = get_rays(w.poses[it % len(w.poses)][None], BLENDER_INTRINSICS, 800, 800, opt.num_rays, generator=w.gen)
    bg = torch.rand(1, opt.num_rays, 3, device=dev, generator=w.gen)
    gt = w.target(r["rays_o"], r["rays_d"], bg)
    torch.cuda.synchronize()
    return tr.train_step(r["rays_o"], r["rays_d"], gt, bg)


for i in range(4):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    r = get_rays(w.poses[5][None], BLENDER_INTRINSICS, 800, 800, opt.num_rays, generator=w.gen)
    bg = torch.rand(1, opt.num_rays, 3, device=dev, generator=w.gen)
    gt = w.target(r["rays_o"], r["rays_d"], bg)
    torch.cuda.synchronize()
    mark = torch.zeros(1, device=dev)  # (a recognisable first launch)
    tr.train_step(r["rays_o"], r["rays_d"], gt, bg)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.kernels]
for e in sorted(evs, key=lambda e: e.time_range.start):
    frames = [f for f in (e.stack or []) if "/root/repo" in f or "aaai2023" in f or "pvd" in f]
    where = frames[0] if frames else (e.stack[0] if e.stack else "?")
    for k in e.kernels:
        print("%-58s %6.1f us   %s   <- %s" % (k.name[:58], k.duration, e.name[:28], where[-110:]))
