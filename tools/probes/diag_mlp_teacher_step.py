#!/usr/bin/env python3
"""Why does an MLP model trained through libpvd_hip.so end ~1 dB below the same training through the reference's kernels + PyTorch
(tools/psnr_vs_reference_kernels.py --teacher-training --teacher-type mlp)?  One training step of the mlp (or hash) model from identical
weights on an identical batch through both stacks: loss, image, every parameter's gradient (unscaled), every parameter after the update."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tools")]
import torch

from bench_reference_kernels_step import reference_kernel_ops
from pvd.config import PVDConfig
from pvd.ops import hip_ops
from pvd.scene import BLENDER_INTRINSICS, get_rays
from pvd.trainer import TeacherTrainer
from pvd.workload import DistillWorkload, measure_mean_count

kind = sys.argv[1] if len(sys.argv) > 1 else "mlp"
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
res = {}
for name, ops in (("ref", reference_kernel_ops()), ("hip", hip_ops())):
    torch.manual_seed(0)
    torch.cuda.manual_seed(1234)
    opt = PVDConfig(model_type="vm", teacher_type=kind, fp16=True)
    w = DistillWorkload(ops, dev, opt, teacher_pretrain_steps=0, seed=0)
    topt = PVDConfig(**{**opt.__dict__, "model_type": kind, "iters": 3000, "update_extra_interval": 10 ** 9, "stage_iters": {"stage1": -1, "stage2": -1}})
    w.tea.teacher_variant = True
    w.tea.args = w.tea.opt = topt
    for p in w.tea.parameters():
        p.requires_grad = True
    tr = TeacherTrainer(topt, w.tea, dev, fp16=True)
    w.tea.mean_count = measure_mean_count(w.tea, w.poses, opt, generator=w.gen)
    w.gen.manual_seed(99)
    grads = {}
    orig = tr._optimize

    def grab(tr=tr, grads=grads, orig=orig):
        sc = float(tr.scaler.get_scale()) if hasattr(tr.scaler, "get_scale") else 1.0
        for n, p in tr.model.named_parameters():
            if p.grad is not None:
                grads[n] = (p.grad.detach().float() / sc).clone()
        grads["_scale"] = sc
        orig()
    tr._optimize = grab
    init = {n: p.detach().float().clone() for n, p in w.tea.named_parameters()}
    for it in range(n_steps):
        r = get_rays(w.poses[it % len(w.poses)][None], BLENDER_INTRINSICS, 800, 800, opt.num_rays, generator=w.gen)
        bg = torch.rand(1, opt.num_rays, 3, device=dev, generator=w.gen)
        gt = w.target(r["rays_o"], r["rays_d"], bg)
        loss, pred = tr.train_step(r["rays_o"], r["rays_d"], gt, bg)
    torch.cuda.synchronize()
    if tr.flat_opt:
        tr.optimizer.flush()
    res[name] = dict(loss=float(loss), pred=pred.detach().float().clone(), grads=grads, init=init,
                     after={n: p.detach().float().clone() for n, p in w.tea.named_parameters()},
                     lr=[float(g["lr"]) for g in tr.optimizer.param_groups] if hasattr(tr.optimizer, "param_groups") else None)
a, b = res["ref"], res["hip"]
print("model %s, %d step(s); loss ref %.6f hip %.6f; image max |diff| %.3g; loss scales %s / %s; lrs ref %s hip %s"
      % (kind, n_steps, a["loss"], b["loss"], float((a["pred"] - b["pred"]).abs().max()), a["grads"]["_scale"], b["grads"]["_scale"], a["lr"], b["lr"]))
print("%-32s %12s %14s %14s %16s %16s" % ("parameter", "init equal", "|g| max ref", "g diff / max", "update max ref", "update diff/max"))
for n in a["init"]:
    ga, gb = a["grads"].get(n), b["grads"].get(n)
    ua, ub = a["after"][n] - a["init"][n], b["after"][n] - b["init"][n]
    gs = float(ga.abs().max()) if ga is not None else float("nan")
    gd = float((ga - gb).abs().max()) / max(gs, 1e-30) if (ga is not None and gb is not None) else float("nan")
    us = float(ua.abs().max())
    print("%-32s %12s %14.4g %14.3g %16.4g %16.3g" % (n, bool(torch.equal(a["init"][n], b["init"][n])), gs, gd, us, float((ua - ub).abs().max()) / max(us, 1e-30)))
