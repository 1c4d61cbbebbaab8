"""Teacher training step (BASELINE.json configs[1]: hash teacher, 4096 rays/batch) on one GPU:
fused hash head (grid + MFMA head both ways) against the layer-by-layer torch formulation.
  python tools/bench_teacher.py [--steps 40]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aaai2023-pvd_amd"))

from pvd.config import PVDConfig  # noqa: E402
from pvd.ops import hip_ops  # noqa: E402
from pvd.scene import BLENDER_INTRINSICS, get_rays  # noqa: E402
from pvd.trainer import TeacherTrainer, psnr  # noqa: E402
from pvd.workload import DistillWorkload, measure_mean_count  # noqa: E402


def run(fused, steps, warmup):
    ops = hip_ops()
    if not fused:
        ops.fused_head = None
    dev = torch.device("cuda:0")
    opt = PVDConfig()
    wl = DistillWorkload(ops, dev, opt, teacher_pretrain_steps=0)
    topt = PVDConfig(**{**opt.__dict__, "model_type": opt.teacher_type, "iters": 30000, "update_extra_interval": 10 ** 9,
                        "stage_iters": {"stage1": -1, "stage2": -1}})
    tea = wl.tea
    tea.teacher_variant = True
    tea.requires_grad_(True).train()  # DistillWorkload froze it
    tea.args = tea.opt = topt
    tr = TeacherTrainer(topt, tea, dev, fp16=True)
    tea.mean_count = measure_mean_count(tea, wl.poses, opt, generator=wl.gen)
    batches = []
    for it in range(8):
        r = get_rays(wl.poses[it % len(wl.poses)][None], BLENDER_INTRINSICS, 800, 800, opt.num_rays, generator=wl.gen)
        bg = torch.rand(1, opt.num_rays, 3, device=dev, generator=wl.gen)
        batches.append((r["rays_o"], r["rays_d"], wl.target(r["rays_o"], r["rays_d"], bg), bg))
    for it in range(warmup):
        tr.train_step(*batches[it % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        loss, pred = tr.train_step(*batches[it % 8])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return ms, float(psnr(pred.detach(), batches[(steps - 1) % 8][2])), float(loss)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    a = ap.parse_args()
    for fused in (True, False):
        ms, p, loss = run(fused, a.steps, a.warmup)
        print(f"teacher train step fused_head={fused}: {ms:.3f} ms/step  {4096 / ms * 1e3:.0f} rays/s  psnr {p:.2f} loss {loss:.5f}", flush=True)
