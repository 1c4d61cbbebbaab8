#!/usr/bin/env python3
"""A whole (short) distillation run on one GPU through the three stages of the reference schedule, scaled down:
teacher training on the analytic scene -> stage 1 (feature loss only) -> stage 2 (+ sigma / colour) -> stage 3 (+ RGB),
with PSNR of the student against the teacher and against the analytic ground truth on held-out views, rendered with the
inference path (march_rays / composite_rays / compact_rays).
  python tools/train_distill.py [--teacher-steps 3000 --stage1 500 --stage2 1500 --steps 6000 --student vm]"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")  # forked graphs: see DESIGN section 6 (before the HIP runtime starts)
import argparse
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd")]

from pvd.config import PVDConfig  # noqa: E402
from pvd.ops import hip_ops  # noqa: E402
from pvd.scene import BLENDER_INTRINSICS, get_rays, synthetic_poses  # noqa: E402
from pvd.trainer import psnr  # noqa: E402
from pvd.workload import DistillWorkload  # noqa: E402


@torch.no_grad()
def evaluate(w, poses, res=200):
    """PSNR on full res x res views (scaled intrinsics): student vs teacher, student vs analytic GT, teacher vs GT."""
    k = res / 800.0
    intr = tuple(v * k for v in BLENDER_INTRINSICS)
    out = []
    for m in (w.stu, w.tea):
        m.eval()
    for pose in poses:
        r = get_rays(pose[None], intr, res, res, -1)
        with torch.autocast("cuda", dtype=torch.float16):
            s = w.stu.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False, max_steps=1024)["image"]
            t = w.tea.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False, max_steps=1024)["image"]
        gt = w.target(r["rays_o"], r["rays_d"], torch.ones(1, res * res, 3, device=s.device))
        out.append((float(psnr(s, t)), float(psnr(s, gt)), float(psnr(t, gt))))
    for m in (w.stu, w.tea):
        m.train()
    return np.mean(np.array(out), axis=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--student", default="vm")
    ap.add_argument("--teacher-steps", type=int, default=3000)
    ap.add_argument("--stage1", type=int, default=500)
    ap.add_argument("--stage2", type=int, default=1500)
    ap.add_argument("--steps", type=int, default=6000)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    opt = PVDConfig(model_type=a.student, iters=a.steps, stage_iters={"stage1": a.stage1, "stage2": a.stage2})
    t0 = time.perf_counter()
    w = DistillWorkload(hip_ops(), dev, opt, teacher_pretrain_steps=a.teacher_steps, start_stage="stage1")
    torch.cuda.synchronize()
    print("teacher: %d steps in %.1f s, train PSNR vs analytic GT %.2f dB" % (a.teacher_steps, time.perf_counter() - t0, w.teacher_psnr), flush=True)
    held_out = torch.from_numpy(synthetic_poses(np.random.RandomState(123))[:4]).to(dev)
    tr = w.trainer
    marks = sorted(set([0, a.stage1, a.stage2] + list(range(0, a.steps + 1, max(a.steps // 6, 1))) + [a.steps]))
    t_train = 0.0

    def spg():
        # stage 3 has no further boundary to respect: ten steps per graph launch, the next step's prefix next to the update
        # (a mark may be overshot by up to 9 steps; the printed step count is the real one)
        return 10 if tr._stage_of(tr.global_step + 3) == 3 else 1  # (+3: the capture's eager warm-up steps)
    for lo, hi in zip(marks[:-1], marks[1:]):
        if tr._stage_of(tr.global_step) != getattr(tr, "_captured_stage", None) or not getattr(w, "_graph", False):
            w.enable_graph(steps_per_graph=spg())  # (re-)capture: the set of loss terms changed
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        while tr.global_step < hi:
            if tr._stage_of(tr.global_step) != tr._captured_stage:
                w.enable_graph(steps_per_graph=spg())
            w.step()
        torch.cuda.synchronize()
        t_train += time.perf_counter() - t1
        st, sg, tg = evaluate(w, held_out)
        print("step %5d (stage %d)  train time %.2f s  PSNR student vs teacher %.2f dB, student vs GT %.2f dB (teacher vs GT %.2f dB)"
              % (tr.global_step, tr._stage_of(max(tr.global_step - 1, 0)), t_train, st, sg, tg), flush=True)


if __name__ == "__main__":
    main()
