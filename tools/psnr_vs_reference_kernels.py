#!/usr/bin/env python3
"""north_star's end-to-end bar -- "PSNR within 0.1 dB of the reference" -- measured against the closest thing to the reference this
image allows on an MI355X: the reference's OWN kernels (oracle/_ref: raymarching.cu, shencoder.cu built for gfx950) under this repo's
generic, reference-shaped host code (autograd wrappers, 12 x F.grid_sample, nn.Linear under autocast, torch.optim.AdamW + GradScaler;
the formulation tests/test_golden_step.py pins against the reference's own train_step).  The hash lookup of that stack is this repo's
generic encoder kernel (gridencoder.cu does not build on HIP).

One teacher (trained once, on the analytic chair), one student initialisation, one schedule (the reference's three stages, scaled:
main_distill_mutual.py:387-396), then three distillation runs:

    A  reference kernels + PyTorch, eager, batches from the host-side provider
    B  libpvd_hip.so, eager, THE SAME batches (same poses, same generator seed)
    C  libpvd_hip.so as shipped: batches made on the device, steps replayed from hipGraphs

and, for each, PSNR on 4 held-out 200x200 views through the stack's own inference path (march_rays / composite_rays / compact_rays):
student vs teacher, student vs the analytic ground truth.

    python tools/psnr_vs_reference_kernels.py [--teacher 3000 --stage1 500 --stage2 1500 --steps 6000] > profiles/r06_psnr_vs_reference_kernels.txt"""
import argparse
import copy
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch

from bench_reference_kernels_step import reference_kernel_ops


def held_out_psnr(w, dev):
    from pvd.scene import BLENDER_INTRINSICS, get_rays, synthetic_poses
    from pvd.trainer import psnr
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(123))[:4]).to(dev)
    res = 200
    intr = tuple(v * res / 800.0 for v in BLENDER_INTRINSICS)
    rows = []
    for m in (w.stu, w.tea):
        m.eval()
    imgs = []
    with torch.no_grad():
        for pose in poses:
            r = get_rays(pose[None], intr, res, res, -1)
            with torch.autocast("cuda", dtype=torch.float16):
                s_img = w.stu.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False, max_steps=1024)["image"]
                t_img = w.tea.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False, max_steps=1024)["image"]
            gt = w.target(r["rays_o"], r["rays_d"], torch.ones(1, res * res, 3, device=dev))
            rows.append((float(psnr(s_img, t_img)), float(psnr(s_img, gt)), float(psnr(t_img, gt))))
            imgs.append(s_img.float())
    for m in (w.stu, w.tea):
        m.train()
    return np.array(rows), torch.stack(imgs)


def compare(a, which=("A", "B", "C")):
    """the runs named in `which`: [(name, per-view PSNR rows [4, 3], the student's renders, wall seconds, steps)], and the teacher's PSNR"""
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import DistillWorkload
    dev = torch.device("cuda:0")

    def config():
        return PVDConfig(model_type=a.student, teacher_type=getattr(a, "teacher_type", "hash"), iters=a.steps,
                         stage_iters={"stage1": a.stage1, "stage2": a.stage2}, fp16=True)

    # the teacher and the student's initial state: made once, by the product
    torch.manual_seed(0)
    torch.cuda.manual_seed(1234)
    w0 = DistillWorkload(hip_ops(), dev, config(), teacher_pretrain_steps=a.teacher, start_stage="stage1", seed=0)
    tea_state = copy.deepcopy(w0.tea.state_dict())
    stu_state = copy.deepcopy(w0.stu.state_dict())
    mean_count = w0.tea.mean_count
    teacher_psnr = w0.teacher_psnr
    del w0
    torch.cuda.empty_cache()

    def run(name, ops, graphs):
        torch.manual_seed(0)
        torch.cuda.manual_seed(1234)
        w = DistillWorkload(ops, dev, config(), teacher_pretrain_steps=0, start_stage="stage1", seed=0)
        w.tea.load_state_dict(tea_state)
        w.stu.load_state_dict(stu_state)
        for m in (w.tea, w.stu):
            m.mean_count = mean_count
            if hasattr(m, "note_occupancy_changed"):
                m.note_occupancy_changed()
        if ops.name == "hip":
            import pvd_hip
            for m in (w.tea, w.stu):
                pvd_hip.note_weights_changed(list(m.parameters()))
        w.gen.manual_seed(4242)  # the batches of runs A and B: same poses (step index), same pixel / background draws
        w.step_idx = 0
        tr = w.trainer
        t0 = time.perf_counter()
        if graphs:
            while tr.global_step < a.steps:
                if tr._stage_of(tr.global_step) != getattr(tr, "_captured_stage", None) or not getattr(w, "_graph", False):
                    w.enable_graph(steps_per_graph=10 if tr._stage_of(tr.global_step + 3) == 3 else 1)
                w.step()
        else:
            while tr.global_step < a.steps:
                w.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rows, imgs = held_out_psnr(w, dev)
        steps = int(tr.global_step)
        del w
        torch.cuda.empty_cache()
        return name, rows, imgs, dt, steps

    table = {"A": ("A  reference kernels + PyTorch (eager)", reference_kernel_ops, False),
             "B": ("B  libpvd_hip.so, eager, same batches as A", hip_ops, False),
             "C": ("C  libpvd_hip.so as shipped (device batches, hipGraphs)", hip_ops, True)}
    return [run(table[k][0], table[k][1](), table[k][2]) for k in which], teacher_psnr


def compare_teacher_training(a):
    """configs[1]: the hash (or mlp / vm) model trained on the analytic scene's pixels for a.teacher steps through both stacks -- same
    initial weights (seeded constructor), same batches -- then 4 held-out views against the ground truth."""
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.scene import BLENDER_INTRINSICS, get_rays, synthetic_poses
    from pvd.trainer import psnr
    from pvd.workload import DistillWorkload
    dev = torch.device("cuda:0")
    rows = []
    for name, ops in (("A  reference kernels + PyTorch (eager)", reference_kernel_ops()), ("B  libpvd_hip.so (eager)", hip_ops())):
        seed = int(getattr(a, "seed", 0))
        torch.manual_seed(seed)
        torch.cuda.manual_seed(1234 + seed)
        opt = PVDConfig(model_type="vm", teacher_type=a.teacher_type, fp16=True)
        t0 = time.perf_counter()
        w = DistillWorkload(ops, dev, opt, teacher_pretrain_steps=a.teacher, seed=seed)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        poses = torch.from_numpy(synthetic_poses(np.random.RandomState(123))[:4]).to(dev)
        res = 200
        intr = tuple(v * res / 800.0 for v in BLENDER_INTRINSICS)
        w.tea.eval()
        vals, imgs = [], []
        with torch.no_grad():
            for pose in poses:
                r = get_rays(pose[None], intr, res, res, -1)
                with torch.autocast("cuda", dtype=torch.float16):
                    t_img = w.tea.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False, max_steps=1024)["image"]
                gt = w.target(r["rays_o"], r["rays_d"], torch.ones(1, res * res, 3, device=dev))
                vals.append(float(psnr(t_img, gt)))
                imgs.append(t_img.float())
        rows.append((name, np.array(vals), torch.stack(imgs), dt, w.teacher_psnr))
        del w
        torch.cuda.empty_cache()
    print("%s model trained on the analytic chair's pixels (configs[1]), %d steps x 4096 rays, fp16 AMP, one MI355X; 4 held-out 200x200 views, inference path"
          % (a.teacher_type, a.teacher))
    print("%-44s %9s %22s %18s" % ("run", "wall s", "last training batch dB", "held-out vs GT dB"))
    for name, v, _, dt, last in rows:
        print("%-44s %9.1f %22.3f %18.3f   per view %s" % (name, dt, last, v.mean(), np.array2string(v, precision=2)))
    print("difference of the means: B - A  %+.3f dB;  the two models' renders against each other: %.2f dB"
          % (rows[1][1].mean() - rows[0][1].mean(), float(psnr(rows[1][2], rows[0][2]))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0, help="(--teacher-training) initial weights and batches")
    ap.add_argument("--teacher-training", action="store_true", help="configs[1] instead: train the teacher-type model through both stacks")
    ap.add_argument("--teacher", type=int, default=3000)
    ap.add_argument("--stage1", type=int, default=500)
    ap.add_argument("--stage2", type=int, default=1500)
    ap.add_argument("--steps", type=int, default=6000)
    ap.add_argument("--student", default="vm")
    ap.add_argument("--teacher-type", default="hash", help="mlp: no grid encoder anywhere -- every native call of run A is the reference's own code")
    a = ap.parse_args()
    if a.teacher_training:
        return compare_teacher_training(a)
    from pvd.trainer import psnr
    runs, teacher_psnr = compare(a)
    print("%s -> %s distillation on the synthetic chair, fp16 AMP, 4096 rays/step, one MI355X; teacher: %d steps on the analytic scene "
          "(last-batch PSNR %.2f dB); schedule: stage 1 to %d, stage 2 to %d, stage 3 to %d; 4 held-out 200x200 views, inference path"
          % (a.teacher_type, a.student, a.teacher, teacher_psnr or float("nan"), a.stage1, a.stage2, a.steps))
    print("%-58s %7s %9s %20s %16s %16s" % ("run", "steps", "wall s", "student vs teacher dB", "student vs GT dB", "teacher vs GT dB"))
    for name, rows, _, dt, steps in runs:
        m = rows.mean(0)
        print("%-58s %7d %9.1f %20.3f %16.3f %16.3f" % (name, steps, dt, m[0], m[1], m[2]))
    ra, rb, rc = (r[1].mean(0) for r in runs)
    print("differences of the means: B - A  %+.3f dB (vs teacher) %+.3f dB (vs GT);  C - A  %+.3f / %+.3f dB" % (rb[0] - ra[0], rb[1] - ra[1], rc[0] - ra[0], rc[1] - ra[1]))
    print("per view, student vs teacher:  A %s   B %s   C %s" % tuple(np.array2string(r[1][:, 0], precision=2) for r in runs))
    print("the two students' renders of the same views against each other:  B vs A %.2f dB   C vs A %.2f dB"
          % (float(psnr(runs[1][2], runs[0][2])), float(psnr(runs[2][2], runs[0][2]))))
    print("north_star: within 0.1 dB of the reference -> %s" % ("met" if abs(rb[0] - ra[0]) <= 0.1 and abs(rb[1] - ra[1]) <= 0.1 else "NOT met on these runs"))


if __name__ == "__main__":
    main()
