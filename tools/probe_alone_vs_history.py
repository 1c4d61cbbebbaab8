#!/usr/bin/env python3
"""Why does bench.py's `alone` figure of k_hash_fwd_fused depend on how many steps ran before it (22-24 us after 25 steps,
30-32 us after 220+)?  Measures the kernel alone at several points of one process, on the samples of a fresh batch AND on a
fixed sample set saved at the first point: history of the process vs the batch's camera."""
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd")]
import torch

import fusedhead
from pvd.config import PVDConfig
from pvd.ops import hip_ops
from pvd.workload import DistillWorkload

dev = torch.device("cuda:0")
opt = PVDConfig(num_rays=4096)
w = DistillWorkload(hip_ops(), dev, opt, teacher_pretrain_steps=300, seed=0)
w.enable_graph(steps_per_graph=5)


def alone(xyzs, dirs):
    for _ in range(3):
        fusedhead.hash_head_infer(w.tea, xyzs, dirs)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fusedhead.hash_head_infer(w.tea, xyzs, dirs)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 100 * 1e3


def fresh():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        rays_o, rays_d, bg, *_ = w.device_batch()
        out = w.stu.render(rays_o, rays_d, staged=False, bg_color=bg, perturb=True, force_all_rays=False, dt_gamma=0.0, max_steps=1024)
    x, d = out["inherited_params"][0], out["inherited_params"][1]
    n = int(out["inherited_params"][3][:, 2].sum())
    return x, d, n


fixed = None
done = 0
for upto in (25, 225, 1025, 3025):
    for _ in range((upto - done) // 5):
        w.step()
    done = upto
    torch.cuda.synchronize()
    x, d, n = fresh()
    if fixed is None:
        fixed = (x.clone(), d.clone())
    pose_idx = int(w._batch_state[0].item()) if hasattr(w, "_batch_state") else -1
    t_fresh = alone(x, d)
    t_fixed = alone(*fixed)
    r = x[:n].norm(dim=-1)
    print("after %5d steps: fresh batch (state %d, %d samples in %d rows, mean |x| %.3f, mean elevation z %.3f): %.2f us   fixed first-point samples: %.2f us"
          % (upto, pose_idx, n, x.shape[0], float(r.mean()), float(x[:n, 2].mean()), t_fresh, t_fixed), flush=True)
