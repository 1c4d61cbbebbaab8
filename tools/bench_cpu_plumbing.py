#!/usr/bin/env python3
"""BASELINE.json configs[0]: `main_just_train_tea.py chair --model_type mlp` without cuda_ray -- 64x64 crop (4096 rays),
512 fixed steps per ray, 8 x 256 NeRF MLP -- as a CPU training step through NeRFRenderer.run (the fixed-step torch sampler)
with the oracle SH encoder.  Prints rays/s of full optimisation steps (the number in BASELINE.md's first row).  CPU only:
this is test/baseline tooling, not the product path."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tests")]
import numpy as np
import torch

from test_cpu_plumbing import _crop_batch, _teacher

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
m, tr, opt = _teacher(num_rays=4096)  # defaults: num_steps 512, upsample_steps 0 (main_just_train_tea.py:45-56)
rays_o, rays_d, scene = _crop_batch(64)
bg = torch.rand(1, 4096, 3)
gt = torch.rand(1, 4096, 3)
tr.train_step(rays_o, rays_d, gt, bg)  # warm-up
t0 = time.perf_counter()
for _ in range(steps):
    loss, _ = tr.train_step(rays_o, rays_d, gt, bg)
dt = time.perf_counter() - t0
print("configs[0] mlp teacher, fixed-step sampler, CPU (%d threads): %.2f s/step = %.0f rays/s (%d samples/step), loss %.4f"
      % (torch.get_num_threads(), dt / steps, steps * 4096 / dt, 4096 * 512, float(loss)))
