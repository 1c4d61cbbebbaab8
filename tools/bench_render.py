#!/usr/bin/env python3
"""Time of one 800 x 800 inference render (640 000 rays) of the hash teacher and the VM student: the reference-shaped loop
(one device-to-host read-back per round) vs the rounds whose state stays on the device (pvd_infer_*)."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tests")]
import numpy as np
import torch

from pvd.scene import BLENDER_INTRINSICS, get_rays, synthetic_poses
from test_hip_infer_rounds import _model

dev = torch.device("cuda:0")
poses = torch.from_numpy(synthetic_poses(np.random.RandomState(2))).to(dev)
r = get_rays(poses[9][None], BLENDER_INTRINSICS, 800, 800, -1)
for kind in ("hash", "vm"):
    m = _model(kind)
    for mode in ("0", "1"):
        os.environ["PVD_INFER_DEVICE_ROUNDS"] = mode
        times = []
        for it in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                out = m.render(r["rays_o"], r["rays_d"], staged=False, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        print("%-5s 800x800 render, %s: %.2f ms (best of 3 after warm-up), %s rounds" % (
            kind, "round state on the device" if mode == "1" else "host read-back per round ", min(times[1:]) * 1e3,
            getattr(m, "_last_rounds", "?") if mode == "1" else "n/a"))
