#!/usr/bin/env python3
"""march_rays_train on the bench scene (4096 rays, ~5 % occupancy): HIP-event time of the entry point."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd")]
import numpy as np
import torch

import pvd_hip
import raymarching
from pvd.scene import BLENDER_INTRINSICS, ChairScene, get_rays, packbits_torch, synthetic_poses

dev = torch.device("cuda:0")
poses = torch.from_numpy(synthetic_poses(np.random.RandomState(0))).to(dev)
bits = packbits_torch(ChairScene(thicken=0.08).density_grid(128, 1.0, 1, device=dev), 10.0)
r = get_rays(poses[0:1], BLENDER_INTRINSICS, 800, 800, 4096)
o, d = r["rays_o"].reshape(-1, 3).contiguous(), r["rays_d"].reshape(-1, 3).contiguous()
nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.2)
run = lambda: raymarching.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, None, 92928, True, 128, False)
for _ in range(5):
    out = run()
with pvd_hip.KernelTimer({"pvd_march_rays_train_ws"}) as kt:
    for _ in range(50):
        out = run()
print("march_rays_train: %.1f us per call (count + write-from-records), samples %d" % (kt.mean_ms("pvd_march_rays_train_ws") * 1e3, int(out[3][:, 2].sum())))
