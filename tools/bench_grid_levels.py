#!/usr/bin/env python3
"""Per-level cost of the hash-grid forward at the bench's launch size (measurement knob: level mask)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tools")]
import numpy as np
import torch

import pvd_hip
from gridencoder import GridEncoder

dev = torch.device("cuda:0")
enc = GridEncoder(num_levels=14, desired_resolution=2048).to(dev)
enc.embeddings.data.uniform_(-1, 1)
S = float(np.log2(enc.per_level_scale))
emb = enc.embeddings.detach().to(torch.float16)


def samples(n_rays=4096):
    import raymarching
    from pvd.scene import BLENDER_INTRINSICS, ChairScene, get_rays, packbits_torch, synthetic_poses
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(0))).to(dev)
    bits = packbits_torch(ChairScene(thicken=0.08).density_grid(128, 1.0, 1, device=dev), 10.0)
    r = get_rays(poses[0:1], BLENDER_INTRINSICS, 800, 800, n_rays)
    o, d = r["rays_o"].reshape(-1, 3).contiguous(), r["rays_d"].reshape(-1, 3).contiguous()
    nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.2)
    xyzs, _, _, _ = raymarching.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, None, -1, True, 128, True)
    return ((xyzs + 1) / 2).contiguous()


def time_mask(x01, mask, iters=100):
    B = x01.shape[0]
    out = torch.empty(14, B, 2, dtype=emb.dtype, device=dev)
    pvd_hip.grid_set_variant(mask << 8)
    run = lambda: pvd_hip.grid_encode_forward(x01, emb, enc.offsets, out, B, 3, 2, 14, S, 16, False, out, 0, False)
    for _ in range(5):
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        run()
    b.record()
    torch.cuda.synchronize()
    pvd_hip.grid_set_variant(0)
    return a.elapsed_time(b) / iters * 1e3


if __name__ == "__main__":
    x = samples()
    print("samples", x.shape[0])
    print("all levels      %7.1f us" % time_mask(x, 0))
    print("none (mask bit 20 only: every block exits) %7.1f us" % time_mask(x, 1 << 20))
    print("dense 0-4       %7.1f us" % time_mask(x, 0b11111))
    print("hashed 5-13     %7.1f us" % time_mask(x, 0b11111111100000))
    for l in range(14):
        print("level %2d        %7.1f us" % (l, time_mask(x, 1 << l)))
    xr = torch.rand(x.shape[0], 3, device=dev)
    print("uniform random points, all levels %7.1f us" % time_mask(xr, 0))
