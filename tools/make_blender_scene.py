#!/usr/bin/env python3
"""Write the synthetic chair as a Blender-format scene (transforms_{train,val,test}.json + RGBA PNGs, what
distill_mutual/provider.py:133-326 reads) so that pvd/provider.py can feed a measured run: there is no dataset offline.
  python tools/make_blender_scene.py OUT_DIR [--views 40] [--res 200]
Views are rendered with the analytic scene through the HIP marcher + compositor (pvd.workload.AnalyticTarget): colour over black and
over white give the premultiplied colour and the alpha of every pixel; PNGs hold straight RGBA, 8 bit."""
import argparse
import json
import math
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd")]
import numpy as np
import torch
from PIL import Image

from pvd.config import PVDConfig
from pvd.ops import hip_ops
from pvd.scene import get_rays, nerf_matrix_to_ngp, pose_spherical
from pvd.workload import DistillWorkload


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--views", type=int, default=40)
    ap.add_argument("--res", type=int, default=200)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    w = DistillWorkload(hip_ops(), dev, PVDConfig(), teacher_pretrain_steps=0)
    angle_x = 0.6911112070083618  # the Blender lego / chair camera
    focal = a.res / (2 * math.tan(angle_x / 2))
    intr = (focal, focal, a.res / 2, a.res / 2)
    rng = np.random.RandomState(11)
    os.makedirs(a.out, exist_ok=True)
    for split, n in (("train", a.views), ("val", max(a.views // 8, 2)), ("test", max(a.views // 8, 2))):
        os.makedirs(os.path.join(a.out, split), exist_ok=True)
        frames = []
        for i in range(n):
            c2w = pose_spherical(-180 + 360 * rng.rand(), -5 - 55 * rng.rand(), 4.0)  # NeRF-Blender convention, r = 4
            pose = torch.from_numpy(nerf_matrix_to_ngp(c2w, 0.8)).to(dev)
            r = get_rays(pose[None], intr, a.res, a.res, -1)
            zeros = torch.zeros(1, a.res * a.res, 3, device=dev)
            c0 = w.target(r["rays_o"], r["rays_d"], zeros)[0]         # sum_i w_i c_i
            c1 = w.target(r["rays_o"], r["rays_d"], zeros + 1.0)[0]   # ... + (1 - sum_i w_i)
            alpha = (1.0 - (c1 - c0).mean(-1, keepdim=True)).clamp(0, 1)
            rgb = torch.where(alpha > 1e-4, c0 / alpha.clamp_min(1e-4), torch.zeros_like(c0)).clamp(0, 1)
            img = torch.cat([rgb, alpha], -1).view(a.res, a.res, 4)
            Image.fromarray((img.cpu().numpy() * 255 + 0.5).astype(np.uint8), "RGBA").save(os.path.join(a.out, split, "r_%d.png" % i))
            frames.append({"file_path": "./%s/r_%d" % (split, i), "transform_matrix": c2w.tolist()})
        json.dump({"camera_angle_x": angle_x, "frames": frames}, open(os.path.join(a.out, "transforms_%s.json" % split), "w"))
    print("wrote %s: %d train views of %dx%d" % (a.out, a.views, a.res, a.res))


if __name__ == "__main__":
    main()
