#!/usr/bin/env python3
"""Time of one 800 x 800 inference render (640 000 rays) of the hash teacher, the VM student and the Plenoxel student: the reference-shaped loop
(one device-to-host read-back per round) vs the rounds whose state stays on the device (pvd_infer_*) vs -- hash model -- the
whole loop as one persistent launch (pvd_infer_image_hash / pvd_infer_image_vm / pvd_infer_image_plenoxel)."""
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
ONLY = os.environ.get("PVD_RENDER_ONLY")  # "p": the persistent renders alone (profiling); PVD_RENDER_KIND=hash|vm: one model
for kind in ("hash", "vm", "tensors"):
    if os.environ.get("PVD_RENDER_KIND", kind) != kind:
        continue
    m = _model(kind)
    for mode in ("0", "1", "p"):
        if (ONLY and mode != ONLY) or (kind == "tensors" and mode == "1"):  # (no device-side round state for the Plenoxel model)
            continue
        os.environ["PVD_INFER_DEVICE_ROUNDS"] = "0" if mode == "0" else "1"
        os.environ["PVD_INFER_PERSISTENT"] = "1" if mode == "p" else "0"
        times = []
        for it in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                out = m.render(r["rays_o"], r["rays_d"], staged=False, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        if mode == "p":
            wsp = m._last_infer_workspace
            st = wsp[-10:-6].tolist()
            ph = wsp[-6:-1].tolist()
            if any(ph):
                print("      workgroup 0, us: refill+scan %.0f  march %.0f  lookup %.0f  head %.0f  blend %.0f" % tuple(v / 100.0 for v in ph))
            print("      persistent launch: %d rays queued of %d; %d workgroups, %d local rounds (%d walk-only), %d rows shaded = %.1f rows per shading round"
                  % (int(wsp[0]), r["rays_o"].shape[1], st[3], st[0], st[2], st[1], st[1] / max(st[0] - st[2], 1)))
        print("%-5s 800x800 render, %s: %.2f ms (best of 3 after warm-up), %s rounds" % (
            kind, {"1": "round state on the device", "0": "host read-back per round ", "p": "ONE persistent launch     "}[mode], min(times[1:]) * 1e3,
            getattr(m, "_last_rounds", "?") if mode == "1" else "n/a"))
