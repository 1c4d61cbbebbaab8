"""Cycle-counter stamps from an INSTRUMENTED build of the library (not the product):
  hipcc <Makefile FLAGS> -DPVD_MARCH_PROFILE -c aaai2023-pvd_amd/csrc/raymarching.hip -o /tmp/x.o; link it with the other objects into a
  scratch libpvd_hip.so and put that in place of aaai2023-pvd_amd/libpvd_hip.so on the GPU box before running this.
Used for the section timings quoted in DESIGN.md (weight load / forward / dX / dW / epilogue; cycles per lattice chunk)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd")]
import numpy as np, torch, pvd_hip, raymarching
from pvd.scene import BLENDER_INTRINSICS, ChairScene, get_rays, packbits_torch, synthetic_poses
dev = torch.device("cuda:0")
poses = torch.from_numpy(synthetic_poses(np.random.RandomState(0))).to(dev)
bits = packbits_torch(ChairScene(thicken=0.08).density_grid(128, 1.0, 1, device=dev), 10.0)
r = get_rays(poses[0:1], BLENDER_INTRINSICS, 800, 800, 4096)
o, d = r["rays_o"].reshape(-1, 3).contiguous(), r["rays_d"].reshape(-1, 3).contiguous()
nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.2)
N = 4096
for it in range(3):
    xyzs = torch.zeros(92928, 3, device=dev); dirs = torch.zeros(92928, 3, device=dev); deltas = torch.zeros(92928, 2, device=dev)
    rays = torch.zeros(N, 3, dtype=torch.int32, device=dev); counter = torch.zeros(2, dtype=torch.int32, device=dev)
    pvd_hip.march_rays_train(o, d, bits, 1.0, 0.0, 1024, N, 1, 128, 92928, nears, fars, xyzs, dirs, deltas, rays, counter, True)
    torch.cuda.synchronize()
t = rays.cpu().numpy()
cyc, chunks, num = t[:, 0].astype(np.int64), t[:, 1], t[:, 2]
print("rays", N, "cycles: mean %.0f median %.0f max %d | chunks mean %.1f max %d | samples mean %.1f" % (cyc.mean(), np.median(cyc), cyc.max(), chunks.mean(), chunks.max(), num.mean()))
ok = chunks > 0
print("cycles per chunk: mean %.0f" % (cyc[ok] / chunks[ok]).mean(), " rays with chunks:", ok.sum())
lat = ((fars - nears) / (2 * 3 ** 0.5 / 1024)).cpu().numpy()
print("lattice points between near and far: mean %.0f max %.0f" % (lat[np.isfinite(lat)].mean(), lat[np.isfinite(lat)].max()))
