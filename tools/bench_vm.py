#!/usr/bin/env python3
"""VM plane x line lookup (pvd_vm_forward / pvd_vm_backward) on the bench's ray samples: HIP-event time per launch,
for different sample counts and a shuffled (incoherent) order."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd")]
import numpy as np
import torch

import pvd_hip
import raymarching
import vmencoder
from pvd.scene import BLENDER_INTRINSICS, ChairScene, get_rays, packbits_torch, synthetic_poses

dev = torch.device("cuda:0")
torch.manual_seed(0)
poses = torch.from_numpy(synthetic_poses(np.random.RandomState(0))).to(dev)
bits = packbits_torch(ChairScene(thicken=0.08).density_grid(128, 1.0, 1, device=dev), 10.0)
r = get_rays(poses[0:1], BLENDER_INTRINSICS, 800, 800, 4096)
o, d = r["rays_o"].reshape(-1, 3).contiguous(), r["rays_d"].reshape(-1, 3).contiguous()
nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.2)
xyzs = raymarching.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, None, -1, True, 128, True)[0]
res = 300
tabs = []
for R in (16, 48):
    tabs += [vmencoder.to_channels_last_param(torch.randn(1, R, res, res, device=dev) * 0.1) for _ in range(3)]
    tabs += [vmencoder.to_channels_last_param(torch.randn(1, R, res, 1, device=dev) * 0.1) for _ in range(3)]
tabs = tabs[0:3] + tabs[3:6] + tabs[6:9] + tabs[9:12]
grads = [torch.zeros_like(t) for t in tabs]
aabb = (-1.0, -1.0, -1.0, 1.0, 1.0, 1.0)
# the same factors with sigma and colour interleaved per texel ([H][W][64] / [L][64])
from vmencoder.vm import interleave_factors
itabs, igrads = list(tabs), [None] * 12
for i in range(6):
    itabs[i], itabs[6 + i] = interleave_factors(tabs[i], tabs[6 + i])
    igrads[i], igrads[6 + i] = interleave_factors(torch.zeros_like(tabs[i]), torch.zeros_like(tabs[6 + i]))
M = xyzs.shape[0]
outs = []
for T, G in ((tabs, grads), (itabs, igrads)):
    sig = torch.empty(M, device=dev); prod = torch.empty(M, 144, dtype=torch.float16, device=dev)
    pvd_hip.vm_forward(xyzs, aabb, T, [res] * 3, sig, prod)
    torch.manual_seed(1)
    gs, gp = torch.randn(M, device=dev), torch.randn(M, 144, device=dev).half()
    for g in G:
        g.zero_()
    pvd_hip.vm_backward(xyzs, aabb, T, [res] * 3, gs, gp, G)
    outs.append((sig, prod, [g.clone() for g in G]))
if os.environ.get("PVD_HIP_LIB") is None: assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "interleaved forward differs"
worst = max(float((a - b).abs().max() / (a.abs().max() + 1e-20)) for a, b in zip(outs[0][2], outs[1][2]))
print("interleaved layout: forward bit-identical, gradient difference (atomic order) %.2e of max" % worst)
for name, x in (("ray order", xyzs), ("shuffled", xyzs[torch.randperm(xyzs.shape[0], device=dev)].contiguous()), ("2x samples", torch.cat([xyzs, xyzs]))):
    M = x.shape[0]
    sig = torch.empty(M, device=dev)
    prod = torch.empty(M, 144, dtype=torch.float16, device=dev)
    gs, gp = torch.randn(M, device=dev), torch.randn(M, 144, device=dev).half()
    for lay, T, G in (("separate", tabs, grads), ("interleaved", itabs, igrads)):
        f = lambda: pvd_hip.vm_forward(x, aabb, T, [res] * 3, sig, prod)
        b = lambda: pvd_hip.vm_backward(x, aabb, T, [res] * 3, gs, gp, G)
        for _ in range(3):
            f(); b()
        with pvd_hip.KernelTimer({"pvd_vm_forward", "pvd_vm_backward"}) as kt:
            for _ in range(30):
                f(); b()
        print(f"{name:12s} {lay:12s} M={M:7d}: forward {kt.mean_ms('pvd_vm_forward') * 1e3:7.1f} us   backward {kt.mean_ms('pvd_vm_backward') * 1e3:7.1f} us", flush=True)
