#!/usr/bin/env python3
"""The reference's OWN kernels (oracle/_ref: raymarching.cu / shencoder.cu built for gfx950 by oracle/build_ref.py) timed next to this
repo's kernels on the same MI355X, same inputs: the metric's batch (4096 rays of a training camera through the chair's occupancy grid,
~9e4 samples).  HIP events around 50 launches each, after 5 warm-up launches; outputs compared while at it.

    python tools/bench_vs_reference_kernels.py > profiles/r06_vs_reference_kernels.txt      (needs oracle/_ref and a GPU)"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "aaai2023-pvd_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch

import pvd_hip
import raymarching as RM
from oracle.build_ref import load_module
from pvd.scene import BLENDER_INTRINSICS, ChairScene, get_rays, packbits_torch, synthetic_poses

DEV = torch.device("cuda:0")
rm, sh = load_module("_raymarching_ref"), load_module("_shencoder_ref")


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


poses = torch.from_numpy(synthetic_poses(np.random.RandomState(0))).to(DEV)
bits = packbits_torch(ChairScene().density_grid(128, 1.0, 1, device=DEV), 10.0)
N = 4096
r = get_rays(poses[3:4], BLENDER_INTRINSICS, 800, 800, N, generator=torch.Generator(device=DEV).manual_seed(7))
o, d = r["rays_o"].reshape(-1, 3).contiguous(), r["rays_d"].reshape(-1, 3).contiguous()
aabb = torch.tensor([-1, -1, -1, 1, 1, 1.0], device=DEV)
nears, fars = RM.near_far_from_aabb(o, d, aabb, 0.2)
print("%-58s %12s %12s %8s" % ("kernel, 4096 rays of one camera", "reference us", "this repo us", "ratio"))


def row(name, t_ref, t_hip):
    print("%-58s %12.1f %12.1f %8.2f" % (name, t_ref, t_hip, t_ref / t_hip))


# near / far
n_r, f_r = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
row("near_far_from_aabb", timed(lambda: rm.near_far_from_aabb(o, d, aabb, N, 0.2, n_r, f_r)), timed(lambda: RM.near_far_from_aabb(o, d, aabb, 0.2)))

# march_rays_train: the reference's wrapper zero-fills xyzs / dirs / deltas and the counter before every call (raymarching.py:240-250):
# timed with that fill, as its training step pays it; this repo's call is its operator-level entry (allocation included)
M = 4096 * 32
xr, dr, lr = torch.zeros(M, 3, device=DEV), torch.zeros(M, 3, device=DEV), torch.zeros(M, 2, device=DEV)
rr = torch.empty(N, 3, dtype=torch.int32, device=DEV)
cr = torch.zeros(2, dtype=torch.int32, device=DEV)


def ref_march():
    xr.zero_(); dr.zero_(); lr.zero_(); cr.zero_()
    rm.march_rays_train(o, d, bits, 1.0, 0.0, 1024, N, 1, 128, M, nears, fars, xr, dr, lr, rr, cr, 1)


def ref_march_kernel_only():
    cr.zero_()
    rm.march_rays_train(o, d, bits, 1.0, 0.0, 1024, N, 1, 128, M, nears, fars, xr, dr, lr, rr, cr, 1)


t_full, t_k = timed(ref_march), timed(ref_march_kernel_only)
t_hip = timed(lambda: RM.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, None, M, True, -1, False, 0, 1024))
samples = int(cr[0])
row("march_rays_train (%d samples), with the wrapper's zero-fills" % samples, t_full, t_hip)
row("march_rays_train, kernel + counter reset only", t_k, t_hip)

# compositing on those samples (this repo's rays table: offsets in ray order)
xh, dh, lh, rh = RM.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, None, M, True, -1, False, 0, 1024)
g = torch.Generator(device=DEV).manual_seed(1)
sig = torch.exp(torch.rand(M, device=DEV, generator=g) * 9 - 2)
rgb = torch.rand(M, 3, device=DEV, generator=g)
ws_r, dep_r, img_r = torch.empty(N, device=DEV), torch.empty(N, device=DEV), torch.empty(N, 3, device=DEV)
ws_h, dep_h, img_h = torch.empty(N, device=DEV), torch.empty(N, device=DEV), torch.empty(N, 3, device=DEV)
row("composite_rays_train_forward", timed(lambda: rm.composite_rays_train_forward(sig, rgb, lh, rh, M, N, ws_r, dep_r, img_r)),
    timed(lambda: pvd_hip.composite_rays_train_forward(sig, rgb, lh, rh, M, N, ws_h, dep_h, img_h)))
gws, gimg = torch.randn(N, device=DEV, generator=g), torch.randn(N, 3, device=DEV, generator=g)
gs_r, gr_r, gs_h, gr_h = torch.zeros(M, device=DEV), torch.zeros(M, 3, device=DEV), torch.zeros(M, device=DEV), torch.zeros(M, 3, device=DEV)
row("composite_rays_train_backward",
    timed(lambda: rm.composite_rays_train_backward(gws, gimg, sig, rgb, lh, rh, ws_r, img_r, M, N, gs_r, gr_r)),
    timed(lambda: pvd_hip.composite_rays_train_backward(gws, gimg, sig, rgb, lh, rh, ws_h, img_h, M, N, gs_h, gr_h)))
print("   (outputs: image max |diff| %.2g, grad_sigmas max |diff| %.2g of max %.2g)" % (
    float((img_r - img_h).abs().max()), float((gs_r - gs_h).abs().max()), float(gs_r.abs().max())))

# SH degree 4 on the samples' directions (the reference encodes every sample's direction; the fused heads here encode inside the head kernel)
B = samples
dirs = dh[:B].contiguous()
o_r, o_h = torch.empty(B, 16, device=DEV), torch.empty(B, 16, device=DEV)
dummy = torch.empty(1, device=DEV)
row("sh_encode_forward, degree 4, %d directions" % B, timed(lambda: sh.sh_encode_forward(dirs, o_r, B, 3, 4, False, dummy)),
    timed(lambda: pvd_hip.sh_encode_forward(dirs, o_h, B, 3, 4, False, dummy)))
print("   (outputs: max |diff| %.2g)" % float((o_r - o_h).abs().max()))

# the inference march: 640k alive rays, one step each (the eval branch's first round)
NA = 640000
ri = get_rays(poses[5:6], BLENDER_INTRINSICS, 800, 800, -1)
oi, di = ri["rays_o"].reshape(-1, 3).contiguous(), ri["rays_d"].reshape(-1, 3).contiguous()
ni, fi = RM.near_far_from_aabb(oi, di, aabb, 0.2)
alive = torch.arange(NA, dtype=torch.int32, device=DEV)
rt = ni.clone()
xi, dri, li = torch.zeros(NA, 3, device=DEV), torch.zeros(NA, 3, device=DEV), torch.zeros(NA, 2, device=DEV)
row("march_rays, 640 000 rays x 1 step",
    timed(lambda: rm.march_rays(NA, 1, alive, rt, oi, di, 1.0, 0.0, 1024, 1, 128, bits, ni, fi, xi, dri, li, 0), n=20),
    timed(lambda: pvd_hip.march_rays(NA, 1, alive, rt, oi, di, 1.0, 0.0, 1024, 1, 128, bits, ni, fi, xi, dri, li, 0), n=20))
