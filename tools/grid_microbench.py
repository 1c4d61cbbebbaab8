#!/usr/bin/env python3
"""SURVEY.md section 8(d)'s kernel micro-benchmarks of the hash-grid encoder (a7 / a8): B = 2^18 and 2^20 points, uniformly random in
[0, 1]^3 and scene samples (the marcher's output on the synthetic chair: in ray order, and Morton-sorted = spatially coherent), f16
and f32 tables, forward (pvd_grid_encode_forward) and backward (pvd_grid_encode_backward, the scatter-add into the table gradient).
Per row: us per launch (HIP events around HIP graphs of 10 launches on the launch stream), algorithmic GB/s and its fraction of
8 TB/s, with SURVEY 8(d)'s bytes per sample: forward 516 (f16) / 1020 (f32), backward 964 / 1916.

    python tools/grid_microbench.py > profiles/r06_grid_microbench.txt"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd")]
import numpy as np
import torch

import pvd_hip
import raymarching
from gridencoder import GridEncoder
from pvd.scene import BLENDER_INTRINSICS, ChairScene, get_rays, packbits_torch, synthetic_poses

dev = torch.device("cuda:0")
enc = GridEncoder(num_levels=14, desired_resolution=2048).to(dev)
enc.embeddings.data.uniform_(-1e-1, 1e-1)
S = float(np.log2(enc.per_level_scale))
BYTES = {("fwd", torch.float16): 516, ("fwd", torch.float32): 1020, ("bwd", torch.float16): 964, ("bwd", torch.float32): 1916}


def scene_samples(B):
    """at least B marched samples of the chair scene, in ray order (4096 rays per camera of the epoch), cut to B"""
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(0))).to(dev)
    bits = packbits_torch(ChairScene().density_grid(128, 1.0, 1, device=dev), 10.0)
    aabb = torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev)
    xs, n, k = [], 0, 0
    g = torch.Generator(device=dev).manual_seed(1)
    while n < B:
        r = get_rays(poses[k % len(poses)][None], BLENDER_INTRINSICS, 800, 800, 4096, generator=g)
        o, d = r["rays_o"].reshape(-1, 3).contiguous(), r["rays_d"].reshape(-1, 3).contiguous()
        nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.2)
        xyzs, _, _, rays = raymarching.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, None, -1, True, 128, True)
        used = int((rays[:, 1] + rays[:, 2]).max())
        xs.append(xyzs[:used])
        n += used
        k += 1
    return ((torch.cat(xs)[:B] + 1) / 2).contiguous(), k


def morton_sorted(x):
    c = (x.clamp(0, 1 - 1e-6) * 1024).long()

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return x[torch.argsort(spread(c[:, 0]) | (spread(c[:, 1]) << 1) | (spread(c[:, 2]) << 2))].contiguous()


def timed(run, per_graph=10, reps=5):
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per_graph):
            run()
    g.replay()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / (per_graph * reps) * 1e3
        best = us if best is None else min(best, us)
    return best


print("hash-grid encoder micro-benchmark (SURVEY 8d): L = 14, C = 2, D = 3, table %d rows; alone on one MI355X" % enc.embeddings.shape[0])
print("%-46s %9s %6s %5s %10s %10s %8s" % ("points", "B", "table", "pass", "us/launch", "GB/s alg.", "of 8TB/s"))
for B in (1 << 18, 1 << 20):
    sc, cams = scene_samples(B)
    sets = [("uniform random in [0,1]^3", torch.rand(B, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(0))),
            ("scene samples, ray order (%d cameras x 4096 rays)" % cams, sc), ("scene samples, Morton-sorted (coherent)", morton_sorted(sc))]
    for name, x01 in sets:
        for dt in (torch.float16, torch.float32):
            emb = enc.embeddings.detach().to(dt).contiguous()
            out = torch.empty(14, B, 2, dtype=dt, device=dev)
            dummy = out[:1]
            us = timed(lambda: pvd_hip.grid_encode_forward(x01, emb, enc.offsets, out, B, 3, 2, 14, S, 16, False, dummy, 0, False))
            bps = BYTES[("fwd", dt)]
            print("%-46s %9d %6s %5s %10.1f %10.0f %8.3f" % (name, B, "f16" if dt == torch.float16 else "f32", "fwd", us, bps * B / us / 1e3, bps * B / us / 1e3 / 8000))
            grad = (torch.randn(14, B, 2, device=dev) * 1e-3).to(dt).contiguous()
            gemb = torch.zeros_like(emb)
            us = timed(lambda: pvd_hip.grid_encode_backward(grad, x01, emb, enc.offsets, gemb, B, 3, 2, 14, S, 16, False, dummy, dummy, 0, False))
            bps = BYTES[("bwd", dt)]
            print("%-46s %9d %6s %5s %10.1f %10.0f %8.3f" % (name, B, "f16" if dt == torch.float16 else "f32", "bwd", us, bps * B / us / 1e3, bps * B / us / 1e3 / 8000))
