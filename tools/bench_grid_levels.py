#!/usr/bin/env python3
"""Where the hash-grid lookup's time goes, level by level: the forward over the first L levels only (L = 1..14), on the
bench's ray samples, timed inside HIP graphs.  The difference between consecutive rows is the marginal cost of a level.
Also prints the L1 line-rate model: distinct 128-byte lines per wave instruction, summed over a launch, at one line per
clock per CU."""
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
enc.embeddings.data.uniform_(-1, 1)
emb = enc.embeddings.detach().half()
S = float(np.log2(enc.per_level_scale))


def samples(n_rays=4096, pose=0):
    """The bench's kind of sample set: one training batch of rays of camera `pose` marched through the chair's occupancy grid;
    positions mapped to [0,1]."""
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(0))).to(dev)
    bits = packbits_torch(ChairScene(thicken=0.08).density_grid(128, 1.0, 1, device=dev), 10.0)
    r = get_rays(poses[pose:pose + 1], BLENDER_INTRINSICS, 800, 800, n_rays)
    o, d = r["rays_o"].reshape(-1, 3).contiguous(), r["rays_d"].reshape(-1, 3).contiguous()
    nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.2)
    return ((raymarching.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, None, -1, True, 128, True)[0] + 1) / 2).contiguous()


if __name__ != "__main__":
    x = None
else:
    x = samples()
B = x.shape[0] if x is not None else 0
offs = enc.offsets.cpu().numpy()


def timed(L, variant):
    out = torch.empty(L, B, 2, dtype=torch.float16, device=dev)
    off = enc.offsets[:L + 1].contiguous()
    pvd_hip.grid_set_fwd_kernel(*variant)
    run = lambda: pvd_hip.grid_encode_forward(x, emb, off, out, B, 3, 2, L, S, 16, False, out, 0, False)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            run()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 100 * 1e3


def lines_per_level(level):
    """Distinct 128-byte lines touched by each group of 32 consecutive samples (one wave of the two-lanes-per-sample kernel),
    8 corners each, summed over the batch -- computed on the host from the same index arithmetic (numpy, uint32)."""
    xs = x.cpu().numpy().astype(np.float32)
    scale = np.float32(np.exp2(np.float32(level * S)) * 16 - 1)
    res = int(np.ceil(scale)) + 1
    size = int(offs[level + 1] - offs[level])
    pos = xs * scale + np.float32(0.5)
    pg = np.floor(pos).astype(np.uint32)
    total = 0
    idx = []
    for c in range(8):
        cc = pg + np.array([(c >> 0) & 1, (c >> 1) & 1, (c >> 2) & 1], np.uint32)
        stride, dense, ok = 1, np.zeros(len(xs), np.uint64), True
        for k in range(3):
            dense += cc[:, k].astype(np.uint64) * np.uint64(stride)
            stride *= res + 1
        if stride <= size:
            i = dense % np.uint64(size)
        else:
            pr = (np.uint32(1), np.uint32(2654435761), np.uint32(805459861))
            h = np.zeros(len(xs), np.uint32)
            for k in range(3):
                h ^= cc[:, k] * pr[k]
            i = h.astype(np.uint64) % np.uint64(size)
        idx.append((i * 4) // 128)  # half2 rows: 4 bytes
    idx = np.stack(idx, 1)  # [B, 8] line ids
    n = (len(xs) // 32) * 32
    per_wave = idx[:n].reshape(-1, 32 * 8)
    per_wave.sort(axis=1)
    total = int((np.diff(per_wave, axis=1) != 0).sum() + per_wave.shape[0])
    return total, res, size


if __name__ == "__main__":
    print("samples %d; levels, resolution, rows, dense?, distinct lines per 32-sample wave, cumulative time (two lanes per sample, 4096 persistent workgroups / plain)" % B)
    prev = (0.0, 0.0)
    tot_lines = 0
    for L in range(1, 15):
        t = (timed(L, (2, 4096)), timed(L, (0, 0)))
        lines, res, size = lines_per_level(L - 1)
        tot_lines += lines
        print("L=%2d res %5d rows %7d %s lines/wave %6.1f | lps2 %6.2f us (+%5.2f)  plain %6.2f us (+%5.2f)" % (
            L, res, size, "dense " if (res + 1) ** 3 <= size else "hashed", lines / (B / 32), t[0], t[0] - prev[0], t[1], t[1] - prev[1]), flush=True)
        prev = t
    clk = 2.4e9
    print("distinct lines per launch %.2f M -> %.1f us at one line per clock per CU (256 CUs, %.1f GHz), launch floor ~3 us not included" % (tot_lines / 1e6, tot_lines / 256 / clk * 1e6, clk / 1e9))
    pvd_hip.grid_set_fwd_kernel()
