#!/usr/bin/env python3
"""Micro-benchmark of the hash-grid lookup (pvd_grid_encode_forward) on one MI355X: kernel variants x table dtype
x sample coherence x batch size, timed with HIP events on the launch stream.  Prints a table of us/launch and
algorithmic GB/s (516 B/sample f16, 1020 B/sample f32) -- the numbers DESIGN.md quotes."""
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
S = float(np.log2(enc.per_level_scale))


def ray_samples(n_rays):
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(0))).to(dev)
    bits = packbits_torch(ChairScene().density_grid(128, 1.0, 1, device=dev), 10.0)
    xs = []
    for k in range(max(1, n_rays // 4096)):
        r = get_rays(poses[k:k + 1], BLENDER_INTRINSICS, 800, 800, min(n_rays, 4096))
        o, d = r["rays_o"].reshape(-1, 3).contiguous(), r["rays_d"].reshape(-1, 3).contiguous()
        nears, fars = raymarching.near_far_from_aabb(o, d, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.2)
        xyzs, _, _, _ = raymarching.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, None, -1, True, 128, True)
        xs.append(xyzs)
    return ((torch.cat(xs) + 1) / 2).contiguous()


def time_it(x01, emb, variant, iters=100):
    B = x01.shape[0]
    out = torch.empty(14, B, 2, dtype=emb.dtype, device=dev)
    if isinstance(variant, tuple):  # (lanes per sample, persistent workgroups): k_grid_fwd_lps (f16 tables only)
        pvd_hip.grid_set_variant(0)
        pvd_hip.grid_set_fwd_kernel(*variant)
    else:
        pvd_hip.grid_set_fwd_kernel(0, 0)
        pvd_hip.grid_set_variant(variant)
    run = lambda: pvd_hip.grid_encode_forward(x01, emb, enc.offsets, out, B, 3, 2, 14, S, 16, False, out, 0, False)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    # 20 launches per HIP graph: eager launches from Python are host-bound below ~11 us per call
    per_graph = 20
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per_graph):
            run()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(1, iters // per_graph)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    iters = reps * per_graph
    return a.elapsed_time(b) / iters * 1e3, out


coh = {n: ray_samples(n) for n in ((4096,) if os.environ.get("PVD_BENCH_SMALL_TABLE") else (4096, 16384, 65536))}
AFF = 1 << 30  # XCD-affine item order
VARIANTS = [("plain", 0), ("pair", 2), ("lps2", (2, 0)), ("lps4", (4, 0)), ("l2p1k", (2, 1024)), ("l2p2k", (2, 2048)), ("l2p4k", (2, 4096)),
            ("l2a1k", (2, 1024 | AFF)), ("l2a2k", (2, 2048 | AFF)), ("l2a4k", (2, 4096 | AFF)), ("l2a8k", (2, 8192 | AFF))]
if os.environ.get("PVD_BENCH_SMALL_TABLE"):  # every level fits every L2: what the lookup costs without L2 misses
    enc = GridEncoder(num_levels=14, desired_resolution=2048, log2_hashmap_size=int(os.environ["PVD_BENCH_SMALL_TABLE"])).to(dev)
    enc.embeddings.data.uniform_(-1, 1)
    print("small table: 2^%s rows per hashed level, %d rows total" % (os.environ["PVD_BENCH_SMALL_TABLE"], enc.embeddings.shape[0]))
print("%-28s %10s %6s " % ("samples", "B", "dtype") + " ".join("%8s" % (n + " us") for n, _ in VARIANTS) + " %9s" % "best GB/s")
for name, x in [("ray-coherent %d rays" % n, v) for n, v in coh.items()] + [("uniform random", torch.rand(1 << 18, 3, device=dev)),
                                                                           ("uniform random", torch.rand(1 << 20, 3, device=dev))]:
    for dt in (torch.float16, torch.float32):
        emb = enc.embeddings.detach().to(dt)
        ts, outs = zip(*[time_it(x, emb, v) for _, v in VARIANTS])
        assert all(torch.equal(outs[0], o) for o in outs), "variants must agree bit for bit"
        bps = 516 if dt == torch.float16 else 1020
        print("%-28s %10d %6s " % (name, x.shape[0], "f16" if dt == torch.float16 else "f32") + " ".join("%8.1f" % t for t in ts)
              + " %9.0f" % (bps * x.shape[0] / min(ts) / 1e3))
pvd_hip.grid_set_variant(0)
pvd_hip.grid_set_fwd_kernel()
