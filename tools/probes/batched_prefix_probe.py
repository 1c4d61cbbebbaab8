#!/usr/bin/env python3
"""What would ONE launch of the frozen teacher's forward over the samples of K consecutive steps cost?  (The prefix of a step -- batch,
march, teacher forward, its compositing -- does not depend on the student: K of them can be made at once.)  The product kernel
(pvd_hash_head_forward_fused) on the concatenated samples of K training batches (K different cameras), 20 launches per graph."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tools")]
import torch

import fusedhead
from bench_grid_levels import samples
from pvd.config import PVDConfig
from pvd.ops import hip_ops
from pvd.workload import make_model

dev = torch.device("cuda:0")
m = make_model(hip_ops(), PVDConfig(model_type="hash"), "hash", True, dev).eval()
m.encoder.embeddings.data.uniform_(-0.3, 0.3)
parts = [samples(pose=k) * 2 - 1 for k in range(8)]


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 100 * 1e3
        best = us if best is None else min(best, us)
    return best


print("%3s %9s %10s %12s %10s" % ("K", "rows", "us/launch", "us per step", "of 8 TB/s"))
for K in (1, 2, 3, 4, 5, 8):
    x = torch.cat(parts[:K]).contiguous()
    d = torch.randn_like(x)
    d = (d / d.norm(dim=-1, keepdim=True)).contiguous()
    us = timed(lambda: fusedhead.hash_head_infer(m, x, d))
    print("%3d %9d %10.2f %12.2f %10.3f" % (K, x.shape[0], us, us / K, 516 * x.shape[0] / (us * 1e-6) / 8e12))
