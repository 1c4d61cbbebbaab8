#!/usr/bin/env python3
"""The frozen hash teacher's fused forward (pvd_hash_head_forward_fused) over the samples of K training batches in ONE launch
(K = 1, 2, 5, 10, 20; every batch from a different camera), timed inside HIP graphs: what batching the parameter-independent
prefix of K consecutive distillation steps buys."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
xs = [(samples(4096, pose=p) * 2 - 1).contiguous() for p in range(20)]
for K in (1, 2, 3, 4, 5, 8, 10, 20):
    x = torch.cat(xs[:K]).contiguous()
    d = torch.randn_like(x)
    d = d / d.norm(dim=-1, keepdim=True)
    M = x.shape[0]
    for _ in range(3):
        fusedhead.hash_head_infer(m, x, d)
    torch.cuda.synchronize()
    per = max(2, 20 // K)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per):
            fusedhead.hash_head_infer(m, x, d)
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / (5 * per) * 1e3
    print("K=%2d  %8d samples  %7.1f us per launch = %6.1f us per batch   552 B/sample -> %5.0f GB/s = %.3f of 8 TB/s" % (
        K, M, us, us / K, 552 * M / us / 1e3, 552 * M / us / 1e3 / 8000), flush=True)
