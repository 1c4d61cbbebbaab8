#!/usr/bin/env python3
"""Per-level timing of the hash-grid scatter-add backward (pvd_grid_encode_backward) on ray-coherent samples."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tools")]
import numpy as np
import torch

import pvd_hip
from bench_grid import enc, ray_samples, S, dev  # noqa: E402  (also prints the forward table)

x = ray_samples(4096)
B = x.shape[0]
for dt in (torch.float16, torch.float32):
    emb = enc.embeddings.detach().to(dt)
    grad = (torch.randn(14, B, 2, device=dev) * 1e-3).to(dt)
    ge = torch.zeros_like(emb)

    def run():
        pvd_hip.grid_encode_backward(grad, x, emb, enc.offsets, ge, B, 3, 2, 14, S, 16, False, grad, grad, 0, False)

    def timeit(mask, iters=10):
        pvd_hip.grid_set_variant(mask << 8)
        run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            run()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e3

    print("backward %s B=%d: all levels %.1f us; per level:" % (dt, B, timeit(0)), " ".join("%d:%.0f" % (l, timeit(1 << l)) for l in range(14)))
    ALL = 1 << 21  # bit 29: run merging on the coarse levels only (scale < 300)
    print("  (run-merging on coarse levels only) all levels %.1f us; per level:" % timeit(ALL), " ".join("%d:%.0f" % (l, timeit(ALL | (1 << l))) for l in range(14)))
    OFF = 1 << 22  # bit 30 of the knob after the << 8 below: plain kernel for every level
    print("  (run-merging off) all levels %.1f us; per level:" % timeit(OFF), " ".join("%d:%.0f" % (l, timeit(OFF | (1 << l))) for l in range(14)))
pvd_hip.grid_set_variant(0)
