#!/usr/bin/env python3
"""The f16 scatter-add backward of the hash grid at the bench's sample count: run-merging thread-per-sample kernel
(k_grid_bwd_coarse) vs two lanes per sample (k_grid_bwd_lps2), HIP graphs of 10 launches; results compared."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tools")]
import torch

import pvd_hip
from bench_grid_levels import S, dev, emb, enc, samples

x = samples()
B = x.shape[0]
grad = (torch.randn(14, B, 2, device=dev) * 1e-3).half()
ge = torch.zeros_like(emb)
NO_LPS_BWD = 1 << 29


def run():
    pvd_hip.grid_encode_backward(grad, x, emb, enc.offsets, ge, B, 3, 2, 14, S, 16, False, grad, grad, 0, False)


def timed(label, knob):
    pvd_hip.grid_set_fwd_kernel(0, knob)
    ge.zero_()
    run()
    torch.cuda.synchronize()
    result = ge.float().clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            run()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 50 * 1e3
    print("%-40s %7.1f us  (%d samples, 964 B/sample -> %.0f GB/s)" % (label, us, B, 964 * B / us / 1e3))
    return result


r0 = timed("k_grid_bwd_coarse (thread per sample)", NO_LPS_BWD)
r1 = timed("k_grid_bwd_lps2 (two lanes per sample)", 0)
scale = r0.abs().max().item()
print("max |difference| / max |gradient| = %.2e; sums %.6f vs %.6f" % ((r0 - r1).abs().max().item() / scale, r0.sum().item(), r1.sum().item()))
pvd_hip.grid_set_fwd_kernel()
