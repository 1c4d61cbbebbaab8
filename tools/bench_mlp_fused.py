#!/usr/bin/env python3
"""The frozen NeRF-MLP model's forward (pvd_freq_encode + pvd_mlp_head_forward_fused) on one batch of samples, timed in HIP graphs."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tools")]
import torch

import fusedhead
from pvd.config import PVDConfig
from pvd.ops import hip_ops
from pvd.workload import make_model

dev = torch.device("cuda:0")
opt = PVDConfig(model_type="mlp", fp16=True)
m = make_model(hip_ops(), opt, "mlp", True, dev).train()
M = int(os.environ.get("M", 92928))
x = torch.rand(M, 3, device=dev) * 2 - 1
d = torch.randn(M, 3, device=dev)
d = d / d.norm(dim=-1, keepdim=True)
for label, fn in (("fused trunk + head", lambda: fusedhead.mlp_head_infer(m, x, d)),):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 50 * 1e3
    print("%-22s M=%d  %7.1f us per forward (incl. ~19 us of positional encoding)  %.0f TFLOP/s" % (label, M, us, 868e3 * M / us / 1e6))
