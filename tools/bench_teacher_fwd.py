#!/usr/bin/env python3
"""The frozen hash teacher's forward (lookup + head) on the bench's samples: two launches (pvd_grid_encode_forward_affine +
pvd_head_forward) vs the fused launch (pvd_hash_head_forward_fused), timed inside HIP graphs of 20 forwards each."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tools")]
import numpy as np
import torch

import fusedhead
import pvd_hip
from bench_grid_levels import samples
from pvd.config import PVDConfig
from pvd.ops import hip_ops
from pvd.workload import make_model

dev = torch.device("cuda:0")
opt = PVDConfig(model_type="hash")
m = make_model(hip_ops(), opt, "hash", True, dev).eval()
m.encoder.embeddings.data.uniform_(-0.3, 0.3)
if os.environ.get("PVD_BENCH_TABLE_SCALE"):  # e.g. 1e-4: the initialisation range, mostly f16 subnormals
    m.encoder.embeddings.data.mul_(float(os.environ["PVD_BENCH_TABLE_SCALE"]) / 0.3)
x01 = samples()
x = (x01 * 2 - 1).contiguous()
d = torch.randn_like(x)
d = d / d.norm(dim=-1, keepdim=True)
M = x.shape[0]


def timed(label):
    for _ in range(3):
        out = fusedhead.hash_head_infer(m, x, d)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            out = fusedhead.hash_head_infer(m, x, d)
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 100 * 1e3
    print("%-34s %7.1f us per forward  (%d samples; 552 B/sample -> %.0f GB/s)" % (label, us, M, 552 * M / us / 1e3))
    return out


fusedhead.FUSED_LOOKUP = False
for lps, persist in ((0, 0), (2, 0), (2, 4096)):
    pvd_hip.grid_set_fwd_kernel(lps, persist)
    ref = timed("two launches, lookup kernel lps=%d persist=%d" % (lps, persist))
pvd_hip.grid_set_fwd_kernel()
fusedhead.FUSED_LOOKUP = True
out = timed("fused lookup + head")
assert all(torch.equal(a, b) for a, b in zip(ref, out)), "fused and unfused outputs differ"
