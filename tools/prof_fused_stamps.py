#!/usr/bin/env python3
"""Cycle stamps of k_hash_fwd_fused from an instrumented A/B build (PVD_HIP_LIB=libpvd_hip_prof.so: every 97th workgroup's waves
write 7 s_memrealtime (100 MHz) stamps over the rgb output): start | inputs + weight DMA landed | first group's gathers issued | all levels
blended | tile visible (barrier) | head done + outputs stored | barrier.  Build: make -C aaai2023-pvd_amd/csrc prof.  Prints the phases of a few workgroups in ns (s_memtime ticks at 100 MHz)."""
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
x = (samples() * 2 - 1).contiguous()
d = torch.randn_like(x)
d = d / d.norm(dim=-1, keepdim=True)
for _ in range(3):
    out = fusedhead.hash_head_infer(m, x, d)
torch.cuda.synchronize()
sig, rgb, feat = fusedhead.hash_head_infer(m, x, d)
torch.cuda.synchronize()
n = (x.shape[0] // 128 // 97 + 1) * 4
st = rgb.view(-1)[: n * 8 * 2].view(torch.int64).view(-1, 8).cpu()
t0 = int(st[:, 0][st[:, 0] > 0].min())
names = ["inputs", "issue", "blend", "barrier", "head", "barrier2"]
print("workgroup.wave: start (ns after the first)  " + "  ".join(names))
import numpy as np
rows = []
for i, r in enumerate(st.tolist()):
    if r[0] <= 0:
        continue
    rows.append([(r[0] - t0) * 10] + [(r[k + 1] - r[k]) * 10 for k in range(6)] + [(r[6] - t0) * 10])
    print("%4d.%d  %8d   " % (i // 4 * 97, i % 4, (r[0] - t0) * 10) + "  ".join("%7d" % ((r[k + 1] - r[k]) * 10) for k in range(6)) + "   requested@%d" % ((r[7] - r[0]) * 10))
a = np.array(rows, dtype=np.float64)
print("median   %8d   " % np.median(a[:, 0]) + "  ".join("%7d" % v for v in np.median(a[:, 1:7], axis=0)) + "   end (ns after first start): median %d max %d" % (np.median(a[:, 7]), a[:, 7].max()))
