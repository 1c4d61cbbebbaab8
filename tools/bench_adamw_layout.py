#!/usr/bin/env python3
"""How much of k_adamw's time is the gather?  The bench's optimizer state after a few eager steps, then the update kernel alone
(HIP events) over (a) the real warm-group list (the touched set + the L1 ranges: runs of 64-192 bytes), (b) the same NUMBER of
groups as one contiguous range, (c) a random permutation of the real list's length -- same bytes per launch in all three."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd")]
import torch

import pvd_hip
from pvd.config import PVDConfig
from pvd.ops import hip_ops
from pvd.workload import DistillWorkload

dev = torch.device("cuda:0")
w = DistillWorkload(hip_ops(), dev, PVDConfig(), teacher_pretrain_steps=20)
for _ in range(4):
    w.step()
torch.cuda.synchronize()
o = w.trainer.optimizer
o.flush()
warm = o._warm_groups
n = int(warm.numel())
runs = int((warm[1:] != warm[:-1] + 1).sum()) + 1
print("warm groups %d of %d (%.1f%%), %d runs, mean run %.1f groups = %.0f bytes; warm B %d, A %d" % (
    n, o.flat_p.numel() // 4, 400.0 * n / o.flat_p.numel(), runs, n / runs, 16.0 * n / runs, o._warm_B.numel(), o._warm_A.numel()))
d = o.defaults
lists = {"real list": warm, "contiguous range": torch.arange(n, dtype=torch.int32, device=dev),
         "B then A": torch.cat([o._warm_B, o._warm_A]).contiguous(),
         "shuffled": warm[torch.randperm(n, device=dev)].contiguous()}
lazy = o._lazy_state()
for name, lst in lists.items():
    def step():
        pvd_hip.adamw_step(o.flat_p, o.flat_g, o.flat_m, o.flat_v, o.segment_ends, o.lr_dev, d["betas"][0], d["betas"][1], d["eps"],
                           d["weight_decay"], o.step_count, None, None, l1_ranges=getattr(o, "_l1", None), cold_bits=o._cold_bits,
                           lazy=(lazy[0], lazy[1], lst), zero_after=True)
    for _ in range(3):
        step()
    with pvd_hip.KernelTimer({"pvd_adamw_step_ex"}) as kt:
        for _ in range(20):
            step()
    torch.cuda.synchronize()
    lazy[1].zero_()
    us = kt.mean_ms("pvd_adamw_step_ex") * 1e3
    print("%-18s %7.1f us  (%.0f MB at 28 B per parameter = %.2f TB/s)" % (name, us, 112e-6 * n, 112.0 * n / us / 1e6), flush=True)
