#!/usr/bin/env python3
"""Why does bench.py's cpu_baseline LOSE throughput from 16 to 128 threads (14.8 k rays/s at 16 threads,
3.3 k on all 128 physical cores)?  The CPU step is two thread pools working in turn: the oracle's C kernels (OpenMP: march,
grid / SH encoders, compositing) and PyTorch-CPU (MLP GEMMs, VM grid_sample, autograd, AdamW).  This tool pins them SEPARATELY
(oracle.set_num_threads vs torch.set_num_threads) and times (a) the whole distillation step for every pair of counts, (b) the
oracle kernels alone and the torch pieces alone at every count, so the piece that collapses is named.

    python tools/cpu_baseline_scaling.py [--rays 4096] [--steps 3]        # CPU only, no GPU needed"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import oracle  # noqa: E402  (a measurement tool of the cpu_baseline leg, like bench.py's)
from oracle_ops import oracle_ops  # noqa: E402
from pvd.config import PVDConfig  # noqa: E402
from pvd.workload import DistillWorkload  # noqa: E402


def timed(fn, n):
    fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return sorted(t)[len(t) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--counts", type=str, default="", help="comma-separated thread counts (default: 1, 4, 8, ... up to the logical cpus)")
    a = ap.parse_args()
    ncpu = os.cpu_count() or 1
    counts = sorted({c for c in (1, 4, 8, 16, 32, 64, 128, 256) if c <= ncpu} | {ncpu})
    if a.counts:
        counts = sorted({int(c) for c in a.counts.split(",")})
        ncpu = counts[-1]
    opt = PVDConfig(num_rays=a.rays, fp16=False)
    w = DistillWorkload(oracle_ops(), "cpu", opt, teacher_pretrain_steps=0, seed=0)
    print("host: %d logical cpus; %d rays/step; mean_count %d; torch intra-op default %d, oracle default %d" % (
        ncpu, a.rays, w.stu.mean_count, torch.get_num_threads(), oracle.num_threads()))

    # ---- (a) the whole step, oracle threads x torch threads
    print("\n(a) whole distillation step, ms (median of %d): rows = oracle (OpenMP) threads, columns = torch threads" % a.steps)
    print("%8s " % "" + " ".join("%9d" % c for c in counts))
    grid = {}
    for no in counts:
        row = []
        for nt in counts:
            oracle.set_num_threads(no)
            torch.set_num_threads(nt)
            ms = timed(w.step, a.steps) * 1e3
            grid[(no, nt)] = ms
            row.append(ms)
        print("%8d " % no + " ".join("%9.1f" % v for v in row), flush=True)
    best = min(grid, key=grid.get)
    print("best: oracle %d x torch %d threads = %.1f ms = %.0f rays/s; all-%d x all-%d = %.1f ms = %.0f rays/s" % (
        best[0], best[1], grid[best], a.rays / grid[best] * 1e3, ncpu, ncpu, grid[(ncpu, ncpu)], a.rays / grid[(ncpu, ncpu)] * 1e3))

    # ---- (b) the pieces alone
    rm = w.stu.rm
    r_o, r_d, bg = w.next_batch()
    o, d = r_o.contiguous().view(-1, 3), r_d.contiguous().view(-1, 3)
    nears, fars = rm.near_far_from_aabb(o, d, w.stu.aabb_train, w.stu.min_near)

    def march():
        return rm.march_rays_train(o, d, w.stu.bound, w.stu.density_bitfield, w.stu.cascade, w.stu.grid_size, nears, fars, None, -1, False, 128,
                                   True, 0, 1024)
    xyzs, dirs, deltas, rays = march()
    M = xyzs.shape[0]
    enc = w.tea.encoder
    x01 = ((xyzs + w.tea.bound) / (2 * w.tea.bound)).contiguous()
    sig = torch.rand(M)
    rgb = torch.rand(M, 3)

    def grid_fwd():
        with torch.no_grad():
            return enc(xyzs, bound=w.tea.bound)

    def composite():
        return rm.composite_rays_train(sig, rgb, deltas, rays)

    def teacher_fwd():
        with torch.no_grad():
            return w.tea(xyzs, dirs)

    def student_fwd_bwd():
        for p in w.stu.parameters():
            p.grad = None
        s, c = w.stu(xyzs, dirs)
        (s.sum() + c.sum()).backward()

    lin = torch.nn.Sequential(torch.nn.Linear(32, 64, bias=False), torch.nn.ReLU(), torch.nn.Linear(64, 64, bias=False), torch.nn.ReLU(),
                              torch.nn.Linear(64, 3, bias=False))
    xin = torch.randn(M, 32)

    def mlp_fwd_bwd():
        lin.zero_grad()
        lin(xin).sum().backward()

    opt_params = [p for p in w.stu.parameters() if p.requires_grad]
    for p in opt_params:
        p.grad = torch.zeros_like(p)
    adamw = torch.optim.AdamW(opt_params, lr=1e-3)

    pieces = [("oracle: march_rays_train (%d samples)" % M, march, "omp"), ("oracle: grid_encode fwd 14 levels", grid_fwd, "omp"),
              ("oracle: composite fwd", composite, "omp"), ("mixed: hash teacher forward (oracle grid+SH, torch MLP)", teacher_fwd, "both"),
              ("torch: VM student fwd+bwd (grid_sample x12, MLP)", student_fwd_bwd, "torch"), ("torch: 32-64-64-3 MLP fwd+bwd", mlp_fwd_bwd, "torch"),
              ("torch: AdamW over the student (%.1f M params)" % (sum(p.numel() for p in opt_params) / 1e6), adamw.step, "torch")]
    print("\n(b) the pieces alone, ms (median of %d); the OTHER pool pinned to 1 thread (mixed: both pools move together)" % a.steps)
    print("%-58s " % "" + " ".join("%8d" % c for c in counts))
    for name, fn, pool in pieces:
        row = []
        for n in counts:
            oracle.set_num_threads(n if pool in ("omp", "both") else 1)
            torch.set_num_threads(n if pool in ("torch", "both") else 1)
            row.append(timed(fn, a.steps) * 1e3)
        print("%-58s " % name + " ".join("%8.1f" % v for v in row), flush=True)


if __name__ == "__main__":
    main()
