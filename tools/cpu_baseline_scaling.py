#!/usr/bin/env python3
"""Why does bench.py's cpu_baseline LOSE throughput from 16 to 128 threads (14.8 k rays/s at 16 threads,
3.3 k on all 128 physical cores)?  The CPU step is two thread pools working in turn: the oracle's C kernels (OpenMP: march,
grid / SH encoders, compositing) and PyTorch-CPU (MLP GEMMs, VM grid_sample, autograd, AdamW).  This tool times (a) the whole
distillation step by thread count and (b) the oracle kernels (called directly on preallocated buffers) and the torch pieces alone
at every count, so the piece that collapses is named.  (The first version pinned the two separately and found that it cannot be done:
both run on ONE libgomp whose thread count is a single global setting.)

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

    # ---- (a) the whole step.  FINDING of the first run of this tool (profiles/r06_cpu_baseline_scaling.txt): the two "pools" are ONE --
    # PyTorch-CPU and the oracle library both run on the process's libgomp, whose thread count is a single global setting:
    # whichever of torch.set_num_threads / omp_set_num_threads is called LAST decides for both (the rows of a 2-D table came out
    # identical).  So the table is one-dimensional, and the order below (torch first, the oracle's setter last) is what takes effect.
    def set_threads(n):
        torch.set_num_threads(n)
        oracle.set_num_threads(n)

    print("\n(a) whole distillation step by thread count (one shared OpenMP runtime), median of %d" % a.steps)
    whole = {}
    for n in counts:
        set_threads(n)
        whole[n] = timed(w.step, a.steps) * 1e3
        print("%8d threads %9.1f ms %9.0f rays/s" % (n, whole[n], a.rays / whole[n] * 1e3), flush=True)
    best = min(whole, key=whole.get)
    print("best: %d threads = %.1f ms = %.0f rays/s; %d threads = %.1f ms = %.0f rays/s" % (
        best, whole[best], a.rays / whole[best] * 1e3, ncpu, whole[ncpu], a.rays / whole[ncpu] * 1e3))

    # ---- (b) the pieces alone
    rm = w.stu.rm
    r_o, r_d, bg = w.next_batch()
    o, d = r_o.contiguous().view(-1, 3), r_d.contiguous().view(-1, 3)
    nears, fars = rm.near_far_from_aabb(o, d, w.stu.aabb_train, w.stu.min_near)

    xyzs, dirs, deltas, rays = rm.march_rays_train(o, d, w.stu.bound, w.stu.density_bitfield, w.stu.cascade, w.stu.grid_size, nears, fars, None, -1,
                                                   False, 128, True, 0, 1024)
    M = xyzs.shape[0]
    enc = w.tea.encoder
    x01 = ((xyzs + w.tea.bound) / (2 * w.tea.bound)).contiguous()
    sig = torch.rand(M)
    rgb = torch.rand(M, 3)

    # the oracle's C kernels called directly on preallocated buffers (through the autograd wrappers a piece also times the wrapper's
    # single-threaded torch work -- zero fills, the [L,B,C] -> [B,LC] permute -- which hid the kernels' own scaling in the first run)
    import oracle_backend as ob
    N = o.shape[0]
    bufs = dict(xyzs=torch.zeros(M, 3), dirs=torch.zeros(M, 3), deltas=torch.zeros(M, 2), rays=torch.zeros(N, 3, dtype=torch.int32),
                counter=torch.zeros(2, dtype=torch.int32), enc=torch.empty(14, M, 2), ws=torch.empty(N), depth=torch.empty(N), img=torch.empty(N, 3))
    emb, offsets = enc.embeddings.detach().contiguous(), enc.offsets
    S = float(torch.log2(torch.tensor(float(enc.per_level_scale))))
    bit = w.stu.density_bitfield.contiguous()

    def march():
        bufs["counter"].zero_()
        ob.march_rays_train(o, d, bit, float(w.stu.bound), 0.0, 1024, N, int(w.stu.cascade), int(w.stu.grid_size), M, nears, fars, bufs["xyzs"],
                            bufs["dirs"], bufs["deltas"], bufs["rays"], bufs["counter"], True)

    def grid_fwd():
        ob.grid_encode_forward(x01, emb, offsets, bufs["enc"], M, 3, 2, 14, S, int(enc.base_resolution), False, None, 0, False)

    def composite():
        ob.composite_rays_train_forward(sig, rgb, deltas, rays, M, N, bufs["ws"], bufs["depth"], bufs["img"])

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
    print("\n(b) the pieces alone by thread count, ms (median of %d)" % a.steps)
    print("%-58s " % "" + " ".join("%8d" % c for c in counts))
    for name, fn, pool in pieces:
        row = []
        for n in counts:
            set_threads(n)
            row.append(timed(fn, a.steps) * 1e3)
        print("%-58s " % name + " ".join("%8.1f" % v for v in row), flush=True)


if __name__ == "__main__":
    main()
