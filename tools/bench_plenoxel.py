"""Plenoxel lookup + head, fused HIP kernels vs the reference formulation (3-D F.grid_sample + torch head) on
ray-coherent samples (64 steps of 3.38e-3 along random rays) and on random points.
  python tools/bench_plenoxel.py [--samples 92928]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aaai2023-pvd_amd"))

import plenoxel  # noqa: E402
import pvd_hip  # noqa: E402
import shencoder  # noqa: E402
from pvd.activation import make_trunc_exp  # noqa: E402

AABB = (-1.0, -1.0, -1.0, 1.0, 1.0, 1.0)


def points(M, coherent):
    if coherent:
        n = (M + 63) // 64
        o = torch.rand(n, 1, 3, device="cuda") * 1.6 - 0.8
        d = torch.randn(n, 1, 3, device="cuda")
        d = d / d.norm(dim=-1, keepdim=True)
        t = torch.arange(64, device="cuda").view(1, 64, 1) * 3.3829e-3
        return (o + t * d).reshape(-1, 3)[:M].contiguous(), d.expand(n, 64, 3).reshape(-1, 3)[:M].contiguous()
    d = torch.randn(M, 3, device="cuda")
    return torch.rand(M, 3, device="cuda") * 2 - 1, d / d.norm(dim=-1, keepdim=True)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=92928)
    a = ap.parse_args()
    M = a.samples
    vol = plenoxel.to_channels_last_3d_param(torch.randn(1, 28, 128, 128, 128, device="cuda") * 0.02).requires_grad_(True)
    vol_cm = vol.detach().contiguous().requires_grad_(True)  # the reference's channel-major layout
    trunc_exp = make_trunc_exp("cuda")
    sh = shencoder.SHEncoder(degree=3)
    for coherent in (True, False):
        x, d = points(M, coherent)
        ws, wc = torch.randn(M, device="cuda") * 0.01, torch.randn(M, 3, device="cuda")

        def fused(backward):
            s, c, sl, _ = plenoxel.plenoxel_head(x, d, AABB, vol, 3, -2.0, 7.0)
            if backward:
                vol.grad = None
                ((s * ws).sum() + (c * wc).sum()).backward()

        def ref(backward, v=vol_cm):
            h = F.grid_sample(v, x.view(1, 1, -1, 1, 3), align_corners=True).view(-1, M).permute(1, 0)
            sl = torch.clamp(h[..., 0], -2.0, 7.0)
            s = trunc_exp(sl)
            c = torch.sigmoid((h[..., 1:].view(-1, 3, 9) * sh(d).unsqueeze(1)).sum(-1))
            if backward:
                v.grad = None
                ((s * ws).sum() + (c * wc).sum()).backward()

        with torch.no_grad():
            f_us, r_us = timeit(lambda: fused(False)), timeit(lambda: ref(False))
        fb_us, rb_us = timeit(lambda: fused(True)), timeit(lambda: ref(True))
        # kernel-only durations (HIP events around the two entry points), gradient accumulated in place
        vol.grad = torch.zeros_like(vol)
        names = {"pvd_plenoxel_forward", "pvd_plenoxel_backward"}
        fused_direct = lambda: ((lambda o: ((o[0] * ws).sum() + (o[1] * wc).sum()).backward())(plenoxel.plenoxel_head(x, d, AABB, vol, 3, -2.0, 7.0)))
        for _ in range(3):
            fused_direct()
        with pvd_hip.KernelTimer(names) as kt:
            for _ in range(20):
                fused_direct()
        kf, kb = kt.mean_ms("pvd_plenoxel_forward") * 1e3, kt.mean_ms("pvd_plenoxel_backward") * 1e3
        vol.grad = None
        alg_f, alg_b = 896 * M, (896 * 2 + 24) * M
        print(f"{'coherent' if coherent else 'random  '} M={M}: k_plenoxel_fwd {kf:6.1f} us ({alg_f / kf / 1e6:5.2f} TB/s of 896 B/sample)  "
              f"k_plenoxel_bwd {kb:6.1f} us ({alg_b / kb / 1e6:5.2f} TB/s of {alg_b // M} B/sample) | end-to-end forward fused {f_us:6.1f} us, "
              f"torch {r_us:7.1f} us; fwd+bwd fused {fb_us:7.1f} us, torch {rb_us:7.1f} us (both incl. a 235 MB grad zero-fill)", flush=True)
