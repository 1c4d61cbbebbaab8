#!/usr/bin/env python3
"""Fused head kernels (pvd_head_forward / pvd_head_backward) vs sample count: fixed cost vs per-tile cost."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd")]
import torch

import pvd_hip

dev = torch.device("cuda:0")
torch.manual_seed(0)


def run(kind, M, packed, iters=30):
    f32 = lambda *s: torch.randn(*s, device=dev) * 0.3
    if kind == 1:
        x0 = (torch.randn(M, 144, device=dev) * 0.3).half(); Wa1, Wa2 = f32(15, 144), None
        sraw, gsraw = f32(M), torch.empty(M, device=dev)
        gx = torch.empty(M, 144, dtype=torch.float16, device=dev)
        gWa1, gWa2 = torch.zeros(15, 144, device=dev), None
    else:
        x0 = (torch.randn(14, M, 2, device=dev) * 0.3).half(); Wa1, Wa2 = f32(64, 28), f32(16, 64)
        sraw, gsraw = None, None
        gx = torch.empty(14, M, 2, dtype=torch.float16, device=dev)
        gWa1, gWa2 = torch.zeros(64, 28, device=dev), torch.zeros(16, 64, device=dev)
    d = torch.randn(M, 3, device=dev); d = d / d.norm(dim=-1, keepdim=True)
    Wc1, Wc2, Wc3 = f32(64, 31), f32(64, 64), f32(3, 64)
    sig, rgb, feat = torch.empty(M, device=dev), torch.empty(M, 3, device=dev), torch.empty(M, 16, device=dev)
    gs, gr, gf = f32(M), f32(M, 3), f32(M, 16)
    gW = [torch.zeros_like(w) for w in (Wc1, Wc2, Wc3)]
    ws = torch.empty(pvd_hip.head_backward_workspace_floats(kind, M), device=dev)
    image = pvd_hip.head_pack_weights(kind, Wa1, Wa2, Wc1, Wc2, Wc3) if packed else None
    fwd = lambda: pvd_hip.head_forward(kind, x0, sraw, d, M, Wa1, Wa2, Wc1, Wc2, Wc3, -2.0, -2.0, 7.0, sig, rgb, feat, image=image)
    bwd = lambda: pvd_hip.head_backward(kind, x0, sraw, d, M, Wa1, Wa2, Wc1, Wc2, Wc3, -2.0, -2.0, 7.0, gs, gr, gf, gsraw, gx, gWa1, gWa2,
                                        *gW, ws, image=image)
    for _ in range(3):
        fwd(); bwd()
    with pvd_hip.KernelTimer({"pvd_head_forward", "pvd_head_backward"}) as kt:
        for _ in range(iters):
            fwd(); bwd()
    return kt.mean_ms("pvd_head_forward") * 1e3, kt.mean_ms("pvd_head_backward") * 1e3


for kind, name in ((1, "vm"), (0, "hash")):
    for M in (64, 4096, 23232, 46464, 92928, 185856):
        f, b = run(kind, M, False)
        fp, bp = run(kind, M, True)
        print(f"{name:5s} M={M:7d}  forward {f:7.1f} us (packed weights {fp:6.1f})   backward+dW reduce {b:7.1f} us (packed {bp:6.1f})", flush=True)
