"""Cycle-counter stamps from an INSTRUMENTED build of the library (not the product):
  hipcc <Makefile FLAGS> -DPVD_HEAD_PROFILE -c aaai2023-pvd_amd/csrc/fusedhead.hip -o /tmp/x.o; link it with the other objects into a
  scratch libpvd_hip.so and put that in place of aaai2023-pvd_amd/libpvd_hip.so on the GPU box before running this.
Used for the section timings quoted in DESIGN.md (weight load / forward / dX / dW / epilogue; cycles per lattice chunk)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd")]
import torch, pvd_hip
dev = torch.device("cuda:0")
for kind in (1, 0):
    for M in (64, 92928):
        f32 = lambda *s: torch.randn(*s, device=dev) * 0.3
        if kind == 1:
            x0 = (torch.randn(M, 144, device=dev) * 0.3).half(); Wa1, Wa2 = f32(15, 144), None
            sraw, gsraw = f32(M), torch.empty(M, device=dev); gx = torch.empty(M, 144, dtype=torch.float16, device=dev)
            gWa1, gWa2 = torch.zeros(15, 144, device=dev), None
        else:
            x0 = (torch.randn(14, M, 2, device=dev) * 0.3).half(); Wa1, Wa2 = f32(64, 28), f32(16, 64)
            sraw, gsraw = None, None; gx = torch.empty(14, M, 2, dtype=torch.float16, device=dev)
            gWa1, gWa2 = torch.zeros(64, 28, device=dev), torch.zeros(16, 64, device=dev)
        d = torch.randn(M, 3, device=dev); d = d / d.norm(dim=-1, keepdim=True)
        Wc1, Wc2, Wc3 = f32(64, 31), f32(64, 64), f32(3, 64)
        gs, gr, gf = f32(M), f32(M, 3), f32(M, 16)
        gW = [torch.zeros_like(w) for w in (Wc1, Wc2, Wc3)]
        n = pvd_hip.head_backward_workspace_floats(kind, M)
        ws = torch.zeros(n + 128, device=dev)
        for _ in range(3):
            pvd_hip.head_backward(kind, x0, sraw, d, M, Wa1, Wa2, Wc1, Wc2, Wc3, -2.0, -2.0, 7.0, gs, gr, gf, gsraw, gx, gWa1, gWa2, *gW, ws)
        torch.cuda.synchronize()
        st = ws[n:n + 60].view(torch.int64).cpu().tolist()
        st = [s for s in st if s != 0]
        print("kind", kind, "M", M, "stamps(delta cycles):", [b - a for a, b in zip(st, st[1:])], "total", st[-1] - st[0])
