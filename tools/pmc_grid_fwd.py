"""One process = 30 launches of the hash-grid lookup on the bench's samples, for a rocprofv3 --pmc pass.
PVD_GRID_LPS = 0 / 2 / 4 selects the forward kernel, PVD_GRID_PERSIST the persistent grid size."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tools")]
import numpy as np, torch, pvd_hip
from bench_grid_levels import samples, enc, emb, S, dev
x = samples()
B = x.shape[0]
pvd_hip.grid_set_fwd_kernel(int(os.environ.get("PVD_GRID_LPS", "0")), int(os.environ.get("PVD_GRID_PERSIST", "0")))
out = torch.empty(14, B, 2, dtype=emb.dtype, device=dev)
for _ in range(30):
    pvd_hip.grid_encode_forward(x, emb, enc.offsets, out, B, 3, 2, 14, S, 16, False, out, 0, False)
torch.cuda.synchronize()
print("B", B)
