"""One process = 30 forwards of the frozen hash teacher (pvd_hash_head_forward_fused) on the bench's samples, for a
rocprofv3 --pmc pass; prints the sample count."""
import os, sys
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
for _ in range(30):
    fusedhead.hash_head_infer(m, x, d)
torch.cuda.synchronize()
print("samples_per_launch", x.shape[0])
