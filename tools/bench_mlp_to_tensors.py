import sys, time
import os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd")]
import torch
from pvd.config import PVDConfig
from pvd.ops import hip_ops
from pvd.workload import DistillWorkload
dev = torch.device("cuda:0")
opt = PVDConfig(num_rays=4096, model_type="tensors", teacher_type="mlp")
w = DistillWorkload(hip_ops(), dev, opt, teacher_pretrain_steps=20)
K = int(os.environ.get("PVD_STEPS_PER_GRAPH", "10"))  # steps per graph launch (> 1: the next step's prefix rides next to the update)
w.enable_graph(steps_per_graph=K)
for _ in range(max(1, 10 // K)): w.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = max(1, 40 // K)
for _ in range(n): loss, info, ps, pt = w.step()
torch.cuda.synchronize()
print("mlp->tensors distill step: %.3f ms, loss %.4f (%d steps per graph launch)" % ((time.perf_counter() - t0) / (n * K) * 1e3, float(loss), K))
