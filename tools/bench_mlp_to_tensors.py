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
w.enable_graph()
for _ in range(5): w.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): loss, info, ps, pt = w.step()
torch.cuda.synchronize()
print("mlp->tensors distill step: %.3f ms, loss %.4f" % ((time.perf_counter() - t0) / 20 * 1e3, float(loss)))
