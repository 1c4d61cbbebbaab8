import os, sys, time
import os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd")]
import torch
from pvd.config import PVDConfig
from pvd.ops import hip_ops
from pvd.workload import DistillWorkload
dev = torch.device("cuda:0")
w = DistillWorkload(hip_ops(), dev, PVDConfig(), teacher_pretrain_steps=0)
tea = w.tea
for it in range(3):
    tea.iter_density = 0 if it == 0 else 20
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        tea.update_extra_state()
    torch.cuda.synchronize(); print("update_extra_state iter_density=%d: %.2f ms" % (tea.iter_density - 1, (time.perf_counter() - t0) * 1e3))
