"""Why is the fused teacher forward 32 us inside bench.py and 26 us in tools/bench_teacher_fwd.py?  Same kernel, one
variable changed at a time."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tools")]
import numpy as np, torch
import fusedhead, pvd_hip
from pvd.config import PVDConfig
from pvd.ops import hip_ops
from pvd.workload import DistillWorkload, make_model

dev = torch.device("cuda:0")
opt = PVDConfig(num_rays=4096)
w = DistillWorkload(hip_ops(), dev, opt, teacher_pretrain_steps=100, seed=0)
w.enable_graph(steps_per_graph=5)
for _ in range(20):
    w.step()
torch.cuda.synchronize()


def timed(model, x, d, label):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        for _ in range(3):
            fusedhead.hash_head_infer(model, x, d)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                fusedhead.hash_head_infer(model, x, d)
        g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            g.replay()
        b.record()
        torch.cuda.synchronize()
    print("%-70s %6.1f us  (%d rows)" % (label, a.elapsed_time(b) / 100 * 1e3, x.shape[0]), flush=True)


with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    rays_o, rays_d, bg, *_ = w.device_batch()
    out = w.stu.render(rays_o, rays_d, staged=False, bg_color=bg, perturb=True, force_all_rays=False, dt_gamma=0, max_steps=1024)
xyzs, dirs = out["inherited_params"][0], out["inherited_params"][1]
n_real = int(out["rays"][:, 2].sum())
timed(w.tea, xyzs, dirs, "bench objects as they are (trained teacher, marcher's padded buffer)")
timed(w.tea, xyzs.clone(), dirs.clone(), "... samples cloned into fresh buffers")
timed(w.tea, xyzs[:n_real].clone(), dirs[:n_real].clone(), "... only the real samples (no padding rows)")
w.tea._emb_half_cache = None
timed(w.tea, xyzs, dirs, "... f16 table re-cast into a fresh allocation")
fresh = make_model(hip_ops(), opt, "hash", True, dev).eval()
fresh.encoder.embeddings.data.uniform_(-0.3, 0.3)
timed(fresh, xyzs, dirs, "fresh random-weight teacher, bench samples")
from bench_grid_levels import samples
x2 = (samples() * 2 - 1).contiguous()
d2 = torch.randn_like(x2); d2 = d2 / d2.norm(dim=-1, keepdim=True)
timed(fresh, x2, d2, "fresh teacher, tool samples (one pose, no sample budget)")
timed(w.tea, x2, d2, "trained teacher, tool samples")
print("padding rows in the bench batch:", xyzs.shape[0] - n_real, "; fraction of bench samples at exactly 0:", float((xyzs == 0).all(1).float().mean()))
