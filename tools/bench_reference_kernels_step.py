#!/usr/bin/env python3
"""What the REFERENCE's own native code + PyTorch does with the metric's step on this MI355X -- the closest thing to "the reference on
MI355X" this image allows -- next to this repo's step.

The reference's Python cannot travel to the GPU box, but its kernels can: oracle/_ref holds raymarching.cu and shencoder.cu of
/root/reference built for gfx950 by oracle/build_ref.py (torch.utils.cpp_extension.load, as the reference's backend.py files do).  This
tool binds THOSE modules under this repo's restatement of the reference's host code -- pvd/renderer.py, network.py, trainer.py in
their GENERIC form, i.e. what the reference's Python does: its autograd wrappers around the native calls (raymarching/raymarching.py),
12 x F.grid_sample for the VM student, nn.Linear heads under torch.autocast, torch.optim.AdamW + GradScaler, eager launches -- the
formulation tests/test_golden_step.py pins against the reference's own train_step.  The one native piece that is NOT the
reference's is the hash-grid encoder (gridencoder.cu does not build on HIP: atomicAdd(__half2*)): the teacher's lookup runs this
repo's generic encoder kernel through the reference-shaped GridEncoder wrapper.

    python tools/bench_reference_kernels_step.py [--steps 30] [--rays 4096]      (needs oracle/_ref and a GPU)"""
import argparse
import os
import sys
import time
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "aaai2023-pvd_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch

from oracle.build_ref import load_module


def reference_kernel_ops():
    """the operator set: the reference's raymarching / SH kernels under the reference-shaped wrappers; everything else torch"""
    import gridencoder
    from raymarching.raymarching import make_ops
    from shencoder.sphere_harmonics import SHEncoderBase, make_sh_encode
    rm, sh = load_module("_raymarching_ref"), load_module("_shencoder_ref")

    class RefSHEncoder(SHEncoderBase):
        _sh_encode = staticmethod(make_sh_encode(sh, device_type="cuda"))

    return types.SimpleNamespace(raymarching=make_ops(rm, device_type="cuda"), GridEncoder=gridencoder.GridEncoder, SHEncoder=RefSHEncoder,
                                 device_type="cuda", name="reference kernels + torch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--student", default="vm")
    a = ap.parse_args()
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import DistillWorkload
    dev = torch.device("cuda:0")
    opt = PVDConfig(num_rays=a.rays, model_type=a.student, fp16=True)
    rows = []
    for name, ops in (("reference's kernels + PyTorch (eager, as the reference's Python issues them)", reference_kernel_ops()),
                      ("this repo (libpvd_hip.so; 20 steps per hipGraph launch)", hip_ops())):
        torch.manual_seed(0)
        torch.cuda.manual_seed(1234)
        w = DistillWorkload(ops, dev, opt, teacher_pretrain_steps=0, seed=0)
        if ops.name == "hip":
            w.enable_graph(steps_per_graph=10)
        for _ in range(5):
            w.step()
        torch.cuda.synchronize()
        n_calls = max(1, a.steps // w.steps_per_call)
        t0 = time.perf_counter()
        for _ in range(n_calls):
            loss = w.step()[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = n_calls * w.steps_per_call
        samples = int(w.stu.step_counter[:, 0].float().mean().item())
        rows.append((name, dt / steps * 1e3, a.rays * steps / dt, samples, float(loss)))
        del w
        torch.cuda.empty_cache()
    print("hash -> %s distillation step, %d rays/step, fp16 AMP, synthetic chair, one MI355X" % (a.student, a.rays))
    for name, ms, rps, samples, loss in rows:
        print("%-84s %9.3f ms/step %12.0f rays/s   (%d samples/step, loss %.4f)" % (name, ms, rps, samples, loss))
    print("ratio: %.1fx" % (rows[0][1] / rows[1][1]))


if __name__ == "__main__":
    main()
