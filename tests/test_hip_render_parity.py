"""End-to-end render parity: the same weights rendered by the HIP operator set on the GPU and by the CPU oracle
operator set (tests/oracle_ops.py), fp32, for every model family -- the training branch (march_rays_train +
composite_rays_train) and the inference branch (march_rays / composite_rays / compact_rays rounds).
north_star's bar: RGB within 1e-4, PSNR within 0.1 dB."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(kind):
    from oracle_ops import oracle_ops
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.scene import ChairScene
    from pvd.workload import install_occupancy, make_model
    torch.manual_seed(0)
    opt = PVDConfig(model_type=kind, resolution0=48, plenoxel_res="[32,32,32]", fp16=False)
    opt.stage_iters = {"stage1": -1, "stage2": -1}  # stage 3: the training branch composites an image
    opt.global_step = 0
    cpu = make_model(oracle_ops(), opt, kind, False, torch.device("cpu"))
    for p in cpu.parameters():
        if p.dim() == 2:
            p.data.mul_(2.0)
    if kind == "hash":
        cpu.encoder.embeddings.data.uniform_(-0.5, 0.5)
    gpu = make_model(hip_ops(), opt, kind, False, torch.device("cuda:0"))
    gpu.load_state_dict(cpu.state_dict())
    scene = ChairScene(thicken=0.08)
    for m in (cpu, gpu):
        install_occupancy(m, scene, opt)
    return cpu, gpu


def _rays(n, seed):
    from pvd.scene import BLENDER_INTRINSICS, get_rays, synthetic_poses
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(seed)))
    r = get_rays(poses[3][None], BLENDER_INTRINSICS, 800, 800, n, generator=torch.Generator().manual_seed(seed))
    return r["rays_o"], r["rays_d"]


@pytest.mark.parametrize("kind", ["hash", "vm", "tensors", "mlp"])
def test_render_matches_cpu_oracle_path(kind):
    cpu, gpu = _pair(kind)
    o, d = _rays(1024, 5)
    bg = torch.rand(1, 1024, 3, generator=torch.Generator().manual_seed(1))
    # the product paths that only exist on the GPU (fused heads) are autocast-only: fp32 runs the shared formulation
    for training in (True, False):
        outs = []
        for m, dev in ((cpu, "cpu"), (gpu, "cuda:0")):
            m.train(training)
            with torch.no_grad():
                kw = dict(staged=False, bg_color=bg.to(dev), perturb=training, max_steps=1024)
                if training:
                    kw.update(force_all_rays=True, dt_gamma=0)
                out = m.render(o.to(dev), d.to(dev), **kw)
            outs.append((out["image"].float().cpu(), out["depth"].float().cpu()))
        (img_c, dep_c), (img_g, dep_g) = outs
        assert torch.isfinite(img_g).all()
        err = (img_c - img_g).abs().max().item()
        assert err <= 1e-4, (kind, training, err)
        mse = ((img_c - img_g) ** 2).mean().item()
        assert mse == 0.0 or -10 * np.log10(mse) > 80.0  # i.e. far inside "PSNR within 0.1 dB"
        assert (dep_c - dep_g).abs().max().item() <= 1e-4
        assert img_c.std().item() > 0.05  # a non-trivial image
