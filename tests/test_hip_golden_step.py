"""The HIP path against the REFERENCE's own distillation step (tests/golden/reference_step.npz, made by
tests/golden/make_golden_step.py from `Trainer.train_step` over `run_cuda` run on the CPU): same rays, weights, occupancy grid
and background draw, fp32, through libpvd_hip.so -- loss, images and every gradient of the student, for the pairs and stages
of tests/test_golden_step.py.  Tolerances are those of the kernels against the oracle (march bit-exact; compositing, lookups
and their atomically accumulated gradients at fp32 rounding)."""
import numpy as np
import pytest
import torch

from test_golden_step import CASES, G, check_table_grad, config, load, rays_of

pytestmark = pytest.mark.gpu


# every pair and stage of the fixture (all eight pairs: each student family, each teacher family, either marching order)


@pytest.mark.parametrize("case,stage", CASES)
def test_hip_distillation_step_matches_the_references_own_train_step(case, stage):
    from pvd.ops import hip_ops
    from pvd.trainer import DistillTrainer
    from pvd.workload import make_model
    dev = torch.device("cuda:0")
    import types
    # (without the flat optimizer: it applies the VM L1 term's gradient inside its update kernel, and this test reads p.grad)
    ops, opt = types.SimpleNamespace(**{**vars(hip_ops()), "flat_adamw": None}), config(case)
    torch.manual_seed(0)
    tea = make_model(ops, opt, opt.teacher_type, True, dev)
    stu = make_model(ops, opt, opt.model_type, False, dev)
    load(tea, case, "tea"), load(stu, case, "stu")
    tr = DistillTrainer(opt, tea, stu, dev, fp16=False)
    pre = "%s__s%d__" % (case, stage)
    tr.global_step = tr.opt.global_step = int(G[pre + "global_step"])
    tr.loss_rate_fea_sc = float(G[pre + "fea_rate_before"])
    tr.rates[1] = tr.loss_rate_fea_sc  # (the rate the objective multiplies with lives next to the other three, on the device)
    rays_o, rays_d = [t.to(dev) for t in rays_of(case)]
    stu.train(), tea.train()
    tr._zero_grads()
    torch.manual_seed(int(G[pre + "seed"]))
    bg = torch.rand([1, rays_o.shape[1], 3], dtype=torch.float32).to(dev)  # drawn on the CPU, like the reference's run
    loss, info, pred_stu, pred_tea = tr.compute_loss(rays_o, rays_d, bg)
    loss.backward()
    torch.cuda.synchronize()
    marcher = stu if opt.render_stu_first else tea
    assert marcher.step_counter[(marcher.local_step - 1) % 16].tolist() == G[pre + "samples"].tolist()  # bit-exact marcher
    assert float(loss.detach()) == pytest.approx(float(G[pre + "loss"]), rel=3e-4), (float(loss.detach()), float(G[pre + "loss"]))
    if stage == 3:
        for got, name in ((pred_stu, "pred_stu"), (pred_tea, "pred_tea")):
            np.testing.assert_allclose(got.detach().float().cpu().numpy().reshape(G[pre + name].shape), G[pre + name], rtol=0, atol=1e-4)
    for n, p in stu.named_parameters():
        got = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().float().cpu()
        if "embeddings" in n:
            check_table_grad(got, pre, n, 1e-3, (case, stage))
            continue
        ref = G[pre + "grad__" + n]
        assert tuple(got.shape) == ref.shape, n
        scale = max(np.abs(ref).max(), 1e-12)
        err = np.abs(got.numpy() - ref).max()
        assert err <= 1e-3 * scale, (case, stage, n, err, scale)
