"""The REFERENCE's own distillation step -- `Trainer.train_step` (distill_mutual/utils.py:954-1189) over `run_cuda`
(distill_mutual/renderer.py:319-448) of two of its `NeRFNetwork`s, run on the CPU in tests/golden/make_golden_step.py with the
CPU oracle underneath as the native operators -- reproduced by this repo's renderer + DistillTrainer on the same rays, weights,
occupancy grid and background draw: loss, the fea-rate decay, both rendered images and every gradient of the student, for

    hash -> vm        stages 1, 2, 3   (the bench's pair, BASELINE configs[2])
    hash -> vm        stage 3 with the TEACHER marching first (render_stu_first = False, renderer.py:392-411)
    mlp  -> tensors   stages 2, 3      (configs[3]; no feature vector, so no stage 1)
    hash -> hash      stages 1, 3      (configs[4]; and stage 3 with bound 2 / two cascades / dt_gamma 1/256)
    hash -> mlp       stages 1, 3      (a NeRF-MLP student)
    vm -> tensors, tensors -> vm       stage 3 (the other two teacher families)

Pins the Python restatement of the path: stage gating, who marches / who inherits, density_scale, background mix, the four
terms and their rates, the VM L1 term, autograd through the wrappers.  Kernel arithmetic is the oracle's on both sides."""
import os

import numpy as np
import pytest
import torch

from oracle_ops import oracle_ops
from pvd.config import PVDConfig
from pvd.trainer import DistillTrainer
from pvd.workload import make_model

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_step.npz"), allow_pickle=False)
CASES = [(str(c), int(s)) for c in G["cases"] for s in G[str(c) + "__stages"]]


def config(case):
    tea_type, stu_type, stu_first, bound, dt_gamma = [str(v) for v in G[case + "__cfg"]]
    return PVDConfig(model_type=stu_type, teacher_type=tea_type, bound=float(bound), PE=6, skip=2, nerf_layer_num=5, nerf_layer_wide=32, resolution0=12,
                     plenoxel_res="[12,12,12]", grid_size=int(G["grid_size"]), density_thresh=10.0, fp16=False,
                     stage_iters={"stage1": 2000, "stage2": 5000}, global_step=0, num_rays=G["rays_o"].shape[1],
                     max_steps=int(G["max_steps"]), dt_gamma=float(dt_gamma), loss_type="normL2", l1_reg_weight=float(G["l1_reg_weight"]),
                     loss_rate_rgb=float(G["loss_rate_rgb"]), loss_rate_fea_sc=float(G["loss_rate_fea_sc"]),
                     loss_rate_color=float(G["loss_rate_color"]), loss_rate_sigma=float(G["loss_rate_sigma"]),
                     render_stu_first=stu_first == "1")


def load(net, case, role):
    sd = {}
    for k in [str(k) for k in G["%s__%s_keys" % (case, role)]]:
        if "embeddings" in k:
            torch.manual_seed(777)  # the 42 MB table is regenerated, as in make_golden_step.py
            sd[k] = (torch.rand(net.state_dict()[k].shape) - 0.5) * 0.6
        else:
            sd[k] = torch.from_numpy(G["%s__%s_sd__%s" % (case, role, k)])
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    net.density_bitfield.copy_(torch.from_numpy(G["bitfield" if net.cascade == 1 else "bitfield2"]))
    net.mean_count = int(G["mean_count"])
    net.note_occupancy_changed()


def rays_of(case):
    two = str(G[case + "__cfg"][3]) == "2"
    return torch.from_numpy(G["rays_o2" if two else "rays_o"]), torch.from_numpy(G["rays_d2" if two else "rays_d"])


def check_table_grad(got, pre, name, tol, where):
    """The hash table's gradient against the fixture: every 7th non-zero row, the column sums, the number of rows reached."""
    rows, ref = torch.from_numpy(G[pre + "grad_rows__" + name]).long(), G[pre + "grad_vals__" + name]
    scale = max(np.abs(ref).max(), 1e-12)
    assert np.abs(got[rows].numpy() - ref).max() <= tol * scale, (where, name, np.abs(got[rows].numpy() - ref).max(), scale)
    assert np.abs(got.double().sum(0).numpy() - G[pre + "grad_colsum__" + name]).max() <= tol * float(G[pre + "grad_abssum__" + name]), (where, name)
    n_ref = int(G[pre + "grad_nrows__" + name])
    assert abs(int((got.abs().sum(1) > 0).sum()) - n_ref) <= max(2, n_ref // 200), (where, name)


_trainers = {}


def trainer_of(case):
    if case not in _trainers:
        ops, opt, dev = oracle_ops(), config(case), torch.device("cpu")
        torch.manual_seed(0)
        tea = make_model(ops, opt, opt.teacher_type, True, dev)
        stu = make_model(ops, opt, opt.model_type, False, dev)
        load(tea, case, "tea"), load(stu, case, "stu")
        _trainers[case] = DistillTrainer(opt, tea, stu, dev, fp16=False)
    return _trainers[case]


@pytest.mark.parametrize("case,stage", CASES)
def test_distillation_step_matches_the_references_own_train_step(case, stage):
    tr = trainer_of(case)
    pre = "%s__s%d__" % (case, stage)
    tr.global_step = tr.opt.global_step = int(G[pre + "global_step"])
    tr.loss_rate_fea_sc = float(G[pre + "fea_rate_before"])
    tr.rates[1] = tr.loss_rate_fea_sc  # (the rate the objective multiplies with lives next to the other three, on the device)
    rays_o, rays_d = rays_of(case)
    for p in tr.model_stu.parameters():
        p.grad = None
    tr.model_stu.train(), tr.model_tea.train()
    torch.manual_seed(int(G[pre + "seed"]))
    bg = torch.rand([1, rays_o.shape[1], 3], dtype=torch.float32)  # the step's random background (utils.py:989-996): first draw
    loss, info, pred_stu, pred_tea = tr.compute_loss(rays_o, rays_d, bg)
    loss.backward()
    assert tr.loss_rate_fea_sc == pytest.approx(float(G[pre + "fea_rate_after"]), rel=1e-12)
    marcher = tr.model_stu if tr.opt.render_stu_first else tr.model_tea
    assert marcher.step_counter[(marcher.local_step - 1) % 16].tolist() == G[pre + "samples"].tolist()  # same samples, same rays kept
    assert float(loss.detach()) == pytest.approx(float(G[pre + "loss"]), rel=2e-5), (float(loss.detach()), float(G[pre + "loss"]))
    if stage == 3:
        for got, name in ((pred_stu, "pred_stu"), (pred_tea, "pred_tea")):
            np.testing.assert_allclose(got.detach().numpy().reshape(G[pre + name].shape), G[pre + name], rtol=0, atol=2e-6)
    reached = 0
    for n, p in tr.model_stu.named_parameters():
        got = (p.grad if p.grad is not None else torch.zeros_like(p)).detach()
        if "embeddings" in n:
            check_table_grad(got, pre, n, 5e-5, (case, stage))
            reached += 1
            continue
        ref = G[pre + "grad__" + n]
        got = got.numpy()
        assert got.shape == ref.shape, n
        scale = max(np.abs(ref).max(), 1e-12)
        assert np.abs(got - ref).max() <= 5e-5 * scale, (case, stage, n, np.abs(got - ref).max(), scale)
        reached += int(np.abs(ref).max() > 0)
        if np.abs(ref).max() == 0:
            assert np.abs(got).max() == 0, (case, stage, n)  # a parameter the stage does not reach stays untouched
    assert reached >= 1, reached


@pytest.mark.parametrize("case", [str(c) for c in G["teacher_cases"]])
def test_teacher_training_step_matches_the_references_own_train_step(case):
    """just_train_tea/utils.py:746-846 over just_train_tea/renderer.py's run_cuda, run on the CPU by make_golden_step.py: one model,
    MSE against alpha-composited pixels on a random background (+ the VM L1 term), reproduced by TeacherTrainer's step body."""
    from pvd.trainer import TeacherTrainer
    mt = case.split("_")[1]
    opt = PVDConfig(model_type=mt, teacher_type=mt, PE=6, skip=2, nerf_layer_num=5, nerf_layer_wide=32, resolution0=12,
                    plenoxel_res="[12,12,12]", grid_size=int(G["grid_size"]), density_thresh=10.0, fp16=False, num_rays=G["rays_o"].shape[1],
                    max_steps=int(G["max_steps"]), dt_gamma=0.0, l1_reg_weight=float(G["l1_reg_weight"]), stage_iters={"stage1": -1, "stage2": -1})
    dev = torch.device("cpu")
    torch.manual_seed(0)
    net = make_model(oracle_ops(), opt, mt, False, dev, teacher_variant=True)
    sd = {}
    for k in [str(k) for k in G[case + "__keys"]]:
        if "embeddings" in k:
            torch.manual_seed(777)
            sd[k] = (torch.rand(net.state_dict()[k].shape) - 0.5) * 0.6
        else:
            sd[k] = torch.from_numpy(G["%s__sd__%s" % (case, k)])
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    net.density_bitfield.copy_(torch.from_numpy(G["bitfield"]))
    net.mean_count = int(G["mean_count"])
    net.note_occupancy_changed()
    net.train()
    tr = TeacherTrainer(opt, net, dev, fp16=False)
    images = torch.from_numpy(G["images"])
    torch.manual_seed(int(G[case + "__seed"]))
    bg = torch.rand_like(images[..., :3])  # pixel-wise random background: the step's first draw (just_train_tea/utils.py:782)
    gt = images[..., :3] * images[..., 3:] + bg * (1 - images[..., 3:])
    np.testing.assert_allclose(gt.numpy(), G[case + "__gt"], rtol=0, atol=1e-7)
    loss, pred = tr._block_body([(torch.from_numpy(G["rays_o"]), torch.from_numpy(G["rays_d"]), gt, bg)])()
    loss.backward()
    assert net.step_counter[(net.local_step - 1) % 16].tolist() == G[case + "__samples"].tolist()
    np.testing.assert_allclose(pred.detach().numpy().reshape(G[case + "__pred"].shape), G[case + "__pred"], rtol=0, atol=2e-6)
    assert float(loss.detach()) == pytest.approx(float(G[case + "__loss"]), rel=2e-5)
    for n, p in net.named_parameters():
        got = (p.grad if p.grad is not None else torch.zeros_like(p)).detach()
        if "embeddings" in n:
            check_table_grad(got, case + "__", n, 5e-5, case)
            continue
        ref = G[case + "__grad__" + n]
        scale = max(np.abs(ref).max(), 1e-12)
        assert np.abs(got.numpy() - ref).max() <= 5e-5 * scale, (case, n, np.abs(got.numpy() - ref).max(), scale)


@pytest.mark.parametrize("case", ["hash_vm", "mlp_tensors", "hash_hash_b2"])
def test_inference_rounds_match_the_references_own_loop(case):
    """The inference branch of run_cuda (renderer.py:450-543: rounds of march_rays / composite_rays / compact_rays over the rays
    still alive) of both models of the pair, as the reference's own loop computed it on the CPU: image and depth per ray (rays that
    miss the box have depth 0/0 there, and here)."""
    tr = trainer_of(case)
    rays_o, rays_d = rays_of(case)
    for role, net in (("tea", tr.model_tea), ("stu", tr.model_stu)):
        net.eval()
        with torch.no_grad():
            res = net.render(rays_o, rays_d, staged=False, bg_color=None, perturb=False, dt_gamma=tr.opt.dt_gamma, max_steps=int(G["max_steps"]))
        net.train()
        ref_i, ref_d = G["%s__eval_%s_image" % (case, role)], G["%s__eval_%s_depth" % (case, role)]
        np.testing.assert_allclose(res["image"].numpy().reshape(ref_i.shape), ref_i, rtol=0, atol=3e-6)
        np.testing.assert_allclose(res["depth"].numpy().reshape(ref_d.shape), ref_d, rtol=1e-5, atol=3e-6, equal_nan=True)
        assert np.isnan(ref_d).any() and np.isfinite(ref_d).sum() > 60


def test_vm_utilities_and_parameter_groups_match_the_reference():
    """density_loss (network.py:549-558), upsample_model (:560-587) of the reference's VM student, and the optimizer parameter
    groups of all four model types (get_params, :646-700) as (lr, parameter names)."""
    from pvd.checkpoint import upsample_vm
    ops, opt, dev = oracle_ops(), config("hash_vm"), torch.device("cpu")
    torch.manual_seed(0)
    stu = make_model(ops, opt, "vm", False, dev)
    load(stu, "hash_vm", "stu")
    assert float(stu.density_loss().detach()) == pytest.approx(float(G["vm__density_loss"]), rel=1e-6)
    upsample_vm(stu, [int(v) for v in G["vm__upsampled_to"]])
    sd = stu.state_dict()
    n = 0
    for k in G.files:
        if k.startswith("vm__up__"):
            got = sd[k[len("vm__up__"):]].detach().numpy()
            assert got.shape == G[k].shape, k
            np.testing.assert_allclose(got, G[k], rtol=0, atol=1e-6)
            n += 1
    assert n >= 12, n  # 3 x (sigma_mat, sigma_vec, color_mat, color_vec) (+ basis_mat.weight: untouched)
    assert float(stu.density_loss().detach()) == pytest.approx(float(G["vm__density_loss_up"]), rel=1e-6)
    for mt in ("hash", "mlp", "vm", "tensors"):
        o = PVDConfig(**{**opt.__dict__, "model_type": mt})
        net = make_model(ops, o, mt, False, dev)
        names = {id(p): n for n, p in net.named_parameters()}
        groups = ["%r|%s" % (float(g["lr"]), ",".join(names[id(p)] for p in g["params"])) for g in net.get_params(0.02)]
        groups = [g for g in groups if g.split("|")[1]]  # (the reference lists the parameter-free direction encoder as a group)
        ref = [str(g) for g in G["groups__" + mt] if str(g).split("|")[1]]
        assert groups == ref, (mt, groups, ref)


def test_fixed_step_sampler_matches_the_references_own_run():
    """configs[0]: the reference's fixed-step sampler `run` (just_train_tea/renderer.py:139-317 shape) of an `mlp` model, run on the
    CPU by make_golden_step.py with its dead `color()` replaced by the model's forward on the rows above the weight threshold:
    uniform steps, perturbation and importance resampling (torch's generator consumed in the reference's order), sort / gather,
    cumprod compositing, depth, white background -- train mode with gradients of every parameter, eval mode (deterministic
    resampling)."""
    opt = PVDConfig(model_type="mlp", teacher_type="mlp", PE=6, skip=2, nerf_layer_num=5, nerf_layer_wide=32, resolution0=12,
                    plenoxel_res="[12,12,12]", grid_size=int(G["grid_size"]), density_thresh=10.0, fp16=False, cuda_ray=False,
                    stage_iters={"stage1": -1, "stage2": -1})
    torch.manual_seed(0)
    net = make_model(oracle_ops(), opt, "mlp", False, torch.device("cpu"), teacher_variant=True)
    sd = {k: torch.from_numpy(G["run_mlp__sd__" + k]) for k in [str(k) for k in G["run_mlp__keys"]]}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    rays_o, rays_d = torch.from_numpy(G["rays_o"]), torch.from_numpy(G["rays_d"])
    T, t = int(G["run_mlp__num_steps"]), int(G["run_mlp__upsample_steps"])
    for mode, perturb in (("train", True), ("eval", False)):
        pre = "run_mlp__%s__" % mode
        net.train(mode == "train")
        for p in net.parameters():
            p.grad = None
        torch.manual_seed(int(G[pre + "seed"]))
        res = net.render(rays_o, rays_d, staged=False, bg_color=None, perturb=perturb, num_steps=T, upsample_steps=t)
        np.testing.assert_allclose(res["image"].detach().numpy(), G[pre + "image"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(res["depth"].detach().numpy(), G[pre + "depth"], rtol=1e-5, atol=2e-6, equal_nan=True)
        if mode == "train":
            (res["image"] * torch.from_numpy(G["run_mlp__g_image"])).sum().backward()
            for n, p in net.named_parameters():
                ref = G[pre + "grad__" + n]
                got = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
                assert np.isfinite(ref).all(), n
                scale = max(np.abs(ref).max(), 1e-12)
                assert np.abs(got - ref).max() <= 5e-5 * scale, (n, np.abs(got - ref).max(), scale)


@pytest.mark.parametrize("fname,case,role", [("reference_ckpt_vm_student.pth", "hash_vm", "stu"), ("reference_ckpt_mlp_teacher.pth", "mlp_tensors", "tea")])
def test_checkpoints_written_by_the_references_own_trainer_load_and_render(fname, case, role):
    """.pth files written by the REFERENCE's `Trainer.save_checkpoint` (utils.py:1405-1475; just_train_tea's for the teacher) in
    make_golden_step.py, read by this repo's loaders (pvd/checkpoint.py, the reference's strict=False rules): every key lands,
    the bookkeeping comes along, and the loaded model renders the image the reference's model rendered (inference rounds)."""
    from pvd.checkpoint import load_student_checkpoint, load_teacher_checkpoint
    path = os.path.join(os.path.dirname(__file__), "golden", fname)
    opt = config(case)
    torch.manual_seed(123)  # (different initial weights: everything must come from the file)
    mt = opt.model_type if role == "stu" else opt.teacher_type
    net = make_model(oracle_ops(), opt, mt, role == "tea", torch.device("cpu"))
    if role == "stu":
        missing, unexpected = load_student_checkpoint(net, ckpt_teacher=None, ckpt_student=path)
    else:
        missing, unexpected = load_teacher_checkpoint(net, path)
    assert not missing and not unexpected, (missing, unexpected)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert ck["epoch"] == 7 and ck["global_step"] == 4321 and ("resolution" in ck) == (mt == "vm")
    assert net.mean_count == ck["mean_count"] == int(G["mean_count"]) and net.mean_density == pytest.approx(0.125)
    np.testing.assert_array_equal(net.density_bitfield.numpy(), G["bitfield"])  # the occupancy grid travels in the state-dict
    net.note_occupancy_changed()
    rays_o, rays_d = rays_of(case)
    net.eval()
    with torch.no_grad():
        res = net.render(rays_o, rays_d, staged=False, bg_color=None, perturb=False, dt_gamma=0, max_steps=int(G["max_steps"]))
    ref = G["%s__eval_%s_image" % (case, role)]
    np.testing.assert_allclose(res["image"].numpy().reshape(ref.shape), ref, rtol=0, atol=3e-6)


def test_reverse_direction_record_the_reference_loader_took_this_repos_checkpoint():
    """Recorded by make_golden_step.py (it can only run where the reference is): a VM student of THIS repo, resampled to a
    non-cubic resolution and saved by pvd/checkpoint.py, was read by the reference's own `Trainer.load_student_checkpoint`
    (utils.py:1529-1556) into the reference's `NeRFNetwork` without a missing / unexpected key, and the two models then rendered
    the same image."""
    logs = [str(m) for m in G["reverse__logs"]]
    assert logs == ["[INFO] loaded student model."], logs
    np.testing.assert_array_equal(G["reverse__image_reference_model"].reshape(-1), G["reverse__image_this_repo"].reshape(-1))
    assert G["reverse__image_this_repo"].std() > 0.05
