#!/usr/bin/env python3
"""Generate tests/golden/reference_step.npz: the REFERENCE's own distillation step -- `Trainer.train_step`
(distill_mutual/utils.py:954-1189) driving `NeRFRenderer.run_cuda` (distill_mutual/renderer.py:319-448) of two of its
`NeRFNetwork`s (hash teacher, VM student) -- run HERE on the CPU, in its three stages, on fixed rays / weights / occupancy.

Everything that is Python / torch in the reference on that path is the reference's own code and arithmetic in these numbers:
the stage gating, which model marches and which inherits, `sigmas * density_scale`, the background mix, the four loss terms
with their rates and the 0.995 decay, the VM L1 term, and autograd through all of it.  The native operators underneath
(march_rays_train, composite_rays_train, grid_encode, sh_encode) are the CPU oracle standing in for the reference's CUDA
extensions here on the CPU (their kernels only run on a GPU: oracle/build_ref.py + make_golden_ref_kernels.py pin those), exactly as in make_golden.py, and `Tensor.cuda()` is
made the identity for the run (the wrappers call it; there is no GPU here) -- so this pins the repo's renderer / trainer
restatement, not kernel arithmetic.

Run in the build container only:   PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_step.py
Only data (inputs + what the reference computed) is written; no reference source is copied."""
import os
import sys
import types
from unittest.mock import MagicMock

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
for p in (REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)

import numpy as np
import torch

import oracle_backend as ob  # the CPU oracle dressed as the three _backend modules

# this repo's own stack (for the reverse-direction checkpoint check below) is imported FIRST and bound to this repo's
# `gridencoder` / `shencoder` / `raymarching` packages; those names are then handed over to the reference's packages
from oracle_ops import oracle_ops as _our_oracle_ops  # noqa: E402
from pvd.checkpoint import save_checkpoint as our_save_checkpoint, upsample_vm as our_upsample_vm  # noqa: E402
from pvd.config import PVDConfig as OurConfig  # noqa: E402
from pvd.workload import make_model as our_make_model  # noqa: E402

OUR_OPS = _our_oracle_ops()

for name, be in (("_raymarching", ob.raymarching_backend), ("_gridencoder", ob.gridencoder_backend), ("_shencoder", ob.shencoder_backend)):
    m = types.ModuleType(name)
    m.__dict__.update(be.__dict__)
    sys.modules[name] = m
for name in ("cv2", "trimesh", "mcubes", "lpips", "tensorboardX", "torch_ema", "imageio", "IPython", "torch_efficient_distloss"):
    sys.modules[name] = MagicMock()
for k in list(sys.modules):
    if k.split(".")[0] in ("gridencoder", "shencoder", "raymarching"):
        sys.modules.pop(k)
sys.path.insert(0, REF)
from distill_mutual.network import NeRFNetwork as RefNet  # noqa: E402
from distill_mutual import utils as ref_utils  # noqa: E402
import raymarching as ref_rm  # noqa: E402

assert ref_rm.__file__.startswith(REF) and ref_utils.__file__.startswith(REF)
# the reference's raymarching wrappers move their inputs with `.cuda()` (raymarching/raymarching.py:36, :221): there is no GPU
# in this container and the operators underneath are the CPU oracle, so `.cuda()` is the identity for this run
torch.Tensor.cuda = lambda self, *a, **k: self

N_RAYS, GRID, MAX_STEPS = 96, 16, 96
RATES = dict(loss_rate_rgb=1.0, loss_rate_fea_sc=0.02, loss_rate_color=0.03, loss_rate_sigma=0.05)
out = {}


def make_args(student, stu_first=True, dt_gamma=0, teacher="hash"):
    a = dict(plenoxel_degree=3, plenoxel_res="[12,12,12]", PE=6, skip=2, nerf_layer_num=5, nerf_layer_wide=32, resolution0=12,
             sigma_clip_min=-2, sigma_clip_max=7, global_step=0,
             stage_iters={"stage1": -1 if "tensors" in (student, teacher) else 2000, "stage2": 5000},  # no feature head: main_distill_mutual.py:243-246
             enable_edit_plenoxel=False, render_stu_first=stu_first, loss_type="normL2", l1_reg_weight=1e-3, model_type=student,
             dt_gamma=dt_gamma, max_steps=MAX_STEPS, **RATES)
    return types.SimpleNamespace(**a)


def occupancy():
    """128-cell-wide would be 2 MB; a 16^3 grid: a ball of radius 0.62 plus a slab, in Morton order, packed to bits."""
    import oracle
    c = (np.arange(GRID) + 0.5) / GRID * 2 - 1
    X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
    occ = (X ** 2 + Y ** 2 + Z ** 2 < 0.62 ** 2) | ((np.abs(Z + 0.7) < 0.1) & (np.abs(X) < 0.8))
    idx = np.stack(np.nonzero(occ), axis=1).astype(np.int32)
    mort = oracle.morton3D(idx) if hasattr(oracle, "morton3D") else None
    if mort is None:
        def part(v):
            v = v.astype(np.uint32) & 0x3ff
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            v = (v | (v << 2)) & 0x09249249
            return v
        mort = part(idx[:, 0]) | (part(idx[:, 1]) << 1) | (part(idx[:, 2]) << 2)
    bits = np.zeros(GRID ** 3, dtype=np.uint8)
    bits[np.asarray(mort, dtype=np.int64)] = 1
    return np.packbits(bits.reshape(-1, 8)[:, ::-1], axis=1).reshape(-1)  # bit k of byte j = cell 8 j + k


def occupancy2():
    """Two cascades (bound 2): the ball of `occupancy()` in the inner level, a thick shell 1.1 < |x| < 1.7 in the outer one."""
    inner = np.unpackbits(occupancy().reshape(-1, 1), axis=1)[:, ::-1].reshape(-1)
    c = ((np.arange(GRID) + 0.5) / GRID * 2 - 1) * 2
    X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
    r = np.sqrt(X ** 2 + Y ** 2 + Z ** 2)
    occ = (r > 1.1) & (r < 1.7) & (Z > -0.5)
    idx = np.stack(np.nonzero(occ), axis=1).astype(np.uint32)

    def part(v):
        v = v & 0x3ff
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    mort = part(idx[:, 0]) | (part(idx[:, 1]) << 1) | (part(idx[:, 2]) << 2)
    outer = np.zeros(GRID ** 3, dtype=np.uint8)
    outer[mort.astype(np.int64)] = 1
    bits = np.concatenate([inner, outer])
    return np.packbits(bits.reshape(-1, 8)[:, ::-1], axis=1).reshape(-1)


def save_table_grad(out, pre, name, g):
    rows = g.abs().sum(1).nonzero().squeeze(1)
    sub = rows[::7]
    out[pre + "grad_rows__" + name] = sub.numpy().astype(np.int32)
    out[pre + "grad_vals__" + name] = g[sub].numpy().copy()
    out[pre + "grad_nrows__" + name] = np.int64(rows.numel())
    out[pre + "grad_colsum__" + name] = g.double().sum(0).numpy()
    out[pre + "grad_abssum__" + name] = np.float64(g.double().abs().sum().item())


def rays(rs):
    o = rs.standard_normal((N_RAYS, 3))
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * rs.uniform(1.6, 2.4, size=(N_RAYS, 1))
    target = rs.uniform(-0.7, 0.7, size=(N_RAYS, 3))
    target[:6] = o[:6] * 1.5 + rs.standard_normal((6, 3))  # a few rays that miss the box
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o.astype(np.float32)[None], d.astype(np.float32)[None]


def build(mt, args, is_teacher, seed, bound=1):
    torch.manual_seed(seed)
    net = RefNet(encoding="hashgrid", bound=bound, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1,
                 grid_size=GRID, model_type=mt, args=args, is_teacher=is_teacher)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if "embeddings" in n:
                torch.manual_seed(777)
                p.copy_((torch.rand(p.shape) - 0.5) * 0.6)  # 42 MB: regenerated from this seed by the test, not stored
            elif p.dim() >= 2:
                p.mul_(3.0 if mt == "vm" and p.dim() == 4 else 1.6)
    return net


RefTrainer = ref_utils.Trainer


def tea_trainer():
    from just_train_tea import utils as tu
    return tu.Trainer


rs = np.random.RandomState(5)
ro, rd = rays(rs)
images = rs.uniform(0, 1, size=(1, N_RAYS, 4)).astype(np.float32)  # 4 channels: the step draws a random background
bitfield = torch.from_numpy(occupancy())
out.update(rays_o=ro, rays_d=rd, images=images, bitfield=bitfield.numpy(), mean_count=np.int64(3000), grid_size=np.int64(GRID),
           max_steps=np.int64(MAX_STEPS), l1_reg_weight=np.float64(1e-3))
for k, v in RATES.items():
    out[k] = np.float64(v)
data = dict(rays_o=torch.from_numpy(ro), rays_d=torch.from_numpy(rd), images=torch.from_numpy(images))

# (name, teacher, student, student renders first, stages)
CASES = [("hash_vm", "hash", "vm", True, (1, 2, 3), 1, 0),
         ("hash_vm_teafirst", "hash", "vm", False, (3,), 1, 0),      # renderer.py:392-411: the teacher marches, the student inherits
         ("mlp_tensors", "mlp", "tensors", True, (2, 3), 1, 0),       # configs[3]; stage 1 does not exist without a feature vector
         ("hash_hash", "hash", "hash", True, (1, 3), 1, 0),           # configs[4]
         ("hash_hash_b2", "hash", "hash", True, (3,), 2, 1 / 256),    # configs[4] as on Tanks&Temples: two cascades, growing step
         ("hash_mlp", "hash", "mlp", True, (1, 3), 1, 0),             # a NeRF-MLP student (gradients through FreqEncoder + trunk)
         ("vm_tensors", "vm", "tensors", True, (3,), 1, 0),           # a VM teacher
         ("tensors_vm", "tensors", "vm", True, (3,), 1, 0)]           # a Plenoxel teacher (no feature vector on the teacher's side)
GSTEP = {1: 100, 2: 3000, 3: 9000}
out["cases"] = np.array([c[0] for c in CASES])
bitfield2 = torch.from_numpy(occupancy2())
out["bitfield2"] = bitfield2.numpy()
ro2, rd2 = rays(np.random.RandomState(6))
ro2 = (ro2 * 1.6).astype(np.float32)  # cameras outside the outer shell
out.update(rays_o2=ro2, rays_d2=rd2)
data2 = dict(rays_o=torch.from_numpy(ro2), rays_d=torch.from_numpy(rd2), images=torch.from_numpy(images))
for case, tea_type, stu_type, stu_first, stages, bound, dt_gamma in CASES:
    args = make_args(stu_type, stu_first, dt_gamma, tea_type)
    tea = build(tea_type, args, True, 11, bound)
    stu = build(stu_type, args, False, 12, bound)
    bf = bitfield if bound == 1 else bitfield2
    data_c = data if bound == 1 else data2
    assert bf.numel() == stu.density_bitfield.numel(), (bf.numel(), stu.density_bitfield.numel())
    for net in (tea, stu):
        net.density_bitfield.copy_(bf)
        net.mean_count = 3000
        net.train()  # (the reference trainer keeps both models in train mode during train_one_epoch; run_cuda branches on it)
    out[case + "__cfg"] = np.array([tea_type, stu_type, str(int(stu_first)), str(bound), repr(float(dt_gamma))])
    out[case + "__stages"] = np.array(stages)
    for role, net in (("tea", tea), ("stu", stu)):
        keys = []
        for k, v in net.state_dict().items():
            keys.append(k)
            if "embeddings" not in k:
                out["%s__%s_sd__%s" % (case, role, k)] = v.detach().numpy().copy()
        out["%s__%s_keys" % (case, role)] = np.array(keys)
    me = types.SimpleNamespace(model_stu=stu, model_tea=tea, model=stu, opt=args, error_map=None, criterion=torch.nn.MSELoss(reduction="none"))
    me.get_loss = lambda pred, gt, me=me: RefTrainer.get_loss(me, pred, gt)
    for stage in stages:
        args.global_step = GSTEP[stage]
        rate0 = args.loss_rate_fea_sc
        for p in stu.parameters():
            p.grad = None
        torch.manual_seed(1000 + stage)
        res = RefTrainer.train_step(me, data_c)
        loss = res[2]
        loss.backward()
        pre = "%s__s%d__" % (case, stage)
        out[pre + "global_step"] = np.int64(GSTEP[stage])
        out[pre + "seed"] = np.int64(1000 + stage)
        out[pre + "fea_rate_before"] = np.float64(rate0)
        out[pre + "fea_rate_after"] = np.float64(args.loss_rate_fea_sc)
        out[pre + "loss"] = np.float64(loss.item())
        out[pre + "parts"] = np.array([float(x) for x in res[3:]], dtype=np.float64)  # rgb (mse, shown), fea, color, sigma
        if res[0] is not None:
            out[pre + "pred_stu"] = res[0].detach().numpy().copy()
            out[pre + "pred_tea"] = res[1].detach().numpy().copy()
        marcher = stu if stu_first else tea
        out[pre + "samples"] = marcher.step_counter[(marcher.local_step - 1) % 16].numpy().copy()
        for n, p in stu.named_parameters():
            g = (p.grad if p.grad is not None else torch.zeros_like(p)).detach()
            if "embeddings" in n:  # 42 MB of mostly zeros: every 7th non-zero row, the column sums and the non-zero row count
                save_table_grad(out, pre, n, g)
            else:
                out[pre + "grad__" + n] = g.numpy().copy()
        print(case, "stage", stage, "loss", loss.item(), "parts", res[3:], "samples", out[pre + "samples"], "fea rate", rate0, "->", args.loss_rate_fea_sc)
    if case in ("hash_vm", "mlp_tensors", "hash_hash_b2"):
        # ---- the inference branch of run_cuda (renderer.py:450-543: rounds of march_rays / composite_rays / compact_rays over the
        # alive rays, n_step = clamp(N // n_alive, 1, 8)) of both models, white background, no perturbation
        for role, net in (("tea", tea), ("stu", stu)):
            net.eval()
            with torch.no_grad():
                res = net.render(data_c["rays_o"], data_c["rays_d"], staged=False, bg_color=None, perturb=False, **vars(args))
            out["%s__eval_%s_image" % (case, role)] = res["image"].numpy().copy()
            out["%s__eval_%s_depth" % (case, role)] = res["depth"].numpy().copy()
            net.train()
            print(case, "eval", role, "image mean", float(res["image"].mean()), "depth mean", float(res["depth"].mean()))
    if case in ("hash_vm", "mlp_tensors"):
        # ---- checkpoints written by the reference's own Trainer.save_checkpoint (utils.py:1405-1475 / just_train_tea's): the VM
        # student of hash->vm as the distillation trainer saves it, the mlp teacher of mlp->tensors as the teacher trainer does
        import tempfile
        for trainer_cls, role, net, fname in ((RefTrainer, "stu", stu, "reference_ckpt_vm_student.pth"), (tea_trainer(), "tea", tea, "reference_ckpt_mlp_teacher.pth")):
            if (case, role) not in (("hash_vm", "stu"), ("mlp_tensors", "tea")):
                continue
            net.mean_density = 0.125
            with tempfile.TemporaryDirectory() as d:
                saver = types.SimpleNamespace(name="ngp", epoch=7, global_step=4321, opt=types.SimpleNamespace(model_type=net.model_type),
                                              stats={"loss": [0.5], "valid_loss": [], "results": [], "checkpoints": [], "best_result": None},
                                              model_stu=net, ckpt_path=d, max_keep_ckpt=2, ema=None)
                trainer_cls.save_checkpoint(saver, name="golden")
                import shutil
                shutil.copy(os.path.join(d, "golden.pth"), os.path.join(HERE, fname))
            print("checkpoint", fname, os.path.getsize(os.path.join(HERE, fname)), "bytes")
    if case == "hash_vm":
        # ---- VM utilities of the reference network: density_loss (network.py:549-558), upsample_model (:560-587), and the
        # optimizer's parameter groups of every model type (get_params, :646-700) as (lr, parameter names)
        out["vm__density_loss"] = np.float64(stu.density_loss().item())
        stu.upsample_model([20, 16, 24])
        out["vm__upsampled_to"] = np.array([20, 16, 24])
        for k, v in stu.state_dict().items():
            if "_mat." in k or "_vec." in k:
                out["vm__up__" + k] = v.detach().numpy().copy()
        out["vm__density_loss_up"] = np.float64(stu.density_loss().item())

for mt in ("hash", "mlp", "vm", "tensors"):
    net = build(mt, make_args(mt), False, 5)
    names = {id(p): n for n, p in net.named_parameters()}
    groups = []
    for g in net.get_params(0.02):
        groups.append("%r|%s" % (float(g["lr"]), ",".join(names[id(p)] for p in g["params"])))
    out["groups__" + mt] = np.array(groups)
    print("get_params", mt, [g.split("|")[0] for g in groups])

# ---- the reference's teacher-training step (just_train_tea/utils.py:746-846 over just_train_tea/renderer.py's run_cuda): one
# model, MSE against alpha-composited ground-truth pixels on a random background, plus the VM L1 term for a VM model
from just_train_tea.network import NeRFNetwork as TeaNet  # noqa: E402
from just_train_tea import utils as tea_utils  # noqa: E402

assert tea_utils.__file__.startswith(REF)
out["teacher_cases"] = np.array(["teacher_hash", "teacher_vm"])
for case, mt in (("teacher_hash", "hash"), ("teacher_vm", "vm")):
    args = make_args(mt)
    args.just_train_a_model, args.color_space = True, "srgb"
    torch.manual_seed(31)
    net = TeaNet(encoding="hashgrid", bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1,
                 grid_size=GRID, model_type=mt, args=args, is_teacher=False)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if "embeddings" in n:
                torch.manual_seed(777)
                p.copy_((torch.rand(p.shape) - 0.5) * 0.6)
            elif p.dim() >= 2:
                p.mul_(3.0 if mt == "vm" and p.dim() == 4 else 1.6)
    net.density_bitfield.copy_(bitfield)
    net.mean_count = 3000
    net.train()
    keys = []
    for k, v in net.state_dict().items():
        keys.append(k)
        if "embeddings" not in k:
            out["%s__sd__%s" % (case, k)] = v.detach().numpy().copy()
    out[case + "__keys"] = np.array(keys)
    me = types.SimpleNamespace(model_stu=net, model_tea=None, model=net, opt=args, criterion=torch.nn.MSELoss(reduction="none"))
    torch.manual_seed(2000)
    loss, pred, gt = tea_utils.Trainer.train_step(me, dict(rays_o=torch.from_numpy(ro), rays_d=torch.from_numpy(rd), images=torch.from_numpy(images.copy())))
    loss.backward()
    pre = case + "__"
    out[pre + "seed"] = np.int64(2000)
    out[pre + "loss"] = np.float64(loss.item())
    out[pre + "pred"], out[pre + "gt"] = pred.detach().numpy().copy(), gt.detach().numpy().copy()
    out[pre + "samples"] = net.step_counter[(net.local_step - 1) % 16].numpy().copy()
    for n, p in net.named_parameters():
        g = (p.grad if p.grad is not None else torch.zeros_like(p)).detach()
        if "embeddings" in n:
            save_table_grad(out, pre, n, g)
        else:
            out[pre + "grad__" + n] = g.numpy().copy()
    print(case, "loss", loss.item(), "samples", out[pre + "samples"])

# ---- the other direction: a checkpoint written by THIS repo (pvd/checkpoint.py, a VM student resampled to a non-cubic
# resolution, channels-last tables in memory) read by the reference's own Trainer.load_student_checkpoint (utils.py:1529-1556) into
# the reference's NeRFNetwork: no missing / unexpected key, and the reference's model then renders what this repo's model renders
def reverse_direction():
    import tempfile
    save_checkpoint, upsample_vm, make_model = our_save_checkpoint, our_upsample_vm, our_make_model
    opt = OurConfig(model_type="vm", teacher_type="hash", PE=6, skip=2, nerf_layer_num=5, nerf_layer_wide=32, resolution0=12,
                    plenoxel_res="[12,12,12]", grid_size=GRID, density_thresh=10.0, fp16=False, max_steps=MAX_STEPS)
    torch.manual_seed(77)
    mine = make_model(OUR_OPS, opt, "vm", False, torch.device("cpu"))
    with torch.no_grad():
        for n, p in mine.named_parameters():
            if p.dim() >= 2:
                p.mul_(3.0 if p.dim() == 4 else 1.6)
    upsample_vm(mine, [14, 12, 16])
    mine.density_bitfield.copy_(bitfield)
    mine.mean_count, mine.mean_density = 2816, 0.3
    mine.note_occupancy_changed()
    logs = []
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "mine.pth")
        save_checkpoint(path, mine, epoch=3, global_step=999)
        a = make_args("vm")
        ref = build("vm", a, False, 99)
        loader = types.SimpleNamespace(opt=types.SimpleNamespace(ckpt_student=path, ckpt_teacher=None, model_type="vm"), model_stu=ref,
                                       device="cpu", ema=None, log=lambda *m, **k: logs.append(" ".join(str(x) for x in m)))
        RefTrainer.load_student_checkpoint(loader)
    assert not any("WARN" in m for m in logs), logs
    assert ref.mean_count == 2816 and list(ref.resolution) == [14, 12, 16], (ref.mean_count, ref.resolution)
    ref.eval(), mine.eval()
    with torch.no_grad():
        r = ref.render(data["rays_o"], data["rays_d"], staged=False, bg_color=None, perturb=False, **vars(a))["image"]
        m = mine.render(data["rays_o"], data["rays_d"], staged=False, bg_color=None, perturb=False, dt_gamma=0, max_steps=MAX_STEPS)["image"]
    err = float((r - m.view_as(r)).abs().max())
    assert err <= 3e-6, err
    out["reverse__logs"] = np.array(logs)
    out["reverse__image_reference_model"], out["reverse__image_this_repo"] = r.numpy().copy(), m.numpy().copy()
    print("reverse direction: reference loader took this repo's checkpoint:", logs, "max |image difference|", err)


reverse_direction()

# ---- configs[0]: the fixed-step sampler `run` (just_train_tea/renderer.py, the non-cuda_ray branch of render) of an `mlp` model:
# uniform steps between the box intersections, perturbation, importance resampling through the reference's own sample_pdf, sort /
# gather, alpha compositing by cumprod, depth, background mix -- all the reference's code.  Its `color()` asserts out before doing
# anything (network.py:515-516), so for this run the masked colour query is the model's own forward on the selected rows (zeros
# elsewhere), which is what the dead body computes.
def masked_color(self, x, d, mask=None, **kwargs):
    rgbs = torch.zeros(x.shape[0], 3, dtype=x.dtype, device=x.device)
    if mask is None:
        return self.forward(x, d)[1]
    if mask.any():
        rgbs[mask] = self.forward(x[mask], d[mask])[1].to(rgbs.dtype)
    return rgbs


args = make_args("mlp")
args.just_train_a_model, args.color_space = True, "srgb"
torch.manual_seed(41)
net = TeaNet(encoding="hashgrid", bound=1, cuda_ray=False, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1,
             grid_size=GRID, model_type="mlp", args=args, is_teacher=False)
with torch.no_grad():
    for n, p in net.named_parameters():
        if p.dim() >= 2:
            p.mul_(1.6)
        if n == "sigma_net.1.weight":
            p[0].add_(0.35)  # denser medium: compositing weights above the 1e-4 colour threshold on most rays
net.color = types.MethodType(masked_color, net)
keys = []
for k, v in net.state_dict().items():
    keys.append(k)
    out["run_mlp__sd__" + k] = v.detach().numpy().copy()
out["run_mlp__keys"] = np.array(keys)
rs3 = np.random.RandomState(9)
g_img = rs3.standard_normal((1, N_RAYS, 3)).astype(np.float32)
g_dep = rs3.standard_normal((1, N_RAYS)).astype(np.float32)
out.update(run_mlp__g_image=g_img, run_mlp__g_depth=g_dep, run_mlp__num_steps=np.int64(24), run_mlp__upsample_steps=np.int64(12))
for mode, perturb, seed in (("train", True, 3000), ("eval", False, 3001)):
    net.train(mode == "train")
    for p in net.parameters():
        p.grad = None
    torch.manual_seed(seed)
    res = net.render(torch.from_numpy(ro), torch.from_numpy(rd), staged=False, bg_color=None, perturb=perturb, num_steps=24, upsample_steps=12)
    pre = "run_mlp__%s__" % mode
    out[pre + "seed"] = np.int64(seed)
    out[pre + "image"], out[pre + "depth"] = res["image"].detach().numpy().copy(), res["depth"].detach().numpy().copy()
    if mode == "train":
        (res["image"] * torch.from_numpy(g_img)).sum().backward()  # (depth is 0/0 on the rays that miss the box: left out of the objective)
        for n, p in net.named_parameters():
            out[pre + "grad__" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy().copy()
    print("run()", mode, "image mean", float(res["image"].mean()), "min", float(res["image"].min()), "depth nan", int(torch.isnan(res["depth"]).sum()))

# ---- occupancy-grid maintenance: the reference's own mark_untrained_grid / update_extra_state (renderer.py:561-775) of a hash
# model, one and two cascades: full sweeps (iter_density < 16), partial updates (uniform + occupied cells), the EMA maximum, the
# mean / threshold, packbits, and the refresh of mean_count from the step counter -- torch's generator seeded before each call
for bound in (1, 2):
    args = make_args("hash")
    torch.manual_seed(21)
    net = RefNet(encoding="hashgrid", bound=bound, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1,
                 grid_size=GRID, model_type="hash", args=args, is_teacher=True)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if "embeddings" in n:
                torch.manual_seed(777)
                p.copy_((torch.rand(p.shape) - 0.5) * 0.6)
            elif p.dim() >= 2:
                p.mul_(1.6)
    pre = "occ_b%d__" % bound
    keys = []
    for k, v in net.state_dict().items():
        keys.append(k)
        if "embeddings" not in k:
            out[pre + "sd__" + k] = v.detach().numpy().copy()
    out[pre + "keys"] = np.array(keys)
    poses = np.stack([ref_utils.nerf_matrix_to_ngp(ref_utils.pose_spherical(th, ph, 4.0), scale=0.8)
                      for th, ph in ((30.0, -20.0), (-100.0, -35.0), (170.0, -5.0))]).astype(np.float32)
    intrinsic = np.array([1111.1, 1111.1, 400.0, 400.0])
    out[pre + "poses"], out[pre + "intrinsic"] = poses, intrinsic
    net.mark_untrained_grid(poses, intrinsic)
    out[pre + "marked"] = net.density_grid.numpy().copy()
    calls = []
    for i, (it, counts) in enumerate(((0, None), (1, [(700, 96), (900, 96), (650, 90)]), (16, None), (17, [(1200, 96)] * 16))):
        net.iter_density = it
        if counts is not None:  # the marcher filled these slots of the step counter since the last update
            net.step_counter.zero_()
            for j, c in enumerate(counts):
                net.step_counter[j, 0], net.step_counter[j, 1] = c
            net.local_step = len(counts)
        torch.manual_seed(500 + i)
        net.update_extra_state()
        c = pre + "u%d__" % i
        out[c + "iter_density"] = np.int64(it)
        out[c + "seed"] = np.int64(500 + i)
        out[c + "counts"] = np.array(counts if counts is not None else np.zeros((0, 2)), dtype=np.int32).reshape(-1, 2)
        out[c + "grid"] = net.density_grid.numpy().copy()
        out[c + "bitfield"] = net.density_bitfield.numpy().copy()
        out[c + "mean_density"] = np.float64(net.mean_density)
        out[c + "mean_count"] = np.int64(net.mean_count)
        calls.append(i)
        print("occupancy bound", bound, "call", i, "iter", it, "mean density", net.mean_density, "occupied", int((net.density_grid > 0).sum()),
              "bits", int(np.unpackbits(net.density_bitfield.numpy()).sum()), "mean_count", net.mean_count)
    out[pre + "calls"] = np.array(calls)

np.savez_compressed(os.path.join(HERE, "reference_step.npz"), **out)
print("wrote reference_step.npz with", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "reference_step.npz")), "bytes")
