#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the REFERENCE's Python (read-only, /root/reference).

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden.py

What can be pinned this way: everything in the reference that runs on CPU
without its CUDA extensions --
  * GridEncoder's level-offset table / parameter count / per_level_scale   (gridencoder/grid.py:142-205)
  * trunc_exp forward + backward                                           (tools/activation.py)
  * FreqEncoder                                                            (tools/encoding.py:6-49)
  * NeRFNetwork parameter names and shapes for mlp / hash / vm             (distill_mutual/network.py)
  * pose_spherical / nerf_matrix_to_ngp / get_rays                         (distill_mutual/utils.py:53-98, 324-404)
  * the reference's GridEncoder / SHEncoder *Python wrappers* (input mapping, [L,B,C]->[B,L*C]
    permute, backward reshapes) driven with the CPU oracle standing in for `_gridencoder` /
    `_shencoder`: pins this repo's wrappers against the reference's wrapper logic (the kernel
    arithmetic in those fixtures is the oracle's, not the reference's).
  * the reference's `composite_rays_train` autograd wrapper (raymarching/raymarching.py:292-357:
    allocation, saved tensors, which gradients exist) the same way, forward + backward.
Kernel arithmetic is NOT pinned by THESE fixtures; since round 6 the reference's own raymarching / SH kernels are built for gfx950
(oracle/build_ref.py) and pin it through tests/golden/make_golden_ref_kernels.py -> reference_kernels.npz (the grid encoder's source
does not build on HIP: "parity unpinned by the reference", see oracle/pvd_oracle.h).
Only data (inputs + expected outputs) is written; no reference source is copied.
"""
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

# stand-ins for the reference's native modules and for python deps missing from this image
for name, be in (("_raymarching", ob.raymarching_backend), ("_gridencoder", ob.gridencoder_backend), ("_shencoder", ob.shencoder_backend)):
    m = types.ModuleType(name)
    m.__dict__.update(be.__dict__)
    sys.modules[name] = m
for name in ("cv2", "trimesh", "mcubes", "lpips", "tensorboardX", "torch_ema", "imageio", "IPython", "torch_efficient_distloss"):
    sys.modules[name] = MagicMock()

# import the reference under an alias package path so its `gridencoder` etc. do not shadow ours
our = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split(".")[0] in ("gridencoder", "shencoder", "raymarching")}
sys.path.insert(0, REF)
import gridencoder as ref_grid  # noqa: E402
import shencoder as ref_sh  # noqa: E402
from tools.activation import trunc_exp as ref_trunc_exp  # noqa: E402
from tools.encoding import FreqEncoder as RefFreq  # noqa: E402

out = {}

# ---- GridEncoder tables
cfgs = [
    dict(input_dim=3, num_levels=14, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048),
    dict(input_dim=3, num_levels=14, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=4096),
    dict(input_dim=2, num_levels=4, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048),
    dict(input_dim=3, num_levels=16, level_dim=4, base_resolution=16, log2_hashmap_size=15, per_level_scale=2, align_corners=True),
    dict(input_dim=3, num_levels=8, level_dim=1, base_resolution=8, log2_hashmap_size=12, per_level_scale=1.5, gridtype="tiled"),
]
for i, c in enumerate(cfgs):
    e = ref_grid.GridEncoder(**c)
    out["grid%d_offsets" % i] = e.offsets.numpy().copy()
    out["grid%d_pls" % i] = np.float64(e.per_level_scale)
    out["grid%d_shape" % i] = np.array(e.embeddings.shape)
out["grid_cfgs"] = np.array([repr(c) for c in cfgs])

# ---- reference wrapper logic over the oracle backend (forward + backward through autograd)
torch.manual_seed(0)
wcfg = dict(input_dim=3, num_levels=8, level_dim=2, base_resolution=8, log2_hashmap_size=12, desired_resolution=256)  # small table
enc = ref_grid.GridEncoder(**wcfg)
out['gw_cfg'] = np.array(repr(wcfg))
emb = (torch.rand_like(enc.embeddings) * 2 - 1) * 0.1
enc.embeddings.data.copy_(emb)
x = (torch.rand(257, 3) * 2 - 1) * 1.0
x[0] = torch.tensor([1.0, -1.0, 0.0]); x[1] = torch.tensor([1.5, 0.0, 0.0])
y = enc(x, bound=1)
g = torch.randn_like(y)
y.backward(g)
out.update(gw_emb=emb.numpy(), gw_x=x.numpy(), gw_y=y.detach().numpy(), gw_g=g.numpy(), gw_gemb=enc.embeddings.grad.numpy())
xb = x.clone() * 2
yb = enc(xb, bound=2)
out.update(gw_xb=xb.numpy(), gw_yb=yb.detach().numpy())

sh = ref_sh.SHEncoder(input_dim=3, degree=4)
d = torch.randn(129, 3); d = d / d.norm(dim=-1, keepdim=True)
d.requires_grad_(True)
ys = sh(d)
gs = torch.randn_like(ys)
ys.backward(gs)
out.update(sh_d=d.detach().numpy(), sh_y=ys.detach().numpy(), sh_g=gs.numpy(), sh_gd=d.grad.numpy())

# ---- reference composite_rays_train wrapper (raymarching/raymarching.py:292-357) over the oracle backend
import raymarching as ref_rm  # noqa: E402  (the reference's package: REF is first on sys.path)
assert ref_rm.__file__.startswith(REF)
rs = np.random.RandomState(7)
counts = rs.randint(0, 40, size=48).astype(np.int32)
counts[5] = 0      # a ray that produced no samples
counts[11] = 1
counts[20] = 97    # longer than one wavefront's 64 lanes
offs = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32)
M_used = int(counts.sum())
M = M_used + 9     # slack after the last ray (stays untouched)
rays_t = np.stack([np.arange(48, dtype=np.int32), offs, counts], axis=1)
rays_t[47, 1] = M - 3  # offset + count >= M: the overflow rule treats it as empty (raymarching.cu:525)
rays_t[47, 2] = 5
rays_t = rays_t[rs.permutation(48)]  # row order != ray id order (the reference's atomics give any order)
sig = torch.from_numpy(np.exp(rs.uniform(-2, 7, size=M)).astype(np.float32)).requires_grad_(True)
rgb = torch.from_numpy(rs.uniform(0, 1, size=(M, 3)).astype(np.float32)).requires_grad_(True)
dl = torch.from_numpy(np.stack([np.full(M, 2 * 3 ** 0.5 / 1024, dtype=np.float32),
                                rs.uniform(0.003, 0.05, size=M).astype(np.float32)], axis=1))
ws, dep, img = ref_rm.composite_rays_train(sig, rgb, dl, torch.from_numpy(rays_t))
g_ws = torch.from_numpy(rs.standard_normal(48).astype(np.float32))
g_dep = torch.from_numpy(rs.standard_normal(48).astype(np.float32))  # ignored by the reference's backward
g_img = torch.from_numpy(rs.standard_normal((48, 3)).astype(np.float32))
torch.autograd.backward([ws, dep, img], [g_ws, g_dep, g_img])
out.update(comp_sigmas=sig.detach().numpy(), comp_rgbs=rgb.detach().numpy(), comp_deltas=dl.numpy(), comp_rays=rays_t,
           comp_ws=ws.detach().numpy(), comp_depth=dep.detach().numpy(), comp_image=img.detach().numpy(),
           comp_g_ws=g_ws.numpy(), comp_g_depth=g_dep.numpy(), comp_g_image=g_img.numpy(),
           comp_g_sigmas=sig.grad.numpy(), comp_g_rgbs=rgb.grad.numpy())

# ---- trunc_exp
xs = torch.linspace(-20, 20, 401, requires_grad=True)
ys = ref_trunc_exp(xs)
gg = torch.linspace(-1, 2, 401)
ys.backward(gg)
out.update(te_x=xs.detach().numpy(), te_y=ys.detach().numpy(), te_g=gg.numpy(), te_gx=xs.grad.numpy())

# ---- FreqEncoder
for mr in (10, 2, 6):
    fe = RefFreq(input_dim=3, max_freq_log2=mr - 1, N_freqs=mr, log_sampling=True)
    xi = torch.linspace(-1, 1, 3 * 17).view(17, 3)
    out["freq%d_x" % mr] = xi.numpy()
    out["freq%d_y" % mr] = fe(xi).numpy()
    out["freq%d_dim" % mr] = np.int64(fe.output_dim)

# ---- network parameter names / shapes
sys.modules.pop("raymarching", None)
from distill_mutual.network import NeRFNetwork as RefNet  # noqa: E402
from distill_mutual import utils as ref_utils  # noqa: E402

args = types.SimpleNamespace(plenoxel_degree=3, plenoxel_res="[128,128,128]", PE=10, skip=3, nerf_layer_num=8, nerf_layer_wide=256,
                             resolution0=300, sigma_clip_min=-2, sigma_clip_max=7, global_step=0,
                             stage_iters={"stage1": 2000, "stage2": 5000}, enable_edit_plenoxel=False, render_stu_first=True)
for mt in ("hash", "mlp", "vm"):
    net = RefNet(encoding="hashgrid", bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1,
                 grid_size=128, model_type=mt, args=args, is_teacher=False)
    sd = net.state_dict()
    out["net_%s_keys" % mt] = np.array(sorted(sd.keys()))
    out["net_%s_shapes" % mt] = np.array([repr(tuple(sd[k].shape)) for k in sorted(sd.keys())])
    out["net_%s_dtypes" % mt] = np.array([str(sd[k].dtype) for k in sorted(sd.keys())])

# ---- cameras and rays
out["pose_sph"] = np.stack([ref_utils.pose_spherical(th, ph, r) for th, ph, r in ((30.0, -20.0, 4.0), (-170.0, -5.0, 4.0), (0.0, -89.0, 3.0))])
out["pose_ngp"] = np.stack([ref_utils.nerf_matrix_to_ngp(p, scale=0.8) for p in out["pose_sph"]])
poses = torch.from_numpy(out["pose_ngp"])
r = ref_utils.get_rays(poses, np.array([1111.1, 1111.1, 24.0, 20.0]), 40, 48, -1)
out.update(rays_o=r["rays_o"].numpy(), rays_d=r["rays_d"].numpy())
torch.manual_seed(3)
r = ref_utils.get_rays(poses[:1], np.array([1111.1, 1111.1, 400.0, 400.0]), 800, 800, 64)
out.update(rays_n_inds=r["inds"].numpy(), rays_n_o=r["rays_o"].numpy(), rays_n_d=r["rays_d"].numpy())

# ---- the reference's NeRFNetwork.forward / density run HERE on the CPU (distill_mutual/network.py:335-494): the torch parts
# of the path -- VM plane x line lookup (12 x F.grid_sample, :216-309), FreqEncoder + NeRF MLP, sigma_net / color_net /
# basis_mat heads, clamp, trunc_exp, sigmoid -- are the reference's own code and arithmetic, forward AND backward (torch
# autograd).  (`hash`: the table lookup inside it goes through the oracle standing in for _gridencoder; SH likewise.)
# Small instances so that the fixtures stay small: the state-dict, the inputs and what the reference computed.
def small_args(**kw):
    a = dict(plenoxel_degree=3, plenoxel_res="[128,128,128]", PE=6, skip=2, nerf_layer_num=5, nerf_layer_wide=32, resolution0=12,
             sigma_clip_min=-2, sigma_clip_max=7, global_step=10 ** 6, stage_iters={"stage1": 2000, "stage2": 5000},
             enable_edit_plenoxel=False, render_stu_first=True)
    a.update(kw)
    return types.SimpleNamespace(**a)


rs = np.random.RandomState(21)
xq = rs.uniform(-1, 1, size=(193, 3)).astype(np.float32)
xq[0] = (1.0, -1.0, 0.25); xq[1] = (0.0, 0.0, 0.0)  # box corner / centre
dq = rs.standard_normal((193, 3)).astype(np.float32)
dq /= np.linalg.norm(dq, axis=1, keepdims=True)
out.update(refnet_x=xq, refnet_d=dq)
for mt in ("vm", "mlp", "hash"):
    torch.manual_seed(100 + len(mt))
    a = small_args()
    net = RefNet(encoding="hashgrid", bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1,
                 grid_size=16, model_type=mt, args=a, is_teacher=False)
    with torch.no_grad():  # away from the initialisation: features that exercise the clamps and a non-flat colour head
        for n, p in net.named_parameters():
            if "embeddings" in n:
                torch.manual_seed(777)
                p.copy_((torch.rand(p.shape) - 0.5) * 0.6)  # 42 MB: regenerated from this seed by the test, not stored
            elif p.dim() >= 2:
                p.mul_(3.0 if mt == "vm" and p.dim() == 4 else 1.6)
    net.train()
    x, d = torch.from_numpy(xq), torch.from_numpy(dq)
    sigma, color = net(x, d)
    g_s = torch.from_numpy(rs.standard_normal(193).astype(np.float32))
    g_c = torch.from_numpy(rs.standard_normal((193, 3)).astype(np.float32))
    g_f = None
    fea = net.feature_sigma_color
    if fea is not None:
        g_f = torch.from_numpy(rs.standard_normal(tuple(fea.shape)).astype(np.float32)) * 0.1
    loss = (sigma * g_s).sum() + (color * g_c).sum() + (0 if g_f is None else (fea * g_f).sum())
    loss.backward()
    pre = "refnet_%s__" % mt
    out[pre + "sigma"] = sigma.detach().numpy()
    out[pre + "color"] = color.detach().numpy()
    out[pre + "feature_sigma_color"] = fea.detach().numpy()
    out[pre + "g_sigma"], out[pre + "g_color"], out[pre + "g_fea"] = g_s.numpy(), g_c.numpy(), g_f.numpy()
    with torch.no_grad():
        dens = net.density(x)
    out[pre + "density_sigma"] = dens["sigma"].detach().numpy()
    keys = []
    for k, v in net.state_dict().items():
        keys.append(k)
        if "embeddings" not in k:
            out[pre + "sd__" + k] = v.detach().numpy()
    out[pre + "keys"] = np.array(keys)
    for n, p in net.named_parameters():
        if "embeddings" in n:
            # the table gradient is 42 MB of mostly zeros: keep its non-zero rows
            rows = p.grad.abs().sum(1).nonzero().squeeze(1)
            out[pre + "grad_rows__" + n] = rows.numpy()
            out[pre + "grad_vals__" + n] = p.grad[rows].numpy()
        else:
            out[pre + "grad__" + n] = p.grad.detach().numpy()

# ---- get_rays with an error map (utils.py:357-381, --error_map): weighted draw of 128 x 128 cells + jitter inside the cell
torch.manual_seed(11)
emap = torch.rand(2, 128 * 128) ** 4  # uneven weights
emap[0, :4000] = 0.0                   # cells that can never be drawn
r = ref_utils.get_rays(poses[:2], np.array([1111.1, 1111.1, 400.0, 400.0]), 800, 800, 96, emap)
out.update(rays_e_map=emap.numpy(), rays_e_inds=r["inds"].numpy(), rays_e_inds_coarse=r["inds_coarse"].numpy(),
           rays_e_o=r["rays_o"].numpy(), rays_e_d=r["rays_d"].numpy())

np.savez_compressed(os.path.join(HERE, "reference_python.npz"), **out)
print("wrote", os.path.join(HERE, "reference_python.npz"), "with", len(out), "arrays")

# ---- the reference's head under autocast (network.py:413-437 hash, :344-381 vm): NeRFNetwork.forward run HERE under
# torch.autocast("cpu", dtype=torch.float16).  CPU autocast applies the same policy to the head's operations as CUDA autocast does
# (nn.Linear in f16 with fp32 accumulation and an f16 result; relu / clamp / sigmoid / cat by type promotion; grid_sample in fp32),
# so what the sigma_net / basis_mat / color_net chain produces for given f16 inputs is the reference's own code and arithmetic.
# Two things differ from the device the reference trains on and are therefore recorded as INPUTS of the head, not as its results:
# the encoders run through the oracle stand-ins in fp32 (their custom_fwd is a CUDA-autocast construct), and trunc_exp is not
# cast to fp32 on the CPU (same reason) -- the fixture's `sigma` is exp evaluated on the f16 feature, stored f16.
amp = {}
rs = np.random.RandomState(33)
xq = rs.uniform(-1, 1, size=(517, 3)).astype(np.float32)
dq = rs.standard_normal((517, 3)).astype(np.float32)
dq /= np.linalg.norm(dq, axis=1, keepdims=True)
amp.update(x=xq, d=dq)
for mt in ("vm", "hash"):
    torch.manual_seed(200 + len(mt))
    a = small_args()
    net = RefNet(encoding="hashgrid", bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1,
                 grid_size=16, model_type=mt, args=a, is_teacher=False)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if "embeddings" in n:
                torch.manual_seed(778)
                p.copy_((torch.rand(p.shape) - 0.5) * 0.6)
            elif p.dim() >= 2:  # features that reach both clamps, a colour head away from sigmoid(0)
                p.mul_(6.0 if mt == "vm" and p.dim() == 4 else 4.0)
    net.train()
    seen = {}
    if mt == "hash":
        net.encoder.register_forward_hook(lambda m, i, o: seen.__setitem__("x0", o.detach().clone()))
    else:
        net.basis_mat.register_forward_pre_hook(lambda m, i: seen.__setitem__("x0", i[0].detach().clone()))
        inner = net.get_sigma_feat
        net.get_sigma_feat = lambda xn: seen.setdefault("sigma_raw", inner(xn))
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
        sigma, color = net(torch.from_numpy(xq), torch.from_numpy(dq))
    pre = "amp_%s__" % mt
    assert color.dtype == torch.float16 and seen["x0"].dtype == torch.float32
    amp[pre + "x0"] = seen["x0"].numpy()                       # fp32, as it reaches the first Linear (which rounds it to f16)
    if mt == "vm":
        amp[pre + "sigma_raw"] = seen["sigma_raw"].detach().float().numpy()
    amp[pre + "sigma"] = sigma.float().numpy()
    amp[pre + "color"] = color.float().numpy()
    amp[pre + "feature_sigma_color"] = net.feature_sigma_color.float().numpy()
    amp[pre + "feature_dtype"] = np.array(str(net.feature_sigma_color.dtype))
    for k, v in net.state_dict().items():
        if "embeddings" not in k and ("sigma_net" in k or "color_net" in k or "basis_mat" in k):
            amp[pre + "sd__" + k] = v.detach().numpy()
np.savez_compressed(os.path.join(HERE, "reference_head_amp.npz"), **amp)
print("wrote", os.path.join(HERE, "reference_head_amp.npz"), "with", len(amp), "arrays")
