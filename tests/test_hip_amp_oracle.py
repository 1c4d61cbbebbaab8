"""The configuration bench.py TIMES -- fp16 AMP, the fused MFMA heads, the frozen hash teacher's lookup + head in one launch --
against the ORACLE directly (it used to be tied to the oracle through a chain: fp32 HIP == oracle, AMP fused
== AMP generic HIP, heads vs torch autocast on the GPU).  oracle/pvd_oracle.c: pvdo_head_forward_amp restates the head under
autocast (network.py:413-437, 344-381) and is pinned by the reference's own NeRFNetwork.forward under torch.autocast
(tests/test_oracle_head_amp.py); the f16 table lookup in front of it is the oracle's pvdo_grid_encode_forward (bit-exact with
the HIP lookup, tests/test_hip_parity.py).

Bar.  Both sides form the same exact f16 x f16 products and round every layer's fp32 sum to f16; the matrix cores add in another
order than the oracle's ascending k, so a sum that lands within rounding of an f16 tie comes out one f16 ulp apart and the next
layers see that.  Measured on the full-size sample set: > 97 % of all outputs bit-identical, the rest within a few f16 ulps --
the bars below are 4e-3 (1 + |feature|), 2e-3 on rgb (two ulps at 1.0), 8e-3 relative on sigma = exp(feature 0)."""
import numpy as np
import pytest
import torch

import oracle
from test_hip_head import _inputs, _model

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().float().cpu().numpy()


def _check(tag, sig, rgb, feat, sig_o, rgb_o, feat_o, min_equal):
    sig, rgb, feat = _np(sig), _np(rgb), _np(feat)
    assert np.isfinite(sig).all() and np.isfinite(rgb).all() and np.isfinite(feat).all()
    fd = np.abs(feat - feat_o)
    assert (fd <= 4e-3 * (1 + np.abs(feat_o))).all(), (tag, float(fd.max()))
    assert np.abs(rgb - rgb_o).max() <= 2e-3 and np.abs(rgb - rgb_o).mean() <= 1e-4, (tag, float(np.abs(rgb - rgb_o).max()))
    rel = np.abs(sig - sig_o) / (np.abs(sig_o) + 1e-6)
    assert rel.max() <= 8e-3 and rel.mean() <= 5e-4, (tag, float(rel.max()), float(rel.mean()))
    eq_f, eq_c = float((feat == feat_o).mean()), float((rgb == rgb_o).mean())
    print("%s: feature_sigma_color %.2f %% bit-identical (max |d| %.2e), rgb %.2f %% (max |d| %.2e), sigma max rel %.2e"
          % (tag, 100 * eq_f, fd.max(), 100 * eq_c, np.abs(rgb - rgb_o).max(), rel.max()))
    assert eq_f >= min_equal and eq_c >= min_equal, (tag, eq_f, eq_c)


def _hash_oracle(m, x, d):
    """encoder (grid.py:113-140 under autocast: f16 table, f16 output) + head, on the CPU oracle"""
    enc = m.encoder
    x01 = (_np(x) + m.bound) / (2 * m.bound)  # grid.py:129
    emb = enc.embeddings.detach().half().cpu().numpy()
    out, _ = oracle.grid_encode_forward(x01, emb, enc.offsets.cpu().numpy(), float(np.log2(enc.per_level_scale)), enc.base_resolution)
    x0 = np.ascontiguousarray(out.transpose(1, 0, 2)).reshape(x01.shape[0], -1)  # [L,B,C] -> [B, L*C] (grid.py:75)
    assert x0.dtype == np.float16 and x0.shape[1] == 28
    a = m.args
    W = [_np(w) for w in (m.sigma_net[0].weight, m.sigma_net[1].weight, m.color_net[0].weight, m.color_net[1].weight, m.color_net[2].weight)]
    return oracle.head_forward_amp(0, x0, None, _np(d), *W, clip_sigma_min=a.sigma_clip_min, clip_feat_min=a.sigma_clip_min, clip_max=a.sigma_clip_max)


@pytest.mark.parametrize("M,bound", [(4099, 1), (92928, 1), (30000, 2)])
def test_fused_hash_lookup_and_head_match_the_amp_oracle(M, bound):
    """k_hash_fwd_fused (the frozen teacher of the timed step; 92 928 rows = the size the roofline is quoted at) and the two-launch
    form (lookup, then k_head_fwd) against oracle lookup + oracle AMP head."""
    import fusedhead
    m = _model("hash").eval()
    m.bound = bound
    x, d = _inputs(M)
    x = x * bound
    x[-5:] = bound * 1.5  # outside the box: zero features on every level (gridencoder.cu:113-124)
    sig_o, rgb_o, feat_o = _hash_oracle(m, x, d)
    sig, rgb, feat = fusedhead.hash_head_infer(m, x, d)
    _check("fused lookup + head, M = %d, bound %d" % (M, bound), sig, rgb, feat, sig_o, rgb_o, feat_o, 0.97)


def test_vm_head_matches_the_amp_oracle_on_the_kernels_own_products():
    """The VM student's head (k_head_fwd<VM>): basis_mat + clamps + colour head on the f16 products the HIP lookup hands it (the
    lookup itself is compared in fp32 with grid_sample and the oracle-side trainer: tests/test_hip_vm.py, test_hip_fullsize.py)."""
    import fusedhead
    m = _model("vm").eval()
    x, d = _inputs(92928)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        sraw, prod = m.ops.vm_encode(x, m._aabb(), *m.sigma_mat, *m.sigma_vec, *m.color_mat, *m.color_vec)
    assert prod.dtype == torch.float16 and sraw.dtype == torch.float32
    sig, rgb, feat = fusedhead.vm_head_infer(m, sraw, prod, d)
    a = m.args
    W = [_np(w) for w in (m.basis_mat.weight, m.color_net[0].weight, m.color_net[1].weight, m.color_net[2].weight)]
    sig_o, rgb_o, feat_o = oracle.head_forward_amp(1, prod.cpu().numpy(), _np(sraw), _np(d), W[0], None, W[1], W[2], W[3],
                                                   clip_sigma_min=a.sigma_clip_min, clip_feat_min=a.sigma_clip_min, clip_max=a.sigma_clip_max)
    assert np.array_equal(_np(feat)[:, 0], feat_o[:, 0])  # the fp32 sigma feature, clamped in fp32
    _check("VM head, M = 92928", sig, rgb, feat, sig_o, rgb_o, feat_o, 0.97)


def _vm_tables_reference_layout(m):
    return [t.detach().float().contiguous().cpu().numpy() for t in (*m.sigma_mat, *m.sigma_vec, *m.color_mat, *m.color_vec)]


def test_vm_lookup_is_bit_exact_against_the_oracles_c_restatement():
    """k_vm_fwd against oracle/pvd_oracle.c: pvdo_vm_forward (network.py:216-309 restated tap by tap in grid_sample's order; pinned by
    the reference's own vm forward, tests/test_oracle_vm.py): the 144 plane x line products are the same floats -- in fp32 and, rounded
    once, in the f16 the AMP head reads -- and the sigma feature (a sum of 48 products whose association is the kernel's butterfly)
    agrees to fp32 summation noise.  300^2 tables (the bench's), 92 928 samples incl. the marcher's padding rows and points outside the box."""
    m = _model("vm").eval()
    x, d = _inputs(92928)
    x[100:200] *= 1.7  # outside [-1, 1]: zero padding
    res = (m.sigma_mat[0].shape[3], m.sigma_mat[0].shape[2], m.sigma_mat[1].shape[2])
    sig_o, prod_o = oracle.vm_forward(_np(x), m._aabb(), _vm_tables_reference_layout(m), res)
    with torch.no_grad():
        sraw32, prod32 = m.ops.vm_encode(x, m._aabb(), *m.sigma_mat, *m.sigma_vec, *m.color_mat, *m.color_vec)
        with torch.autocast("cuda", dtype=torch.float16):
            sraw16, prod16 = m.ops.vm_encode(x, m._aabb(), *m.sigma_mat, *m.sigma_vec, *m.color_mat, *m.color_vec)
    assert prod32.dtype == torch.float32 and prod16.dtype == torch.float16
    assert np.array_equal(_np(prod32), prod_o), float(np.abs(_np(prod32) - prod_o).max())
    assert np.array_equal(prod16.cpu().numpy(), prod_o.astype(np.float16))
    assert torch.equal(sraw32, sraw16)
    assert np.abs(_np(sraw32) - sig_o).max() <= 4e-6 * max(1.0, float(np.abs(sig_o).max())) and np.abs(prod_o).max() > 0.01


def test_config2_amp_student_forward_matches_the_amp_oracle_end_to_end():
    """The VM student of the timed step under AMP, no HIP value on the oracle side: oracle VM lookup (fp32) -> products rounded to
    f16 (basis_mat's autocast cast) -> oracle AMP head -> oracle compositor, against k_vm_fwd + k_head_fwd<VM> + the HIP compositor on
    the bench's 4096-ray batch.  The products are bit-identical (above); the sigma feature differs by fp32 summation noise, which the
    clamp / exp / compositing carry through: features within 4e-3 (1 + |f|), image within 1e-4 max / 1e-6 mean (measured 5.3e-6 / 1.9e-8, 99.91 % of the colour features bit-identical)."""
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import DistillWorkload
    torch.manual_seed(0)
    w = DistillWorkload(hip_ops(), torch.device("cuda:0"), PVDConfig(num_rays=4096), teacher_pretrain_steps=0, seed=0)
    stu = w.stu
    with torch.no_grad():  # a student away from its initialisation (after init the colour head sees ~0.01-sized features)
        for n, p in stu.named_parameters():
            if p.dim() == 4:
                p.mul_(2.5)
            elif p.dim() == 2:
                p.mul_(2.0)
    import pvd_hip
    pvd_hip.note_weights_changed(list(stu.parameters()))
    rays_o, rays_d, bg = w.next_batch()
    w.opt.global_step = w.trainer.global_step
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        out = stu.render(rays_o, rays_d, staged=False, bg_color=bg, perturb=True, force_all_rays=False)
    xyzs, dirs, deltas, rays = out["inherited_params"]
    n = int((rays[:, 1] + rays[:, 2]).max())
    assert n > 60000, n
    res = (stu.sigma_mat[0].shape[3], stu.sigma_mat[0].shape[2], stu.sigma_mat[1].shape[2])
    sig_raw, prod = oracle.vm_forward(_np(xyzs[:n]), stu._aabb(), _vm_tables_reference_layout(stu), res)
    a = stu.args
    W = [_np(t) for t in (stu.basis_mat.weight, stu.color_net[0].weight, stu.color_net[1].weight, stu.color_net[2].weight)]
    sig_o, rgb_o, feat_o = oracle.head_forward_amp(1, prod.astype(np.float16), sig_raw, _np(dirs[:n]), W[0], None, W[1], W[2], W[3],
                                                   clip_sigma_min=a.sigma_clip_min, clip_feat_min=a.sigma_clip_min, clip_max=a.sigma_clip_max)
    feat = _np(stu.feature_sigma_color[:n])
    fd = np.abs(feat - feat_o)
    assert (fd <= 4e-3 * (1 + np.abs(feat_o))).all() and float((feat[:, 1:] == feat_o[:, 1:]).mean()) >= 0.97, float(fd.max())
    M = xyzs.shape[0]
    sig_full, rgb_full = np.zeros(M, np.float32), np.zeros((M, 3), np.float32)
    sig_full[:n], rgb_full[:n] = sig_o * stu.density_scale, rgb_o
    ws, depth, img = oracle.composite_rays_train_forward(sig_full, rgb_full, _np(deltas), rays.cpu().numpy(), N=rays.shape[0])
    img = img + (1 - ws[:, None]) * _np(bg).reshape(-1, 3)
    err = np.abs(_np(out["image"]).reshape(-1, 3) - img)
    print("configs[2] AMP student render, %d samples: colour features %.2f %% bit-identical, image max |d| %.2e, mean %.2e"
          % (n, 100 * float((feat[:, 1:] == feat_o[:, 1:]).mean()), err.max(), err.mean()))
    assert err.max() <= 1e-4 and err.mean() <= 1e-6 and img.std() > 0.02, (float(err.max()), float(err.mean()))


def test_config2_amp_render_of_the_timed_step_matches_the_amp_oracle():
    """configs[2] at full size under AMP: the bench's 4096-ray batch marched by the HIP marcher (bit-exact with the oracle's), the
    frozen hash teacher's samples through k_hash_fwd_fused and the teacher image through the HIP compositor -- against oracle
    lookup + oracle AMP head + oracle compositor on the same samples."""
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import DistillWorkload
    torch.manual_seed(0)
    w = DistillWorkload(hip_ops(), torch.device("cuda:0"), PVDConfig(num_rays=4096), teacher_pretrain_steps=0, seed=0)
    assert w.opt.fp16 and w.tea.model_type == "hash" and w.stu.model_type == "vm"
    with torch.no_grad():
        g = torch.Generator(device="cuda:0").manual_seed(3)
        for n, p in w.tea.named_parameters():
            if "embeddings" in n:
                p.copy_((torch.rand(p.shape, device="cuda:0", generator=g) - 0.5) * 0.6)
            elif n.startswith(("sigma_net", "color_net")):
                p.mul_(1.5)
    rays_o, rays_d, bg = w.next_batch()
    w.opt.global_step = w.trainer.global_step  # stage 3 (what compute_loss sets before it renders: renderer.py:421-438 gates on it)
    tea = w.tea  # (train mode: run_cuda's train branch, on the student's samples)
    assert tea.training
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        inh, nf = w.stu.march(rays_o, rays_d, perturb=True, force_all_rays=False)
        out = tea.render(rays_o, rays_d, staged=False, bg_color=bg, perturb=True, force_all_rays=False, inherited_params=inh,
                         nears_fars=nf, premarched=True)
    xyzs, dirs, deltas, rays = inh
    n = int((rays[:, 1] + rays[:, 2]).max())
    assert n > 60000, n
    sig_o, rgb_o, feat_o = _hash_oracle(tea, xyzs[:n], dirs[:n])
    feat = tea.feature_sigma_color[:n]
    fd = np.abs(_np(feat) - feat_o)
    assert (fd <= 4e-3 * (1 + np.abs(feat_o))).all() and float((_np(feat) == feat_o).mean()) >= 0.97
    # teacher image: oracle compositor (raymarching.cu:504-582) on the oracle's sigma / rgb; density_scale = 1
    M = xyzs.shape[0]
    sig_full, rgb_full = np.zeros(M, np.float32), np.zeros((M, 3), np.float32)
    sig_full[:n], rgb_full[:n] = sig_o * tea.density_scale, rgb_o
    ws, depth, img = oracle.composite_rays_train_forward(sig_full, rgb_full, _np(deltas), rays.cpu().numpy(), N=rays.shape[0])
    img = img + (1 - ws[:, None]) * _np(bg).reshape(-1, 3)  # renderer.py:419
    err = np.abs(_np(out["image"]).reshape(-1, 3) - img)
    print("configs[2] AMP teacher render, %d samples: image max |d| %.2e, mean %.2e" % (n, err.max(), err.mean()))
    assert err.max() <= 2e-3 and err.mean() <= 1e-4, (float(err.max()), float(err.mean()))
