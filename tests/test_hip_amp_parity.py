"""End-to-end parity of the configuration bench.py times: fp16 autocast + the fused MFMA heads + the fused
normL2 objective + FlatAdamW (folded L1 term, device-side schedule) + FlatGradScaler, eager and under hipGraph
replay -- against the SAME training run through the generic formulation: the reference's layer-by-layer autocast
network (bias-free nn.Linear chain, clamp, trunc_exp, SH, sigmoid -- distill_mutual/network.py:335-437), torch.norm
losses (utils.py:941-952, 1109-1189), torch.optim.AdamW + CosineAnnealingLR + torch.amp.GradScaler
(main_distill_mutual.py:334-348), with only the encoders / marcher / compositor shared.

Two kinds of bars:
  * what both paths compute exactly alike up to f16 rounding and summation order -> tight relative bars on the
    loss trajectory, the gradient direction and the update;
  * north_star's end-to-end bars: PSNR within 0.1 dB between the two runs, and of the AMP render against the fp32
    render of the same weights.  For the RGB bar the f16 formulation's OWN noise is measured first (the same generic
    network evaluated with permuted hidden units = permuted accumulation order), and the fused head must sit inside
    a small multiple of it.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _generic_ops():
    """The HIP operator set without the fused head, the fused objective and the flat optimizer."""
    from pvd.ops import hip_ops
    ops = hip_ops()
    for name in ("fused_head", "distill_loss", "flat_adamw"):
        delattr(ops, name)
    return ops


def _workload(ops, student="vm", seed=0, scene_scale=1.0, **kw):
    from pvd.config import PVDConfig
    from pvd.workload import DistillWorkload
    opt = PVDConfig(num_rays=1024, resolution0=64, iters=300, model_type=student, **kw)
    torch.cuda.manual_seed(1234)
    return DistillWorkload(ops, torch.device(DEV), opt, teacher_pretrain_steps=0, seed=seed, scene_scale=scene_scale)


def _pair(student="vm", **kw):
    """(fused workload, generic workload) with identical weights, occupancy, sample budget and batch stream."""
    from pvd.ops import hip_ops
    wa = _workload(hip_ops(), student, **kw)
    wb = _workload(_generic_ops(), student, **kw)
    # a teacher that is not at its initialisation: sizeable table entries and MLP weights (what training produces)
    g = torch.Generator(device=DEV).manual_seed(5)
    with torch.no_grad():
        if wa.tea.model_type == "hash":
            wa.tea.encoder.embeddings.copy_((torch.rand(wa.tea.encoder.embeddings.shape, device=DEV, generator=g) - 0.5) * 0.6)
        for p in list(wa.tea.sigma_net.parameters()) + list(wa.tea.color_net.parameters()):
            p.mul_(1.7)
    import pvd_hip
    pvd_hip.note_weights_changed(list(wa.tea.parameters()))
    wb.tea.load_state_dict(wa.tea.state_dict())
    wb.stu.load_state_dict(wa.stu.state_dict())
    wb.stu.mean_count = wb.tea.mean_count = wa.stu.mean_count
    for w in (wa, wb):
        w._eager_device_batches = True  # the one-kernel batch generator, keyed by (seed, counter): same rays in both runs
    return wa, wb


def _flat_params(model):
    getattr(model, "_pvd_flush_params", lambda: None)()  # FlatAdamW: deferred weight decay of rows nothing reads
    return torch.cat([p.detach().float().permute(0, 2, 3, 1).reshape(-1) if p.dim() == 4 else p.detach().float().reshape(-1)
                      for p in model.parameters() if p.requires_grad])


def _flat_grads(model):
    return torch.cat([p.grad.detach().float().permute(0, 2, 3, 1).reshape(-1) if p.dim() == 4 else p.grad.detach().float().reshape(-1)
                      for p in model.parameters() if p.requires_grad])


def _psnr(a, b):
    mse = torch.mean((a.float() - b.float()) ** 2).item()
    return float("inf") if mse == 0 else -10.0 * np.log10(mse)


CASES = {
    "vm": dict(),                                                   # configs[2]: hash -> vm (what bench.py times)
    "hash": dict(),                                                 # hash -> hash, bound 1
    "hash_bound2": dict(bound=2.0, dt_gamma=1.0 / 256, scene_scale=1.9),  # configs[4]: two cascades, distance-proportional steps
    "tensors_from_mlp": dict(teacher_type="mlp", plenoxel_res="[48,48,48]"),  # configs[3]
}


def _case(name, **extra):
    kw = dict(CASES[name], **extra)
    return _pair(name.split("_")[0], **kw)


@pytest.mark.parametrize("student", list(CASES))
def test_one_backward_fused_vs_generic(student):
    """Same weights, same batch: loss and the whole student gradient of the fused AMP path vs the generic AMP path."""
    wa, wb = _case(student, l1_reg_weight=0.0)  # (the folded L1 gradient never reaches .grad: covered by the training-run test)
    outs = []
    for w in (wa, wb):
        tr = w.trainer
        tr._zero_grads()
        with torch.autocast("cuda", dtype=torch.float16):
            loss, info, ps, pt = tr.compute_loss(*w.device_batch())
        scale = 1024.0
        (loss * scale).backward()
        g = _flat_grads(w.stu) / scale
        hg = getattr(tr.optimizer, "_half_grad", None) if tr.flat_opt else None
        if hg is not None and w.stu.model_type == "hash":  # the hash table's f16 gradient waits for the update kernel: fold it in for the comparison
            b, e, h = hg
            emb = w.stu.encoder.embeddings
            off = 0
            for p in w.stu.parameters():
                if p is emb:
                    break
                off += p.numel() if p.requires_grad else 0
            g[off:off + emb.numel()] += h.float() / scale
        outs.append((float(loss), g, ps.detach().float(), pt.detach().float()))
    (la, ga, psa, pta), (lb, gb, psb, ptb) = outs
    assert abs(la - lb) <= 3e-3 * abs(lb), (la, lb)
    assert (psa - psb).abs().max().item() <= 4e-3 and (pta - ptb).abs().max().item() <= 4e-3
    cos = torch.nn.functional.cosine_similarity(ga, gb, dim=0).item()
    rel = ((ga - gb).norm() / gb.norm()).item()
    print("one backward (%s): loss %.6f vs %.6f, grad cos %.6f rel %.3e, image diff %.2e" % (student, la, lb, cos, rel, (psa - psb).abs().max().item()))
    assert cos > 0.999 and rel < 4e-2, (cos, rel)
    assert gb.abs().max().item() > 0


@pytest.mark.parametrize("student", list(CASES))
def test_training_run_fused_graph_vs_generic_eager(student):
    """N distillation steps three ways from identical states: (a) fused AMP eager, (b) fused AMP under hipGraph replay,
    (c) generic AMP eager.  (a) and (b) must agree to summation order; (a) and (c) within f16 noise: loss trajectory,
    parameter update and PSNR (within 0.1 dB).  One step runs with a loss scale that overflows f16: all three skip it
    and back the scale off."""
    wa, wc = _case(student)
    wg, _ = _case(student)
    wg.stu.load_state_dict(wa.stu.state_dict())
    p0 = _flat_params(wa.stu).clone()
    n_warm, n = 3, 8  # enable_graph runs 3 eager warm-up steps before it records
    la, lg, lc = [], [], []
    for _ in range(n_warm):
        la.append(float(wa.step()[0]))
        lc.append(float(wc.step()[0]))
    wg.enable_graph()
    overflow_step = 4 if student != "tensors_from_mlp" else -1  # (the Plenoxel student has no f16 value on its gradient path)
    for k in range(n):
        if k == overflow_step:  # overflow: every gradient of this step is inf / nan -> the step must be skipped, the scale halved
            for w in (wa, wg, wc):
                w.trainer.scaler._scale.fill_(2.0 ** 40)
            before = [_flat_params(w.stu).clone() for w in (wa, wg, wc)]
        ra, rg, rc = wa.step(), wg.step(), wc.step()
        if k == overflow_step:
            for w, b in zip((wa, wg, wc), before):
                assert torch.equal(_flat_params(w.stu), b), "an overflowing step must leave the parameters alone"
                assert float(w.trainer.scaler._scale) == 2.0 ** 39
                w.trainer.scaler._scale.fill_(65536.0)
            continue
        la.append(float(ra[0])); lg.append(float(rg[0])); lc.append(float(rc[0]))
    la, lg, lc = np.array(la), np.array(lg), np.array(lc)
    assert np.isfinite(la).all() and np.isfinite(lg).all() and np.isfinite(lc).all()
    # graph replay == eager (same kernels, same batches; float atomics reorder sums)
    assert np.allclose(la[n_warm:], lg, rtol=2e-3), (la, lg)
    # fused == generic within f16 noise, step after step (the trajectories would drift apart if a gradient were mis-wired)
    assert np.allclose(la, lc, rtol=1.5e-2), (la, lc)
    pa, pg, pc = _flat_params(wa.stu), _flat_params(wg.stu), _flat_params(wc.stu)
    upd = (pc - p0).norm().item()
    print("training run (%s): losses fused %s\n graph %s\n generic %s\n update %.4e, |fused-graph|/upd %.3e, |fused-generic|/upd %.3e"
          % (student, la, lg, lc, upd, (pa - pg).norm().item() / upd, (pa - pc).norm().item() / upd))
    assert upd > 0
    assert ((pa - pg).norm().item() / upd) < 0.08, (pa - pg).norm().item() / upd
    assert ((pa - pc).norm().item() / upd) < 0.25, (pa - pc).norm().item() / upd
    # PSNR of the student against the teacher on one more batch: within 0.1 dB across the three runs
    ps = []
    for w in (wa, wg, wc):
        w._graph = False
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            _, _, pred_s, pred_t = w.trainer.compute_loss(*w.device_batch())
        ps.append(_psnr(pred_s, pred_t))
    print("psnr student vs teacher:", ps)
    assert abs(ps[0] - ps[1]) <= 0.1 and abs(ps[0] - ps[2]) <= 0.1, ps


def _permute_hidden_units(model, seed):
    """In place: the same function with the hidden units of every MLP layer permuted -- identical in exact arithmetic, a
    different accumulation order in f16."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for net in (getattr(model, "sigma_net", None), model.color_net):
            if net is None:
                continue
            for l in range(len(net) - 1):
                perm = torch.randperm(net[l].weight.shape[0], generator=g).to(net[l].weight.device)
                net[l].weight.copy_(net[l].weight[perm])
                net[l + 1].weight.copy_(net[l + 1].weight[:, perm])


@pytest.mark.parametrize("kind", ["hash", "vm"])
def test_amp_render_error_is_the_f16_formulations_own(kind):
    """Renders of the same weights: fp32, generic AMP (the reference's autocast formulation), generic AMP with permuted
    hidden units (x2), fused AMP.  The spread of the generic AMP renders around fp32 is the noise the reference's own
    fp16 path carries; the fused head must be inside 2x that (+1e-4), and its PSNR against the fp32 render within 0.1 dB
    of theirs.  Stated bound: RGB of the AMP path differs from fp32 by at most a few 1e-3 (printed by the assertion)."""
    from pvd.ops import hip_ops
    from pvd.scene import BLENDER_INTRINSICS, get_rays, synthetic_poses
    wa = _workload(hip_ops(), "vm" if kind == "vm" else "hash")
    model = wa.stu if kind == "vm" else wa.tea
    with torch.no_grad():
        g = torch.Generator(device=DEV).manual_seed(9)
        if kind == "hash":
            model.encoder.embeddings.copy_((torch.rand(model.encoder.embeddings.shape, device=DEV, generator=g) - 0.5) * 0.6)
        for p in model.parameters():
            if p.dim() == 2:
                p.mul_(1.7)
    import pvd_hip
    pvd_hip.note_weights_changed(list(model.parameters()))
    generic = _generic_ops()
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(4))).to(DEV)
    r = get_rays(poses[2][None], BLENDER_INTRINSICS, 800, 800, 2048, generator=torch.Generator(device=DEV).manual_seed(3))
    bg = torch.rand(1, 2048, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    model.args.global_step = 10 ** 6  # stage 3

    def render(m, ops, amp):
        old = m.ops
        m.ops = ops
        m.train()
        try:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=amp):
                out = m.render(r["rays_o"], r["rays_d"], staged=False, bg_color=bg, perturb=False, force_all_rays=True, dt_gamma=0,
                               max_steps=1024)
        finally:
            m.ops = old
        return out["image"].float()

    def render_permuted(seed):
        saved = {k: v.clone() for k, v in model.state_dict().items()}
        _permute_hidden_units(model, seed)
        try:
            return render(model, generic, True)
        finally:
            model.load_state_dict(saved)
            pvd_hip.note_weights_changed(list(model.parameters()))

    was_teacher = model.is_teacher
    model.is_teacher = False  # the model marches for itself
    try:
        img32 = render(model, generic, False)
        img_gen = [render(model, generic, True)] + [render_permuted(s) for s in (1, 2)]
        img_fused = render(model, hip_ops(), True)
    finally:
        model.is_teacher = was_teacher
    print("AMP render error vs fp32 (%s): fused %.2e, generic %.2e, generic spread under permuted accumulation %.2e"
          % (kind, (img_fused - img32).abs().max().item(), max((i - img32).abs().max().item() for i in img_gen),
             max((img_gen[0] - i).abs().max().item() for i in img_gen[1:])))
    assert img32.std().item() > 0.02
    e_gen = max((i - img32).abs().max().item() for i in img_gen)
    e_fused = (img_fused - img32).abs().max().item()
    spread = max((img_gen[0] - i).abs().max().item() for i in img_gen[1:])
    assert e_fused <= 2.0 * max(e_gen, spread) + 1e-4, (e_fused, e_gen, spread)
    assert e_fused <= 1e-4, e_fused  # north_star's RGB bar, met by the AMP path itself on these weights
    p_gen = min(_psnr(i, img32) for i in img_gen)
    p_fused = _psnr(img_fused, img32)
    assert p_fused >= p_gen - 0.1 and p_fused > 45.0, (p_fused, p_gen)
    # "PSNR within 0.1 dB": against a common target (the analytic scene), AMP fused vs fp32
    target = wa.target(r["rays_o"], r["rays_d"], bg).float()
    assert abs(_psnr(img_fused, target) - _psnr(img32, target)) <= 0.1
