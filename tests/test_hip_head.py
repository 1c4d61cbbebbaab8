"""GPU parity of the fused MFMA sigma/colour head against the reference formulation -- the chain of half
Linear layers, clamp, trunc_exp, SH, sigmoid that NeRFNetwork.forward runs under autocast(fp16)
(distill_mutual/network.py:335-437), evaluated layer by layer in PyTorch on the same device.
Tolerances are f16-level: both sides round every layer output to f16; the GEMM accumulation order differs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(kind, seed=0):
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import make_model
    torch.manual_seed(seed)
    opt = PVDConfig(model_type=kind, resolution0=64)
    opt.stage_iters = {"stage1": -1, "stage2": -1}
    ops = hip_ops()
    ops.fused_head = None  # layer-by-layer torch formulation
    m = make_model(ops, opt, kind, False, torch.device("cuda:0"))
    for p in m.parameters():
        if p.dim() == 2:
            p.data.mul_(2.0)  # livelier activations than the default init
    if kind == "hash":
        m.encoder.embeddings.data.uniform_(-1.0, 1.0)
    return m


def _inputs(M, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.rand(M, 3, device="cuda", generator=g) * 2 - 1
    d = torch.randn(M, 3, device="cuda", generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    x[:8] = 0.0
    d[:8] = 0.0  # the marcher's padding rows
    return x, d


def _check(sig, rgb, feat, sig_r, rgb_r, feat_r):
    assert torch.isfinite(sig).all() and torch.isfinite(rgb).all()
    # f16 has 11 significant bits: one ulp at 1.0 is 9.8e-4, sigmoid output is in [0, 1]
    assert (rgb - rgb_r.float()).abs().max().item() <= 2e-3
    assert (rgb - rgb_r.float()).abs().mean().item() <= 2e-4
    fd = (feat - feat_r.float()).abs()
    assert (fd <= 4e-3 * (1 + feat_r.float().abs())).all(), fd.max().item()
    rel = ((sig - sig_r.float()).abs() / (sig_r.float().abs() + 1e-6))
    assert rel.max().item() <= 8e-3 and rel.mean().item() <= 1e-3, (rel.max().item(), rel.mean().item())


@pytest.mark.parametrize("M", [16 * 1000, 4099])
def test_hash_head_matches_layerwise_autocast(M):
    import fusedhead
    m = _model("hash").eval()
    x, d = _inputs(M)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        sig_r, rgb_r = m(x, d)
        feat_r = m.feature_sigma_color
    sig, rgb, feat = fusedhead.hash_head_infer(m, x, d)
    _check(sig, rgb, feat, sig_r, rgb_r, feat_r)


@pytest.mark.parametrize("M", [16 * 1000, 4099])
def test_vm_head_matches_layerwise_autocast(M):
    import fusedhead
    m = _model("vm").eval()
    x, d = _inputs(M)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        sig_r, rgb_r = m(x, d)
        feat_r = m.feature_sigma_color
        sraw, prod = m.ops.vm_encode(x, m._aabb(), *m.sigma_mat, *m.sigma_vec, *m.color_mat, *m.color_vec)
    sig, rgb, feat = fusedhead.vm_head_infer(m, sraw, prod, d)
    _check(sig, rgb, feat, sig_r, rgb_r, feat_r)


def test_vm_head_backward_matches_layerwise_autograd():
    """Gradients of the fused head (prod, sigma_raw, all four weight matrices) vs autograd through the
    layer-by-layer autocast formulation on the same inputs."""
    import fusedhead
    from pvd.ops import hip_ops
    m = _model("vm", seed=3).train()
    M = 16 * 700 + 5
    x, d = _inputs(M, seed=4)
    g = torch.Generator(device="cuda").manual_seed(9)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        sraw0, prod0 = m.ops.vm_encode(x, m._aabb(), *m.sigma_mat, *m.sigma_vec, *m.color_mat, *m.color_vec)
    w_sig = torch.randn(M, device="cuda", generator=g) * 1e-3
    w_rgb = torch.randn(M, 3, device="cuda", generator=g)
    w_fea = torch.randn(M, 16, device="cuda", generator=g) * 0.1
    names = ["basis_mat.weight", "color_net.0.weight", "color_net.1.weight", "color_net.2.weight"]
    params = dict(m.named_parameters())

    w_rgb_b = torch.randn(M, 3, device="cuda", generator=g)

    def run(fused, split="both"):
        sraw = sraw0.clone().requires_grad_(True)
        prod = prod0.clone().requires_grad_(True)
        for n in names:
            params[n].grad = None
        with torch.autocast("cuda", dtype=torch.float16):
            if fused:
                sig, rgb, feat, rgb_l = fusedhead.vm_head_train(m, sraw, prod, d)
                assert rgb_l.data_ptr() == rgb.data_ptr()  # the same values, a second autograd output
            else:
                a = m.args
                cf = torch.clamp(m.linear(prod, m.basis_mat.weight), a.sigma_clip_min, a.sigma_clip_max)
                sf = torch.clamp(sraw, a.sigma_clip_min, a.sigma_clip_max)
                feat = torch.cat([sf.unsqueeze(-1), cf], dim=-1)
                sig = m.trunc_exp(sf)
                rgb = rgb_l = m._color_head(m.encoder_dir(d), cf)
        # rgb has two consumers (compositing, colour term): the fused backward adds their gradients while loading
        loss = (sig.float() * w_sig).sum() + (feat.float() * w_fea).sum()
        if split in ("both", "first"):
            loss = loss + (rgb.float() * w_rgb).sum()
        if split in ("both", "second"):
            loss = loss + (rgb_l.float() * w_rgb_b).sum()
        loss.backward()
        return sraw.grad.clone(), prod.grad.float().clone(), [params[n].grad.clone() for n in names]

    for split in ("first", "second"):  # one of the two gradients absent (None reaches the backward)
        _, gp_r1, _ = run(False, split)
        _, gp_f1, _ = run(True, split)
        assert (gp_f1 - gp_r1).abs().max().item() <= 2e-2 * gp_r1.abs().max().item()
    gs_r, gp_r, gw_r = run(False)
    gs_f, gp_f, gw_f = run(True)
    assert torch.isfinite(gp_f).all()
    assert torch.allclose(gs_f, gs_r, rtol=2e-3, atol=1e-6)
    scale = gp_r.abs().max().item()
    assert (gp_f - gp_r).abs().max().item() <= 2e-2 * scale, ((gp_f - gp_r).abs().max().item(), scale)
    assert (gp_f - gp_r).abs().mean().item() <= 2e-3 * gp_r.abs().mean().item() + 1e-8
    for n, a, b in zip(names, gw_f, gw_r):
        s = b.abs().max().item()
        assert (a - b).abs().max().item() <= 2e-2 * s, (n, (a - b).abs().max().item(), s)
        assert (a - b).abs().mean().item() <= 5e-3 * b.abs().mean().item(), n


def test_vm_head_backward_accumulates_into_existing_grads():
    import fusedhead
    m = _model("vm", seed=5).train()
    M = 4096
    x, d = _inputs(M, seed=6)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        sraw, prod = m.ops.vm_encode(x, m._aabb(), *m.sigma_mat, *m.sigma_vec, *m.color_mat, *m.color_vec)
    ps = [m.basis_mat.weight, m.color_net[0].weight, m.color_net[1].weight, m.color_net[2].weight]
    outs = []
    for pre in (0.0, 1.0):
        for p in ps:
            p.grad = torch.full_like(p, pre)
        with torch.autocast("cuda", dtype=torch.float16):
            sig, rgb, feat = fusedhead.vm_head_train(m, sraw.clone().requires_grad_(True), prod.clone().requires_grad_(True), d)[:3]
        (rgb.sum() + feat.sum()).backward()
        outs.append([p.grad.clone() for p in ps])
    for a, b in zip(*outs):
        assert torch.allclose(b - 1.0, a, rtol=1e-4, atol=1e-3)


def test_hash_head_backward_matches_layerwise_autograd():
    """Fused hash path (grid lookup + MFMA head, MFMA head backward + grid scatter) vs autograd through
    GridEncoder + the layer-by-layer autocast formulation: gradients of the embedding table, sigma_net and
    color_net (network.py:395-437)."""
    import fusedhead
    m = _model("hash", seed=7).train()
    M = 16 * 700 + 5
    x, d = _inputs(M, seed=8)
    g = torch.Generator(device="cuda").manual_seed(10)
    w_sig = torch.randn(M, device="cuda", generator=g) * 1e-3
    w_rgb = torch.randn(M, 3, device="cuda", generator=g)
    w_fea = torch.randn(M, 16, device="cuda", generator=g) * 0.1
    names = ["encoder.embeddings", "sigma_net.0.weight", "sigma_net.1.weight", "color_net.0.weight", "color_net.1.weight",
             "color_net.2.weight"]
    params = dict(m.named_parameters())

    def run(fused):
        for n in names:
            params[n].grad = None
        with torch.autocast("cuda", dtype=torch.float16):
            if fused:
                sig, rgb, feat = fusedhead.hash_head_train(m, x, d)
            else:
                sig, rgb = m(x, d)
                feat = m.feature_sigma_color
        loss = (sig.float() * w_sig).sum() + (rgb.float() * w_rgb).sum() + (feat.float() * w_fea).sum()
        loss.backward()
        return sig.detach().float(), rgb.detach().float(), feat.detach().float(), [params[n].grad.float().clone() for n in names]

    sig_r, rgb_r, feat_r, gw_r = run(False)
    sig_f, rgb_f, feat_f, gw_f = run(True)
    _check(sig_f, rgb_f, feat_f, sig_r, rgb_r, feat_r)
    for n, a, b in zip(names, gw_f, gw_r):
        assert torch.isfinite(a).all(), n
        s = b.abs().max().item()
        assert s > 0, n
        assert (a - b).abs().max().item() <= 2e-2 * s, (n, (a - b).abs().max().item(), s)
        assert (a - b).abs().mean().item() <= 5e-3 * b.abs().mean().item() + 1e-9, n


def test_hash_head_backward_accumulates_into_existing_grads():
    import fusedhead
    m = _model("hash", seed=11).train()
    M = 4096
    x, d = _inputs(M, seed=12)
    ps = [m.encoder.embeddings, m.sigma_net[0].weight, m.sigma_net[1].weight, m.color_net[0].weight, m.color_net[1].weight,
          m.color_net[2].weight]
    outs = []
    for pre in (0.0, 1.0):
        for p in ps:
            p.grad = torch.full_like(p, pre)
        with torch.autocast("cuda", dtype=torch.float16):
            sig, rgb, feat = fusedhead.hash_head_train(m, x, d)
        (rgb.sum() + feat.sum()).backward()
        outs.append([p.grad.clone() for p in ps])
    for a, b in zip(*outs):
        assert torch.allclose(b - 1.0, a, rtol=1e-3, atol=2e-3)


@pytest.mark.parametrize("kind", ["vm", "hash"])
def test_packed_weight_image_is_bit_identical(kind):
    """pvd_head_pack_weights + image path == converting the fp32 masters inside the kernels (same f16 values, same
    LDS layout): outputs and every gradient bit for bit."""
    import pvd_hip
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    M = 16 * 300 + 7
    K = 1 if kind == "vm" else 0
    f32 = lambda *s: torch.randn(*s, device=dev) * 0.3
    if K == 1:
        x0 = (torch.randn(M, 144, device=dev) * 0.3).half(); Wa1, Wa2 = f32(15, 144), None
        sraw = f32(M)
    else:
        x0 = (torch.randn(14, M, 2, device=dev) * 0.3).half(); Wa1, Wa2 = f32(64, 28), f32(16, 64)
        sraw = None
    d = torch.randn(M, 3, device=dev); d = d / d.norm(dim=-1, keepdim=True)
    Wc1, Wc2, Wc3 = f32(64, 31), f32(64, 64), f32(3, 64)
    gs, gr, gf = f32(M), f32(M, 3), f32(M, 16)
    image = pvd_hip.head_pack_weights(K, Wa1, Wa2, Wc1, Wc2, Wc3)
    assert image.dtype == torch.float16 and image.numel() == pvd_hip.head_image_halfs(K)
    res = []
    for img in (None, image):
        sig, rgb, feat = torch.empty(M, device=dev), torch.empty(M, 3, device=dev), torch.empty(M, 16, device=dev)
        pvd_hip.head_forward(K, x0, sraw, d, M, Wa1, Wa2, Wc1, Wc2, Wc3, -2.0, -2.0, 7.0, sig, rgb, feat, image=img)
        gx = torch.empty_like(x0)
        gsraw = torch.empty(M, device=dev) if K == 1 else None
        gWa1 = torch.zeros_like(Wa1)
        gWa2 = torch.zeros_like(Wa2) if Wa2 is not None else None
        gW = [torch.zeros_like(w) for w in (Wc1, Wc2, Wc3)]
        ws = torch.empty(pvd_hip.head_backward_workspace_floats(K, M), device=dev)
        pvd_hip.head_backward(K, x0, sraw, d, M, Wa1, Wa2, Wc1, Wc2, Wc3, -2.0, -2.0, 7.0, gs, gr, gf, gsraw, gx, gWa1, gWa2, *gW, ws, image=img)
        res.append([sig, rgb, feat, gx, gWa1] + gW + ([gsraw] if K == 1 else [gWa2]))
    for a, b in zip(*res):
        # the weight gradients are summed by atomics over 16 slices: order-dependent in the last bits
        if a.dtype == torch.float32 and a.dim() == 2 and a.shape[0] != M:
            assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item()
        else:
            assert torch.equal(a, b)


def test_teacher_image_cache_follows_weight_updates():
    """hash_head_infer caches the packed image and the f16 table; an optimizer kernel that rewrites the parameters
    (no autograd version bump) must invalidate both."""
    import fusedhead
    import pvd_hip
    m = _model("hash", seed=13).eval()
    x, d = _inputs(4096, seed=14)
    s0, c0, _ = fusedhead.hash_head_infer(m, x, d)
    s1, c1, _ = fusedhead.hash_head_infer(m, x, d)
    assert torch.equal(c0, c1)
    ptr = m.color_net[2].weight.data_ptr()
    torch.ops.aten.mul_(m.color_net[2].weight.data, 0.5)  # .data: no version bump on the Parameter, like a raw kernel write
    m.encoder.embeddings.data.mul_(0.5)
    assert m.color_net[2].weight.data_ptr() == ptr
    pvd_hip.note_weights_changed([m.color_net[2].weight, m.encoder.embeddings])
    s2, c2, _ = fusedhead.hash_head_infer(m, x, d)
    assert not torch.equal(c0, c2) and not torch.equal(s0, s2)


def test_hash_density_fused_matches_layerwise():
    """NeRFNetwork.density (used by update_extra_state, network.py:439-494) through the fused lookup + head vs the
    layer-by-layer autocast formulation."""
    m = _model("hash", seed=21).eval()
    x, _ = _inputs(16 * 500 + 3, seed=22)
    import fusedhead
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        m.ops.fused_head = fusedhead
        assert m._fused_ok(x)
        fused = m.density(x)
        m.ops.fused_head = None
        ref = m.density(x)
    rel = (fused["sigma"] - ref["sigma"].float()).abs() / (ref["sigma"].float().abs() + 1e-6)
    assert rel.max().item() <= 8e-3 and rel.mean().item() <= 1e-3
    assert (fused["geo_feat"] - ref["geo_feat"].float()).abs().max().item() <= 4e-3 * (1 + ref["geo_feat"].float().abs().max().item())


@pytest.mark.parametrize("M,bound", [(16 * 1000, 1), (4099, 1), (37, 1), (20000, 2), (300001, 1)])  # the last: several tiles per workgroup
def test_fused_lookup_and_head_is_bit_identical_to_the_two_launches(M, bound, monkeypatch):
    """pvd_hash_head_forward_fused (lookup + head in one launch, no [14][M][2] intermediate) vs pvd_grid_encode_forward_affine
    followed by pvd_head_forward: same table values, same blend order, same MFMA chain -> the same bits.  Ragged sizes (not a
    multiple of the 128-sample workgroup tile or of the 16-sample MFMA tile), samples outside the box, a second cascade."""
    import fusedhead
    m = _model("hash").eval()
    m.bound = bound
    x, d = _inputs(M)
    x = x * bound
    x[-5:] = bound * 1.5  # outside [-bound, bound]: zero features on every level (gridencoder.cu:113-124)
    monkeypatch.setattr(fusedhead, "FUSED_LOOKUP", False)
    s0, c0, f0 = fusedhead.hash_head_infer(m, x, d)
    monkeypatch.setattr(fusedhead, "FUSED_LOOKUP", True)
    s1, c1, f1 = fusedhead.hash_head_infer(m, x, d)
    assert torch.equal(s0, s1) and torch.equal(c0, c1) and torch.equal(f0, f1)
    assert torch.isfinite(s1).all() and f1.abs().max().item() > 0


@pytest.mark.parametrize("dma", ["0", "1"])
def test_every_variant_of_the_fused_kernel_produces_the_same_bits(dma, monkeypatch):
    """k_hash_fwd_fused (14 levels per memory round trip; offsets in SGPRs, Level3 index shapes, blended as the loads retire) x
    PVD_FUSED_DMA (where the weight image's LDS-DMA is issued): against lookup + head as two launches, incl. a second cascade,
    out-of-box samples and a multi-chunk workgroup.  (The round-3 kernel and the 7-levels-per-round-trip form were removed in round 5.)"""
    import fusedhead
    m = _model("hash").eval()
    for M, bound in ((4099, 1), (20000, 2), (140001, 1)):
        m.bound = bound
        x, d = _inputs(M)
        x = x * bound
        x[-5:] = bound * 1.5
        monkeypatch.setattr(fusedhead, "FUSED_LOOKUP", False)
        ref = fusedhead.hash_head_infer(m, x, d)
        monkeypatch.setattr(fusedhead, "FUSED_LOOKUP", True)
        monkeypatch.setenv("PVD_FUSED_DMA", dma)
        got = fusedhead.hash_head_infer(m, x, d)
        assert all(torch.equal(a, b) for a, b in zip(ref, got)), (dma, M)


def test_the_fused_launch_stamps_its_own_extent():
    """pvd_hash_head_forward_fused_span: the same outputs bit for bit, and {min start, max end} over the launch's workgroups in the
    device's 100 MHz counter -- bench.py reads the in-step duration of the roofline kernel from it, inside a replayed graph where no
    host event can sit.  The span of a launch alone on the chip must agree with a HIP-event bracket of the same launch."""
    import fusedhead
    import pvd_hip
    m = _model("hash").eval()
    x, d = _inputs(92928)
    ref = fusedhead.hash_head_infer(m, x, d)
    span = torch.tensor(list(pvd_hip.FUSED_SPAN_INIT), dtype=torch.int64, device=x.device)
    assert np.isnan(pvd_hip.fused_span_us(span))  # an untouched record reads as "not written"
    m._fused_span = span
    got = fusedhead.hash_head_infer(m, x, d)
    assert "_fused_span" not in m.__dict__  # consumed by the launch it was meant for
    assert all(torch.equal(a, b) for a, b in zip(ref, got))
    torch.cuda.synchronize()
    us = float(pvd_hip.fused_span_us(span))
    assert 5.0 < us < 500.0, us
    # against HIP events around the same launch (events add the launch's ramp and a kernel boundary: a few us more)
    times = []
    for _ in range(5):
        span.copy_(torch.tensor(list(pvd_hip.FUSED_SPAN_INIT), dtype=torch.int64, device=x.device))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        m._fused_span = span
        a.record()
        fusedhead.hash_head_infer(m, x, d)
        b.record()
        torch.cuda.synchronize()
        times.append((float(pvd_hip.fused_span_us(span)), 1e3 * a.elapsed_time(b)))
    s_us, e_us = min(t[0] for t in times), min(t[1] for t in times)
    assert s_us <= e_us + 1.0 and s_us >= 0.4 * e_us, times  # (the bracket also covers the launch of the output allocations' fills, if any)


def test_vm_head_weight_gradient_reduction_riding_on_the_table_scatter():
    """The reduction of the VM head's per-workgroup weight-gradient tiles inside the table scatter's launch
    (pvd_head_backward_defer + pvd_vm_backward_rider, csrc/head_dw_reduce.h) against its own launch (PVD_HEAD_DW_RIDE=0): the same
    device function over the same tiles, so head and table gradients agree to the order of their float atomics."""
    import os
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import make_model
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    opt = PVDConfig(model_type="vm", resolution0=64)
    opt.stage_iters = {"stage1": -1, "stage2": -1}
    m = make_model(hip_ops(), opt, "vm", False, dev)
    for p in m.parameters():
        if p.dim() == 2:
            p.data.mul_(2.0)
    x, d = _inputs(16 * 1000 + 37)
    x = x * 0.8
    gen = torch.Generator(device="cuda").manual_seed(9)
    gs, gc, gf = (torch.randn(x.shape[0], device=dev, generator=gen), torch.randn(x.shape[0], 3, device=dev, generator=gen),
                  torch.randn(x.shape[0], 16, device=dev, generator=gen))
    res = {}
    for ride in ("1", "0"):
        old = os.environ.get("PVD_HEAD_DW_RIDE")
        os.environ["PVD_HEAD_DW_RIDE"] = ride
        try:
            for p in m.parameters():
                p.grad = torch.zeros_like(p)  # final buffers exist: the kernels accumulate straight into them (the trainer's situation)
            with torch.autocast("cuda", dtype=torch.float16):
                sigma, color = m(x, d)
                feat = m.feature_sigma_color
            ((sigma.float() * gs).sum() + (color.float() * gc).sum() + (feat.float() * gf).sum()).backward()
            torch.cuda.synchronize()
            res[ride] = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
        finally:
            if old is None:
                os.environ.pop("PVD_HEAD_DW_RIDE", None)
            else:
                os.environ["PVD_HEAD_DW_RIDE"] = old
    heads = [n for n in res["1"] if n.startswith(("basis_mat", "color_net"))]
    assert len(heads) == 4
    for n in res["1"]:
        a, b = res["1"][n], res["0"][n]
        scale = float(b.abs().max())
        assert scale > 0 and torch.isfinite(a).all(), n
        assert float((a - b).abs().max()) <= 2e-5 * scale, (n, float((a - b).abs().max()), scale)
