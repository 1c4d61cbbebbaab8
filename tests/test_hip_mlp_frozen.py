"""The frozen `mlp` (NeRF MLP) model on the GPU: positional encoding in one kernel (pvd_freq_encode), cached f16 weights, Linear +
ReLU as one library GEMM each, fused sigma / colour head -- against the layer-by-layer autocast formulation of the same
weights (reference: tools/encoding.py:6-49, distill_mutual/network.py:335-437)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype,stride", [(torch.float32, None), (torch.float16, 64), (torch.float32, 72)])
@pytest.mark.parametrize("include_input", [True, False])
def test_freq_encode_kernel_matches_the_torch_formulation(dtype, stride, include_input):
    import pvd_hip
    from pvd.encoding import FreqEncoder
    enc = FreqEncoder(3, 9, 10, include_input=include_input)
    x = (torch.rand(5001, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(0)) * 2 - 1)
    ref = enc(x)  # torch formulation (hip_encode is only attached by get_encoder)
    got = pvd_hip.freq_encode(x, enc.freq_bands, include_input, dtype, stride)
    w = ref.shape[1]
    assert got.shape == (5001, stride or w) and got.dtype == dtype
    tol = 2e-6 if dtype == torch.float32 else 1e-3
    assert (got[:, :w].float() - ref).abs().max().item() <= tol
    assert not got[:, w:].any()
    if dtype == torch.float32 and include_input:
        assert torch.equal(got[:, :3], x)


@pytest.mark.parametrize("layers,skip", [(8, 4), (6, 2)])
def test_frozen_mlp_model_matches_the_layerwise_autocast_formulation(layers, skip):
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import make_model
    ops = hip_ops()
    opt = PVDConfig(model_type="mlp", nerf_layer_num=layers, skip=skip, fp16=True)
    opt.stage_iters = {"stage1": -1, "stage2": -1}
    torch.manual_seed(3)
    m = make_model(ops, opt, "mlp", True, torch.device(DEV)).train()
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 2:
                p.mul_(1.5)
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.rand(20011, 3, device=DEV, generator=g) * 2 - 1
    d = torch.randn(20011, 3, device=DEV, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        sig, rgb = m(x, d)
        fea, sl, cl = m.feature_sigma_color.float(), m.sigma_l.float(), m.color_l.float()
        dens = m.density(x)["sigma"].float()
    # the generic formulation: the same model with the fast paths taken away (torch FreqEncoder, nn.Linear under autocast,
    # layer-by-layer head)
    generic = hip_ops()
    delattr(generic, "freq_encode"), delattr(generic, "fused_head")
    m.ops = generic
    m.encoder_nerf_pe.hip_encode = None
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        sig_r, rgb_r = m(x, d)
        fea_r = m.feature_sigma_color.float()
        dens_r = m.density(x)["sigma"].float()
    assert torch.isfinite(sig).all() and sig.shape == sig_r.shape and rgb.shape == rgb_r.shape
    # both sides round every layer to f16; GEMM kernels and the head's accumulation order differ
    assert (rgb.float() - rgb_r.float()).abs().max().item() <= 4e-3
    assert (fea - fea_r).abs().max().item() <= 2e-2 * (1 + fea_r.abs().max().item())
    assert ((sig.float() - sig_r.float()).abs() <= 2e-2 * sig_r.float().abs() + 1e-3).all()
    assert ((dens - dens_r).abs() <= 2e-2 * dens_r.abs() + 1e-3).all()
    assert torch.equal(sl, fea[:, 0]) and cl.shape == (20011, 3)


@pytest.mark.parametrize("M", [1, 191, 192, 20011])
def test_fused_trunk_kernel_matches_the_library_gemm_path(M, monkeypatch):
    """pvd_mlp_head_forward_fused (trunk streamed through LDS, activations in registers) vs the same frozen model through
    library GEMMs + the fused head: both are f16 GEMMs with f32 accumulation and a rounding per layer; accumulation order
    differs."""
    import fusedhead
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import make_model
    opt = PVDConfig(model_type="mlp", fp16=True)
    opt.stage_iters = {"stage1": -1, "stage2": -1}
    torch.manual_seed(7)
    m = make_model(hip_ops(), opt, "mlp", True, torch.device(DEV)).train()
    assert fusedhead.mlp_supported(m)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 2:
                p.mul_(1.4)
        for layer in m.nerf_mlp:
            layer.bias.uniform_(-0.2, 0.2)
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.rand(M, 3, device=DEV, generator=g) * 2 - 1
    d = torch.randn(M, 3, device=DEV, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    outs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("PVD_MLP_FUSED", fused)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            sig, rgb = m(x, d)
        outs.append((sig.float().clone(), rgb.float().clone(), m.feature_sigma_color.float().clone()))
    (sa, ra, fa), (sb, rb, fb) = outs
    assert torch.isfinite(sa).all() and fa.abs().max().item() > 0.05
    assert (ra - rb).abs().max().item() <= 3e-3
    assert (fa - fb).abs().max().item() <= 1e-2 * (1 + fb.abs().max().item())
    assert ((sa - sb).abs() <= 1.5e-2 * sb.abs() + 1e-3).all()
