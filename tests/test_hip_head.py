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
