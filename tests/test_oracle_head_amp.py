"""The oracle's restatement of the sigma / colour head under autocast (oracle/pvd_oracle.c: pvdo_head_forward_amp) against the
REFERENCE's own NeRFNetwork.forward run under torch.autocast("cpu", float16) (tests/golden/make_golden.py ->
reference_head_amp.npz: distill_mutual/network.py:413-437 for the hash model, :344-381 for vm).  The fixture holds what reached
the head's first Linear (encoder output / plane x line products, fp32), the head's weights and what the reference computed.
Bar: the same f16 values except where an fp32 sum lands within rounding of an f16 tie (the reference's CPU GEMM adds in blocks,
the oracle in ascending k): at most one f16 ulp, on at most 1 % of the entries."""
import os

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def _ulp16(v):
    v = np.abs(v.astype(np.float32))
    return np.maximum(np.spacing(v.astype(np.float16)).astype(np.float32), np.float32(2.0 ** -24))


def _close_f16(got, ref, what):
    diff = np.abs(got.astype(np.float32) - ref.astype(np.float32))
    assert (diff <= _ulp16(ref) * 1.001).all(), (what, float(diff.max()))
    assert (diff != 0).mean() <= 0.01, (what, float((diff != 0).mean()))


@pytest.mark.parametrize("mt", ["hash", "vm"])
def test_oracle_amp_head_is_the_references_head_under_autocast(mt):
    g = np.load(os.path.join(HERE, "golden", "reference_head_amp.npz"))
    pre = "amp_%s__" % mt
    W = lambda k: g[pre + "sd__" + k]
    x0 = g[pre + "x0"].astype(np.float16)  # the first Linear's cast
    if mt == "hash":
        sigma, rgb, feat = oracle.head_forward_amp(0, x0, None, g["d"], W("sigma_net.0.weight"), W("sigma_net.1.weight"),
                                                   W("color_net.0.weight"), W("color_net.1.weight"), W("color_net.2.weight"))
        assert str(g[pre + "feature_dtype"]) == "torch.float16"
    else:
        sigma, rgb, feat = oracle.head_forward_amp(1, x0, g[pre + "sigma_raw"], g["d"], W("basis_mat.weight"), None,
                                                   W("color_net.0.weight"), W("color_net.1.weight"), W("color_net.2.weight"))
        # feature 0 is the fp32 sigma feature, clamped in fp32 (network.py:357-360): exact
        assert np.array_equal(feat[:, 0], g[pre + "feature_sigma_color"][:, 0])
    _close_f16(feat, g[pre + "feature_sigma_color"], "feature_sigma_color")
    _close_f16(rgb, g[pre + "color"], "color")
    assert (g[pre + "color"] > 0.02).any() and (g[pre + "color"] < 0.98).any() and np.ptp(g[pre + "feature_sigma_color"][:, 1:]) > 1.0
    # sigma: the reference's CPU run applies exp to the f16 feature (hash) / the fp32 one (vm) without trunc_exp's cast to fp32
    # (a CUDA-autocast construct); the oracle's fp32 exp of the same feature must round to it
    ref_sigma = g[pre + "sigma"]
    if mt == "hash":
        _close_f16(sigma.astype(np.float16).astype(np.float32), ref_sigma, "sigma")
    else:
        assert np.allclose(sigma, ref_sigma, rtol=2e-6)
