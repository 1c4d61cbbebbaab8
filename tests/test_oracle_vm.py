"""The oracle's C restatement of the TensoRF "VM" plane x line lookup (oracle/pvd_oracle.c: pvdo_vm_forward, network.py:216-309)
pinned by (a) PyTorch's own F.grid_sample formulation of get_sigma_feat / get_color_feat on non-cubic tables with points outside the
box, and (b) the REFERENCE's NeRFNetwork.forward (model_type vm) run on the CPU -- tests/golden/reference_python.npz `refnet_vm__*`:
its state dict, its inputs and the feature_sigma_color it computed (sigma feature clamped, colour features = basis_mat of the 144
products, clamped)."""
import os

import numpy as np
import torch
import torch.nn.functional as F

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
MAT_IDS, VEC_IDS = ((0, 1), (0, 2), (1, 2)), (2, 1, 0)


def _grid_sample_formulation(x, tabs):
    xt = torch.from_numpy(x)

    def feats(mats, vecs):
        outs = []
        for i, (m0, m1) in enumerate(MAT_IDS):
            pc = torch.stack([xt[:, m0], xt[:, m1]], -1).view(1, -1, 1, 2)
            lc = torch.stack([torch.zeros_like(xt[:, 0]), xt[:, VEC_IDS[i]]], -1).view(1, -1, 1, 2)
            pv = F.grid_sample(torch.from_numpy(mats[i]), pc, align_corners=True).view(mats[i].shape[1], -1)
            lv = F.grid_sample(torch.from_numpy(vecs[i]), lc, align_corners=True).view(vecs[i].shape[1], -1)
            outs.append(pv * lv)
        return torch.cat(outs, 0)
    return feats(tabs[0:3], tabs[3:6]).sum(0).numpy(), feats(tabs[6:9], tabs[9:12]).T.numpy()


def test_oracle_vm_lookup_is_the_grid_sample_formulation():
    rs = np.random.RandomState(0)
    res = (7, 9, 11)  # three different resolutions: planes are (x, y), (x, z), (y, z), lines z, y, x
    tabs = []
    for R in (16, 48):
        tabs += [rs.standard_normal((1, R, res[m1], res[m0])).astype(np.float32) for m0, m1 in MAT_IDS]
        tabs += [rs.standard_normal((1, R, res[v], 1)).astype(np.float32) for v in VEC_IDS]
    x = rs.uniform(-1.3, 1.3, size=(2000, 3)).astype(np.float32)  # a third of the points outside the box: zero padding
    x[:4] = [[1, 1, 1], [-1, -1, -1], [0, 0, 0], [1, -1, 0.999999]]
    sig, prod = oracle.vm_forward(x, (-1, -1, -1, 1, 1, 1), tabs, res)
    s_r, p_r = _grid_sample_formulation(x, tabs)
    assert np.abs(prod - p_r).max() <= 2e-6 and np.abs(sig - s_r).max() <= 1e-5
    assert (np.abs(prod).max(axis=1) == 0).any() and np.abs(prod).max() > 1.0  # some points see nothing, most see the tables
    # an aabb that is not the unit cube: x_n = 2 (x - lo) / (hi - lo) - 1 (network.py:345-350)
    aabb = (-0.5, -1.0, -2.0, 1.5, 1.0, 0.0)
    lo, hi = np.array(aabb[:3], np.float32), np.array(aabb[3:], np.float32)
    xw = ((x + 1) / 2 * (hi - lo) + lo).astype(np.float32)
    sig2, prod2 = oracle.vm_forward(xw, aabb, tabs, res)
    xn = (2 * (xw - lo) / (hi - lo) - 1).astype(np.float32)
    s_r2, p_r2 = _grid_sample_formulation(xn, tabs)
    assert np.abs(prod2 - p_r2).max() <= 2e-6 and np.abs(sig2 - s_r2).max() <= 1e-5


def test_oracle_vm_lookup_reproduces_the_references_vm_forward():
    g = np.load(os.path.join(HERE, "golden", "reference_python.npz"))
    pre = "refnet_vm__"
    sd = {k[len(pre) + 4:]: g[k] for k in g.files if k.startswith(pre + "sd__")}
    tabs = [sd["sigma_mat.%d" % i] for i in range(3)] + [sd["sigma_vec.%d" % i] for i in range(3)] + \
           [sd["color_mat.%d" % i] for i in range(3)] + [sd["color_vec.%d" % i] for i in range(3)]
    res = (tabs[0].shape[3], tabs[0].shape[2], tabs[1].shape[2])
    assert tabs[3].shape[2] == res[2] and tabs[2].shape[2:] == (res[2], res[1])
    aabb = sd["aabb_train"]
    sig, prod = oracle.vm_forward(g["refnet_x"], aabb, tabs, res)
    color_feat = prod @ sd["basis_mat.weight"].T  # network.py:306-308 (fp32 on the CPU)
    fea = np.concatenate([np.clip(sig, -2, 7)[:, None], np.clip(color_feat, -2, 7)], axis=1)  # :357-366
    want = g[pre + "feature_sigma_color"]
    assert fea.shape == want.shape and np.abs(fea - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
    assert np.ptp(want[:, 0]) > 1.0 and np.ptp(want[:, 1:]) > 0.2  # (a field that varies)
