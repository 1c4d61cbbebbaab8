#!/usr/bin/env python3
"""Generate tests/golden/reference_poses.npz: the reference's three random-pose generators (distill_mutual/utils.py:100-197,
`get_rand_poses`, selected by --data_type: main_distill_mutual.py:208-212) run HERE with a seeded np.random, `.cuda()` being the
identity for the run.  Data only (seeds, the original-loader poses fed to the llff generator, the poses that came out).

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_poses.py
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

import oracle_backend as ob

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
from distill_mutual import utils as ref_utils  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self  # no GPU here: the generators only move their results
ref_utils.device = torch.device("cpu")

out = {}
for seed in (0, 7):
    np.random.seed(seed)
    out["synthetic_seed%d" % seed] = ref_utils.get_rand_poses("synthetic").numpy()
    np.random.seed(seed)
    out["tank_seed%d" % seed] = ref_utils.get_rand_poses("tank").numpy()
    # an LLFF-like original loader: forward-facing cameras on a small patch, cam2world in the NGP convention
    rs = np.random.RandomState(100 + seed)
    orig = np.tile(np.eye(4, dtype=np.float32), (20, 1, 1))
    orig[:, :3, 3] = rs.uniform([-0.4, -0.25, 0.9], [0.4, 0.25, 1.1], size=(20, 3)).astype(np.float32)
    out["llff_original_seed%d" % seed] = orig
    np.random.seed(seed)
    out["llff_seed%d" % seed] = ref_utils.get_rand_poses("llff", original_loader=orig.copy()).numpy()
np.savez_compressed(os.path.join(HERE, "reference_poses.npz"), **out)
print({k: v.shape for k, v in out.items()})
