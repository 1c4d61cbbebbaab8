"""Occupancy-grid maintenance against the REFERENCE's own `mark_untrained_grid` / `update_extra_state`
(distill_mutual/renderer.py:561-775), run on the CPU in tests/golden/make_golden_step.py (hash model, grid 16^3, one and two
cascades; the oracle underneath as morton3D / packbits / grid_encode): cells no camera sees, full sweeps, partial updates
(uniform + occupied cells: torch's generator is consumed in the reference's order), the EMA maximum, mean / threshold,
packbits, and the refresh of mean_count from the step counter."""
import os

import numpy as np
import pytest
import torch

from oracle_ops import oracle_ops
from pvd.config import PVDConfig
from pvd.workload import make_model

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_step.npz"), allow_pickle=False)


@pytest.mark.parametrize("bound", [1, 2])
def test_mark_untrained_grid_and_update_extra_state_follow_the_reference(bound):
    pre = "occ_b%d__" % bound
    opt = PVDConfig(model_type="hash", teacher_type="hash", bound=float(bound), PE=6, skip=2, nerf_layer_num=5, nerf_layer_wide=32,
                    resolution0=12, plenoxel_res="[12,12,12]", grid_size=int(G["grid_size"]), density_thresh=10.0, fp16=False)
    torch.manual_seed(0)
    net = make_model(oracle_ops(), opt, "hash", True, torch.device("cpu"))
    sd = {}
    for k in [str(k) for k in G[pre + "keys"]]:
        if "embeddings" in k:
            torch.manual_seed(777)
            sd[k] = (torch.rand(net.state_dict()[k].shape) - 0.5) * 0.6
        else:
            sd[k] = torch.from_numpy(G[pre + "sd__" + k])
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    assert net.cascade == (1 if bound == 1 else 2)

    net.mark_untrained_grid(G[pre + "poses"], G[pre + "intrinsic"])
    np.testing.assert_array_equal(net.density_grid.numpy(), G[pre + "marked"])  # the same cells are -1
    assert (G[pre + "marked"] < 0).any() == (bound == 2)

    for i in G[pre + "calls"]:
        c = pre + "u%d__" % int(i)
        net.iter_density = int(G[c + "iter_density"])
        counts = G[c + "counts"]
        if len(counts):
            net.step_counter.zero_()
            net.step_counter[:len(counts)] = torch.from_numpy(counts)
            net.local_step = len(counts)
        torch.manual_seed(int(G[c + "seed"]))
        net.update_extra_state()
        ref = G[c + "grid"]
        got = net.density_grid.numpy()
        np.testing.assert_array_equal(got < 0, ref < 0)
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-6)
        assert float(net.mean_density) == pytest.approx(float(G[c + "mean_density"]), rel=1e-6)
        flips = np.unpackbits(net.density_bitfield.numpy() ^ G[c + "bitfield"]).sum()
        assert flips <= 2, flips  # (a cell within an ulp of the threshold may land on the other side)
        assert int(net.mean_count) == int(G[c + "mean_count"])
        assert net.local_step == 0 and net.iter_density == int(G[c + "iter_density"]) + 1
