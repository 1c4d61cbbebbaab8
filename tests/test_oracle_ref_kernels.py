"""The CPU oracle against the REFERENCE'S OWN KERNELS.

tests/golden/reference_kernels.npz holds seeded inputs and what raymarching.cu / shencoder.cu of the reference computed for them -- the
reference's sources built for gfx950 by oracle/build_ref.py (PyTorch-ROCm's torch.utils.cpp_extension.load, the recipe of the
reference's backend.py files) and run on an MI355X by tests/golden/make_golden_ref_kernels.py.  This file pins the oracle with them, on
the CPU: everything that is integer / index / position work bit for bit, the transcendental and summation-order pieces to a few ulp.
(The grid encoder is not in the fixture: its backward calls atomicAdd(__half2*), which HIP does not provide -- unbuildable here.)"""
import os

import numpy as np
import pytest

import oracle

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kernels.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(PATH), reason="fixture not generated yet")


@pytest.fixture(scope="module")
def G():
    return dict(np.load(PATH))


def _by_ray(rays, xyzs, dirs, deltas, N):
    counts = np.zeros(N, np.int32)
    parts = {}
    for idx, off, num in rays:
        counts[idx] = num
        parts[int(idx)] = (xyzs[off:off + num], dirs[off:off + num], deltas[off:off + num])
    order = [parts[i] for i in range(N) if i in parts]
    c = lambda k: np.concatenate([p[k] for p in order])  # noqa: E731
    return counts, c(0), c(1), c(2)


def test_near_far_morton_packbits_are_the_reference_kernels_bits(G):
    n, f = oracle.near_far_from_aabb(G["nf_o"], G["nf_d"], G["nf_aabb"], 0.2)
    assert np.array_equal(n, G["nf_nears"]) and np.array_equal(f, G["nf_fars"])  # incl. the misses (FLT_MAX) and the axis-parallel rays
    assert (G["nf_nears"] > 1e30).sum() >= 1 and (G["nf_fars"] < 0).sum() >= 4  # a miss (FLT_MAX both) and rays looking away
    assert np.array_equal(oracle.morton3D(G["mo_coords"]), G["mo_idx"]) and np.array_equal(oracle.morton3D_invert(G["mo_idx"]), G["mo_back"])
    assert np.array_equal(oracle.packbits(G["pb_grid"], 10.0), G["pb_bits"])  # every 7th value sits exactly on the threshold
    pol = oracle.polar_from_ray(G["nf_o"], G["nf_d"], 2.0)  # (rays that miss the sphere: NaN on both sides; acos / atan2: libm vs the device's)
    assert np.array_equal(np.isnan(pol), np.isnan(G["polar"])) and np.nanmax(np.abs(pol - G["polar"])) <= 5e-7


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("perturb", [0, 1])
def test_march_rays_train_is_the_reference_kernel_bit_for_bit(G, tag, perturb):
    """kernel_march_rays_train (raymarching.cu:313-483) itself: per ray the same number of samples, the same clamped positions, the same
    (dt, t - last_t), with the PCG32 start jitter -- bound 1 / one cascade / constant step (a) and bound 2 / two cascades / dt_gamma
    1/256 (b).  Order-free: the reference hands out rows and offsets with atomics."""
    bound, C, dtg = float(G["m%s_cfg" % tag][0]), int(G["m%s_cfg" % tag][1]), float(G["m%s_cfg" % tag][2])
    o, d = G["m%s_o" % tag], G["m%s_d" % tag]
    cnt = G["m%s%d_counts" % (tag, perturb)]
    M = int(cnt.sum()) + 128
    x, dd, dl, rays, counter = oracle.march_rays_train(o, d, G["m%s_bits" % tag], bound, C, 128, G["m%s_nears" % tag], G["m%s_fars" % tag], M,
                                                       perturb=bool(perturb), dt_gamma=dtg)
    got = _by_ray(rays, x, dd, dl, o.shape[0])
    assert int(cnt.sum()) > 5000 and int((cnt == 0).sum()) >= 4
    assert np.array_equal(got[0], cnt) and int(counter[0]) == int(cnt.sum())
    assert np.array_equal(got[1], G["m%s%d_xyzs" % (tag, perturb)]) and np.array_equal(got[3], G["m%s%d_deltas" % (tag, perturb)])
    assert np.array_equal(got[2], np.repeat(d, cnt, axis=0))


def test_composite_rays_train_matches_the_reference_kernels(G):
    ws, dep, img = oracle.composite_rays_train_forward(G["cp_sig"], G["cp_rgb"], G["cp_deltas"], G["cp_rays"])
    assert np.abs(ws - G["cp_ws"]).max() <= 1e-6 and np.abs(dep - G["cp_depth"]).max() <= 2e-6 and np.abs(img - G["cp_image"]).max() <= 1e-6
    gs, gr = oracle.composite_rays_train_backward(G["cp_gws"], G["cp_gimg"], G["cp_sig"], G["cp_rgb"], G["cp_deltas"], G["cp_rays"], G["cp_ws"], G["cp_image"])
    assert np.abs(gr - G["cp_grgb"]).max() <= 1e-6
    assert np.abs(gs - G["cp_gsig"]).max() <= 2e-6 * np.abs(G["cp_gsig"]).max() + 1e-9 and np.abs(G["cp_gsig"]).max() > 0


@pytest.mark.parametrize("deg", range(1, 9))
def test_sh_encoder_matches_the_reference_kernels(G, deg):
    out, dy = oracle.sh_encode_forward(G["sh_dirs"], deg, True)
    gi = oracle.sh_encode_backward(G["sh%d_g" % deg], G["sh_dirs"], deg, G["sh%d_dy" % deg])
    assert np.abs(out - G["sh%d_out" % deg]).max() <= 5e-6       # (fp32 evaluation order; the oracle's own bar vs fp64 is 3e-6)
    assert np.abs(dy - G["sh%d_dy" % deg]).max() <= 5e-5
    assert np.abs(gi - G["sh%d_gi" % deg]).max() <= 2e-5 * (1 + np.abs(G["sh%d_gi" % deg]).max())


def test_inference_trio_matches_the_reference_kernels(G):
    alive = np.arange(1024, dtype=np.int32)
    for perturb in (0, 1):
        x, dd, dl = oracle.march_rays(1024, 4, alive, G["inf_nears"].copy(), G["inf_o"], G["inf_d"], 1.0, G["inf_bits"], 1, 128, G["inf_nears"], G["inf_fars"],
                                      perturb=perturb)
        assert np.array_equal(x, G["inf%d_xyzs" % perturb]) and np.array_equal(dl, G["inf%d_deltas" % perturb])
    rt, ws, dep, img = G["inf_nears"].copy(), np.zeros(1024, np.float32), np.zeros(1024, np.float32), np.zeros((1024, 3), np.float32)
    al = alive.copy()
    oracle.composite_rays(1024, 4, al, rt, G["inf_sig"], G["inf_rgb"], G["inf1_deltas"], ws, dep, img)
    assert np.array_equal(al, G["inf_alive_after"]) and np.array_equal(rt, G["inf_t_after"])  # which rays ended, and where the others stand
    assert np.abs(ws - G["inf_ws"]).max() <= 1e-6 and np.abs(dep - G["inf_depth"]).max() <= 2e-6 and np.abs(img - G["inf_image"]).max() <= 1e-6
    ca, ct, k = oracle.compact_rays(1024, G["inf_alive_after"], G["inf_t_after"])
    order = np.argsort(ca[:k])
    assert k == G["inf_compact_alive"].shape[0] and 0 < k < 1024
    assert np.array_equal(ca[:k][order], G["inf_compact_alive"]) and np.array_equal(ct[:k][order], G["inf_compact_t"])
