"""Edge cases through the C ABI on the GPU: empty inputs, rays that miss the box, an empty and a full occupancy grid,
a sample budget of zero rows, wrong dtypes / devices, unsupported shapes -- results checked against the oracle
where there is something to compute, and the reference's error behaviour where there is not
(gridencoder.cu:346-373 "GridEncoding: C must be 1, 2, 4, or 8.", TORCH_CHECKs at :420-436)."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def test_empty_inputs_are_no_ops():
    import pvd_hip as hip
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
    hip.near_far_from_aabb(z(0, 3), z(0, 3), t(np.array([-1, -1, -1, 1, 1, 1], np.float32)), 0, 0.2, z(0), z(0))
    hip.morton3D(z(0, 3, dt=torch.int32), 0, z(0, dt=torch.int32))
    hip.packbits(z(0), 0, 0.5, z(0, dt=torch.uint8))
    counter = z(2, dt=torch.int32)
    hip.march_rays_train(z(0, 3), z(0, 3), z(128 ** 3 // 8, dt=torch.uint8), 1.0, 0.0, 1024, 0, 1, 128, 128, z(0), z(0), z(128, 3), z(128, 3),
                         z(128, 2), z(0, 3, dt=torch.int32), counter, 0)
    assert counter.tolist() == [0, 0]
    hip.composite_rays_train_forward(z(0), z(0, 3), z(0, 2), z(0, 3, dt=torch.int32), 0, 0, z(0), z(0), z(0, 3))
    emb = z(16, 2)
    offs = t(np.array([0, 8, 16], np.int32))
    hip.grid_encode_forward(z(0, 3), emb, offs, z(2, 0, 2), 0, 3, 2, 2, 1.0, 2, False, z(1), 0, False)
    hip.sh_encode_forward(z(0, 3), z(0, 16), 0, 3, 4, False, z(1))
    sig, rgb, feat = z(0), z(0, 3), z(0, 16)
    hip.head_forward(1, z(0, 144, dt=torch.float16), z(0), z(0, 3), 0, z(15, 144), None, z(64, 31), z(64, 64), z(3, 64), -2.0, -2.0, 7.0,
                     sig, rgb, feat)
    flag = z(1)
    hip.check_finite(z(0), flag)
    assert float(flag) == 0.0


def test_rays_that_miss_and_empty_or_full_grid():
    import pvd_hip as hip
    N = 512
    rng = np.random.RandomState(0)
    o = np.tile(np.array([[0.0, 0.0, 3.0]], np.float32), (N, 1))
    d = rng.randn(N, 3).astype(np.float32)
    d[:, 2] = np.abs(d[:, 2]) + 0.1  # pointing away from the box: every ray misses
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[: N // 2, 2] *= -1  # first half points at the box
    d[0] = [0.0, 0.0, -1.0]  # axis-parallel: 1/0 in the slab test
    d[1] = [1.0, 0.0, 0.0]   # passes above the box: early-out to FLT_MAX (raymarching.cu:128-141)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n_ref, f_ref = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
    hip.near_far_from_aabb(t(o), t(d), t(aabb), N, 0.2, nears, fars)
    assert np.array_equal(nears.cpu().numpy(), n_ref) and np.array_equal(fars.cpu().numpy(), f_ref)
    # pointing away: the slabs are crossed behind the origin (far < near: nothing to march); sideways: a true miss
    assert (f_ref[N // 2:] < n_ref[N // 2:]).all() or (n_ref[N // 2:] == np.finfo(np.float32).max).any()
    assert n_ref[1] == np.finfo(np.float32).max and f_ref[1] == np.finfo(np.float32).max
    for fill in (0, 255):
        bits = np.full(128 ** 3 // 8, fill, np.uint8)
        M = N * 1024 if fill else 128
        ref = oracle.march_rays_train(o, d, bits, 1.0, 1, 128, n_ref, f_ref, M, perturb=1)
        xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        counter = torch.zeros(2, dtype=torch.int32, device=dev)
        hip.march_rays_train(t(o), t(d), t(bits), 1.0, 0.0, 1024, N, 1, 128, M, nears, fars, xyzs, dirs, deltas, rays, counter, 1)
        assert np.array_equal(ref[3], rays.cpu().numpy()) and np.array_equal(ref[4], counter.cpu().numpy())
        assert np.array_equal(ref[0], xyzs.cpu().numpy()) and np.array_equal(ref[2], deltas.cpu().numpy())
        r = rays.cpu().numpy()
        assert (r[N // 2:, 2] == 0).all() and r[1, 2] == 0  # missing rays produce no samples
        if fill == 0:
            assert int(counter[0]) == 0
            ws, dep, img = torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, 3, device=dev)
            hip.composite_rays_train_forward(torch.zeros(M, device=dev), torch.zeros(M, 3, device=dev), deltas, rays, M, N, ws, dep, img)
            assert float(ws.abs().max()) == 0.0 and float(img.abs().max()) == 0.0  # rays with no samples composite to zero
        else:
            assert r[: N // 2, 2].max() > 500  # a full grid: hundreds of samples per ray, the record-overflow path


def test_sample_budget_too_small_drops_everything_cleanly():
    import pvd_hip as hip
    from test_hip_parity import _scene_rays
    N = 256
    o, d, bits, C = _scene_rays(N, 9)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n_ref, f_ref = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    M = 1  # no ray fits (strict off + num >= M)
    ref = oracle.march_rays_train(o, d, bits, 1.0, C, 128, n_ref, f_ref, M)
    xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    hip.march_rays_train(t(o), t(d), t(bits), 1.0, 0.0, 1024, N, C, 128, M, t(n_ref), t(f_ref), xyzs, dirs, deltas, rays, counter, 0)
    assert np.array_equal(ref[3], rays.cpu().numpy()) and float(xyzs.abs().max()) == 0.0
    ws, dep, img = torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, 3, device=dev)
    hip.composite_rays_train_forward(torch.ones(M, device=dev), torch.ones(M, 3, device=dev), deltas, rays, M, N, ws, dep, img)
    assert float(ws.abs().max()) == 0.0  # overflowing rays are skipped by the compositor as well (raymarching.cu:523-531)


def test_error_behaviour_matches_the_reference_checks():
    import pvd_hip as hip
    z = lambda *s, dt=torch.float32, d=dev: torch.zeros(*s, dtype=dt, device=d)
    offs = t(np.array([0, 8, 16], np.int32))
    with pytest.raises(RuntimeError, match="C must be 1, 2, 4, or 8"):  # gridencoder.cu:361
        hip.grid_encode_forward(z(4, 3), z(16, 3), offs, z(2, 4, 3), 4, 3, 3, 2, 1.0, 2, False, z(1), 0, False)
    with pytest.raises(RuntimeError):  # D must be 2 or 3 (:369)
        hip.grid_encode_forward(z(4, 4), z(16, 2), offs, z(2, 4, 2), 4, 4, 2, 2, 1.0, 2, False, z(1), 0, False)
    with pytest.raises(RuntimeError, match="no CPU path"):  # CHECK_CUDA (:420)
        hip.grid_encode_forward(z(4, 3, d="cpu"), z(16, 2), offs, z(2, 4, 2), 4, 3, 2, 2, 1.0, 2, False, z(1), 0, False)
    with pytest.raises(RuntimeError):  # CHECK_IS_INT offsets (:432)
        hip.grid_encode_forward(z(4, 3), z(16, 2), offs.long(), z(2, 4, 2), 4, 3, 2, 2, 1.0, 2, False, z(1), 0, False)
    with pytest.raises(RuntimeError, match="contiguous"):  # CHECK_CONTIGUOUS (:425)
        hip.grid_encode_forward(z(4, 6)[:, ::2], z(16, 2), offs, z(2, 4, 2), 4, 3, 2, 2, 1.0, 2, False, z(1), 0, False)
    with pytest.raises(RuntimeError):  # SH degree outside 1..8 (sphere_harmonics.py:75-78)
        hip.sh_encode_forward(z(4, 3), z(4, 81), 4, 3, 9, False, z(1))
    with pytest.raises(RuntimeError):  # plenoxel: C must be 3 * degree^2 + 1
        hip.plenoxel_forward(z(4, 3), None, (-1, -1, -1, 1, 1, 1), torch.zeros(1, 27, 4, 4, 4, device=dev).contiguous(
            memory_format=torch.channels_last_3d), 3, -2.0, 7.0, z(4, 27), None, None, None, None)


def test_round_two_entry_points_reject_bad_arguments_and_accept_empty_inputs():
    """pvd_freq_encode / pvd_mlp_head_forward_fused / pvd_distill_* (fea_width) / pvd_adamw_step_ex (warm list without the deferred
    decay): empty inputs are no-ops, malformed ones are refused by the binding or by the library (never a launch on bad sizes)."""
    import pvd_hip
    dev = torch.device("cuda:0")
    # positional encoding: empty batch, too many frequencies, a row stride shorter than the encoding
    out = pvd_hip.freq_encode(torch.empty(0, 3, device=dev), [1.0, 2.0], True, torch.float16, 16)
    assert out.shape == (0, 16)
    with pytest.raises(pvd_hip.PvdHipError):
        pvd_hip.freq_encode(torch.zeros(4, 3, device=dev), [1.0] * 17)
    with pytest.raises(pvd_hip.PvdHipError):
        pvd_hip.freq_encode(torch.zeros(4, 3, device=dev), [1.0, 2.0], True, torch.float32, 14)
    with pytest.raises(pvd_hip.PvdHipError):
        pvd_hip.freq_encode(torch.zeros(4, 3, device=dev).half(), [1.0])
    # fused NeRF-MLP forward: a weight stream of the wrong length for the stated layer structure
    f = lambda *s: torch.zeros(*s, device=dev)
    with pytest.raises(pvd_hip.PvdHipError):
        pvd_hip.mlp_head_forward_fused(torch.zeros(8, 64, device=dev).half(), torch.zeros(1000, device=dev).half(), 4, 1, f(8, 3), 8,
                                       f(64, 28), f(16, 64), f(64, 31), f(64, 64), f(3, 64), -2.0, 7.0, f(8), f(8, 3), f(8, 16))
    # objective: feature rows must be 16 wide or 1 wide
    S = torch.zeros(4 + 4 * 1024, device=dev)
    img = f(1, 4, 3)
    with pytest.raises(pvd_hip.PvdHipError):
        pvd_hip.distill_sumsq(img, img, f(8, 7), f(8, 7), f(8, 3), f(8, 3), S)
    pvd_hip.distill_sumsq(img, img, f(8, 1), f(8, 1), f(8, 3), f(8, 3), S, reduce=True)
    assert float(S[:4].abs().sum()) == 0.0
    # optimizer: a list of warm groups only makes sense when the cold groups' decay is deferred
    n = 1024
    p, g, m, v = torch.randn(n, device=dev), f(n), f(n), f(n)
    bits = torch.zeros(n // 128, dtype=torch.int32, device=dev)
    step, lr = f(1), torch.tensor([1e-3], device=dev)
    log, cnt = torch.zeros(4, 1, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    with pytest.raises(pvd_hip.PvdHipError):
        pvd_hip.adamw_step(p, g, m, v, [n], lr, 0.9, 0.99, 1e-15, 0.01, step, cold_bits=bits, lazy=(torch.zeros(4, 2, device=dev), cnt))
    warm = torch.arange(n // 4, dtype=torch.int32, device=dev)
    p0 = p.clone()
    pvd_hip.adamw_step(p, g, m, v, [n], lr, 0.9, 0.99, 1e-15, 0.01, step, cold_bits=bits, lazy=(log, cnt, warm))
    ref = (p0.double() - 1e-3 * 0.01 * p0.double())
    assert torch.allclose(p.double(), ref, rtol=1e-6) and int(cnt[0]) == 1 and float(step[0]) == 1.0
