"""GPU parity: every entry point of libpvd_hip.so against the CPU oracle on the same seeded inputs.

Bars (DESIGN.md "Parity contract"):
  * integer / index work and everything deterministic that feeds it -- near/far, Morton, packbits,
    the whole marcher output (rays table, xyz, dirs, deltas), grid-encoder forward in f32 AND f16 --
    is BIT-EXACT;
  * floating point with a different evaluation order (SH polynomials, __expf in the compositor,
    atomics in the encoder backward) is within the tolerance written next to each assert;
    north_star's bar is RGB/sigma within 1e-4.
"""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def hip():
    import pvd_hip
    return pvd_hip


def _scene_rays(n_rays, seed, bound=1.0, thicken=0.08):
    from pvd.scene import BLENDER_INTRINSICS, ChairScene, get_rays, packbits_torch, synthetic_poses
    rng = np.random.RandomState(seed)
    poses = torch.from_numpy(synthetic_poses(rng))
    g = torch.Generator().manual_seed(seed)
    r = get_rays(poses[seed % 300][None], BLENDER_INTRINSICS, 800, 800, n_rays, generator=g)
    C = 1 + int(np.ceil(np.log2(bound)))
    grid = ChairScene(thicken=thicken).density_grid(128, bound, C)
    bits = packbits_torch(grid, 10.0)
    o = (r["rays_o"].reshape(-1, 3) * (bound if bound > 1 else 1.0)).contiguous()
    return o.numpy(), r["rays_d"].reshape(-1, 3).contiguous().numpy(), bits.numpy(), C


def t(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def test_near_far_bit_exact(hip, dev):
    o, d, _, _ = _scene_rays(4096, 1)
    d[:7] = [[1, 0, 0], [0, 1, 0], [0, 0, -1], [1e-30, 1, 0], [0, 0, 1], [-1, 0, 0], [0.6, 0.8, 0]]  # axis-parallel / misses
    o[7:9] = [[0.5, 0.5, 5.0], [3.0, 3.0, 3.0]]
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n_ref, f_ref = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    nears, fars = torch.empty(4096, device=dev), torch.empty(4096, device=dev)
    hip.near_far_from_aabb(t(o, dev), t(d, dev), t(aabb, dev), 4096, 0.2, nears, fars)
    assert np.array_equal(n_ref, nears.cpu().numpy()) and np.array_equal(f_ref, fars.cpu().numpy())


def test_morton_packbits_bit_exact(hip, dev):
    rng = np.random.RandomState(0)
    c = rng.randint(0, 1024, (100000, 3)).astype(np.int32)
    idx = torch.empty(c.shape[0], dtype=torch.int32, device=dev)
    hip.morton3D(t(c, dev), c.shape[0], idx)
    assert np.array_equal(oracle.morton3D(c), idx.cpu().numpy())
    back = torch.empty(c.shape[0], 3, dtype=torch.int32, device=dev)
    hip.morton3D_invert(idx, c.shape[0], back)
    assert np.array_equal(back.cpu().numpy(), c)
    g = rng.randn(2 * 128 ** 3).astype(np.float32)
    g[::97] = 0.25
    bf = torch.empty(g.size // 8, dtype=torch.uint8, device=dev)
    hip.packbits(t(g, dev), g.size // 8, 0.25, bf)
    assert np.array_equal(oracle.packbits(g, 0.25), bf.cpu().numpy())


@pytest.mark.parametrize("perturb", [0, 1])
@pytest.mark.parametrize("bound,dt_gamma", [(1.0, 0.0), (2.0, 1.0 / 256)])
def test_march_rays_train_bit_exact(hip, dev, perturb, bound, dt_gamma):
    N = 4096
    o, d, bits, C = _scene_rays(N, 3, bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    n_ref, f_ref = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    M = N * 64
    ref = oracle.march_rays_train(o, d, bits, bound, C, 128, n_ref, f_ref, M, perturb=perturb, dt_gamma=dt_gamma)
    xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    hip.march_rays_train(t(o, dev), t(d, dev), t(bits, dev), bound, dt_gamma, 1024, N, C, 128, M, t(n_ref, dev), t(f_ref, dev),
                         xyzs, dirs, deltas, rays, counter, perturb)
    assert ref[4][0] > 1000, "scene produced too few samples to be a test"
    assert np.array_equal(ref[4], counter.cpu().numpy())
    assert np.array_equal(ref[3], rays.cpu().numpy()), "rays table (id, offset, count) must be bit-exact"
    assert np.array_equal(ref[0], xyzs.cpu().numpy())
    assert np.array_equal(ref[1], dirs.cpu().numpy())
    assert np.array_equal(ref[2], deltas.cpu().numpy())


@pytest.mark.parametrize("N,counter0", [(4096, 0), (4099, 17), (1, 0), (3, 5)])
def test_march_record_pass_equals_three_kernel_path(hip, dev, N, counter0):
    """pvd_march_rays_train_ws (count with chunk records -> write from records, scan folded in) against the
    count / scan / re-march path and the oracle: every output bit for bit, including a non-zero running counter, a
    ray count that is not a multiple of the workgroup's 4 rays, and rays whose chunk records overflow (a dense
    volume: more than 15 sample-bearing chunks per ray)."""
    for dense in (False, True):
        o, d, bits, C = _scene_rays(N, 7)
        if dense:
            bits = np.full_like(bits, 255)
        aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
        n_ref, f_ref = oracle.near_far_from_aabb(o, d, aabb, 0.2)
        M = N * (1024 if dense else 64) + 128
        ref = oracle.march_rays_train(o, d, bits, 1.0, C, 128, n_ref, f_ref, M, perturb=1, counter=np.array([counter0, 3], np.int32))
        outs = []
        for use_ws in (True, False):
            xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
            rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
            counter = torch.tensor([counter0, 3], dtype=torch.int32, device=dev)
            hip.march_rays_train(t(o, dev), t(d, dev), t(bits, dev), 1.0, 0.0, 1024, N, C, 128, M, t(n_ref, dev), t(f_ref, dev),
                                 xyzs, dirs, deltas, rays, counter, 1, use_workspace=use_ws)
            outs.append([x.cpu().numpy() for x in (xyzs, dirs, deltas, rays, counter)])
        for a, b in zip(*outs):
            assert np.array_equal(a, b)
        assert outs[0][4][1] == 3 + N
        for a, b in zip(ref, outs[0]):
            assert np.array_equal(a, b)
        if dense and N >= 3:
            assert outs[0][3][:, 2].max() > 15 * 64 // 2  # long rays: the record overflow path ran


def test_march_overflow_drops_trailing_rays(hip, dev):
    N = 2048
    o, d, bits, C = _scene_rays(N, 5)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n_ref, f_ref = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    full = oracle.march_rays_train(o, d, bits, 1.0, 1, 128, n_ref, f_ref, N * 64)
    M = int(full[4][0]) // 2 // 128 * 128  # budget for about half the samples
    ref = oracle.march_rays_train(o, d, bits, 1.0, 1, 128, n_ref, f_ref, M)
    xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    hip.march_rays_train(t(o, dev), t(d, dev), t(bits, dev), 1.0, 0.0, 1024, N, 1, 128, M, t(n_ref, dev), t(f_ref, dev),
                         xyzs, dirs, deltas, rays, counter, 0)
    assert np.array_equal(ref[3], rays.cpu().numpy()) and np.array_equal(ref[0], xyzs.cpu().numpy())
    r = rays.cpu().numpy()
    dropped = (r[:, 2] > 0) & (r[:, 1] + r[:, 2] >= M)
    assert dropped.any() and not dropped[: np.argmax(dropped)].any()  # strictly the tail


def _samples(N, seed, dev):
    o, d, bits, C = _scene_rays(N, seed)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n_ref, f_ref = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    M = N * 48
    xyzs, dirs, deltas, rays, counter = oracle.march_rays_train(o, d, bits, 1.0, 1, 128, n_ref, f_ref, M, perturb=1)
    m = int(counter[0])
    m += 128 - m % 128
    return xyzs[:m], dirs[:m], deltas[:m], rays


def test_composite_train_fwd_bwd(hip, dev):
    N = 4096
    xyzs, dirs, deltas, rays = _samples(N, 7, dev)
    M = xyzs.shape[0]
    rng = np.random.RandomState(0)
    sig = np.exp(rng.uniform(-2, 7, M)).astype(np.float32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    ws_r, dep_r, img_r = oracle.composite_rays_train_forward(sig, rgb, deltas, rays)
    ws, dep, img = torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, 3, device=dev)
    hip.composite_rays_train_forward(t(sig, dev), t(rgb, dev), t(deltas, dev), t(rays, dev), M, N, ws, dep, img)
    # __expf vs libm expf and fp32 accumulation: well inside north_star's 1e-4
    np.testing.assert_allclose(ws.cpu().numpy(), ws_r, atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(img.cpu().numpy(), img_r, atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(dep.cpu().numpy(), dep_r, atol=1e-5, rtol=1e-5)
    gws = rng.randn(N).astype(np.float32)
    gim = rng.randn(N, 3).astype(np.float32)
    gs_r, gr_r = oracle.composite_rays_train_backward(gws, gim, sig, rgb, deltas, rays, ws_r, img_r)
    gs, gr = torch.zeros(M, device=dev), torch.zeros(M, 3, device=dev)
    hip.composite_rays_train_backward(t(gws, dev), t(gim, dev), t(sig, dev), t(rgb, dev), t(deltas, dev), t(rays, dev),
                                      t(ws_r, dev), t(img_r, dev), M, N, gs, gr)
    np.testing.assert_allclose(gr.cpu().numpy(), gr_r, atol=2e-6, rtol=1e-5)
    scale = np.abs(gs_r).max()
    assert np.abs(gs.cpu().numpy() - gs_r).max() <= 1e-5 * scale


def _table(L_off, C, rng, dtype):
    emb = rng.uniform(-1, 1, (L_off, C)).astype(np.float32) * 0.1
    return emb.astype(dtype)


def _offsets(D, L, per_level_scale, H, log2_hash, align=False):
    from gridencoder.grid import level_offsets
    return np.array(level_offsets(D, L, per_level_scale, H, log2_hash, align), np.int32)


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
@pytest.mark.parametrize("D,C,gridtype,align", [(3, 2, 0, False), (3, 2, 1, False), (2, 2, 0, False), (3, 4, 0, True), (3, 1, 0, False), (3, 8, 0, False)])
def test_grid_encode_forward_bit_exact(hip, dev, dtype, D, C, gridtype, align):
    rng = np.random.RandomState(1)
    L, H = 14, 16
    pls = np.exp2(np.log2(2048 / H) / (L - 1))
    S = float(np.log2(pls))
    offs = _offsets(D, L, pls, H, 19, align)
    emb = _table(offs[-1], C, rng, dtype)
    B = 20000
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[:8] = [[0.0] * D, [1.0] * D, [0.5] * D, [-0.1] + [0.5] * (D - 1), [1.0001] + [0.5] * (D - 1), [1.0] + [0.0] * (D - 1), [0.25] * D, [0.999999] * D]
    ref, dref = oracle.grid_encode_forward(x, emb, offs, S, H, calc_grad_inputs=True, gridtype=gridtype, align_corners=align)
    td = torch.float32 if dtype == np.float32 else torch.float16
    out = torch.empty(L, B, C, dtype=td, device=dev)
    dy = torch.empty(B, L * D * C, dtype=td, device=dev)
    hip.grid_encode_forward(t(x, dev), t(emb, dev), t(offs, dev), out, B, D, C, L, S, H, True, dy, gridtype, align)
    assert np.array_equal(ref.view(np.uint32 if dtype == np.float32 else np.uint16), out.cpu().numpy().view(np.uint32 if dtype == np.float32 else np.uint16))
    assert np.array_equal(dref, dy.cpu().numpy())
    assert np.all(out[:, 3].cpu().numpy() == 0) and np.all(out[:, 4].cpu().numpy() == 0)  # out-of-range inputs -> zeros


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
@pytest.mark.parametrize("D,C", [(3, 2), (2, 4), (3, 1)])
def test_grid_encode_backward(hip, dev, dtype, D, C):
    rng = np.random.RandomState(2)
    L, H = 14, 16
    pls = np.exp2(np.log2(2048 / H) / (L - 1))
    S = float(np.log2(pls))
    offs = _offsets(D, L, pls, H, 19)
    emb = _table(offs[-1], C, rng, dtype)
    B = 30000
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    g = (rng.randn(L, B, C) * (1.0 if dtype == np.float32 else 0.01)).astype(dtype)
    _, dref = oracle.grid_encode_forward(x, emb, offs, S, H, calc_grad_inputs=True)
    ge_ref, gi_ref = oracle.grid_encode_backward(g, x, emb, offs, S, H, dy_dx=dref)
    td = torch.float32 if dtype == np.float32 else torch.float16
    ge = torch.zeros(offs[-1], C, dtype=td, device=dev)
    gi = torch.zeros(B, D, dtype=td, device=dev)
    hip.grid_encode_backward(t(g, dev), t(x, dev), t(emb, dev), t(offs, dev), ge, B, D, C, L, S, H, True, t(dref, dev), gi, 0, False)
    a, b = ge.float().cpu().numpy(), ge_ref.astype(np.float32)
    # order of the atomic adds differs from the oracle's ascending-b order: rounding-level differences only
    tol = 2e-5 if dtype == np.float32 else 2e-2
    assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-6), (np.abs(a - b).max(), np.abs(b).max())
    # conservation: sum of scattered weights == sum of incoming grads of in-range points (per level, channel)
    np.testing.assert_allclose(a.sum(), g.astype(np.float32).sum(), rtol=2e-3 if dtype == np.float32 else 5e-2, atol=1e-2)
    assert np.array_equal(gi.cpu().numpy(), gi_ref)  # input gradient is deterministic -> bit-exact


@pytest.mark.parametrize("coherent", [False, True])
def test_grid_encode_backward_two_lane_kernel(hip, dev, coherent):
    """k_grid_bwd_lps2 (f16, D 3, C 2, no input gradient: the training path) and the thread-per-sample run-merging kernel it
    replaces, both against the oracle.  `coherent`: samples marching along rays, so that whole runs of lanes hit one row (the
    segmented merge among lanes of equal parity) -- and a ragged size."""
    rng = np.random.RandomState(12)
    L, H, D, C = 14, 16, 3, 2
    pls = np.exp2(np.log2(2048 / H) / (L - 1))
    S = float(np.log2(pls))
    offs = _offsets(D, L, pls, H, 19)
    emb = _table(offs[-1], C, rng, np.float16)
    B = 20011
    if coherent:
        n_rays = 400
        o = rng.uniform(0.1, 0.9, (n_rays, 1, 3))
        d = rng.standard_normal((n_rays, 1, 3))
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        steps = np.arange(B // n_rays + 1)[None, :, None] * 1.7e-3
        x = (o + d * steps).reshape(-1, 3)[:B].astype(np.float32)  # some leave [0,1]: skipped by the kernel (gridencoder.cu:254-259)
    else:
        x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    g = (rng.randn(L, B, C) * 0.01).astype(np.float16)
    ge_ref, _ = oracle.grid_encode_backward(g, x, emb, offs, S, H)
    ref = ge_ref.astype(np.float32)
    for knob in (0, 1 << 29):  # two lanes per sample (default) / thread per sample
        hip.grid_set_fwd_kernel(2, 4096 | knob)
        ge = torch.zeros(offs[-1], C, dtype=torch.float16, device=dev)
        dummy = torch.zeros(1, dtype=torch.float16, device=dev)
        hip.grid_encode_backward(t(g, dev), t(x, dev), t(emb, dev), t(offs, dev), ge, B, D, C, L, S, H, False, dummy, dummy, 0, False)
        a = ge.float().cpu().numpy()
        assert np.abs(a - ref).max() <= 2e-2 * np.abs(ref).max(), (knob, np.abs(a - ref).max(), np.abs(ref).max())
        np.testing.assert_allclose(a.sum(), ref.sum(), rtol=5e-2, atol=1e-2)
    hip.grid_set_fwd_kernel()


@pytest.mark.parametrize("gridtype,align", [(0, False), (1, False), (0, True)])
def test_grid_encode_forward_paired_gathers_bit_exact(hip, dev, gridtype, align):
    """k_grid_fwd_pair (knob bit 1: x / x+1 corners from one 8-byte access) == the oracle and == the default kernel."""
    rng = np.random.RandomState(8)
    L, H, D, C = 14, 16, 3, 2
    pls = np.exp2(np.log2(2048 / H) / (L - 1))
    S = float(np.log2(pls))
    offs = _offsets(D, L, pls, H, 19, align)
    emb = _table(offs[-1], C, rng, np.float16)
    B = 30000
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[:5] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 1], [0.999999, 0.25, 0.75]]
    x[5] = [1.5, 0.5, 0.5]  # out of range: zeros
    ref, _ = oracle.grid_encode_forward(x, emb, offs, S, H, gridtype=gridtype, align_corners=align)
    outs = []
    for knob in (2, 0):
        hip.grid_set_variant(knob)
        out = torch.empty(L, B, C, dtype=torch.float16, device=dev)
        hip.grid_encode_forward(t(x, dev), t(emb, dev), t(offs, dev), out, B, D, C, L, S, H, False, out, gridtype, align)
        outs.append(out)
    hip.grid_set_variant(0)
    assert torch.equal(outs[0], outs[1])
    assert np.array_equal(outs[0].cpu().numpy().view(np.uint16), np.ascontiguousarray(ref).view(np.uint16))


@pytest.mark.parametrize("gridtype,align", [(0, False), (1, False), (0, True)])
def test_grid_encode_forward_lanes_per_sample_kernels_bit_exact(hip, dev, gridtype, align):
    """k_grid_fwd_lps<2> (the default for f16 / D 3 / C 2), <4>, one workgroup per work item or persistent, level-major or
    XCD-affine item order, and the thread-per-sample k_grid_fwd: all == the oracle bit for bit (the corners of a sample sit on
    2 / 4 adjacent lanes, products cross by DPP, the sum keeps the reference's corner order); ragged sizes."""
    rng = np.random.RandomState(18)
    L, H, D, C = 14, 16, 3, 2
    pls = np.exp2(np.log2(2048 / H) / (L - 1))
    S = float(np.log2(pls))
    offs = _offsets(D, L, pls, H, 19, align)
    emb = _table(offs[-1], C, rng, np.float16)
    for B in (30001, 77, 1):
        x = rng.uniform(0, 1, (B, D)).astype(np.float32)
        x[:1] = [[1.5, 0.5, 0.5]]  # out of range: zeros
        if B > 10:
            x[1:6] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 1], [0.999999, 0.25, 0.75]]
        ref, _ = oracle.grid_encode_forward(x, emb, offs, S, H, gridtype=gridtype, align_corners=align)
        ref_bits = np.ascontiguousarray(ref).view(np.uint16)
        for lps, persist in ((2, 4096), (2, 0), (2, 96), (4, 0), (4, 1000), (2, 64 | (1 << 30)), (0, 0)):
            hip.grid_set_fwd_kernel(lps, persist)
            out = torch.full((L, B, C), 7.0, dtype=torch.float16, device=dev)
            hip.grid_encode_forward(t(x, dev), t(emb, dev), t(offs, dev), out, B, D, C, L, S, H, False, out, gridtype, align)
            assert np.array_equal(out.cpu().numpy().view(np.uint16), ref_bits), (B, lps, persist)
    hip.grid_set_fwd_kernel()


@pytest.mark.parametrize("dtype,bound", [(np.float32, 1.0), (np.float16, 2.0), (np.float16, 1.0)])
def test_grid_encode_forward_affine_is_the_mapped_forward(hip, dev, dtype, bound):
    """pvd_grid_encode_forward_affine(x, bound, 2*bound) == pvd_grid_encode_forward((x + bound) / (2*bound)) bit for bit,
    and == the oracle on the mapped positions (GridEncoder.forward's input mapping, grid.py:211), incl. out-of-range points."""
    rng = np.random.RandomState(5)
    L, H, D, C = 14, 16, 3, 2
    pls = np.exp2(np.log2(2048 / H) / (L - 1))
    S = float(np.log2(pls))
    offs = _offsets(D, L, pls, H, 19)
    emb = _table(offs[-1], C, rng, dtype)
    B = 20000
    x = rng.uniform(-1.05 * bound, 1.05 * bound, (B, D)).astype(np.float32)
    x[:4] = [[-bound] * 3, [bound] * 3, [0, 0, 0], [bound, -bound, 0.5 * bound]]
    x01 = ((x + np.float32(bound)) / np.float32(2 * bound)).astype(np.float32)
    td = torch.float32 if dtype == np.float32 else torch.float16
    out_a = torch.empty(L, B, C, dtype=td, device=dev)
    out_p = torch.empty(L, B, C, dtype=td, device=dev)
    hip.grid_encode_forward_affine(t(x, dev), bound, 2 * bound, t(emb, dev), t(offs, dev), out_a, B, D, C, L, S, H, 0, False)
    hip.grid_encode_forward(t(x01, dev), t(emb, dev), t(offs, dev), out_p, B, D, C, L, S, H, False, out_p, 0, False)
    assert torch.equal(out_a, out_p)
    ref, _ = oracle.grid_encode_forward(x01, emb, offs, S, H)
    assert np.array_equal(out_a.cpu().numpy().view(np.uint16 if dtype == np.float16 else np.uint32),
                          np.ascontiguousarray(ref).view(np.uint16 if dtype == np.float16 else np.uint32))


@pytest.mark.parametrize("B", [1, 4097, 92928])
def test_grid_encode_forward_affine_pack_is_lookup_plus_the_pack_kernels_image(hip, dev, B):
    """pvd_grid_encode_forward_affine_pack (ABI 6): the same lookup bit for bit, and the hash head's f16 weight image as
    pvd_head_pack_weights writes it -- also when the launch that runs is not the two-lanes-per-sample kernel (the image then comes from a
    launch of its own behind the lookup)."""
    rng = np.random.RandomState(8)
    L, H, D, C = 14, 16, 3, 2
    pls = np.exp2(np.log2(2048 / H) / (L - 1))
    S = float(np.log2(pls))
    offs = _offsets(D, L, pls, H, 19)
    emb = _table(offs[-1], C, rng, np.float16)
    x = rng.uniform(-1.02, 1.02, (B, D)).astype(np.float32)
    W = [torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(dev) for s in ((64, 28), (16, 64), (64, 31), (64, 64), (3, 64))]
    want = hip.head_pack_weights(0, *W)
    ref = torch.empty(L, B, C, dtype=torch.float16, device=dev)
    hip.grid_encode_forward_affine(t(x, dev), 1.0, 2.0, t(emb, dev), t(offs, dev), ref, B, D, C, L, S, H, 0, False)
    try:
        for variant in ((2, 4096), (4, 4096), (0, 0)):  # lanes per sample, persistent workgroups (0: the generic kernel)
            hip.grid_set_fwd_kernel(*variant)
            out = torch.empty(L, B, C, dtype=torch.float16, device=dev)
            image = torch.full((hip.head_image_halfs(0),), float("nan"), dtype=torch.float16, device=dev)
            hip.grid_encode_forward_affine_pack(t(x, dev), 1.0, 2.0, t(emb, dev), t(offs, dev), out, B, D, C, L, S, H, 0, False, (*W, image))
            assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), variant
            assert torch.equal(image.view(torch.int16), want[:image.numel()].view(torch.int16)), variant
    finally:
        hip.grid_set_fwd_kernel()


@pytest.mark.parametrize("bound", [1.0, 2.0, 1.5])
def test_grid_encode_backward_affine_is_the_mapped_backward(hip, dev, bound):
    """pvd_grid_encode_backward_affine(x, bound, 2*bound) (ABI 6) against pvd_grid_encode_backward((x + bound) / (2*bound)) and the oracle on the
    mapped positions: the same rows and weights (f16 atomics: order-dependent rounding, the existing backward bar), identical where a row
    receives ONE contribution; f32 tables / other shapes are refused."""
    rng = np.random.RandomState(6)
    L, H, D, C = 14, 16, 3, 2
    pls = np.exp2(np.log2(2048 / H) / (L - 1))
    S = float(np.log2(pls))
    offs = _offsets(D, L, pls, H, 19)
    B = 20000
    x = rng.uniform(-1.05 * bound, 1.05 * bound, (B, D)).astype(np.float32)
    x[:4] = [[-bound] * 3, [bound] * 3, [0, 0, 0], [bound, -bound, 0.5 * bound]]
    x01 = ((x + np.float32(bound)) / np.float32(2 * bound)).astype(np.float32)
    grad = (rng.standard_normal((L, B, C)) * 0.05).astype(np.float16)
    g_a = torch.zeros(int(offs[-1]), C, dtype=torch.float16, device=dev)
    g_p = torch.zeros(int(offs[-1]), C, dtype=torch.float16, device=dev)
    dummy = g_p[:1]
    hip.grid_encode_backward_affine(t(grad, dev), t(x, dev), bound, 2 * bound, g_a, t(offs, dev), g_a, B, D, C, L, S, H, 0, False)
    hip.grid_encode_backward(t(grad, dev), t(x01, dev), g_p, t(offs, dev), g_p, B, D, C, L, S, H, False, dummy, dummy, 0, False)
    a, p = g_a.float().cpu().numpy(), g_p.float().cpu().numpy()
    scale = np.abs(p).max()
    assert scale > 0 and np.abs(a - p).max() <= 2e-2 * scale
    ref = oracle.grid_encode_backward(grad, x01, np.zeros((int(offs[-1]), C), np.float16), offs, S, H)[0].astype(np.float32)
    assert np.abs(a - ref).max() <= 2e-2 * max(np.abs(ref).max(), 1e-30)
    g32 = torch.zeros(int(offs[-1]), C, dtype=torch.float32, device=dev)
    with pytest.raises(hip.PvdHipError):
        hip.grid_encode_backward_affine(t(grad.astype(np.float32), dev), t(x, dev), bound, 2 * bound, g32, t(offs, dev), g32, B, D, C, L, S, H, 0, False)


@pytest.mark.parametrize("degree", list(range(1, 9)))
def test_sh_encode(hip, dev, degree):
    rng = np.random.RandomState(degree)
    B = 5000
    d = rng.randn(B, 3).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:3] = [[0, 0, 0], [0, 0, 1], [0.3, -0.2, 0.1]]  # padding rows (d = 0) and non-unit input see the same polynomials
    ref, dref = oracle.sh_encode_forward(d, degree, calc_grad_inputs=True)
    out = torch.empty(B, degree ** 2, device=dev)
    dy = torch.empty(B, 3 * degree ** 2, device=dev)
    hip.sh_encode_forward(t(d, dev), out, B, 3, degree, True, dy)
    # fp32 Horner vs the oracle's fp64 evaluation rounded once
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=3e-6, rtol=0)
    np.testing.assert_allclose(dy.cpu().numpy(), dref, atol=5e-5, rtol=1e-5)
    g = rng.randn(B, degree ** 2).astype(np.float32)
    gi = torch.zeros(B, 3, device=dev)
    hip.sh_encode_backward(t(g, dev), t(d, dev), B, 3, degree, t(dref, dev), gi)
    np.testing.assert_allclose(gi.cpu().numpy(), oracle.sh_encode_backward(g, d, degree, dref), atol=1e-4, rtol=1e-5)


def test_inference_trio(hip, dev):
    """Full march -> composite -> compact rounds, HIP vs oracle, comparing the final image per ray."""
    N = 4096
    o, d, bits, C = _scene_rays(N, 11)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    rng = np.random.RandomState(0)

    def shade(xyzs):  # deterministic stand-in for the network
        s = 30.0 * (np.sin(xyzs * 9.0).sum(-1) > 0.2).astype(np.float32) + 0.5
        c = 0.5 + 0.5 * np.sin(xyzs * 5.0 + 1.0)
        return s.astype(np.float32), c.astype(np.float32)

    def run(backend):
        ws, dep, img = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
        alive = np.arange(N, dtype=np.int32)
        rt = nears.copy()
        n_alive, step = N, 0
        while step < 1024 and n_alive > 0:
            n_step = max(min(N // n_alive, 8), 1)
            if backend == "oracle":
                xyzs, dirs, deltas = oracle.march_rays(n_alive, n_step, alive, rt, o, d, 1.0, bits, 1, 128, nears, fars, align=128)
                s, c = shade(xyzs)
                oracle.composite_rays(n_alive, n_step, alive, rt, s, c, deltas, ws, dep, img)
                alive, rt, n_alive = oracle.compact_rays(n_alive, alive, rt)
            else:
                M = n_alive * n_step
                M += 128 - M % 128
                xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
                ta, tt = t(alive, dev), t(rt, dev)
                hip.march_rays(n_alive, n_step, ta, tt, t(o, dev), t(d, dev), 1.0, 0.0, 1024, 1, 128, t(bits, dev), t(nears, dev), t(fars, dev),
                               xyzs, dirs, deltas, 0)
                s, c = shade(xyzs.cpu().numpy())
                tws, tdep, timg = t(ws, dev), t(dep, dev), t(img, dev)
                hip.composite_rays(n_alive, n_step, ta, tt, t(s, dev), t(c, dev), deltas, tws, tdep, timg)
                ws, dep, img = tws.cpu().numpy(), tdep.cpu().numpy(), timg.cpu().numpy()
                na, nt = torch.zeros_like(ta), torch.zeros_like(tt)
                cnt = torch.zeros(1, dtype=torch.int32, device=dev)
                hip.compact_rays(n_alive, na, ta, nt, tt, cnt)
                n_new = int(cnt.item())
                # canonicalise the (legal) cross-workgroup order by ray id
                a, r = na[:n_new].cpu().numpy(), nt[:n_new].cpu().numpy()
                order = np.argsort(a, kind="stable")
                alive, rt, n_alive = np.ascontiguousarray(a[order]), np.ascontiguousarray(r[order]), n_new
            step += n_step
        return ws, dep, img

    ws_r, dep_r, img_r = run("oracle")
    ws_h, dep_h, img_h = run("hip")
    assert ws_r.max() > 0.5
    np.testing.assert_allclose(ws_h, ws_r, atol=1e-5)
    np.testing.assert_allclose(img_h, img_r, atol=1e-5)
    np.testing.assert_allclose(dep_h, dep_r, atol=1e-4, rtol=1e-5)


def test_full_size_properties(hip, dev):
    """BASELINE-size run through the Python operators: size-independent properties."""
    import raymarching
    N = 4096 * 4
    o, d, bits, C = _scene_rays(N, 13)
    to, td_, tb = t(o, dev), t(d, dev), t(bits, dev)
    aabb = torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev)
    nears, fars = raymarching.near_far_from_aabb(to, td_, aabb, 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(to, td_, 1.0, tb, 1, 128, nears, fars, counter, -1, True, 128, True, 0, 1024)
    r = rays.cpu().numpy()
    assert np.array_equal(r[:, 0], np.arange(N))
    assert np.array_equal(r[:, 1], np.concatenate([[0], np.cumsum(r[:-1, 2])]))  # offsets are the exclusive prefix sum
    assert int(counter[0]) == r[:, 2].sum() and int(counter[1]) == N and xyzs.shape[0] % 128 == 0
    m = int(counter[0])
    x = xyzs[:m]
    assert bool((x.abs() <= 1).all()) and bool((deltas[:m, 0] > 0).all()) and bool((deltas[m:] == 0).all())
    # every emitted sample sits in an occupied cell of the bitfield
    cell = ((x + 1) * 64).clamp(0, 127).int()
    idx = raymarching.morton3D(cell).long()
    occ = (tb[idx // 8].int() >> (idx % 8).int()) & 1
    assert bool(occ.all())
    # compositing: an opaque constant-colour field reproduces the colour with weights_sum -> 1
    sig = torch.full((xyzs.shape[0],), 1e4, device=dev)
    rgb = torch.full((xyzs.shape[0], 3), 0.25, device=dev)
    ws, depth, img = raymarching.composite_rays_train(sig, rgb, deltas, rays)
    hit = torch.from_numpy(r[:, 2] > 0).to(dev)
    assert torch.allclose(ws[hit], torch.ones_like(ws[hit]), atol=1e-6) and torch.allclose(img[hit], torch.full_like(img[hit], 0.25), atol=1e-6)
    assert bool((ws[~hit] == 0).all())


@pytest.mark.parametrize("budget", ["half", "slack", "tight", "tiny", "all"])
def test_march_fresh_outputs_equal_zero_filled_outputs(hip, dev, budget):
    """PVD_MARCH_FRESH: outputs and counter handed over UNINITIALISED (here: NaN / garbage) come back bit-identical to the
    zero-filled call and to the oracle -- budgets that drop trailing rays, leave a tail, hold no ray at all, or are the
    worst case N * max_steps; record path and the fallback (no workspace: the library clears up front)."""
    N = 1027
    o, d, bits, C = _scene_rays(N, 11)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n_ref, f_ref = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    total = int(oracle.march_rays_train(o, d, bits, 1.0, C, 128, n_ref, f_ref, N * 1024)[4][0])
    M = {"half": total // 2 // 128 * 128, "slack": (total * 5 // 4) // 128 * 128 + 128, "tight": total, "tiny": 4, "all": N * 1024}[budget]
    ref = oracle.march_rays_train(o, d, bits, 1.0, C, 128, n_ref, f_ref, M, perturb=1)
    for use_ws in (True, False):
        xyzs, dirs, deltas = (torch.full((M, k), float("nan"), device=dev) for k in (3, 3, 2))
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        counter = torch.tensor([123456, -7], dtype=torch.int32, device=dev)
        hip.march_rays_train(t(o, dev), t(d, dev), t(bits, dev), 1.0, 0.0, 1024, N, C, 128, M, t(n_ref, dev), t(f_ref, dev),
                             xyzs, dirs, deltas, rays, counter, 1, use_workspace=use_ws, fresh=True)
        got = [x.cpu().numpy() for x in (xyzs, dirs, deltas, rays, counter)]
        for name, a, b in zip(("xyzs", "dirs", "deltas", "rays", "counter"), ref, got):
            assert np.array_equal(a, b), (name, use_ws)
    r = ref[3]
    dropped = (r[:, 2] > 0) & (r[:, 1] + r[:, 2] >= M)
    assert dropped.any() == (budget in ("half", "tight", "tiny"))


def test_march_wrapper_scratch_counter_takes_the_fresh_path(hip, dev):
    """raymarching.march_rays_train(..., scratch_counter=True) == the reference protocol (counter.zero_() + zero-filled
    outputs), and it really is the no-fill path."""
    import pvd_hip
    import raymarching
    N = 777
    o, d, bits, C = _scene_rays(N, 4)
    to, td_, tb = t(o, dev), t(d, dev), t(bits, dev)
    nears, fars = raymarching.near_far_from_aabb(to, td_, torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev), 0.2)
    calls = []
    real = pvd_hip.raymarching_backend.march_rays_train
    pvd_hip.raymarching_backend.march_rays_train = lambda *a, **k: (calls.append(k.get("fresh", False)), real(*a, **k))[1]
    try:
        outs = []
        for scratch in (True, False):
            junk = [torch.full((40000 * 8,), float("nan"), device=dev) for _ in range(3)]
            del junk
            counter = torch.tensor([999, 999], dtype=torch.int32, device=dev) if scratch else torch.zeros(2, dtype=torch.int32, device=dev)
            res = raymarching.march_rays_train(to, td_, 1.0, tb, C, 128, nears, fars, counter, 40000, True, 128, False, 0, 1024, scratch)
            outs.append([x.clone() for x in res] + [counter.clone()])
    finally:
        pvd_hip.raymarching_backend.march_rays_train = real
    assert calls == [True, False]
    for a, b in zip(*outs):
        assert torch.equal(a, b)
