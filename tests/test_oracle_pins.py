"""What pins the CPU oracle (the reference ships no tests or golden vectors and its kernels cannot
be built here): published known answers, independent third-party arithmetic and closed-form
identities.  CPU only, seconds."""
import os
import numpy as np
import pytest
import torch

import oracle


# ------------------------------------------------------------------ pcg32 / half
def test_pcg32_published_known_answer():
    # pcg32_srandom(42, 54) demo output of the PCG reference implementation (pcg-random.org, pcg32-demo)
    u, _ = oracle.pcg32_stream(42, 0, 6, initseq=54)
    assert [hex(x) for x in u] == ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"]


def test_pcg32_advance_equals_stepping():
    u_all, f_all = oracle.pcg32_stream(42, 0, 300)
    for n in (0, 1, 2, 63, 64, 255, 299):
        u, f = oracle.pcg32_stream(42, n, 1)
        assert u[0] == u_all[n] and f[0] == f_all[n]
    assert (f_all >= 0).all() and (f_all < 1).all()


def test_half_conversion_matches_numpy():
    rng = np.random.RandomState(0)
    xs = np.concatenate([(rng.randn(20000) * 10.0 ** rng.randint(-9, 6, 20000)).astype(np.float32),
                         np.array([0, -0.0, 65504, 65519.99, 65520, 1e-8, 5.96e-8, 2.98e-8, 6.1e-5, np.inf, -np.inf], np.float32)])
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    got = np.array([oracle.f32_to_f16_bits(x) for x in xs], np.uint16)
    assert np.array_equal(got, want)
    hs = np.arange(0, 0x7c01, 7, dtype=np.uint16)
    assert np.array_equal(np.array([oracle.f16_bits_to_f32(h) for h in hs], np.float32), hs.view(np.float16).astype(np.float32))


# ------------------------------------------------------------------ morton / packbits / near-far
def test_morton_roundtrip_and_definition():
    rng = np.random.RandomState(1)
    c = rng.randint(0, 1024, (5000, 3)).astype(np.int32)
    m = oracle.morton3D(c)
    assert np.array_equal(oracle.morton3D_invert(m), c)
    def interleave(x, y, z):
        r = 0
        for b in range(10):
            r |= ((x >> b) & 1) << (3 * b) | ((y >> b) & 1) << (3 * b + 1) | ((z >> b) & 1) << (3 * b + 2)
        return r
    for i in range(50):
        assert int(m[i]) == interleave(*[int(v) for v in c[i]])


def test_packbits_definition():
    g = np.random.RandomState(2).randn(8 * 1000).astype(np.float32)
    bits = oracle.packbits(g, 0.1)
    assert np.array_equal(np.unpackbits(bits, bitorder="little").astype(bool), g > 0.1)


def test_near_far_slab():
    rng = np.random.RandomState(3)
    o = rng.uniform(-3, 3, (2000, 3)).astype(np.float32)
    d = rng.randn(2000, 3).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n, f = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    with np.errstate(divide="ignore"):
        t0, t1 = (aabb[:3] - o) / d, (aabb[3:] - o) / d
    tn, tf = np.minimum(t0, t1).max(1), np.maximum(t0, t1).min(1)
    hit = tn <= tf
    assert np.array_equal(n == np.finfo(np.float32).max, ~hit)
    np.testing.assert_allclose(n[hit], np.maximum(tn[hit], 0.2), rtol=1e-5)
    np.testing.assert_allclose(f[hit], tf[hit], rtol=1e-5)


# ------------------------------------------------------------------ marcher
def _scene(n, seed, bound=1.0):
    from pvd.scene import BLENDER_INTRINSICS, ChairScene, get_rays, packbits_torch, synthetic_poses
    poses = torch.from_numpy(synthetic_poses(np.random.RandomState(seed)))
    r = get_rays(poses[5][None], BLENDER_INTRINSICS, 800, 800, n, generator=torch.Generator().manual_seed(seed))
    C = 1 + int(np.ceil(np.log2(bound)))
    grid = ChairScene().density_grid(128, bound, C)
    return r["rays_o"].reshape(-1, 3).numpy(), r["rays_d"].reshape(-1, 3).numpy(), packbits_torch(grid, 10.0).numpy(), grid.numpy()


def test_march_rays_train_invariants():
    o, d, bits, grid = _scene(1024, 0)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n, f = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    M = 1024 * 128
    for perturb in (0, 1):
        xyzs, dirs, deltas, rays, counter = oracle.march_rays_train(o, d, bits, 1.0, 1, 128, n, f, M, perturb=perturb)
        tot = int(counter[0])
        assert counter[1] == 1024 and tot == rays[:, 2].sum() and tot > 5000
        assert np.array_equal(rays[:, 0], np.arange(1024))
        assert np.array_equal(rays[:, 1], np.concatenate([[0], np.cumsum(rays[:-1, 2])]))
        dt_min = np.float32(2 * np.float32(1.7320508075688772) / np.float32(1024))
        assert np.all(deltas[:tot, 0] == dt_min)  # dt_gamma = 0: constant step (raymarching.cu:346,368)
        assert np.all(deltas[tot:] == 0) and np.all(xyzs[tot:] == 0)
        # every sample lies in an occupied cell and on its ray
        cell = np.clip(((xyzs[:tot] + 1) * 64).astype(np.int32), 0, 127)
        m = oracle.morton3D(cell)
        assert np.all((bits[m // 8] >> (m % 8)) & 1)
        for ray in np.nonzero(rays[:, 2])[0][:40]:
            s, c = rays[ray, 1], rays[ray, 2]
            assert np.array_equal(dirs[s:s + c], np.repeat(d[ray][None], c, 0))
            t = ((xyzs[s:s + c] - o[ray]) * d[ray]).sum(-1)
            assert np.all(np.diff(t) > 0) and t[0] >= n[ray] - 1e-4 and t[-1] < f[ray]
            np.testing.assert_allclose(np.cumsum(deltas[s:s + c, 1]) - deltas[s, 1], t - t[0], atol=2e-4)  # deltas[:,1] telescopes
        if perturb:  # the noise is a pure function of the ray index (pcg32{42}.advance(n))
            first = rays[:, 2] > 0
            again = oracle.march_rays_train(o, d, bits, 1.0, 1, 128, n, f, M, perturb=1)
            assert np.array_equal(again[0], xyzs)


def _bruteforce_walk(o, d, near, far, bits, t0):
    """One ray, dt_gamma = 0, bound = 1, one cascade, H = 128: the reference's loop (raymarching.cu:430-481) restated in numpy
    float32 scalar arithmetic, one operation at a time -- independent of oracle/pvd_oracle.c (no shared code, no lattice /
    probe restructuring).  Returns the samples' (xyz, dt, t - last_t)."""
    f32 = np.float32
    dt = f32(2) * f32(1.7320508075688772) / f32(1024)
    rd = f32(1) / d
    t, last_t, out = f32(t0), f32(t0), []
    while t < far and len(out) < 1024:
        p = np.clip((np.float64(t) * d.astype(np.float64) + o).astype(np.float32), -1, 1)  # fmaf == exact product then ONE rounding
        cell = np.clip((0.5 * (p.astype(np.float64) + 1.0).astype(np.float32).astype(np.float64) * 128).astype(np.float32), 0, 127).astype(np.int32)
        m = int(oracle.morton3D(cell[None])[0])
        if (bits[m // 8] >> (m % 8)) & 1:
            t = f32(t + dt)
            out.append((p.copy(), dt, f32(t - last_t)))
            last_t = t
        else:
            sgn = np.copysign(f32(1), d).astype(np.float32)
            face = ((cell.astype(np.float32) + f32(0.5) + f32(0.5) * sgn) * f32(1 / 128) * f32(2) - f32(1)).astype(np.float32)
            tt = t + max(f32(0), ((face - p).astype(np.float32) * rd).astype(np.float32).min())
            while True:
                t = f32(t + dt)
                if not t < tt:
                    break
    return out


@pytest.mark.parametrize("perturb", [0, 1])
def test_march_samples_match_bruteforce_fixed_step_walk(perturb):
    """dt_gamma = 0, bound = 1: an independent numpy walk in float32 reproduces, per ray, num_steps AND every sample's
    clamped position, dt and `t - last_t` (bit for bit), with and without the per-ray start jitter
    t0 = fma(dt_min, pcg32{42}.advance(n).next_float(), near) (raymarching.cu:346-352; the generator is pinned by its published
    known-answer vector above)."""
    o, d, bits, _ = _scene(256, 1)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n, f = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    xyzs, dirs, deltas, rays, counter = oracle.march_rays_train(o, d, bits, 1.0, 1, 128, n, f, 256 * 1024, perturb=perturb)
    f32 = np.float32
    dt = f32(2) * f32(1.7320508075688772) / f32(1024)
    checked = 0
    for ray in range(256):
        if f[ray] == np.finfo(np.float32).max:
            assert rays[ray, 2] == 0
            continue
        t0 = f32(n[ray])
        if perturb:
            _, noise = oracle.pcg32_stream(42, ray, 1)
            t0 = f32(np.float64(t0) + np.float64(dt) * np.float64(noise[0]))  # fmaf(dt_min, rnd, near): fused in the reference's builds (oracle/_ref's ISA)
        walk = _bruteforce_walk(o[ray], d[ray], n[ray], f[ray], bits, t0)
        s, c = int(rays[ray, 1]), int(rays[ray, 2])
        assert c == len(walk), ray
        if c:
            assert np.array_equal(xyzs[s:s + c], np.stack([w[0] for w in walk])), ray
            assert np.array_equal(deltas[s:s + c, 0], np.array([w[1] for w in walk], np.float32)), ray
            assert np.array_equal(deltas[s:s + c, 1], np.array([w[2] for w in walk], np.float32)), ray
            assert np.array_equal(dirs[s:s + c], np.repeat(d[ray][None], c, 0))
            checked += c
    assert checked == int(counter[0]) and checked > 2000


def _bruteforce_walk_general(o, d, far, bits, t0, bound, C, H, dt_gamma, max_steps=1024, limit=None):
    """The reference's marching loop (raymarching.cu:44-56, 340-481) for ANY bound / cascade count / dt_gamma, one float32
    operation at a time in numpy scalars (fp64 only where the C source promotes: `0.5 * (...) * H`), independent of
    oracle/pvd_oracle.c.  Returns [(xyz, dt, t - last_t)]."""
    f32 = np.float32
    s3 = f32(1.7320508075688772)
    dt_min = f32(f32(2) * s3) / f32(max_steps)
    dt_max = f32(f32(f32(2) * s3) * f32(1 << (C - 1))) / f32(H)
    rd = (f32(1) / d).astype(np.float32)
    rH = f32(1) / f32(H)
    bound = f32(bound)
    gam = f32(dt_gamma)

    def step_of(t):
        return min(max(f32(t * gam), dt_min), dt_max)

    def exponent(v):  # frexpf(v, &e): v = m * 2^e with m in [0.5, 1); 0 -> 0
        return int(np.frexp(f32(v))[1])
    t, last_t, out = f32(t0), f32(t0), []
    limit = max_steps if limit is None else limit  # (the inference march stops after n_step samples; dt_min still comes from max_steps)
    while t < far and len(out) < limit:
        p = np.clip((np.float64(t) * d.astype(np.float64) + o).astype(np.float32), -bound, bound)  # fmaf
        dt = step_of(t)
        lvl_pos = min(C - 1, max(0, exponent(np.abs(p).max())))
        lvl_dt = min(C - 1, max(0, exponent(f32(f32(dt * f32(H)) * f32(0.5)))))
        level = max(lvl_pos, lvl_dt)
        mip_bound = min(f32(1 << level), bound)
        mip_rbound = f32(1) / mip_bound
        v = (p.astype(np.float64) * np.float64(mip_rbound) + 1.0).astype(np.float32)  # fmaf(x, mip_rbound, 1): ONE rounding, as the reference's builds fuse it
        cell = np.clip((0.5 * v.astype(np.float64) * H).astype(np.float32), 0, H - 1).astype(np.int32)
        m = level * H ** 3 + int(oracle.morton3D(cell[None])[0])
        if (bits[m // 8] >> (m % 8)) & 1:
            t = f32(t + dt)
            out.append((p.copy(), dt, f32(t - last_t)))
            last_t = t
        else:
            sgn = np.copysign(f32(1), d).astype(np.float32)
            a = (cell.astype(np.float32) + f32(0.5) + (f32(0.5) * sgn).astype(np.float32)).astype(np.float32)
            face = (((a * rH).astype(np.float32) * f32(2)).astype(np.float32) - f32(1)).astype(np.float32)
            txyz = ((face.astype(np.float64) * np.float64(mip_bound) - p.astype(np.float64)).astype(np.float32) * rd).astype(np.float32)  # fmaf(face, mip_bound, -x) * rd
            tt = f32(t + max(f32(0), txyz.min()))
            while True:
                t = f32(t + step_of(t))
                if not t < tt:
                    break
    return out


@pytest.mark.parametrize("bound,C,dt_gamma", [(2.0, 2, 1.0 / 256), (1.5, 2, 1.0 / 256), (2.0, 2, 0.0), (4.0, 3, 1.0 / 128)])
def test_march_samples_match_bruteforce_walk_with_cascades_and_growing_steps(bound, C, dt_gamma):
    """configs[4]'s marcher (bound > 1: several cascades, mip level from position AND from step size, dt = clamp(t * dt_gamma)):
    the independent numpy walk reproduces every ray's samples -- positions, dt, t - last_t -- bit for bit, for a power-of-two and
    a non-power-of-two bound, with the start jitter on."""
    from pvd.scene import ChairScene, packbits_torch
    H = 128
    o, d, _, _ = _scene(96, 2, bound=bound)
    # the scene scaled into the outer cascades (as tests/test_hip_workloads.py does for configs[4]), thickened so that the
    # coarse cells of the outer cascades hold it
    bits = packbits_torch(ChairScene(thicken=0.1, scale=1.9 if bound <= 2 else 0.95 * bound).density_grid(H, bound, C), 10.0).numpy()
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    n, f = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    xyzs, dirs, deltas, rays, counter = oracle.march_rays_train(o, d, bits, bound, C, H, n, f, 96 * 1024, perturb=1, dt_gamma=dt_gamma)
    f32 = np.float32
    dt_min = f32(f32(2) * f32(1.7320508075688772)) / f32(1024)
    checked, outer = 0, 0
    for ray in range(96):
        if f[ray] == np.finfo(np.float32).max:
            assert rays[ray, 2] == 0
            continue
        _, noise = oracle.pcg32_stream(42, ray, 1)
        t0 = f32(np.float64(n[ray]) + np.float64(dt_min) * np.float64(noise[0]))  # fmaf(dt_min, rnd, near): fused in the reference's builds
        walk = _bruteforce_walk_general(o[ray], d[ray], f[ray], bits, t0, bound, C, H, dt_gamma)
        s, c = int(rays[ray, 1]), int(rays[ray, 2])
        assert c == len(walk), (ray, c, len(walk))
        if c:
            assert np.array_equal(xyzs[s:s + c], np.stack([w[0] for w in walk])), ray
            assert np.array_equal(deltas[s:s + c, 0], np.array([w[1] for w in walk], np.float32)), ray
            assert np.array_equal(deltas[s:s + c, 1], np.array([w[2] for w in walk], np.float32)), ray
            checked += c
            outer += int((np.abs(xyzs[s:s + c]).max(1) > 1.0).sum())
    assert checked == int(counter[0]) and checked > 500 and outer > 50, (checked, outer)
    if dt_gamma > 0:
        assert np.unique(deltas[:checked, 0]).size > 3  # the step does grow with t


@pytest.mark.parametrize("bound,C,dt_gamma", [(1.0, 1, 0.0), (2.0, 2, 1.0 / 256)])
def test_inference_march_matches_bruteforce_walk(bound, C, dt_gamma):
    """march_rays (raymarching.cu:704-811; the inference rounds of run_cuda): from every alive ray's current t, at most n_step
    samples into slots [n * n_step, ...) -- the same independent walk, stopped after n_step samples, reproduces positions, dt and
    `t - last_t` bit for bit, for two consecutive rounds (the second from the t the first one ended at)."""
    from pvd.scene import ChairScene, packbits_torch
    H, N, n_step = 128, 256, 8
    o, d, _, _ = _scene(N, 4, bound=bound)
    bits = packbits_torch(ChairScene(thicken=0.1, scale=1.9 if bound > 1 else 1.0).density_grid(H, bound, C), 10.0).numpy()
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    alive = np.arange(N, dtype=np.int32)
    t_now = nears.copy()
    total = 0
    for rnd in range(2):
        xyzs, dirs, deltas = oracle.march_rays(N, n_step, alive, t_now, o, d, bound, bits, C, H, nears, fars, dt_gamma=dt_gamma)
        t_next = t_now.copy()
        for n in range(N):
            if not t_now[n] < fars[n]:
                assert not deltas[n * n_step:(n + 1) * n_step].any()
                continue
            walk = _bruteforce_walk_general(o[n], d[n], fars[n], bits, t_now[n], bound, C, H, dt_gamma, limit=n_step)
            k = len(walk)
            s = n * n_step
            if k:
                assert np.array_equal(xyzs[s:s + k], np.stack([w[0] for w in walk])), (rnd, n)
                assert np.array_equal(deltas[s:s + k, 0], np.array([w[1] for w in walk], np.float32)), (rnd, n)
                assert np.array_equal(deltas[s:s + k, 1], np.array([w[2] for w in walk], np.float32)), (rnd, n)
                assert np.array_equal(dirs[s:s + k], np.repeat(d[n][None], k, 0))
                # composite_rays advances the ray's t by deltas[:, 1], one float32 add per sample (raymarching.cu:858-870): where
                # the next round starts
                tt = np.float32(t_now[n])
                for w in walk:
                    tt = np.float32(tt + w[2])
                t_next[n] = tt
            assert not deltas[s + k:s + n_step].any() and not xyzs[s + k:s + n_step].any()  # unused slots stay zero
            if k < n_step:
                t_next[n] = fars[n]  # the ray left the box before filling its slots: nothing left to march
            total += k
        t_now = t_next
    assert total > 300


def test_march_overflow_rule():
    o, d, bits, _ = _scene(512, 2)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n, f = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    full = oracle.march_rays_train(o, d, bits, 1.0, 1, 128, n, f, 512 * 1024)
    M = int(full[4][0]) // 2
    xyzs, _, deltas, rays, counter = oracle.march_rays_train(o, d, bits, 1.0, 1, 128, n, f, M)
    assert np.array_equal(rays, full[3]) and np.array_equal(counter, full[4])  # table is independent of M
    for r in range(512):
        s, c = rays[r, 1], rays[r, 2]
        if c and s + c < M:
            assert np.array_equal(xyzs[s:s + c], full[0][s:s + c])
        elif c and s < M:
            assert np.all(xyzs[s:min(s + c, M)] == 0)  # dropped (strict <, raymarching.cu:419)


# ------------------------------------------------------------------ compositing
def _torch_composite(sig, rgb, deltas, rays, N):
    ws, dep, img = [], [], []
    for n in range(N):
        s, c = int(rays[n, 1]), int(rays[n, 2])
        a = 1 - torch.exp(-sig[s:s + c] * deltas[s:s + c, 0])
        T = torch.cumprod(torch.cat([torch.ones(1, dtype=a.dtype), 1 - a]), 0)[:-1]
        w = a * T
        t = torch.cumsum(deltas[s:s + c, 1], 0)
        ws.append(w.sum()); dep.append((w * t).sum()); img.append((w[:, None] * rgb[s:s + c]).sum(0))
    return torch.stack(ws), torch.stack(dep), torch.stack(img)


def test_composite_forward_backward_vs_autograd_float64():
    o, d, bits, _ = _scene(200, 3)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n, f = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    xyzs, dirs, deltas, rays, counter = oracle.march_rays_train(o, d, bits, 1.0, 1, 128, n, f, 200 * 256, perturb=1)
    M = int(counter[0]) + 7
    deltas = deltas[:M]
    rng = np.random.RandomState(0)
    sig = np.exp(rng.uniform(-2, 5, M)).astype(np.float32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    ws, dep, img = oracle.composite_rays_train_forward(sig, rgb, deltas, rays)
    ts, tr = torch.tensor(sig, dtype=torch.float64, requires_grad=True), torch.tensor(rgb, dtype=torch.float64, requires_grad=True)
    ws_t, dep_t, img_t = _torch_composite(ts, tr, torch.tensor(deltas, dtype=torch.float64), rays, 200)
    np.testing.assert_allclose(ws, ws_t.detach().numpy(), atol=2e-6)
    np.testing.assert_allclose(img, img_t.detach().numpy(), atol=2e-6)
    np.testing.assert_allclose(dep, dep_t.detach().numpy(), atol=2e-5, rtol=1e-5)
    assert (rays[:, 2] == 0).any() and np.all(ws[rays[:, 2] == 0] == 0)  # empty rays -> zeros
    gws, gim = rng.randn(200).astype(np.float32), rng.randn(200, 3).astype(np.float32)
    gs, gr = oracle.composite_rays_train_backward(gws, gim, sig, rgb, deltas, rays, ws, img)
    (ws_t * torch.tensor(gws, dtype=torch.float64)).sum().add((img_t * torch.tensor(gim, dtype=torch.float64)).sum()).backward()
    np.testing.assert_allclose(gr, tr.grad.numpy(), atol=1e-5)
    assert np.abs(gs - ts.grad.numpy()).max() <= 2e-5 * np.abs(ts.grad.numpy()).max()


def test_composite_overflowing_ray_is_empty():
    sig = np.ones(10, np.float32); rgb = np.ones((10, 3), np.float32); dl = np.full((10, 2), 0.1, np.float32)
    rays = np.array([[0, 0, 4], [1, 4, 6]], np.int32)  # second ray: offset+count == M -> treated as empty (raymarching.cu:525)
    ws, dep, img = oracle.composite_rays_train_forward(sig, rgb, dl, rays)
    assert ws[0] > 0 and ws[1] == 0 and np.all(img[1] == 0)


# ------------------------------------------------------------------ grid encoder
def _np_index(pg, res, size, hashed):
    pg = pg.astype(np.uint64)
    if hashed:
        idx = (pg[:, 0] * 1) ^ ((pg[:, 1] * 2654435761) & 0xFFFFFFFF) ^ ((pg[:, 2] * 805459861) & 0xFFFFFFFF)
    else:
        idx = pg[:, 0] + pg[:, 1] * (res + 1) + pg[:, 2] * (res + 1) ** 2
    return ((idx & 0xFFFFFFFF) % np.uint64(size)).astype(np.int64)


def test_grid_forward_against_numpy_restatement():
    from gridencoder.grid import level_offsets
    rng = np.random.RandomState(4)
    L, H = 14, 16
    pls = np.exp2(np.log2(2048 / H) / (L - 1))
    S = np.float32(np.log2(pls))
    offs = np.array(level_offsets(3, L, pls, H, 19, False), np.int32)
    emb = (rng.uniform(-1, 1, (offs[-1], 2)) * 0.1).astype(np.float32)
    x = rng.uniform(0, 1, (300, 3)).astype(np.float32)
    out, _ = oracle.grid_encode_forward(x, emb, offs, float(S), H)
    scales, ress = oracle.grid_level_params(L, float(S), H)
    for l in (0, 3, 4, 5, 13):  # dense levels (0-4) and hashed (5-13); SURVEY.md section 4 (ii),(iii)
        size = offs[l + 1] - offs[l]
        hashed = (ress[l] + 1) ** 3 > size
        assert hashed == (l >= 5)
        pos = x.astype(np.float64) * np.float64(scales[l]) + 0.5
        pos = pos.astype(np.float32)
        cell = np.floor(pos)
        fr = (pos - cell).astype(np.float64)
        acc = np.zeros((300, 2))
        for c in range(8):
            bit = np.array([(c >> k) & 1 for k in range(3)])
            w = np.prod(np.where(bit, fr, 1 - fr), axis=1)
            idx = _np_index(cell + bit, int(ress[l]), int(size), hashed)
            acc += w[:, None] * emb[offs[l] + idx]
        np.testing.assert_allclose(out[l], acc, atol=2e-7)
    # level 13 sits on a knife edge (13*S ~ 7 +- 1 ulp -> 2048 or 2049); it is a hashed level, where the
    # resolution only gates the (always exceeded) stride test, so either value gives the same indices
    assert int(ress[0]) == 16 and int(ress[13]) in (2048, 2049)


def test_grid_dense_level_is_trilinear_interpolation():
    """Level 0 of a tiled grid == torch grid_sample (align_corners=True) on the (res+1)^3 lattice shifted by 0.5 cell."""
    from gridencoder.grid import level_offsets
    rng = np.random.RandomState(5)
    H, L = 8, 1
    offs = np.array(level_offsets(3, L, 2.0, H, 19, False), np.int32)
    emb = rng.randn(offs[-1], 2).astype(np.float32)
    x = rng.uniform(0.1, 0.9, (500, 3)).astype(np.float32)
    out, _ = oracle.grid_encode_forward(x, emb, offs, 1.0, H, gridtype=1)
    scales, ress = oracle.grid_level_params(L, 1.0, H)
    n = int(ress[0]) + 1
    vol = torch.tensor(emb[: n ** 3].reshape(n, n, n, 2)).permute(3, 0, 1, 2)[None]  # [1,C,z,y,x]
    pos = torch.tensor(x) * float(scales[0]) + 0.5
    g = (pos / (n - 1) * 2 - 1).view(1, 1, 1, -1, 3)
    ref = torch.nn.functional.grid_sample(vol, g, mode="bilinear", align_corners=True).view(2, -1).T
    np.testing.assert_allclose(out[0], ref.numpy(), atol=2e-6)


def test_grid_backward_is_adjoint_of_forward_and_conserves_mass():
    from gridencoder.grid import level_offsets
    rng = np.random.RandomState(6)
    L, H = 6, 8
    offs = np.array(level_offsets(3, L, 1.7, H, 12, False), np.int32)
    S = float(np.log2(1.7))
    emb = rng.randn(offs[-1], 2).astype(np.float32)
    x = rng.uniform(-0.05, 1.05, (400, 3)).astype(np.float32)
    inside = np.all((x >= 0) & (x <= 1), axis=1)
    g = rng.randn(L, 400, 2).astype(np.float32)
    out, _ = oracle.grid_encode_forward(x, emb, offs, S, H)
    ge, _ = oracle.grid_encode_backward(g, x, emb, offs, S, H)
    # <forward(E), g> == <E, backward(g)> (the encoder is linear in the table)
    np.testing.assert_allclose((out.astype(np.float64) * g).sum(), (emb.astype(np.float64) * ge).sum(), rtol=1e-4)
    ones = np.ones_like(g)
    ge1, _ = oracle.grid_encode_backward(ones, x, emb, offs, S, H)
    np.testing.assert_allclose(ge1.sum(), inside.sum() * L * 2, rtol=1e-5)  # weights of a point sum to 1 per level/channel
    assert np.all(out[:, ~inside] == 0)


def test_grid_input_gradient_matches_finite_differences():
    from gridencoder.grid import level_offsets
    rng = np.random.RandomState(7)
    L, H = 4, 4
    offs = np.array(level_offsets(3, L, 2.0, H, 19, False), np.int32)
    emb = rng.randn(offs[-1], 2).astype(np.float32)
    x = rng.uniform(0.2, 0.8, (50, 3)).astype(np.float32)
    out, dy = oracle.grid_encode_forward(x, emb, offs, 1.0, H, calc_grad_inputs=True)
    dy = dy.reshape(50, L, 3, 2)
    eps = 1e-3
    for dim in range(3):
        xp, xm = x.copy(), x.copy()
        xp[:, dim] += eps; xm[:, dim] -= eps
        fd = (oracle.grid_encode_forward(xp, emb, offs, 1.0, H)[0] - oracle.grid_encode_forward(xm, emb, offs, 1.0, H)[0]) / (2 * eps)
        same_cell = np.abs(fd - dy[:, :, dim].transpose(1, 0, 2)) < 0.05 * np.abs(fd).max()
        assert same_cell.mean() > 0.9  # piecewise-linear: exact except where +-eps crosses a cell face


def test_grid_f16_matches_numpy_float16_emulation():
    from gridencoder.grid import level_offsets
    rng = np.random.RandomState(8)
    L, H = 3, 8
    offs = np.array(level_offsets(3, L, 2.0, H, 19, False), np.int32)
    emb = (rng.randn(offs[-1], 2) * 0.1).astype(np.float16)
    x = rng.uniform(0, 1, (64, 3)).astype(np.float32)
    out, _ = oracle.grid_encode_forward(x, emb, offs, 1.0, H)
    assert out.dtype == np.float16
    scales, ress = oracle.grid_level_params(L, 1.0, H)
    for l in range(L):
        pos = (x.astype(np.float64) * np.float64(scales[l]) + 0.5).astype(np.float32)
        cell = np.floor(pos)
        fr = (pos - cell).astype(np.float32)
        acc = np.zeros((64, 2), np.float16)
        for c in range(8):
            w = np.ones(64, np.float32)
            for k in range(3):
                w = w * (fr[:, k] if (c >> k) & 1 else (np.float32(1) - fr[:, k]))
            bit = np.array([(c >> k) & 1 for k in range(3)])
            idx = _np_index(cell + bit, int(ress[l]), int(offs[l + 1] - offs[l]), False)
            prod = (w[:, None] * emb[offs[l] + idx].astype(np.float32)).astype(np.float16)  # product rounded to half
            acc = (acc.astype(np.float32) + prod.astype(np.float32)).astype(np.float16)     # half add
        assert np.array_equal(out[l].view(np.uint16), acc.view(np.uint16))


def test_grid_unsupported_shapes_raise():
    with pytest.raises(RuntimeError):
        oracle.grid_encode_forward(np.zeros((4, 3), np.float32), np.zeros((64, 3), np.float32), np.array([0, 64], np.int32), 1.0, 4)  # C = 3
    with pytest.raises(RuntimeError):
        oracle.grid_encode_forward(np.zeros((4, 4), np.float32), np.zeros((64, 2), np.float32), np.array([0, 64], np.int32), 1.0, 4)  # D = 4


# ------------------------------------------------------------------ SH
def test_sh_against_scipy_and_reference_spot_values():
    from scipy.special import sph_harm_y
    rng = np.random.RandomState(9)
    d = rng.randn(200, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    out, _ = oracle.sh_encode_forward(d.astype(np.float32), 8)
    d = d.astype(np.float32).astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    theta, phi = np.arccos(np.clip(d[:, 2], -1, 1)), np.arctan2(d[:, 1], d[:, 0])
    for l in range(8):
        for m in range(-l, l + 1):
            Y = sph_harm_y(l, abs(m), theta, phi)  # complex, Condon-Shortley phase included
            ref = Y.real if m == 0 else (np.sqrt(2) * (Y.real if m > 0 else Y.imag))
            np.testing.assert_allclose(out[:, l * l + l + m], ref, atol=3e-6)
    # non-unit inputs see the polynomials the reference hard-codes (r = 1 baked in): shencoder.cu:50,56,:69
    z, _ = oracle.sh_encode_forward(np.zeros((1, 3), np.float32), 4)
    assert abs(z[0, 0] - 0.28209479177387814) < 1e-7 and abs(z[0, 6] + 0.31539156525251999) < 1e-7
    assert np.all(z[0, [1, 2, 3, 4, 5, 7, 8]] == 0)
    p, _ = oracle.sh_encode_forward(np.array([[0.3, -0.2, 0.5]], np.float32), 4)
    x, y, zz = 0.3, -0.2, 0.5
    assert abs(p[0, 1] - (-0.48860251190291987 * y)) < 1e-6 and abs(p[0, 3] - (-0.48860251190291987 * x)) < 1e-6
    assert abs(p[0, 8] - 0.54627421529603959 * (x * x - y * y)) < 1e-6
    assert abs(p[0, 9] - 0.59004358992664352 * y * (-3 * x * x + y * y)) < 1e-6
    assert abs(p[0, 12] - 0.3731763325901154 * zz * (5 * zz * zz - 3)) < 1e-6


def test_sh_gradient_matches_finite_differences():
    rng = np.random.RandomState(10)
    d = rng.randn(50, 3).astype(np.float32)
    _, dy = oracle.sh_encode_forward(d, 6, calc_grad_inputs=True)
    dy = dy.reshape(50, 3, 36)
    eps = 1e-3
    for k in range(3):
        dp, dm = d.copy(), d.copy()
        dp[:, k] += eps; dm[:, k] -= eps
        fd = (oracle.sh_encode_forward(dp, 6)[0] - oracle.sh_encode_forward(dm, 6)[0]) / (2 * eps)
        np.testing.assert_allclose(dy[:, k], fd, atol=2e-2 * max(1.0, np.abs(fd).max()))
    g = rng.randn(50, 36).astype(np.float32)
    gi = oracle.sh_encode_backward(g, d, 6, dy.reshape(50, -1))
    np.testing.assert_allclose(gi, np.einsum("bc,bkc->bk", g, dy), rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------ inference trio
def test_inference_composite_matches_train_composite_when_nothing_terminates():
    o, d, bits, _ = _scene(300, 4)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n, f = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    N = 300
    xyzs, dirs, deltas, rays, counter = oracle.march_rays_train(o, d, bits, 1.0, 1, 128, n, f, N * 1024)
    M = int(counter[0]) + 1
    sig_fn = lambda p: (0.2 + 0.1 * np.sin(p.sum(-1) * 3)).astype(np.float32)  # thin medium: T never drops below 1e-4
    rgb_fn = lambda p: (0.5 + 0.5 * np.sin(p * 4)).astype(np.float32)
    ws_t, dep_t, img_t = oracle.composite_rays_train_forward(sig_fn(xyzs[:M]), rgb_fn(xyzs[:M]), deltas[:M], rays)
    ws, dep, img = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    alive, rt, n_alive, step = np.arange(N, dtype=np.int32), n.copy(), N, 0
    while step < 1024 and n_alive > 0:
        n_step = max(min(N // n_alive, 8), 1)
        x, _, dl = oracle.march_rays(n_alive, n_step, alive, rt, o, d, 1.0, bits, 1, 128, n, f)
        oracle.composite_rays(n_alive, n_step, alive, rt, sig_fn(x), rgb_fn(x), dl, ws, dep, img)
        alive, rt, n_alive = oracle.compact_rays(n_alive, alive, rt)
        step += n_step
    np.testing.assert_allclose(ws, ws_t, atol=2e-6)
    np.testing.assert_allclose(img, img_t, atol=2e-6)
    # the inference compositor integrates absolute t (it starts from rays_t = near, raymarching.cu:840,876),
    # the training one t relative to the first sample (:542,557): they differ by weights_sum * near
    hit = rays[:, 2] > 0
    np.testing.assert_allclose(dep[hit], dep_t[hit] + ws_t[hit] * n[hit], atol=3e-5, rtol=1e-5)


# ------------------------------------------------------------------ model of the HIP marcher's exact lattice jump
def _lattice_advance_model(t, dt, n):
    """numpy-float32 transcription of lattice_advance() in aaai2023-pvd_amd/csrc/raymarching.hip."""
    f32 = np.float32
    bits = lambda v: int(np.array(v, dtype=np.float32).view(np.uint32))
    fb = lambda b: np.array(b, dtype=np.uint32).view(np.float32)[()]
    t, dt = f32(t), f32(dt)
    while n > 0:
        t1 = f32(t + dt); n -= 1
        if n == 0:
            return t1
        t2 = f32(t1 + dt)
        b1, b2 = bits(t1), bits(t2)
        if (b1 >> 23) != (b2 >> 23):
            t = t1
            continue
        n -= 1
        if n == 0:
            return t2
        t3 = f32(t2 + dt)
        b3 = bits(t3)
        if (b3 >> 23) != (b2 >> 23):
            t = t2
            continue
        c = b3 - b2
        if c == 0:
            return t2
        kmax = ((((b2 >> 23) + 1) << 23) - 1 - b2) // c
        k = min(n, kmax)
        t = fb(b2 + k * c)
        n -= k
    return t


def test_lattice_advance_model():
    """The wave-parallel marcher evaluates t_k = fl(...fl(fl(t0+dt)+dt)...) for all lanes at once with an
    integer jump; this pins the jump rule against serial float32 accumulation (incl. ties, crossings)."""
    f32 = np.float32
    rng = np.random.RandomState(0)
    bits = lambda v: int(np.array(v, dtype=np.float32).view(np.uint32))
    for trial in range(4000):
        t0 = f32(rng.uniform(0.0, 8.0)) if trial % 3 else f32(2.0 ** rng.randint(-3, 4) - rng.uniform(0, 0.05))
        dt = f32(2 * f32(1.7320508075688772) / f32(rng.choice([1024, 512, 256, 1000, 777, 4096, 64])))
        if trial % 7 == 0:  # dt whose remainder is exactly half an ulp two binades up: forces the tie rule
            dt = np.array((bits(dt) & ~0x3ff) | 0x200, dtype=np.uint32).view(np.float32)[()]
        n = int(rng.randint(0, 300))
        t = t0
        for _ in range(n):
            t = f32(t + dt)
        assert bits(_lattice_advance_model(t0, dt, n)) == bits(t), (t0, dt, n)


def _probe_model(t, o, d, rd, bits):
    """float32 transcription of Dda::probe_impl<true> for bound = 1, C = 1, H = 128 (raymarching.hip)."""
    f32 = np.float32
    p = np.clip((np.float64(t) * d.astype(np.float64) + o).astype(np.float32), -1, 1)  # fmaf
    cell = np.clip((0.5 * (p + f32(1)).astype(np.float64) * 128).astype(np.float32), 0, 127).astype(np.int32)
    m = int(oracle.morton3D(cell[None])[0])
    occ = bool((bits[m // 8] >> (m % 8)) & 1)
    sgn = np.copysign(f32(1), d).astype(np.float32)
    face = ((cell.astype(np.float32) + f32(0.5) + f32(0.5) * sgn) * f32(1 / 128) * f32(2) - f32(1)).astype(np.float32)
    tt = f32(t + max(f32(0), ((face - p).astype(np.float32) * rd).astype(np.float32).min()))
    return occ, tt, p


def _march_ray_wave_model(t0, far, limit, o, d, bits, dt):
    """Python model of march_ray_wave<WRITE> in raymarching.hip: 64 lattice points per round, control flow
    replayed on the occupancy / skip-target 'ballots'.  Returns [(xyz, dt, t_after - last_t)]."""
    f32 = np.float32
    rd = (f32(1) / d).astype(np.float32)
    out, emitted, last_t, t_base = [], 0, t0, t0
    pending, pending_tt = False, f32(0)
    if limit == 0:
        return out
    while True:
        ts = [_lattice_advance_model(t_base, dt, k) for k in range(64)]
        after = [f32(t + dt) for t in ts]
        valid = [bool(t < far) for t in ts]
        pr = [_probe_model(t, o, d, rd, bits) if v else (False, f32(0), None) for t, v in zip(ts, valid)]
        occ = [v and p[0] for v, p in zip(valid, pr)]
        cur, done, emit = 0, False, []
        if pending:
            ge = [k for k in range(64) if not (ts[k] < pending_tt)]
            if not ge:
                cur = 64
                done = not valid[63]
            else:
                cur, pending = ge[0], False
        while cur < 64 and not done:
            if not valid[cur]:
                done = True
                break
            if occ[cur]:
                e = cur
                while e < 64 and occ[e]:
                    e += 1
                run, room = e - cur, limit - emitted
                if run >= room:
                    run, done = room, True
                emit += list(range(cur, cur + run))
                emitted += run
                cur = e
            else:
                tt = pr[cur][1]
                ge = [k for k in range(cur + 1, 64) if not (ts[k] < tt)]
                if not ge:
                    pending, pending_tt, cur = True, tt, 64
                    done = not valid[63]
                else:
                    cur = ge[0]
        for k in emit:
            out.append((pr[k][2], dt, f32(after[k] - last_t)))
            last_t = after[k]
        if done:
            return out
        t_base = after[63]


def test_wave_marcher_model_equals_serial_oracle():
    """The HIP marcher's wave-parallel formulation (lattice + ballot replay) selects exactly the samples the
    serial reference loop does -- checked here in a float32 Python model against the C oracle, ray by ray."""
    o, d, bits, _ = _scene(96, 6)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n, f = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    xyzs, dirs, deltas, rays, counter = oracle.march_rays_train(o, d, bits, 1.0, 1, 128, n, f, 96 * 1024, perturb=1)
    _, t0f = oracle.pcg32_stream(42, 0, 1)
    dt = np.float32(2) * np.float32(1.7320508075688772) / np.float32(1024)
    checked = 0
    for ray in range(96):
        if f[ray] == np.finfo(np.float32).max:
            assert rays[ray, 2] == 0
            continue
        noise = oracle.pcg32_stream(42, ray, 1)[1][0]
        t0 = np.float32(n[ray] + dt * noise)
        for limit in (1024, 5):
            got = _march_ray_wave_model(t0, f[ray], limit, o[ray], d[ray], bits, dt)
            s, c = rays[ray, 1], min(rays[ray, 2], limit)
            assert len(got) == c, (ray, len(got), c)
            for k, (p, dtk, dl1) in enumerate(got):
                assert np.array_equal(p, xyzs[s + k]) and dtk == deltas[s + k, 0] and dl1 == deltas[s + k, 1], (ray, k)
        checked += 1
    assert checked > 10 and counter[0] > 500


def test_polar_from_ray_closed_form_sphere_hit():
    """pvdo_polar_from_ray (raymarching.cu:164-200) against the geometry it encodes, in float64: the larger root of
    |o + t d| = r is a point ON the sphere in FRONT of the origin, and (theta, phi) decode back to that point with
    y as the up axis (theta from +y, phi = atan2(z, x)); coordinates are normalised to [-1, 1]."""
    import oracle
    rs = np.random.RandomState(11)
    o = (rs.uniform(-1, 1, size=(2000, 3)) * 0.6).astype(np.float32)
    d = rs.standard_normal((2000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::7] *= 2.5  # the quadratic's A term: directions need not be unit
    for r in (2.0, 3.2):
        c = oracle.polar_from_ray(o, d, r).astype(np.float64)
        assert c.shape == (2000, 2) and np.abs(c).max() <= 1.0 + 1e-6
        o64, d64 = o.astype(np.float64), d.astype(np.float64)
        A, B, C = (d64 * d64).sum(1), (o64 * d64).sum(1), (o64 * o64).sum(1) - r * r
        t = (-B + np.sqrt(B * B - A * C)) / A
        assert (t > 0).all()
        p = o64 + t[:, None] * d64
        np.testing.assert_allclose(np.linalg.norm(p, axis=1), r, rtol=1e-12)
        theta, phi = (c[:, 0] + 1) * np.pi / 2, c[:, 1] * np.pi
        q = r * np.stack([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)], axis=1)
        assert np.abs(q - p).max() <= 2e-5 * r
    # axis cases from the centre: +y is the pole (theta 0), +x is (pi/2, 0), +z is (pi/2, pi/2), -y the other pole
    z3 = np.zeros((4, 3), np.float32)
    ax = np.array([[0, 1, 0], [1, 0, 0], [0, 0, 1], [0, -1, 0]], np.float32)
    np.testing.assert_allclose(oracle.polar_from_ray(z3, ax, 2.0), [[-1, 0], [0, 0], [0, 0.5], [1, 0]], atol=1e-7)


@pytest.mark.parametrize("bound,C", [(1.5, 2), (3.0, 3), (2.0, 2), (1.0, 1)])
def test_fma_contraction_sensitivity_of_the_marcher_with_a_non_power_of_two_bound(tmp_path, bound, C):
    """The canonical arithmetic fuses a product into an add only where the source says fmaf().  A CUDA
    build of the reference contracts more (nvcc -fmad=true): `x * mip_rbound + 1` and `(..) * mip_bound - x` in the marcher,
    which is exact for power-of-two bounds only.  How much can that matter?  The same oracle source built with
    -ffp-contract=fast (every contractible product fused) marches a bound-1.5 scene (mip_bound = 1 and 1.5): the two builds
    must agree on which cells are occupied for all but a vanishing fraction of knife-edge samples, and on coordinates to
    float rounding.  (Measured here: no ray of 3000 changes -- sample positions are o + t*d with t advanced in whole dt steps,
    the contractible products only feed the cell index and the skip distance, where an ulp matters on a knife edge only.)"""
    import ctypes
    import shutil
    import subprocess
    import oracle
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("no gcc")
    here = os.path.dirname(os.path.abspath(oracle.__file__))
    so = str(tmp_path / "libpvd_oracle_fma.so")
    cmd = ["gcc", "-O2", "-fPIC", "-std=c11", "-ffp-contract=fast", "-mfma", "-fno-fast-math", "-fopenmp", "-shared", "-o", so,
           os.path.join(here, "pvd_oracle.c"), "-lm"]
    if subprocess.call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) != 0:
        import pytest
        pytest.skip("this host cannot build the -mfma variant")
    rs = np.random.RandomState(5)
    H, N, M = 64, 3000, 3000 * 400  # (bound, C): 1.5 and 3 are the non-power-of-two cases; 2 and 1 the controls
    # occupancy: a thick spherical shell in both cascades
    ax = (np.arange(H) + 0.5) / H * 2 - 1
    grid = np.zeros((C, H ** 3), np.float32)
    for c in range(C):
        b = min(2 ** c, bound)
        X, Y, Z = np.meshgrid(ax * b, ax * b, ax * b, indexing="ij")
        r = np.sqrt(X ** 2 + Y ** 2 + Z ** 2)
        occ = ((r > 0.45 * min(bound, 1.5) / 1.5) & (r < 1.3 * bound / 1.5)).astype(np.float32)
        coords = np.stack(np.meshgrid(np.arange(H), np.arange(H), np.arange(H), indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
        grid[c, oracle.morton3D(coords)] = occ.reshape(-1)
    bits = oracle.packbits(grid, 0.5)
    o = rs.standard_normal((N, 3)).astype(np.float32)
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * (3.0 * max(bound, 1.5) / 1.5)
    d = (-o + rs.standard_normal((N, 3)).astype(np.float32) * 0.6)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    results, nf = [], []
    canonical = oracle.lib()
    fused = ctypes.CDLL(so)
    assert ctypes.cast(fused.pvdo_march_rays_train, ctypes.c_void_p).value != ctypes.cast(canonical.pvdo_march_rays_train, ctypes.c_void_p).value
    for lib in (canonical, fused):
        oracle._lib = lib
        try:
            nf.append(oracle.near_far_from_aabb(o, d, aabb, 0.2))
            x, _, dl, rays, cnt = oracle.march_rays_train(o, d, bits, bound, C, H, nf[-1][0], nf[-1][1], M, perturb=False, dt_gamma=1.0 / 256)
        finally:
            oracle._lib = canonical
        results.append((x, dl, rays, cnt))
    # the slab test's (bound - o) * rd products do contract: near/far move by an ulp or two, never more
    hit = np.isfinite(nf[0][1]) & (nf[0][1] < 1e8)
    assert np.abs(nf[0][0][hit] - nf[1][0][hit]).max() <= 4e-7 * 4 and np.abs(nf[0][1][hit] - nf[1][1][hit]).max() <= 4e-7 * 8
    (x0, d0, r0, c0), (x1, d1, r1, c1) = results
    assert int(c0[0]) > 50 * N // 10  # a real march
    differing = int((r0[:, 2] != r1[:, 2]).sum())
    assert differing <= N // 100, "rays whose sample count depends on the contraction mode: %d of %d" % (differing, N)
    same = r0[:, 2] == r1[:, 2]
    # rays marched identically (all but the knife-edge ones): coordinates agree to rounding of the fused / unfused products
    worst = 0.0
    for n in np.nonzero(same)[0][:500]:
        a, b, k = r0[n, 1], r1[n, 1], r0[n, 2]
        if k:
            worst = max(worst, float(np.abs(x0[a:a + k] - x1[b:b + k]).max()))
    assert worst <= 5e-6, worst
    print("contraction sensitivity: %d of %d rays differ in sample count; worst coordinate difference %.3g" % (differing, N, worst))
