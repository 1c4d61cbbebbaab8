"""The HIP path against what the reference's kernel sources fix without being compiled (tests/golden/reference_constants.npz,
see tests/test_reference_constants.py for the oracle side and tests/golden/make_reference_constants.py for the generator):
SH basis + derivatives evaluated from the reference's own expressions, the hash primes, the PCG32 constants."""
import os

import numpy as np
import pytest
import torch

from test_reference_constants import pcg32_from_constants, sh_bar

pytestmark = pytest.mark.gpu
FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_constants.npz")


@pytest.fixture(scope="module")
def ref():
    return np.load(FIX)


@pytest.fixture(scope="module")
def hip():
    import pvd_hip
    return pvd_hip


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("degree", list(range(1, 9)))
def test_hip_sh_equals_the_reference_polynomials_evaluated_in_source_order(hip, ref, degree):
    dirs = ref["sh_dirs"]
    B, n = len(dirs), degree * degree
    out = torch.empty(B, n, device="cuda")
    dy = torch.empty(B, 3 * n, device="cuda")
    hip.sh_encode_forward(t(dirs), out, B, 3, degree, True, dy)
    want = ref["sh_out"][:, :n]
    got = out.cpu().numpy()
    assert np.all(np.abs(got - want) <= sh_bar(dirs)[:, :n]), np.abs(got - want).max()
    dy = dy.cpu().numpy().reshape(B, 3, n)
    for a, key in enumerate(("sh_dx", "sh_dy", "sh_dz")):
        want = ref[key][:, :n]
        assert np.all(np.abs(dy[:, a] - want) <= sh_bar(dirs, True)[:, :n]), (key, np.abs(dy[:, a] - want).max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_hip_hash_index_uses_the_reference_primes(hip, ref, dtype):
    """as test_oracle_hash_index_uses_the_reference_primes: exact-corner positions on one hashed level, a table of row numbers.
    f16 tables hold integers exactly up to 2048, so the f16 case numbers the rows modulo 2048 (C = 2: row & 2047, row >> 11)."""
    primes = ref["hash_primes"].astype(np.uint64)
    H, size = 129, 1 << 16
    rng = np.random.default_rng(3)
    cells = rng.integers(0, 128, size=(4000, 3))
    x = (cells / 128.0).astype(np.float32)
    rows = np.arange(size)
    table = np.stack([rows & 2047, rows >> 11], 1).astype(np.float32)
    out = torch.empty(1, len(x), 2, dtype=dtype, device="cuda")
    hip.grid_encode_forward(t(x), t(table).to(dtype), t(np.array([0, size], np.int32)), out, len(x), 3, 2, 1, 0.0, H, False, out, 0, True)
    got = out.float().cpu().numpy()[0]
    idx = got[:, 0].astype(np.uint64) + (got[:, 1].astype(np.uint64) << np.uint64(11))
    want = np.zeros(len(cells), np.uint64)
    for d in range(3):
        want ^= (cells[:, d].astype(np.uint64) * primes[d]) & np.uint64(0xFFFFFFFF)
    want %= np.uint64(size)
    assert np.array_equal(idx, want)


def test_hip_march_jitter_is_the_generator_the_reference_constants_define(hip, ref):
    """march_rays_train with perturb: t0 = near + dt_min * pcg32{perturb seed}.advance(n).next_float() (raymarching.cu:346-352).
    In a fully occupied grid the first sample of ray n sits at o + (t0 + 0) * d, so the jitter can be read back."""
    H, N = 32, 64
    bitfield = torch.full((H ** 3 // 8,), 255, dtype=torch.uint8, device="cuda")
    o = np.tile(np.array([[0.1, -0.2, -3.0]], np.float32), (N, 1))
    d = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (N, 1))
    nears, fars = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    hip.near_far_from_aabb(t(o), t(d), t(np.array([-1, -1, -1, 1, 1, 1], np.float32)), N, 0.2, nears, fars)
    M = N * 1024
    xyzs, dirs, deltas = torch.zeros(M, 3, device="cuda"), torch.zeros(M, 3, device="cuda"), torch.zeros(M, 2, device="cuda")
    rays = torch.zeros(N, 3, dtype=torch.int32, device="cuda")
    counter = torch.zeros(2, dtype=torch.int32, device="cuda")
    seed = 42
    hip.march_rays_train(t(o), t(d), bitfield, 1.0, 0.0, 1024, N, 1, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, seed)
    rays, xyzs = rays.cpu().numpy(), xyzs.cpu().numpy()
    near = nears.cpu().numpy()
    dt_min = np.float32(2) * np.float32(ref["rm_sqrt3"][0]) / np.float32(1024)
    for n in range(N):
        _, f = pcg32_from_constants(ref, seed, 1, n, 1)
        t0 = np.float32(near[n] + dt_min * f[0])
        z = np.float32(np.float32(o[n, 2]) + t0 * np.float32(1.0))
        assert xyzs[rays[n, 1], 2] == z, (n, xyzs[rays[n, 1], 2], z)
