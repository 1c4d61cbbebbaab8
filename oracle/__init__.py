"""CPU ORACLE for the PVD hot path -- TEST INFRASTRUCTURE ONLY.

numpy-facing ctypes binding of ``oracle/libpvd_oracle.so`` (built from
``oracle/pvd_oracle.c`` by ``oracle/Makefile``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker / timed CPU baseline.  The product
(``aaai2023-pvd_amd/``) never imports it.

PARITY STATUS (round 6): marcher, compositor, Morton / packbits, SH encoder and the
inference trio are pinned by the reference's OWN kernels (``oracle/build_ref.py``
builds raymarching.cu / shencoder.cu for gfx950 the way the reference builds them;
``tests/golden/reference_kernels.npz``, ``tests/test_oracle_ref_kernels.py``); the
grid encoder is "parity unpinned by the reference" (its source does not build on
HIP); see ``oracle/pvd_oracle.h``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpvd_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "pvd_oracle.c")
    hdr = os.path.join(_HERE, "pvd_oracle.h")
    stale = (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    )
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libpvd_oracle.so"])
    return _LIB_PATH


_lib = None

_u32, _i32, _f32 = ctypes.c_uint32, ctypes.c_int32, ctypes.c_float
_p = ctypes.c_void_p


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.pvdo_num_threads.restype = ctypes.c_int
        _lib.pvdo_f32_to_f16.restype = ctypes.c_uint16
        _lib.pvdo_f32_to_f16.argtypes = [_f32]
        _lib.pvdo_f16_to_f32.restype = _f32
        _lib.pvdo_f16_to_f32.argtypes = [ctypes.c_uint16]
        for name in ("pvdo_grid_encode_forward", "pvdo_grid_encode_backward",
                     "pvdo_sh_encode_forward", "pvdo_sh_encode_backward"):
            getattr(_lib, name).restype = ctypes.c_int
    return _lib


def num_threads():
    return int(lib().pvdo_num_threads())


def set_num_threads(n):
    lib().pvdo_set_num_threads(int(n))


def _ptr(a):
    return _p(a.ctypes.data) if a is not None else _p(0)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


# ----------------------------------------------------------------- pcg32
def pcg32_stream(seed, advance, count, initseq=1):
    u = np.empty(count, np.uint32)
    f = np.empty(count, np.float32)
    lib().pvdo_pcg32_stream(ctypes.c_uint64(seed), ctypes.c_uint64(initseq), ctypes.c_int64(advance),
                            _u32(count), _ptr(u), _ptr(f))
    return u, f


# ----------------------------------------------------------------- utils
def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    rays_o, rays_d, aabb = _c(rays_o, np.float32).reshape(-1, 3), _c(rays_d, np.float32).reshape(-1, 3), _c(aabb, np.float32)
    N = rays_o.shape[0]
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    lib().pvdo_near_far_from_aabb(_ptr(rays_o), _ptr(rays_d), _ptr(aabb), _u32(N), _f32(min_near), _ptr(nears), _ptr(fars))
    return nears, fars


def polar_from_ray(rays_o, rays_d, radius):
    rays_o, rays_d = _c(rays_o, np.float32).reshape(-1, 3), _c(rays_d, np.float32).reshape(-1, 3)
    N = rays_o.shape[0]
    coords = np.empty((N, 2), np.float32)
    lib().pvdo_polar_from_ray(_ptr(rays_o), _ptr(rays_d), _f32(radius), _u32(N), _ptr(coords))
    return coords


def morton3D(coords):
    coords = _c(coords, np.int32).reshape(-1, 3)
    out = np.empty(coords.shape[0], np.int32)
    lib().pvdo_morton3D(_ptr(coords), _u32(coords.shape[0]), _ptr(out))
    return out


def morton3D_invert(indices):
    indices = _c(indices, np.int32).reshape(-1)
    out = np.empty((indices.shape[0], 3), np.int32)
    lib().pvdo_morton3D_invert(_ptr(indices), _u32(indices.shape[0]), _ptr(out))
    return out


def packbits(grid, thresh):
    grid = _c(grid, np.float32).reshape(-1)
    assert grid.size % 8 == 0
    out = np.empty(grid.size // 8, np.uint8)
    lib().pvdo_packbits(_ptr(grid), _u32(out.size), _f32(thresh), _ptr(out))
    return out


# ----------------------------------------------------------------- train
def march_rays_train(rays_o, rays_d, bitfield, bound, C, H, nears, fars, M, perturb=False,
                     dt_gamma=0.0, max_steps=1024, counter=None):
    """Returns xyzs[M,3], dirs[M,3], deltas[M,2] (zero padded), rays[N,3], counter[2]."""
    rays_o, rays_d = _c(rays_o, np.float32).reshape(-1, 3), _c(rays_d, np.float32).reshape(-1, 3)
    bitfield = _c(bitfield, np.uint8)
    nears, fars = _c(nears, np.float32), _c(fars, np.float32)
    N = rays_o.shape[0]
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    rays = np.zeros((N, 3), np.int32)
    counter = np.zeros(2, np.int32) if counter is None else _c(counter, np.int32)
    lib().pvdo_march_rays_train(_ptr(rays_o), _ptr(rays_d), _ptr(bitfield), _f32(bound), _f32(dt_gamma), _u32(max_steps),
                                _u32(N), _u32(C), _u32(H), _u32(M), _ptr(nears), _ptr(fars),
                                _ptr(xyzs), _ptr(dirs), _ptr(deltas), _ptr(rays), _ptr(counter), _u32(int(perturb)))
    return xyzs, dirs, deltas, rays, counter


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, N=None):
    sigmas, rgbs, deltas, rays = _c(sigmas, np.float32), _c(rgbs, np.float32), _c(deltas, np.float32), _c(rays, np.int32)
    M, N = sigmas.shape[0], rays.shape[0]
    ws, depth, image = np.empty(N, np.float32), np.empty(N, np.float32), np.empty((N, 3), np.float32)
    lib().pvdo_composite_rays_train_forward(_ptr(sigmas), _ptr(rgbs), _ptr(deltas), _ptr(rays), _u32(M), _u32(N),
                                            _ptr(ws), _ptr(depth), _ptr(image))
    return ws, depth, image


def composite_rays_train_backward(grad_ws, grad_image, sigmas, rgbs, deltas, rays, ws, image):
    grad_ws, grad_image = _c(grad_ws, np.float32), _c(grad_image, np.float32)
    sigmas, rgbs, deltas, rays = _c(sigmas, np.float32), _c(rgbs, np.float32), _c(deltas, np.float32), _c(rays, np.int32)
    ws, image = _c(ws, np.float32), _c(image, np.float32)
    M, N = sigmas.shape[0], rays.shape[0]
    gs, gr = np.zeros(M, np.float32), np.zeros((M, 3), np.float32)
    lib().pvdo_composite_rays_train_backward(_ptr(grad_ws), _ptr(grad_image), _ptr(sigmas), _ptr(rgbs), _ptr(deltas),
                                             _ptr(rays), _ptr(ws), _ptr(image), _u32(M), _u32(N), _ptr(gs), _ptr(gr))
    return gs, gr


# ----------------------------------------------------------------- infer
def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, bitfield, C, H, nears, fars,
               align=-1, perturb=0, dt_gamma=0.0, max_steps=1024):
    rays_alive, rays_t = _c(rays_alive, np.int32), _c(rays_t, np.float32)
    rays_o, rays_d = _c(rays_o, np.float32).reshape(-1, 3), _c(rays_d, np.float32).reshape(-1, 3)
    bitfield, nears, fars = _c(bitfield, np.uint8), _c(nears, np.float32), _c(fars, np.float32)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    lib().pvdo_march_rays(_u32(n_alive), _u32(n_step), _ptr(rays_alive), _ptr(rays_t), _ptr(rays_o), _ptr(rays_d),
                          _f32(bound), _f32(dt_gamma), _u32(max_steps), _u32(C), _u32(H), _ptr(bitfield),
                          _ptr(nears), _ptr(fars), _ptr(xyzs), _ptr(dirs), _ptr(deltas), _u32(int(perturb)))
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    """In place on rays_t, weights_sum, depth, image (must be contiguous float32 numpy arrays)."""
    for a in (rays_t, weights_sum, depth, image):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    rays_alive = _c(rays_alive, np.int32)
    sigmas, rgbs, deltas = _c(sigmas, np.float32), _c(rgbs, np.float32), _c(deltas, np.float32)
    lib().pvdo_composite_rays(_u32(n_alive), _u32(n_step), _ptr(rays_alive), _ptr(rays_t), _ptr(sigmas), _ptr(rgbs),
                              _ptr(deltas), _ptr(weights_sum), _ptr(depth), _ptr(image))


def compact_rays(n_alive, rays_alive_old, rays_t_old):
    rays_alive_old, rays_t_old = _c(rays_alive_old, np.int32), _c(rays_t_old, np.float32)
    rays_alive, rays_t = np.zeros_like(rays_alive_old), np.zeros_like(rays_t_old)
    counter = np.zeros(1, np.int32)
    lib().pvdo_compact_rays(_u32(n_alive), _ptr(rays_alive), _ptr(rays_alive_old), _ptr(rays_t), _ptr(rays_t_old), _ptr(counter))
    return rays_alive, rays_t, int(counter[0])


# ----------------------------------------------------------------- grid
def grid_level_params(L, S, H):
    scales, ress = np.empty(L, np.float32), np.empty(L, np.uint32)
    lib().pvdo_grid_level_params(_u32(L), _f32(S), _u32(H), _ptr(scales), _ptr(ress))
    return scales, ress


def _table(a):
    """float32 -> dtype 0; float16 -> dtype 1 (passed as raw bits)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.float16:
        return a, 1, np.float16
    return _c(a, np.float32), 0, np.float32


def grid_encode_forward(inputs, embeddings, offsets, S, H, calc_grad_inputs=False, gridtype=0, align_corners=False):
    """Returns outputs [L,B,C] (reference layout) and dy_dx [B, L*D*C] or None."""
    inputs = _c(inputs, np.float32)
    B, D = inputs.shape
    emb, dtype, npdt = _table(embeddings)
    C = emb.shape[1]
    offsets = _c(offsets, np.int32)
    L = offsets.shape[0] - 1
    out = np.empty((L, B, C), npdt)
    dy_dx = np.empty((B, L * D * C), npdt) if calc_grad_inputs else None
    rc = lib().pvdo_grid_encode_forward(_ptr(inputs), _ptr(emb), _ptr(offsets), _ptr(out), _u32(B), _u32(D), _u32(C), _u32(L),
                                        _f32(S), _u32(H), ctypes.c_int(int(calc_grad_inputs)), _ptr(dy_dx),
                                        _u32(gridtype), ctypes.c_int(int(align_corners)), ctypes.c_int(dtype))
    if rc != 0:
        raise RuntimeError("GridEncoding: unsupported D/C/L (gridencoder.cu:350-356,367-371)")
    return out, dy_dx


def grid_encode_backward(grad, inputs, embeddings, offsets, S, H, dy_dx=None, gridtype=0, align_corners=False):
    """grad is [L,B,C].  Returns grad_embeddings (same dtype as table) and grad_inputs or None."""
    inputs = _c(inputs, np.float32)
    B, D = inputs.shape
    emb, dtype, npdt = _table(embeddings)
    C = emb.shape[1]
    offsets = _c(offsets, np.int32)
    L = offsets.shape[0] - 1
    grad = np.ascontiguousarray(grad, dtype=npdt)
    assert grad.shape == (L, B, C)
    ge = np.zeros_like(emb)
    calc = dy_dx is not None
    gi = np.zeros((B, D), npdt) if calc else None
    if calc:
        dy_dx = np.ascontiguousarray(dy_dx, dtype=npdt)
    rc = lib().pvdo_grid_encode_backward(_ptr(grad), _ptr(inputs), _ptr(emb), _ptr(offsets), _ptr(ge), _u32(B), _u32(D), _u32(C),
                                         _u32(L), _f32(S), _u32(H), ctypes.c_int(int(calc)), _ptr(dy_dx), _ptr(gi),
                                         _u32(gridtype), ctypes.c_int(int(align_corners)), ctypes.c_int(dtype))
    if rc != 0:
        raise RuntimeError("GridEncoding: unsupported D/C/L")
    return ge, gi


# ----------------------------------------------------------------- SH
def sh_encode_forward(inputs, degree, calc_grad_inputs=False):
    inputs = _c(inputs, np.float32)
    B, D = inputs.shape
    out = np.empty((B, degree * degree), np.float32)
    dy_dx = np.empty((B, D * degree * degree), np.float32) if calc_grad_inputs else None
    rc = lib().pvdo_sh_encode_forward(_ptr(inputs), _ptr(out), _u32(B), _u32(D), _u32(degree),
                                      ctypes.c_int(int(calc_grad_inputs)), _ptr(dy_dx))
    if rc != 0:
        raise RuntimeError("SH encoder: input dim must be 3 and degree in [1, 8]")
    return out, dy_dx


def sh_encode_backward(grad, inputs, degree, dy_dx):
    grad, inputs, dy_dx = _c(grad, np.float32), _c(inputs, np.float32), _c(dy_dx, np.float32)
    B, D = inputs.shape
    gi = np.zeros((B, D), np.float32)
    rc = lib().pvdo_sh_encode_backward(_ptr(grad), _ptr(inputs), _u32(B), _u32(D), _u32(degree), _ptr(dy_dx), _ptr(gi))
    if rc != 0:
        raise RuntimeError("SH encoder: bad shape")
    return gi


def f32_to_f16_bits(x):
    return int(lib().pvdo_f32_to_f16(float(x)))


def f16_bits_to_f32(h):
    return float(lib().pvdo_f16_to_f32(int(h)))


# ----------------------------------------------------------------- the head under autocast (network.py:413-437, 344-381)
def head_forward_amp(kind, x0, sigma_raw, dirs, Wa1, Wa2, Wc1, Wc2, Wc3, clip_sigma_min=-2.0, clip_feat_min=-2.0, clip_max=7.0):
    """kind 0: x0 [M,28] float16 (encoder / trunk output), Wa1 [64,28], Wa2 [16,64];  kind 1 (vm): x0 [M,144] float16 products,
    sigma_raw [M], Wa1 [15,144].  Returns sigma [M], rgb [M,3], feature_sigma_color [M,16] (float32 holding the f16 values)."""
    x0 = np.ascontiguousarray(x0, dtype=np.float16)
    M = x0.shape[0]
    assert x0.shape == (M, 28 if kind == 0 else 144)
    dirs = _c(dirs, np.float32).reshape(M, 3)
    f = lambda a: None if a is None else _c(a, np.float32)
    Wa1, Wa2, Wc1, Wc2, Wc3, sigma_raw = f(Wa1), f(Wa2), f(Wc1), f(Wc2), f(Wc3), f(sigma_raw)
    assert Wa1.shape == ((64, 28) if kind == 0 else (15, 144)) and Wc1.shape == (64, 31) and Wc2.shape == (64, 64) and Wc3.shape == (3, 64)
    assert (Wa2.shape == (16, 64)) if kind == 0 else (sigma_raw is not None and sigma_raw.shape == (M,))
    sigma, rgb, feat = np.empty(M, np.float32), np.empty((M, 3), np.float32), np.empty((M, 16), np.float32)
    L = lib()
    L.pvdo_head_forward_amp.restype = ctypes.c_int
    rc = L.pvdo_head_forward_amp(ctypes.c_int(kind), _ptr(x0.view(np.uint16)), _ptr(sigma_raw), _ptr(dirs), _u32(M), _ptr(Wa1), _ptr(Wa2),
                                 _ptr(Wc1), _ptr(Wc2), _ptr(Wc3), _f32(clip_sigma_min), _f32(clip_feat_min), _f32(clip_max),
                                 _ptr(sigma), _ptr(rgb), _ptr(feat))
    assert rc == 0, rc
    return sigma, rgb, feat


# ----------------------------------------------------------------- the VM plane x line lookup (network.py:216-309)
def vm_forward(xyz, aabb, tables, res):
    """tables: 12 arrays in the reference's layout -- sigma_mat[3] [16,H,W], sigma_vec[3] [16,L], color_mat[3] [48,H,W], color_vec[3] [48,L]
    (leading 1 and trailing 1 dimensions of the reference's [1,R,H,W] / [1,R,L,1] parameters are accepted); res = (res_x, res_y, res_z).
    Returns sigma_feat [M] and color_prod [M,144] (float32)."""
    xyz = _c(xyz, np.float32).reshape(-1, 3)
    M = xyz.shape[0]
    aabb = _c(aabb, np.float32).reshape(6)
    mat_ids, vec_ids = ((0, 1), (0, 2), (1, 2)), (2, 1, 0)
    keep = []
    for k, R in ((0, 16), (1, 48)):
        for i in range(3):
            t = _c(tables[6 * k + i], np.float32).reshape(R, int(res[mat_ids[i][1]]), int(res[mat_ids[i][0]]))
            keep.append(t)
        for i in range(3):
            keep.append(_c(tables[6 * k + 3 + i], np.float32).reshape(R, int(res[vec_ids[i]])))
    ptrs = (ctypes.c_void_p * 12)(*[t.ctypes.data for t in keep])
    resa = (ctypes.c_uint32 * 3)(*[int(r) for r in res])
    sig, prod = np.empty(M, np.float32), np.empty((M, 144), np.float32)
    L = lib()
    L.pvdo_vm_forward.restype = ctypes.c_int
    rc = L.pvdo_vm_forward(_ptr(xyz), _u32(M), _ptr(aabb), ptrs, resa, _ptr(sig), _ptr(prod))
    assert rc == 0
    return sig, prod
