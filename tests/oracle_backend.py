"""Test helper: the CPU oracle dressed as the reference's three ``_backend`` modules, operating in
place on CPU torch tensors.  Lets the test-suite drive the product's host logic (autograd
Functions, renderer, trainer) without a GPU.  TEST INFRASTRUCTURE ONLY."""
import ctypes
import types

import torch

import oracle

_u32, _f32, _int, _vp = ctypes.c_uint32, ctypes.c_float, ctypes.c_int, ctypes.c_void_p


def _p(t):
    if t is None:
        return _vp(0)
    assert not t.is_cuda and t.is_contiguous()
    return _vp(t.data_ptr())


def _L():
    return oracle.lib()


def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
    _L().pvdo_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), _u32(N), _f32(min_near), _p(nears), _p(fars))


def polar_from_ray(rays_o, rays_d, radius, N, coords):
    _L().pvdo_polar_from_ray(_p(rays_o), _p(rays_d), _f32(radius), _u32(N), _p(coords))


def morton3D(coords, N, indices):
    _L().pvdo_morton3D(_p(coords), _u32(N), _p(indices))


def morton3D_invert(indices, N, coords):
    _L().pvdo_morton3D_invert(_p(indices), _u32(N), _p(coords))


def packbits(grid, N, thresh, bitfield):
    _L().pvdo_packbits(_p(grid), _u32(N), _f32(thresh), _p(bitfield))


def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, perturb):
    _L().pvdo_march_rays_train(_p(rays_o), _p(rays_d), _p(grid), _f32(bound), _f32(dt_gamma), _u32(max_steps), _u32(N), _u32(C), _u32(H),
                               _u32(M), _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(deltas), _p(rays), _p(counter), _u32(int(perturb)))


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image):
    _L().pvdo_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(deltas), _p(rays), _u32(M), _u32(N), _p(weights_sum), _p(depth), _p(image))


def composite_rays_train_backward(gws, gimg, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, grad_sigmas, grad_rgbs):
    _L().pvdo_composite_rays_train_backward(_p(gws), _p(gimg), _p(sigmas), _p(rgbs), _p(deltas), _p(rays), _p(weights_sum), _p(image),
                                            _u32(M), _u32(N), _p(grad_sigmas), _p(grad_rgbs))


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, xyzs, dirs, deltas, perturb):
    _L().pvdo_march_rays(_u32(n_alive), _u32(n_step), _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d), _f32(bound), _f32(dt_gamma),
                         _u32(max_steps), _u32(C), _u32(H), _p(grid), _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(deltas), _u32(int(perturb)))


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    _L().pvdo_composite_rays(_u32(n_alive), _u32(n_step), _p(rays_alive), _p(rays_t), _p(sigmas), _p(rgbs), _p(deltas), _p(weights_sum),
                             _p(depth), _p(image))


def compact_rays(n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter):
    _L().pvdo_compact_rays(_u32(n_alive), _p(rays_alive), _p(rays_alive_old), _p(rays_t), _p(rays_t_old), _p(alive_counter))


def _dt(t):
    return {torch.float32: 0, torch.float16: 1}[t.dtype]


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs, dy_dx, gridtype, align_corners):
    rc = _L().pvdo_grid_encode_forward(_p(inputs), _p(embeddings), _p(offsets), _p(outputs), _u32(B), _u32(D), _u32(C), _u32(L), _f32(S), _u32(H),
                                       _int(int(bool(calc_grad_inputs))), _p(dy_dx), _u32(gridtype), _int(int(bool(align_corners))), _int(_dt(embeddings)))
    if rc != 0:
        raise RuntimeError("GridEncoding: C must be 1, 2, 4, or 8.")


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs, dy_dx, grad_inputs, gridtype, align_corners):
    rc = _L().pvdo_grid_encode_backward(_p(grad), _p(inputs), _p(embeddings), _p(offsets), _p(grad_embeddings), _u32(B), _u32(D), _u32(C), _u32(L),
                                        _f32(S), _u32(H), _int(int(bool(calc_grad_inputs))), _p(dy_dx), _p(grad_inputs), _u32(gridtype),
                                        _int(int(bool(align_corners))), _int(_dt(grad_embeddings)))
    if rc != 0:
        raise RuntimeError("GridEncoding: C must be 1, 2, 4, or 8.")


def sh_encode_forward(inputs, outputs, B, D, C, calc_grad_inputs, dy_dx):
    rc = _L().pvdo_sh_encode_forward(_p(inputs), _p(outputs), _u32(B), _u32(D), _u32(C), _int(int(bool(calc_grad_inputs))), _p(dy_dx))
    if rc != 0:
        raise RuntimeError("SH encoder: unsupported shape")


def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
    rc = _L().pvdo_sh_encode_backward(_p(grad), _p(inputs), _u32(B), _u32(D), _u32(C), _p(dy_dx), _p(grad_inputs))
    if rc != 0:
        raise RuntimeError("SH encoder: unsupported shape")


raymarching_backend = types.SimpleNamespace(
    near_far_from_aabb=near_far_from_aabb, polar_from_ray=polar_from_ray, morton3D=morton3D, morton3D_invert=morton3D_invert,
    packbits=packbits, march_rays_train=march_rays_train, composite_rays_train_forward=composite_rays_train_forward,
    composite_rays_train_backward=composite_rays_train_backward, march_rays=march_rays, composite_rays=composite_rays,
    compact_rays=compact_rays)
gridencoder_backend = types.SimpleNamespace(grid_encode_forward=grid_encode_forward, grid_encode_backward=grid_encode_backward)
shencoder_backend = types.SimpleNamespace(sh_encode_forward=sh_encode_forward, sh_encode_backward=sh_encode_backward)
