"""``raymarching`` -- the Python operator surface of the reference's raymarching package
(raymarching/raymarching.py), re-hosted on libpvd_hip.so.

Same names, positional signatures, return values and allocation rules as the reference
(cited per function), so ``renderer.py``-style callers run unchanged:

    near_far_from_aabb, polar_from_ray, morton3D, morton3D_invert, packbits,
    march_rays_train, composite_rays_train, march_rays, composite_rays, compact_rays

The autograd Functions are built by :func:`make_ops` around a *backend* object that has the
reference's ``_backend`` function set.  The module-level operators are bound to the HIP
backend only (``pvd_hip.raymarching_backend``); there is no CPU fallback.  ``make_ops`` exists
so the test-suite can drive the same host logic with the CPU oracle.
"""
import types

import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd


def make_ops(backend, device_type="cuda"):
    """Build the ten operators on top of ``backend``.

    device_type="cuda": inputs that are not on the GPU are moved there, as the reference
    does (raymarching.py:35-38, 225-230).  The test-suite passes "cpu" with the oracle.
    """
    fwd32 = custom_fwd(device_type=device_type, cast_inputs=torch.float32)
    bwd = custom_bwd(device_type=device_type)

    def _to_dev(t):
        if device_type == "cuda" and not t.is_cuda:
            return t.cuda()
        return t

    class _NearFar(Function):
        # reference: _near_far_from_aabb, raymarching.py:20-53
        @staticmethod
        @fwd32
        def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
            rays_o = _to_dev(rays_o).contiguous().view(-1, 3)
            rays_d = _to_dev(rays_d).contiguous().view(-1, 3)
            N = rays_o.shape[0]
            nears = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
            fars = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
            backend.near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars)
            return nears, fars

    class _Polar(Function):
        # reference: _polar_from_ray, raymarching.py:56-87
        @staticmethod
        @fwd32
        def forward(ctx, rays_o, rays_d, radius):
            rays_o = _to_dev(rays_o).contiguous().view(-1, 3)
            rays_d = _to_dev(rays_d).contiguous().view(-1, 3)
            N = rays_o.shape[0]
            coords = torch.empty(N, 2, dtype=rays_o.dtype, device=rays_o.device)
            backend.polar_from_ray(rays_o, rays_d, radius, N, coords)
            return coords

    class _Morton(Function):
        # reference: _morton3D, raymarching.py:90-113
        @staticmethod
        def forward(ctx, coords):
            coords = _to_dev(coords)
            N = coords.shape[0]
            indices = torch.empty(N, dtype=torch.int32, device=coords.device)
            backend.morton3D(coords.int().contiguous(), N, indices)
            return indices

    class _MortonInvert(Function):
        # reference: _morton3D_invert, raymarching.py:116-138
        @staticmethod
        def forward(ctx, indices):
            indices = _to_dev(indices)
            N = indices.shape[0]
            coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
            backend.morton3D_invert(indices.int().contiguous(), N, coords)
            return coords

    class _Packbits(Function):
        # reference: _packbits, raymarching.py:141-169
        @staticmethod
        @fwd32
        def forward(ctx, grid, thresh, bitfield=None):
            grid = _to_dev(grid).contiguous()
            C, H3 = grid.shape[0], grid.shape[1]
            N = C * H3 // 8
            if bitfield is None:
                bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
            backend.packbits(grid, N, thresh, bitfield)
            return bitfield

    class _MarchTrain(Function):
        # reference: _march_rays_train, raymarching.py:176-289
        @staticmethod
        @fwd32
        def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                    perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024, scratch_counter=False, budget=None):
            rays_o = _to_dev(rays_o).contiguous().view(-1, 3)
            rays_d = _to_dev(rays_d).contiguous().view(-1, 3)
            density_bitfield = _to_dev(density_bitfield).contiguous()
            N = rays_o.shape[0]
            M = N * max_steps  # worst case (:232)
            # running-average sample budget; overflowing rays are dropped (:234-238)
            if not force_all_rays and mean_count > 0:
                if align > 0:
                    mean_count += align - mean_count % align
                M = mean_count
            # budget = (rows to allocate, DEVICE int32 logical budget): an extension for captured training steps -- the rows are
            # fixed when the step is captured, the budget rays are dropped against (the reference's M, above) is read on the device
            budget_dev = None
            if budget is not None and not force_all_rays and mean_count > 0:
                M, budget_dev = int(budget[0]), budget[1]
                assert M >= 1 and hasattr(backend, "MARCH_FRESH"), "a device-side budget needs the HIP backend"
            dev, dt = rays_o.device, rays_o.dtype
            # scratch_counter: the caller does not care what step_counter held (run_cuda zeroes it right before, renderer.py:374).
            # A backend with MARCH_FRESH then takes uninitialised outputs and counter and writes the zeros itself.
            fresh = bool(getattr(backend, "MARCH_FRESH", False)) and (scratch_counter or step_counter is None)
            alloc = torch.empty if fresh else torch.zeros
            buf = alloc(M * 8, dtype=dt, device=dev)  # one zero-fill for the three outputs (:240-242)
            xyzs, dirs, deltas = buf[:3 * M].view(M, 3), buf[3 * M:6 * M].view(M, 3), buf[6 * M:].view(M, 2)
            rays = torch.empty(N, 3, dtype=torch.int32, device=dev)  # id, offset, num_steps
            if step_counter is None:
                step_counter = alloc(2, dtype=torch.int32, device=dev)  # point counter, ray counter
            elif scratch_counter and not fresh:
                step_counter.zero_()
            if budget_dev is not None:
                backend.march_rays_train(rays_o, rays_d, density_bitfield, bound, dt_gamma, max_steps, N, C, H, M, nears, fars,
                                         xyzs, dirs, deltas, rays, step_counter, perturb, fresh=fresh, budget_dev=budget_dev)
            elif fresh:
                backend.march_rays_train(rays_o, rays_d, density_bitfield, bound, dt_gamma, max_steps, N, C, H, M, nears, fars,
                                         xyzs, dirs, deltas, rays, step_counter, perturb, fresh=True)
            else:
                backend.march_rays_train(rays_o, rays_d, density_bitfield, bound, dt_gamma, max_steps, N, C, H, M, nears, fars,
                                         xyzs, dirs, deltas, rays, step_counter, perturb)
            # warm-up only: trim to the real count (D2H sync, :276-284)
            if force_all_rays or mean_count <= 0:
                m = step_counter[0].item()
                if align > 0:
                    m += align - m % align
                xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
                if device_type == "cuda":
                    torch.cuda.empty_cache()
            return xyzs, dirs, deltas, rays

    class _CompositeTrain(Function):
        # reference: _composite_rays_train, raymarching.py:292-357
        @staticmethod
        @fwd32
        def forward(ctx, sigmas, rgbs, deltas, rays):
            sigmas = sigmas.contiguous()
            rgbs = rgbs.contiguous()
            M, N = sigmas.shape[0], rays.shape[0]
            weights_sum = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
            depth = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
            image = torch.empty(N, 3, dtype=sigmas.dtype, device=sigmas.device)
            backend.composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image)
            ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
            ctx.dims = [M, N]
            return weights_sum, depth, image

        @staticmethod
        @bwd
        def backward(ctx, grad_weights_sum, grad_depth, grad_image):
            # grad_depth is ignored, exactly like the reference (:330)
            grad_weights_sum = grad_weights_sum.contiguous()
            grad_image = grad_image.contiguous()
            sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
            M, N = ctx.dims
            gbuf = torch.zeros(sigmas.shape[0] * 4, dtype=sigmas.dtype, device=sigmas.device)  # one zero-fill (:339-340)
            grad_sigmas, grad_rgbs = gbuf[:sigmas.shape[0]], gbuf[sigmas.shape[0]:].view(-1, 3)
            backend.composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image,
                                                  M, N, grad_sigmas, grad_rgbs)
            return grad_sigmas, grad_rgbs, None, None

    fused_bg = hasattr(backend, "composite_rays_train_bg_forward")

    class _CompositeTrainBg(Function):
        """composite_rays_train + run_cuda's epilogue in one launch each way (distill_mutual/renderer.py:442-446):
        returns (weights_sum, normalised depth, background-blended image).  bg: python scalar or [.., N, 3] tensor."""

        @staticmethod
        @fwd32
        def forward(ctx, sigmas, rgbs, deltas, rays, bg, nears, fars, depth_eps, packed_rays=False, budget_dev=None, objective=None):
            ctx.packed_rays = bool(packed_rays) and bool(getattr(backend, "MARCH_FRESH", False))
            ctx.budget_dev = budget_dev
            kw = {} if budget_dev is None else {"budget_dev": budget_dev}
            sigmas, rgbs = sigmas.contiguous(), rgbs.contiguous()
            M, N = sigmas.shape[0], rays.shape[0]
            bg_t = bg.reshape(-1, 3).contiguous().float() if torch.is_tensor(bg) else None
            bg_s = 0.0 if torch.is_tensor(bg) else float(bg)
            weights_sum = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
            depth = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
            image = torch.empty(N, 3, dtype=sigmas.dtype, device=sigmas.device)
            # objective: a stage-3 distillation objective that rides on this launch and on the backward one (pvd/losses.py
            # ObjectiveRide: teacher image / features / colours, the student's features / colours; receives the partial sums)
            ctx.objective = objective if (objective is not None and hasattr(backend, "composite_objective_forward") and N > 0 and M > 0) else None
            if ctx.objective is not None:
                ob = ctx.objective
                fixed = bool(getattr(ob, "fixed_parts", False))  # ray-DP: a partial-sum count that is the same on every rank
                fk = {"fixed_parts": True} if fixed else {}
                nparts = backend.composite_objective_blocks(N, ob.fea_s.shape[0], **({"fixed": True} if fixed else {}))
                ob.S = torch.empty(4 + 4 * nparts, dtype=torch.float32, device=sigmas.device)
                dk = {} if getattr(ob, "rates_decay", None) is None else {"rates_decay": ob.rates_decay, "fea_decay": ob.fea_decay}
                backend.composite_objective_forward(sigmas, rgbs, deltas, rays, M, N, bg_t, bg_s, nears, fars, depth_eps, weights_sum, depth,
                                                    image, ob.img_t, ob.fea_s, ob.fea_t, ob.col_s, ob.col_t, ob.S, **kw, **dk, **fk)
                ob.decayed = "rates_decay" in dk
                ob.nparts = nparts
            else:
                backend.composite_rays_train_bg_forward(sigmas, rgbs, deltas, rays, M, N, bg_t, bg_s, nears, fars, depth_eps, weights_sum,
                                                        depth, image, **kw)
            ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, image)
            ctx.bg = (bg_t, bg_s)
            ctx.dims = [M, N]
            ctx.set_materialize_grads(False)  # unused outputs arrive as None instead of freshly zero-filled tensors
            return weights_sum, depth, image

        @staticmethod
        @bwd
        def backward(ctx, grad_weights_sum, grad_depth, grad_image):
            sigmas, rgbs, deltas, rays, weights_sum, image = ctx.saved_tensors
            M, N = ctx.dims
            # packed_rays (a table of march_rays_train: slots in ray order, no gaps): the kernel itself zeroes the slots no
            # ray owns; otherwise one zero-fill (:339-340)
            gbuf = (torch.empty if ctx.packed_rays else torch.zeros)(sigmas.shape[0] * 4, dtype=sigmas.dtype, device=sigmas.device)
            grad_sigmas, grad_rgbs = gbuf[:sigmas.shape[0]], gbuf[sigmas.shape[0]:].view(-1, 3)
            gws = grad_weights_sum.contiguous() if grad_weights_sum is not None else None
            if grad_image is None:  # only weights_sum was used
                grad_image = torch.zeros_like(image)
            kw = {} if ctx.budget_dev is None else {"budget_dev": ctx.budget_dev}
            ob = ctx.objective
            if ob is not None and getattr(ob, "armed", False):
                # the objective's backward ran first and left its coefficients: the image gradient is formed inside this launch
                # (grad_image is a placeholder), the feature / colour gradients are written by extra workgroups of it
                ob.armed = False
                backend.composite_objective_backward(gws, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, ctx.bg[0], ctx.bg[1], grad_sigmas,
                                                     grad_rgbs, ob.img_t, ob.fea_s, ob.fea_t, ob.col_s, ob.col_t, ob.coef, ob.upstream, ob.g_fea,
                                                     ob.g_col, fresh=bool(ctx.packed_rays), finish=getattr(ob, "finish", None),
                                                     **({"fixed_parts": True} if getattr(ob, "fixed_parts", False) else {}), **kw)
            elif ctx.packed_rays:
                backend.composite_rays_train_bg_backward(gws, grad_image.contiguous(), sigmas, rgbs, deltas, rays, weights_sum, image, M, N,
                                                         ctx.bg[0], ctx.bg[1], grad_sigmas, grad_rgbs, fresh=True, **kw)
            else:
                backend.composite_rays_train_bg_backward(gws, grad_image.contiguous(), sigmas, rgbs, deltas, rays, weights_sum, image, M, N,
                                                         ctx.bg[0], ctx.bg[1], grad_sigmas, grad_rgbs, **kw)
            return grad_sigmas, grad_rgbs, None, None, None, None, None, None, None, None, None

    def composite_rays_train_bg(sigmas, rgbs, deltas, rays, bg, nears, fars, depth_eps, packed_rays=False, budget_dev=None, objective=None):
        """packed_rays: `rays` comes straight from march_rays_train (offsets in ray order from 0, no gaps).  budget_dev: the
        device-side logical sample budget the march of these rays was given (march_rays_train(budget=...)), if any.
        objective: see _CompositeTrainBg.forward (ignored by backends without the fused launches)."""
        if fused_bg:
            return _CompositeTrainBg.apply(sigmas, rgbs, deltas, rays, bg, nears, fars, depth_eps, packed_rays, budget_dev, objective)
        assert budget_dev is None, "a device-side budget needs the fused compositing of the HIP backend"
        # reference formulation (used with the CPU oracle backend in the test-suite)
        weights_sum, depth, image = _CompositeTrain.apply(sigmas, rgbs, deltas, rays)
        if torch.is_tensor(bg):
            bg = bg.reshape(-1, 3)
        image = image + (1 - weights_sum).unsqueeze(-1) * bg
        depth = torch.clamp(depth - nears, min=0) / (fars - nears + depth_eps)
        return weights_sum, depth, image

    class _March(Function):
        # reference: _march_rays, raymarching.py:367-454
        @staticmethod
        @fwd32
        def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far,
                    align=-1, perturb=False, dt_gamma=0, max_steps=1024):
            rays_o = _to_dev(rays_o).contiguous().view(-1, 3)
            rays_d = _to_dev(rays_d).contiguous().view(-1, 3)
            M = n_alive * n_step
            if align > 0:
                M += align - (M % align)
            dev, dt = rays_o.device, rays_o.dtype
            xyzs = torch.zeros(M, 3, dtype=dt, device=dev)
            dirs = torch.zeros(M, 3, dtype=dt, device=dev)
            deltas = torch.zeros(M, 2, dtype=dt, device=dev)  # (for alpha, for depth)
            backend.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H,
                               density_bitfield, near, far, xyzs, dirs, deltas, perturb)
            return xyzs, dirs, deltas

    class _Composite(Function):
        # reference: _composite_rays, raymarching.py:457-502 (in place on weights_sum / depth / image / rays_t)
        @staticmethod
        @fwd32
        def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
            backend.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image)
            return tuple()

    class _Compact(Function):
        # reference: _compact_rays, raymarching.py:505-527 (in place on rays_alive / rays_t / alive_counter)
        @staticmethod
        @fwd32
        def forward(ctx, n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter):
            backend.compact_rays(n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter)
            return tuple()

    extra = {}
    if hasattr(backend, "infer_march"):  # inference rounds whose state stays on the device (pvd_infer_*; not in the reference)
        extra = dict(infer_round_begin=backend.infer_round_begin, infer_compact=backend.infer_compact, infer_march=backend.infer_march,
                     infer_composite=backend.infer_composite, INFER_STATE_INTS=backend.INFER_STATE_INTS)
    return types.SimpleNamespace(
        **extra,
        near_far_from_aabb=_NearFar.apply,
        polar_from_ray=_Polar.apply,
        morton3D=_Morton.apply,
        morton3D_invert=_MortonInvert.apply,
        packbits=_Packbits.apply,
        march_rays_train=_MarchTrain.apply,
        composite_rays_train=_CompositeTrain.apply,
        composite_rays_train_bg=composite_rays_train_bg,
        march_rays=_March.apply,
        composite_rays=_Composite.apply,
        compact_rays=_Compact.apply,
    )
