"""Occupancy-grid volume renderer: the harness counterpart of the reference's
``NeRFRenderer`` (distill_mutual/renderer.py:66-814; teacher variant just_train_tea/renderer.py).

The live path -- ``run_cuda`` (train and inference branches), ``update_extra_state``, ``mark_untrained_grid`` and
``render`` -- with the reference's buffer names (``density_grid``, ``density_bitfield``, ``step_counter``,
``aabb_train``, ``aabb_infer``) so that state-dicts line up (checkpoint row); plus ``run``, the
fixed-step pure-torch sampler of the non-``cuda_ray`` configuration (BASELINE.json configs[0], a plumbing check: in the
reference its colour query asserts out, network.py:515-516; here it is a masked query).
"""
import math

import numpy as np
import torch
import torch.nn as nn


def near_far_from_aabb_torch(rays_o, rays_d, aabb, min_near):
    """The slab test of raymarching.cu:93-147 as torch ops (same IEEE operations per element, so it agrees with the kernels
    bit for bit): used by the non-cuda_ray sampler, which must run without any native operator."""
    rd = 1.0 / rays_d
    lo = (aabb[:3] - rays_o) * rd
    hi = (aabb[3:] - rays_o) * rd
    t_lo, t_hi = torch.minimum(lo, hi), torch.maximum(lo, hi)
    near, far = t_lo[:, 0], t_hi[:, 0]
    miss = torch.zeros_like(near, dtype=torch.bool)
    for a in (1, 2):  # the kernel tests an axis against the interval accumulated so far and leaves at the first miss
        miss = miss | (near > t_hi[:, a]) | (t_lo[:, a] > far)
        near, far = torch.maximum(near, t_lo[:, a]), torch.minimum(far, t_hi[:, a])
    big = torch.finfo(near.dtype).max
    near = torch.where(miss, torch.full_like(near, big), near.clamp_min(min_near))
    far = torch.where(miss, torch.full_like(far, big), far)
    return near, far


def sample_pdf(bins, weights, n_samples, det=False):
    """Inverse-transform sampling of a piecewise-constant pdf (hierarchical NeRF sampling; reference: sample_pdf,
    renderer.py:17-52).  bins [B, T], weights [B, T-1] -> [B, n_samples]."""
    pdf = weights + 1e-5
    pdf = pdf / pdf.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    shape = list(cdf.shape[:-1]) + [n_samples]
    if det:
        u = torch.linspace(0.5 / n_samples, 1.0 - 0.5 / n_samples, n_samples, device=cdf.device).expand(shape)
    else:
        u = torch.rand(shape, device=cdf.device)
    u = u.contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = (hi - 1).clamp_min(0)
    hi = hi.clamp_max(cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    b_lo, b_hi = torch.gather(bins, -1, lo), torch.gather(bins, -1, hi)
    width = c_hi - c_lo
    width = torch.where(width < 1e-5, torch.ones_like(width), width)
    return b_lo + (u - c_lo) / width * (b_hi - b_lo)


class NeRFRenderer(nn.Module):
    def __init__(self, ops, bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1,
                 grid_size=128, teacher_variant=False):
        super().__init__()
        self.ops = ops
        self.rm = ops.raymarching
        self.bound = bound
        self.cascade = 1 + math.ceil(math.log2(bound))  # renderer.py:81
        self.grid_size = grid_size
        self.density_scale = density_scale
        self.min_near = min_near
        self.density_thresh = density_thresh
        self.bg_radius = bg_radius
        # just_train_tea/renderer.py differs in two places (depth normalisation, no stage gating)
        self.teacher_variant = teacher_variant

        aabb = torch.FloatTensor([-bound, -bound, -bound, bound, bound, bound])
        self.register_buffer("aabb_train", aabb)
        self.register_buffer("aabb_infer", aabb.clone())

        self.cuda_ray = cuda_ray
        if cuda_ray:
            self.register_buffer("density_grid", torch.zeros([self.cascade, grid_size ** 3]))
            self.register_buffer("density_bitfield", torch.zeros(self.cascade * grid_size ** 3 // 8, dtype=torch.uint8))
            self.mean_density = 0
            self.iter_density = 0
            self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))  # ring of 16 (renderer.py:108-113)
            self.mean_count = 0
            self.local_step = 0
        if bg_radius > 0:
            # the reference's background model is dead code (`assert 1 == 2`, distill_mutual/network.py:496): no encoder_bg /
            # bg_net exists here either, so refuse the configuration up front instead of failing inside the first render
            raise NotImplementedError("bg_radius > 0 (background model) is not supported: the reference's own branch asserts out "
                                      "(distill_mutual/network.py:496)")
        # Bumped by everything that rewrites density_grid / density_bitfield.  Writers go through ctypes on data_ptr(), which
        # autograd's tensor versions do not see, so consumers of a frozen grid (the trainer's touched-row set, the compact
        # gradient exchange) key on this counter instead.
        self.occ_epoch = 0

    def note_occupancy_changed(self):
        self.occ_epoch += 1

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.note_occupancy_changed()  # density_grid / density_bitfield may have been replaced
        return out

    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def reset_extra_state(self):
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.note_occupancy_changed()
        self.mean_density = 0
        self.iter_density = 0
        self.step_counter.zero_()
        self.mean_count = 0
        self.local_step = 0

    # ------------------------------------------------------------------ run (no occupancy grid, no native operator)
    def run(self, rays_o, rays_d, num_steps=128, upsample_steps=128, bg_color=None, perturb=False, **kwargs):
        """reference: run, distill_mutual/renderer.py:139-317.  Uniform steps between the box intersections (+ optional
        importance resampling), densities for every step, colours only where the compositing weight exceeds 1e-4
        (`color(mask=...)`), alpha compositing by cumprod.  rays_o / rays_d [B, N, 3] with B == 1."""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N, device = rays_o.shape[0], rays_o.device
        aabb = self.aabb_train if self.training else self.aabb_infer
        nears, fars = near_far_from_aabb_torch(rays_o, rays_d, aabb, self.min_near)
        nears, fars = nears.unsqueeze(-1), fars.unsqueeze(-1)

        z_vals = nears + (fars - nears) * torch.linspace(0.0, 1.0, num_steps, device=device).unsqueeze(0)  # [N, T]
        sample_dist = (fars - nears) / num_steps
        if perturb:
            z_vals = z_vals + (torch.rand(z_vals.shape, device=device) - 0.5) * sample_dist

        def positions(z):
            x = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z.unsqueeze(-1)  # [N, T, 3]
            return torch.min(torch.max(x, aabb[:3]), aabb[3:])  # clipped into the box (:186)

        def weights_of(z, sigma):
            deltas = torch.cat([z[..., 1:] - z[..., :-1], sample_dist * torch.ones_like(z[..., :1])], dim=-1)
            alphas = 1 - torch.exp(-deltas * self.density_scale * sigma)
            trans = torch.cumprod(torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1), dim=-1)[..., :-1]
            return alphas * trans, deltas

        xyzs = positions(z_vals)
        dens = {k: v.view(N, num_steps, -1) for k, v in self.density(xyzs.reshape(-1, 3)).items()}

        if upsample_steps > 0:  # NeRF-style resampling along the first pass's weights (:195-246)
            with torch.no_grad():
                w0, d0 = weights_of(z_vals, dens["sigma"].squeeze(-1))
                z_mid = z_vals[..., :-1] + 0.5 * d0[..., :-1]
                new_z = sample_pdf(z_mid, w0[:, 1:-1], upsample_steps, det=not self.training).detach()
                new_xyzs = positions(new_z)
            new_dens = {k: v.view(N, upsample_steps, -1) for k, v in self.density(new_xyzs.reshape(-1, 3)).items()}
            z_vals, order = torch.sort(torch.cat([z_vals, new_z], dim=1), dim=1)
            xyzs = torch.gather(torch.cat([xyzs, new_xyzs], dim=1), 1, order.unsqueeze(-1).expand(-1, -1, 3))
            for k in dens:
                both = torch.cat([dens[k], new_dens[k]], dim=1)
                dens[k] = torch.gather(both, 1, order.unsqueeze(-1).expand_as(both))

        weights, _ = weights_of(z_vals, dens["sigma"].squeeze(-1))
        dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
        flat = {k: v.reshape(-1, v.shape[-1]) for k, v in dens.items()}
        mask = weights > 1e-4  # hard-coded in the reference (:283)
        rgbs = self.color(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask=mask.reshape(-1), **flat).view(N, -1, 3)

        weights_sum = weights.sum(dim=-1)
        depth = torch.sum(weights * ((z_vals - nears) / (fars - nears)).clamp(0, 1), dim=-1)
        image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
        if bg_color is None:
            bg_color = 1
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        return {"depth": depth.view(*prefix), "image": image.view(*prefix, 3)}

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        raise NotImplementedError()

    # ------------------------------------------------------------------ run_cuda
    def run_cuda(self, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024,
                 inherited_params=(), **kwargs):
        """reference: run_cuda, distill_mutual/renderer.py:319-559.  rays_o/rays_d [B, N, 3] with B == 1."""
        rm = self.rm
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        device = rays_o.device

        nears_fars = kwargs.get("nears_fars")  # the model that marched first hands its near/far along with the samples
        if nears_fars is not None:
            nears, fars = nears_fars
        else:
            nears, fars = rm.near_far_from_aabb(rays_o, rays_d, self.aabb_train if self.training else self.aabb_infer, self.min_near)

        if self.bg_radius > 0:
            polar = rm.polar_from_ray(rays_o, rays_d, self.bg_radius)
            bg_color = self.background(polar, rays_d)
        elif bg_color is None:
            bg_color = 1

        if self.training:
            # whoever renders first marches; the other model inherits the very same samples
            # (renderer.py:365-411): student first when args.render_stu_first, else teacher first.
            stu_first = bool(getattr(self.args, "render_stu_first", True))
            i_march = (not self.is_teacher) if stu_first else self.is_teacher
            if self.teacher_variant:
                i_march = True
            premarched = kwargs.get("premarched", False)  # march() already ran for this step (see DistillTrainer)
            if premarched:
                i_march = False
            else:
                counter = self.step_counter[self.local_step % 16]  # set to zero (renderer.py:374): the scratch_counter flag below
                self.local_step += 1
            budget = self._budget() if not force_all_rays else None  # (single-model training: the model composites its own march)
            own = i_march or bool(kwargs.get("own_march", False))  # the samples come from THIS model's march (budget applies)
            if i_march:
                if budget is not None:
                    xyzs, dirs, deltas, rays = rm.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield, self.cascade,
                                                                   self.grid_size, nears, fars, counter, self.mean_count, perturb, 128,
                                                                   force_all_rays, dt_gamma, max_steps, True, budget)
                else:
                    xyzs, dirs, deltas, rays = rm.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield, self.cascade,
                                                                   self.grid_size, nears, fars, counter, self.mean_count, perturb, 128,
                                                                   force_all_rays, dt_gamma, max_steps, True)
                inherited_params = [xyzs, dirs, deltas, rays]
            else:
                xyzs, dirs, deltas, rays = inherited_params

            sigmas, rgbs = self(xyzs, dirs)

            if not self.teacher_variant:  # stage gating, renderer.py:421-438
                gs, st = self.args.global_step, self.args.stage_iters
                if gs < st["stage1"]:
                    return {"stage1": gs, "depth": None, "image": None, "inherited_params": inherited_params, "sigmas": sigmas, "rays": rays}
                if gs < st["stage2"]:
                    return {"stage2": gs, "depth": None, "image": None, "inherited_params": inherited_params, "sigmas": sigmas, "rays": rays}

            if self.density_scale != 1:
                sigmas = self.density_scale * sigmas
            eps = 0.0 if self.teacher_variant else 1e-6  # renderer.py:446 vs just_train_tea/renderer.py
            # compositing + `image += (1 - ws) * bg` + depth normalisation (renderer.py:442-446) as one op
            # (trainer) a stage-3 objective riding on the compositing launches: it needs this model's feature / colour rows
            ride = kwargs.get("objective")
            if ride is not None and not (torch.is_grad_enabled() and ride.with_student(getattr(self, "feature_sigma_color", None), getattr(self, "color_l", None))):
                ride = None
            okw = {} if ride is None else {"objective": ride}
            if budget is not None and own:
                weights_sum, depth, image = rm.composite_rays_train_bg(sigmas, rgbs, deltas, rays, bg_color, nears, fars, eps, True, budget[1], **okw)
            else:
                weights_sum, depth, image = rm.composite_rays_train_bg(sigmas, rgbs, deltas, rays, bg_color, nears, fars, eps, True, **okw)  # rays: straight from the march
            image = image.view(*prefix, 3)
            depth = depth.view(*prefix)
            return {"depth": depth, "image": image, "inherited_params": inherited_params, "sigmas": sigmas, "rays": rays,
                    "nears_fars": (nears, fars)}

        # ---- inference: march / shade / composite in rounds with ray compaction (renderer.py:450-543)
        if self._persistent_render(rays_o, perturb):
            # a frozen hash, VM or Plenoxel model: the whole loop as ONE persistent launch (pvd_infer_image_hash / _vm / _plenoxel;
            # PVD_INFER_PERSISTENT=0: the rounds)
            if self.model_type == "tensors":
                run = self.ops.plenoxel.infer_image
            else:
                fh = self.ops.fused_head
                run = fh.hash_infer_image if self.model_type == "hash" else fh.vm_infer_image
            weights_sum, depth, image = run(self, rays_o, rays_d, nears, fars, dt_gamma, max_steps)
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            depth = torch.clamp(depth - nears, min=0) / (fars - nears)
            return {"depth": depth.view(*prefix), "image": image.view(*prefix, 3), "inherited_params": inherited_params}
        if self._rounds_on_device(rays_o, perturb):
            return self._run_rounds_device(rays_o, rays_d, nears, fars, bg_color, dt_gamma, max_steps, prefix, inherited_params)
        dtype = torch.float32
        weights_sum = torch.zeros(N, dtype=dtype, device=device)
        depth = torch.zeros(N, dtype=dtype, device=device)
        image = torch.zeros(N, 3, dtype=dtype, device=device)
        n_alive = N
        alive_counter = torch.zeros([1], dtype=torch.int32, device=device)
        rays_alive = torch.zeros(2, n_alive, dtype=torch.int32, device=device)  # ping-pong
        rays_t = torch.zeros(2, n_alive, dtype=dtype, device=device)
        step, i = 0, 0
        while step < max_steps:
            if step == 0:
                rays_alive[0] = torch.arange(n_alive, dtype=torch.int32, device=device)
                rays_t[0] = nears
            else:
                alive_counter.zero_()
                rm.compact_rays(n_alive, rays_alive[i % 2], rays_alive[(i + 1) % 2], rays_t[i % 2], rays_t[(i + 1) % 2], alive_counter)
                n_alive = alive_counter.item()  # D2H sync per round, as in the reference (:488)
            if n_alive <= 0:
                break
            n_step = max(min(N // n_alive, 8), 1)
            xyzs, dirs, deltas = rm.march_rays(n_alive, n_step, rays_alive[i % 2], rays_t[i % 2], rays_o, rays_d, self.bound,
                                               self.density_bitfield, self.cascade, self.grid_size, nears, fars, 128, perturb,
                                               dt_gamma, max_steps)
            sigmas, rgbs = self(xyzs, dirs)
            sigmas = self.density_scale * sigmas
            rm.composite_rays(n_alive, n_step, rays_alive[i % 2], rays_t[i % 2], sigmas, rgbs, deltas, weights_sum, depth, image)
            step += n_step
            i += 1
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        return {"depth": depth.view(*prefix), "image": image.view(*prefix, 3), "inherited_params": inherited_params}

    def _persistent_render(self, rays_o, perturb):
        import os
        fh = getattr(getattr(self, "ops", None), "fused_head", None)
        mt = getattr(self, "model_type", None)
        if os.environ.get("PVD_INFER_PERSISTENT", "1") == "0":
            return False
        if mt == "tensors":  # fp32 with or without autocast; the editing demo rewrites the volume per call (network.py:313-316): rounds
            px = getattr(getattr(self, "ops", None), "plenoxel", None)
            return (rays_o.is_cuda and not perturb and hasattr(px, "infer_image") and getattr(self, "bg_net", None) is None
                    and not self.args.enable_edit_plenoxel)
        return self._rounds_on_device(rays_o, perturb) and ((mt == "hash" and hasattr(fh, "hash_infer_image")) or
                                                            (mt == "vm" and hasattr(fh, "vm_infer_image")))

    # ------------------------------------------------------------------ inference rounds, state on the device
    def _rounds_on_device(self, rays_o, perturb):
        import os
        return (rays_o.is_cuda and not perturb and hasattr(self.rm, "infer_march") and torch.is_autocast_enabled("cuda")
                and getattr(self, "supports_device_rows", lambda: False)() and os.environ.get("PVD_INFER_DEVICE_ROUNDS", "1") != "0")

    @torch.no_grad()
    def _run_rounds_device(self, rays_o, rays_d, nears, fars, bg_color, dt_gamma, max_steps, prefix, inherited_params, check_every=8):
        """The inference loop of run_cuda (reference: renderer.py:450-543) without its per-round `alive_counter.item()`: the
        round state {alive count, n_step, rows, steps done} lives on the device (pvd_infer_*), every launch of a round takes
        its extent from there, and the host looks at it once every `check_every` rounds -- to stop, and to shrink the upper
        bound that sizes the launches and the scratch rows.  Same n_step rule, same march / composite arithmetic: per-ray
        results are those of the reference loop."""
        rm = self.rm
        N, device = rays_o.shape[0], rays_o.device
        f32 = dict(dtype=torch.float32, device=device)
        weights_sum, depth, image = torch.zeros(N, **f32), torch.zeros(N, **f32), torch.zeros(N, 3, **f32)
        state = torch.zeros(rm.INFER_STATE_INTS, dtype=torch.int32, device=device)
        state[0] = N  # cnt[0]: every ray starts alive
        rays_alive = torch.empty(2, N, dtype=torch.int32, device=device)
        rays_t = torch.empty(2, N, **f32)
        rays_alive[0] = torch.arange(N, dtype=torch.int32, device=device)
        rays_t[0] = nears
        rows = N + (128 - N % 128) % 128  # n_alive * n_step <= N by the n_step rule
        xyzs, dirs, deltas = torch.empty(rows, 3, **f32), torch.empty(rows, 3, **f32), torch.empty(rows, 2, **f32)
        rows_dev = state[4:5]
        n_upper, i = N, 0
        while True:
            for _ in range(check_every):
                cur, old = i & 1, (i & 1) ^ 1
                if i > 0:
                    rm.infer_compact(state, cur, n_upper, rays_alive[cur], rays_alive[old], rays_t[cur], rays_t[old])
                rm.infer_round_begin(state, cur, N, max_steps)
                rm.infer_march(state, n_upper, rays_alive[cur], rays_t[cur], rays_o, rays_d, self.bound, dt_gamma, max_steps, self.cascade,
                               self.grid_size, self.density_bitfield, fars, xyzs, dirs, deltas, False)
                m_upper = min(rows, 8 * n_upper + (128 - (8 * n_upper) % 128) % 128)
                sigmas, rgbs = self.forward_rows(xyzs[:m_upper], dirs[:m_upper], rows_dev)
                rm.infer_composite(state, n_upper, rays_alive[cur], rays_t[cur], sigmas, rgbs.float(), deltas, float(self.density_scale),
                                   weights_sum, depth, image)
                i += 1
            n_alive, steps_done = state[2:6:3].tolist()  # one read-back per `check_every` rounds
            if n_alive <= 0 or steps_done >= max_steps:
                break
            n_upper = min(n_upper, n_alive)  # survivors only shrink
        self._last_rounds = int(state[6])
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        return {"depth": depth.view(*prefix), "image": image.view(*prefix, 3), "inherited_params": inherited_params}

    # ------------------------------------------------------------------ occupancy grid upkeep
    def march(self, rays_o, rays_d, dt_gamma=0, perturb=False, force_all_rays=False, max_steps=1024, nears_fars=None):
        """The sampling half of the training branch of run_cuda on its own: near/far + march_rays_train.  Returns
        (inherited_params, (nears, fars)) for run_cuda(..., inherited_params=, nears_fars=, premarched=True), so that
        the two models' forwards can be issued on different streams."""
        rm = self.rm
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        nears, fars = nears_fars if nears_fars is not None else rm.near_far_from_aabb(rays_o, rays_d, self.aabb_train, self.min_near)
        counter = self.step_counter[self.local_step % 16]  # "counter.zero_()" (renderer.py:374) = the scratch_counter flag below
        self.local_step += 1
        budget = self._budget() if not force_all_rays else None  # fixed allocation + device-side budget (fix_sample_alloc)
        extra = (budget,) if budget is not None else ()
        xyzs, dirs, deltas, rays = rm.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield, self.cascade, self.grid_size,
                                                       nears, fars, counter, self.mean_count, perturb, 128, force_all_rays, dt_gamma,
                                                       max_steps, True, *extra)
        return [xyzs, dirs, deltas, rays], (nears, fars)

    def _cell_centres(self, coords, cas, jitter):
        """Grid coords [n,3] in [0,H) -> world positions of cascade `cas` (renderer.py:680-693)."""
        H = self.grid_size
        xyzs = 2 * coords.float() / (H - 1) - 1
        bound = min(2 ** cas, self.bound)
        hgs = bound / H
        p = xyzs * (bound - hgs)
        if jitter:
            p = p + (torch.rand_like(p) * 2 - 1) * hgs
        return p, hgs

    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        """Cells seen by no training camera get density -1 and are never updated
        (reference: renderer.py:561-645)."""
        if not self.cuda_ray:
            return
        if isinstance(poses, np.ndarray):
            poses = torch.from_numpy(poses)
        dev = self.density_grid.device
        poses = poses.to(dev)
        B = poses.shape[0]
        fx, fy, cx, cy = intrinsic
        H = self.grid_size
        axis = torch.arange(H, dtype=torch.int32, device=dev)
        count = torch.zeros_like(self.density_grid)
        for xs in axis.split(S):
            for ys in axis.split(S):
                for zs in axis.split(S):
                    xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
                    coords = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], dim=-1)
                    indices = self.rm.morton3D(coords).long()
                    for cas in range(self.cascade):
                        world, hgs = self._cell_centres(coords, cas, jitter=False)
                        for head in range(0, B, S):
                            P = poses[head:head + S]
                            cam = (world.unsqueeze(0) - P[:, :3, 3].unsqueeze(1)) @ P[:, :3, :3]  # world -> camera
                            seen = (cam[..., 2] > 0) \
                                & (cam[..., 0].abs() < cx / fx * cam[..., 2] + hgs * 2) \
                                & (cam[..., 1].abs() < cy / fy * cam[..., 2] + hgs * 2)
                            count[cas, indices] += seen.sum(0).reshape(-1)
        self.density_grid[count == 0] = -1
        self.note_occupancy_changed()

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128):
        """EMA-max update of the density grid, re-pack of the bitfield, refresh of mean_count
        (reference: renderer.py:647-775)."""
        if not self.cuda_ray:
            return
        self._flush_deferred_updates()  # the density sweep reads table rows no training sample touches
        occ = getattr(self.ops, "occupancy", None)
        if occ is not None and self.density_grid.is_cuda:
            return self._update_extra_state_device(occ, decay)
        rm = self.rm
        dev = self.density_grid.device
        H = self.grid_size
        tmp_grid = -torch.ones_like(self.density_grid)

        def query(coords, indices, cas):
            p, _ = self._cell_centres(coords, cas, jitter=True)
            sig = self.density(p)["sigma"].reshape(-1).detach().float() * self.density_scale
            tmp_grid[cas, indices] = sig

        if self.iter_density < 16:  # full sweep
            axis = torch.arange(H, dtype=torch.int32, device=dev)
            for xs in axis.split(S):
                for ys in axis.split(S):
                    for zs in axis.split(S):
                        xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
                        coords = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], dim=-1)
                        indices = rm.morton3D(coords).long()
                        for cas in range(self.cascade):
                            query(coords, indices, cas)
        else:  # H^3/4 uniform cells + H^3/4 currently occupied cells
            n = H ** 3 // 4
            for cas in range(self.cascade):
                coords = torch.randint(0, H, (n, 3), device=dev)
                indices = rm.morton3D(coords).long()
                occ = torch.nonzero(self.density_grid[cas] > 0).squeeze(-1)
                if occ.shape[0] > 0:
                    occ = occ[torch.randint(0, occ.shape[0], [n], dtype=torch.long, device=dev)]
                    occ_coords = rm.morton3D_invert(occ)
                    indices = torch.cat([indices, occ], dim=0)
                    coords = torch.cat([coords, occ_coords], dim=0)
                query(coords, indices, cas)

        valid = (self.density_grid >= 0) & (tmp_grid >= 0)
        self.density_grid[valid] = torch.maximum(self.density_grid[valid] * decay, tmp_grid[valid])
        self.mean_density = torch.mean(self.density_grid.clamp(min=0)).item()
        self.iter_density += 1

        thresh = min(self.mean_density, self.density_thresh)
        self.density_bitfield = rm.packbits(self.density_grid, thresh, self.density_bitfield)
        self.note_occupancy_changed()

        total_step = min(16, self.local_step)
        if total_step > 0:
            self.mean_count = int(self.step_counter[:total_step, 0].sum().item() / total_step)
        self.local_step = 0

    def _update_extra_state_device(self, occ, decay):
        """update_extra_state with the cell selection, jitter, running maximum, mean / threshold and packbits on the
        device (csrc/occupancy.hip): no nonzero(), no .item() except the one that sizes next steps' sample budget."""
        dev = self.density_grid.device
        H, C = self.grid_size, self.cascade
        H3 = H ** 3
        st = getattr(self, "_occ_state", None)
        if st is None:
            st = self._occ_state = dict(list=torch.empty(H3, dtype=torch.int32, device=dev), count=torch.zeros(1, dtype=torch.int32, device=dev),
                                        tmp=torch.empty(H3, dtype=torch.float32, device=dev), scratch=torch.empty(1024, device=dev),
                                        mean_thresh=torch.zeros(2, device=dev), calls=0)
        full = self.iter_density < 16
        n_u, n_o = (H3, 0) if full else (H3 // 4, H3 // 4)
        n = n_u + n_o
        indices = torch.empty(n, dtype=torch.int32, device=dev)
        xyz = torch.empty(n, 3, dtype=torch.float32, device=dev)
        for cas in range(C):
            st["calls"] += 1
            if self.occ_replay:
                self._occ_cascade_replay(occ, st, cas, full, n_u, indices, xyz, decay)
                continue
            occ.occ_sample(self.density_grid[cas], H, n_u, n_o, full, float(min(2 ** cas, self.bound)), 0x0cc0 + 7919 * st["calls"], st["list"],
                           st["count"], indices, xyz)
            sig = self.density(xyz)["sigma"].reshape(-1).detach().float().contiguous()
            occ.occ_update(self.density_grid[cas], st["tmp"], indices, sig, H, float(self.density_scale), float(decay))
        occ.occ_finish(self.density_grid.view(-1), float(self.density_thresh), st["mean_thresh"], st["scratch"], self.density_bitfield)
        self.note_occupancy_changed()
        self.mean_density = st["mean_thresh"][0]  # stays on the device (float(...) to read it)
        self.iter_density += 1
        total_step = min(16, self.local_step)
        if total_step > 0:
            self.mean_count = int(self.step_counter[:total_step, 0].sum().item() / total_step)
            self._publish_budget()
        self.local_step = 0

    occ_replay = False  # True: the device path consumes torch's CPU generator exactly as the reference's update_extra_state does

    def _occ_cascade_replay(self, occ, st, cas, full, n_u, indices, xyz, decay):
        """One cascade of the device-side update with the REFERENCE's random draws: the torch calls of renderer.py:700-741 in their
        order and shapes on the CPU generator (so torch.manual_seed(s) here = torch.manual_seed(s) before the reference's own
        CPU run), handed to pvd_occ_sample_replay; duplicate cells resolved like the reference's sequential assignment
        (pvd_occ_update_ordered).  Host round trips as in the reference (nonzero, the draws): a checking mode, not a fast one."""
        dev = self.density_grid.device
        H = self.grid_size
        bound_c = float(min(2 ** cas, self.bound))
        if full:
            assert H <= 128, "the reference sweeps in 128^3 blocks: replay covers the single-block case"
            jitter = torch.rand(H ** 3, 3).to(dev)
            occ.occ_sample_replay(H, 0, 0, True, bound_c, None, None, None, jitter, indices, xyz)
            n = H ** 3
        else:
            cells = torch.randint(0, H, (n_u, 3))
            lst = torch.nonzero(self.density_grid[cas] > 0).squeeze(-1)
            n_o = 0
            picks = None
            if lst.shape[0] > 0:
                picks = torch.randint(0, lst.shape[0], [n_u], dtype=torch.long).to(torch.int32).to(dev)
                n_o = n_u
            n = n_u + n_o
            jitter = torch.rand(n, 3).to(dev)
            occ.occ_sample_replay(H, n_u, n_o, False, bound_c, cells.to(torch.int32).to(dev).contiguous(), lst.to(torch.int32).contiguous(), picks,
                                  jitter, indices, xyz)
        sig = self.density(xyz[:n].contiguous())["sigma"].reshape(-1).detach().float().contiguous()
        occ.occ_update_ordered(self.density_grid[cas], st["tmp"], st["list"], indices[:n].contiguous(), sig, H, float(self.density_scale), float(decay))

    # ---- sample budget in device memory (captured training steps)
    sample_alloc = None  # rows the training branch allocates when set (>= the aligned mean_count); see fix_sample_alloc

    def _publish_budget(self):
        """mean_count as march_rays_train will use it (aligned the reference's way, raymarching.py:234-238) in device memory:
        a captured step reads its budget there, so update_extra_state can move it without a re-capture."""
        if self.sample_alloc is None or self.mean_count <= 0:
            return
        m = self.mean_count + (128 - self.mean_count % 128)
        if getattr(self, "_budget_dev", None) is None:
            self._budget_dev = torch.zeros(1, dtype=torch.int32, device=self.density_grid.device)
        self._budget_dev.fill_(min(m, self.sample_alloc))
        self.budget_exceeded = m > self.sample_alloc  # (the caller re-captures with more rows)

    def fix_sample_alloc(self, headroom=1.25, granule=16384):
        """Allocate a FIXED number of sample rows from now on (mean_count x headroom, rounded up to `granule`) and keep the
        budget rays are dropped against -- mean_count, as in the reference -- in device memory.  For training steps that
        are captured once and replayed across occupancy-grid updates.  Returns the row count."""
        assert self.mean_count > 0, "needs a measured mean_count"
        m = self.mean_count + (128 - self.mean_count % 128)
        self.sample_alloc = (int(m * headroom) + granule - 1) // granule * granule
        self._publish_budget()
        return self.sample_alloc

    def _budget(self):
        if self.sample_alloc is None or self.mean_count <= 0:
            return None
        return (self.sample_alloc, self._budget_dev)

    def _flush_deferred_updates(self):
        """A flat optimizer may hold deferred weight decay for table rows the training samples never touch
        (FlatAdamW.flush, installed by the trainer as `_pvd_flush_params`).  Whoever reads tables OUTSIDE the training
        samples -- the occupancy sweep, an evaluation render, a checkpoint -- brings them up to date first; free when
        nothing is pending."""
        fl = getattr(self, "_pvd_flush_params", None)
        if fl is not None and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            fl()

    def render(self, rays_o, rays_d, staged=False, max_ray_batch=4096, **kwargs):
        """reference: render, renderer.py:777-814 (`staged` is ignored on the cuda_ray path)."""
        if not self.training:
            self._flush_deferred_updates()  # evaluation marches on this model's own grid, not the training marcher's
        if self.cuda_ray:
            return self.run_cuda(rays_o, rays_d, **kwargs)
        kwargs = {k: v for k, v in kwargs.items() if k in ("num_steps", "upsample_steps", "bg_color", "perturb")}
        B, N = rays_o.shape[:2]
        if not staged:
            return self.run(rays_o, rays_d, **kwargs)
        depth = torch.empty((B, N), device=rays_o.device)
        image = torch.empty((B, N, 3), device=rays_o.device)
        for b in range(B):
            for head in range(0, N, max_ray_batch):
                tail = min(head + max_ray_batch, N)
                out = self.run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail], **kwargs)
                depth[b:b + 1, head:tail] = out["depth"]
                image[b:b + 1, head:tail] = out["image"]
        return {"depth": depth, "image": image}
