"""Builds the benchmark / smoke / parity workloads: synthetic chair scene, hash teacher (optionally
pre-trained on the analytic scene), student of any model type, trainers.  Shared by bench.py,
__graft_entry__.smoke() and the tests so they all measure / check the same thing."""
import numpy as np
import torch

from .config import PVDConfig
from .network import NeRFNetwork
from .scene import BLENDER_INTRINSICS, ChairScene, forward_facing_train_poses, get_rays, packbits_torch, rand_poses, synthetic_poses
from .trainer import DistillTrainer, RayDP, TeacherTrainer


def renderer_kwargs(opt):
    # constructor arguments of the reference's two NeRFNetwork() calls (main_distill_mutual.py:261-286)
    return dict(bound=opt.bound, cuda_ray=opt.cuda_ray, density_scale=1, min_near=opt.min_near, density_thresh=opt.density_thresh,
                bg_radius=opt.bg_radius, grid_size=opt.grid_size)


def make_model(ops, opt, model_type, is_teacher, device, teacher_variant=False):
    m = NeRFNetwork(ops, model_type=model_type, args=opt, is_teacher=is_teacher, teacher_variant=teacher_variant,
                    **renderer_kwargs(opt))
    return m.to(device)


def install_occupancy(model, scene, opt):
    """Fill density_grid / density_bitfield from the analytic scene (what a trained teacher's
    update_extra_state converges to), on the model's device."""
    dev = model.density_grid.device
    grid = scene.density_grid(opt.grid_size, opt.bound, model.cascade, device=dev)
    model.density_grid.copy_(grid)
    model.density_bitfield.copy_(packbits_torch(grid, min(float(grid.clamp(min=0).mean()), opt.density_thresh)))
    model.mean_density = float(grid.clamp(min=0).mean())
    model.note_occupancy_changed()
    return grid


class AnalyticTarget:
    """Ground-truth pixels of the analytic scene, rendered with the same marcher + compositor."""

    def __init__(self, ops, scene, proxy):
        self.rm, self.scene, self.proxy = ops.raymarching, scene, proxy

    @torch.no_grad()
    def __call__(self, rays_o, rays_d, bg_color, max_steps=1024):
        p, rm = self.proxy, self.rm
        o, d = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        nears, fars = rm.near_far_from_aabb(o, d, p.aabb_train, p.min_near)
        xyzs, dirs, deltas, rays = rm.march_rays_train(o, d, p.bound, p.density_bitfield, p.cascade, p.grid_size, nears, fars,
                                                       None, -1, False, 128, True, 0, max_steps)
        sig, col = self.scene.sigma(xyzs), self.scene.color(xyzs, dirs)
        ws, _, img = rm.composite_rays_train(sig, col, deltas, rays)
        return (img + (1 - ws).unsqueeze(-1) * bg_color).view(*rays_o.shape[:-1], 3)


def measure_mean_count(model, poses, opt, n_poses=8, generator=None):
    """What update_extra_state would store in mean_count: the average number of samples a batch of
    `num_rays` rays generates (renderer.py:768-773)."""
    rm = model.rm
    tot = 0
    for k in range(n_poses):
        r = get_rays(poses[k:k + 1], BLENDER_INTRINSICS, 800, 800, opt.num_rays, generator=generator)
        o, d = r["rays_o"].contiguous().view(-1, 3), r["rays_d"].contiguous().view(-1, 3)
        nears, fars = rm.near_far_from_aabb(o, d, model.aabb_train, model.min_near)
        counter = torch.zeros(2, dtype=torch.int32, device=o.device)
        rm.march_rays_train(o, d, model.bound, model.density_bitfield, model.cascade, model.grid_size, nears, fars, counter, -1,
                            True, 128, True, opt.dt_gamma, opt.max_steps)
        tot += int(counter[0].item())
    return tot // n_poses


class DistillWorkload:
    """hash -> <student> distillation on the synthetic chair (BASELINE.json configs[2] for student 'vm')."""

    def __init__(self, ops, device, opt=None, teacher_pretrain_steps=0, seed=0, dp=None, start_stage="stage3", thicken=0.08, scene_scale=1.0):
        self.ops, self.device = ops, torch.device(device)
        self.opt = opt or PVDConfig()
        opt = self.opt
        torch.manual_seed(seed)
        self.rng = np.random.RandomState(seed)
        self.gen = torch.Generator(device=self.device)
        self.dp_rank = dp.rank if dp else 0
        self.gen.manual_seed(seed + 1000 * self.dp_rank)  # different rays on every rank
        self.scene = ChairScene(thicken=thicken, scale=scene_scale)
        # one epoch of random distillation cameras, by --data_type (get_rand_poses, utils.py:100-197); "synthetic" is the chair's
        orig = forward_facing_train_poses(self.rng) if opt.data_type == "llff" else None
        self.poses = torch.from_numpy(rand_poses(opt.data_type, self.rng, original_poses=orig, scale=opt.scale)).to(self.device)

        self.tea = make_model(ops, opt, opt.teacher_type, True, self.device)
        install_occupancy(self.tea, self.scene, opt)
        self.target = AnalyticTarget(ops, self.scene, self.tea)
        self.teacher_psnr = None
        if teacher_pretrain_steps > 0:
            self.pretrain_teacher(teacher_pretrain_steps)

        self.stu = make_model(ops, opt, opt.model_type, False, self.device)
        # the student starts from the teacher's checkpoint with strict=False (utils.py:1536-1545):
        # occupancy buffers and every same-named, same-shaped tensor (color_net.* for hash->vm) carry over
        src = self.tea.state_dict()
        dst = self.stu.state_dict()
        self.stu.load_state_dict({k: v for k, v in src.items() if k in dst and dst[k].shape == v.shape}, strict=False)
        mc = measure_mean_count(self.tea, self.poses, opt, generator=self.gen)
        self.tea.mean_count = self.stu.mean_count = mc
        self.trainer = DistillTrainer(opt, self.tea, self.stu, self.device, fp16=opt.fp16, dp=dp)
        if start_stage == "stage3":
            self.trainer.global_step = opt.stage_iters["stage2"]
        elif start_stage == "stage2":
            self.trainer.global_step = max(opt.stage_iters["stage1"], 0)
        self.step_idx = 0

    def pretrain_teacher(self, steps):
        """Teacher training on the analytic scene with a fixed (analytic) occupancy grid."""
        opt = self.opt
        topt = PVDConfig(**{**opt.__dict__, "model_type": opt.teacher_type, "iters": max(steps, 1), "update_extra_interval": 10 ** 9,
                            "stage_iters": {"stage1": -1, "stage2": -1}})  # the teacher entry point has no stage gating
        self.tea.teacher_variant = True
        self.tea.args = self.tea.opt = topt
        tr = TeacherTrainer(topt, self.tea, self.device, fp16=opt.fp16)
        self.tea.mean_count = measure_mean_count(self.tea, self.poses, opt, generator=self.gen)
        last = None
        for it in range(steps):
            r = get_rays(self.poses[it % len(self.poses)][None], BLENDER_INTRINSICS, 800, 800, opt.num_rays, generator=self.gen)
            bg = torch.rand(1, opt.num_rays, 3, device=self.device, generator=self.gen)
            gt = self.target(r["rays_o"], r["rays_d"], bg)
            _, pred = tr.train_step(r["rays_o"], r["rays_d"], gt, bg)
            last = (pred, gt)
        from .trainer import psnr
        self.teacher_psnr = float(psnr(last[0].detach(), last[1]))
        self.tea.teacher_variant = False
        self.tea.args = self.tea.opt = opt
        for p in self.tea.parameters():
            p.grad = None

    def next_batch(self):
        opt = self.opt
        pose = self.poses[self.step_idx % len(self.poses)][None]
        self.step_idx += 1
        r = get_rays(pose, BLENDER_INTRINSICS, 800, 800, opt.num_rays, generator=self.gen)
        # synthetic data has alpha: random background per ray (utils.py:987-995)
        bg = torch.rand(1, opt.num_rays, 3, device=self.device, generator=self.gen)
        return r["rays_o"], r["rays_d"], bg

    def device_batch(self):
        """next_batch() with nothing on the host: pose index and RNG state live on the device, so the batch
        generation can sit inside a captured HIP graph (default CUDA generator, graph-safe)."""
        opt = self.opt
        mk = getattr(self.ops, "make_batch", None)
        if mk is not None:  # pose selection, pixel ids, rays, background and near/far in one kernel
            if not hasattr(self, "_batch_state"):
                self._batch_state = torch.zeros(3, dtype=torch.int64, device=self.device)
                self._poses_c = self.poses.float().contiguous()
                # every ray-DP rank must draw its own pixels: the rank is part of the key (the CUDA seed alone is the same
                # on every rank unless the caller seeds per rank)
                self._batch_seed = 0x5eed + 1000003 * int(torch.cuda.initial_seed() % (2 ** 31)) + 0x9E3779B1 * self.dp_rank
            return mk(self._poses_c, self._batch_state, self._batch_seed, BLENDER_INTRINSICS, 800, 800, opt.num_rays,
                      self.stu.aabb_train, self.stu.min_near)
        if not hasattr(self, "_pose_idx"):
            self._pose_idx = torch.zeros(1, dtype=torch.long, device=self.device)
        pose = self.poses.index_select(0, self._pose_idx)
        self._pose_idx.add_(1).remainder_(len(self.poses))
        fused = getattr(self.ops, "get_rays", None)
        r = fused(pose, BLENDER_INTRINSICS, 800, 800, opt.num_rays) if fused else get_rays(pose, BLENDER_INTRINSICS, 800, 800, opt.num_rays)
        bg = torch.rand(1, opt.num_rays, 3, device=self.device)
        return r["rays_o"], r["rays_d"], bg

    def enable_graph(self, steps_per_graph=1):
        """Whole-step hipGraph capture (GPU only).  steps_per_graph > 1: every step() call then runs that many training steps
        (one graph launch); `steps_per_call` says how many."""
        assert not self.opt.update_stu_extra, "update_stu_extra rewrites the occupancy grid between steps: run eagerly"
        self.trainer.capture_step(self.device_batch, steps_per_graph=steps_per_graph)
        self._graph = True

    @property
    def steps_per_call(self):
        return self.trainer.steps_per_replay if getattr(self, "_graph", False) else 1

    def step(self):
        if getattr(self, "_graph", False):
            return self.trainer.replay_step()
        if getattr(self, "_eager_device_batches", False):
            return self.trainer.train_step(*self.device_batch())
        return self.trainer.train_step(*self.next_batch())
