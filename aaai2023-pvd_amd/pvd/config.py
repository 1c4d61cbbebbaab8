"""Flag names and defaults of the reference's two entry points, as one dataclass.

The reference splats ``vars(opt)`` into ``render(**kwargs)`` (distill_mutual/utils.py:1005), so
the flag names double as keyword names of ``run_cuda``; they are kept verbatim.
Defaults: main_distill_mutual.py:43-254, main_just_train_tea.py:20-227.
"""
from dataclasses import dataclass, field


@dataclass
class PVDConfig:
    # rendering
    bound: float = 1.0
    scale: float = 0.8
    dt_gamma: float = 0.0
    min_near: float = 0.2
    density_thresh: float = 10.0
    bg_radius: float = -1.0
    grid_size: int = 128
    max_steps: int = 1024
    num_steps: int = 512       # fixed steps per ray when NOT cuda_ray (main_just_train_tea.py:45-50)
    upsample_steps: int = 0    # (main_just_train_tea.py:51-56)
    max_ray_batch: int = 4096
    num_rays: int = 4096
    # training
    iters: int = 30000
    lr: float = 1e-2
    fp16: bool = True
    cuda_ray: bool = True
    seed: int = 0
    update_extra_interval: int = 16
    # distillation
    teacher_type: str = "hash"
    model_type: str = "vm"
    loss_type: str = "normL2"
    loss_rate_rgb: float = 1.0
    loss_rate_fea_sc: float = 0.002
    loss_rate_color: float = 0.002
    loss_rate_sigma: float = 0.002
    l1_reg_weight: float = 1e-4
    sigma_clip_min: float = -2.0
    sigma_clip_max: float = 7.0
    render_stu_first: bool = True
    update_stu_extra: bool = False
    data_type: str = "synthetic"  # which random-camera generator feeds the distillation: synthetic | llff | tank (main_distill_mutual.py:206-212)
    stage_iters: dict = field(default_factory=lambda: {"stage1": 2000, "stage2": 5000})
    global_step: int = 0
    # model zoo
    PE: int = 10
    nerf_layer_num: int = 8
    nerf_layer_wide: int = 256
    skip: int = 3
    resolution0: int = 300
    plenoxel_degree: int = 3
    plenoxel_res: str = "[128,128,128]"
    enable_edit_plenoxel: bool = False

    def __post_init__(self):
        # plenoxel has no feature head: stage 1 disabled (main_distill_mutual.py:243-246)
        if "tensors" in (self.model_type, self.teacher_type):
            self.stage_iters = dict(self.stage_iters, stage1=-1)
