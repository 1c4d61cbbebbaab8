"""Drop-in for the reference's ``raymarching`` package (raymarching/__init__.py:1), HIP backend only."""
from pvd_hip import raymarching_backend as _backend

from .raymarching import make_ops

_ops = make_ops(_backend, device_type="cuda")

near_far_from_aabb = _ops.near_far_from_aabb
polar_from_ray = _ops.polar_from_ray
morton3D = _ops.morton3D
morton3D_invert = _ops.morton3D_invert
packbits = _ops.packbits
march_rays_train = _ops.march_rays_train
composite_rays_train = _ops.composite_rays_train
composite_rays_train_bg = _ops.composite_rays_train_bg  # extension: + run_cuda's epilogue
march_rays = _ops.march_rays
composite_rays = _ops.composite_rays
compact_rays = _ops.compact_rays

# inference rounds whose state stays on the device (pvd_infer_*; the reference reads the alive count back every round)
infer_round_begin, infer_compact, infer_march, infer_composite = _ops.infer_round_begin, _ops.infer_compact, _ops.infer_march, _ops.infer_composite
INFER_STATE_INTS = _ops.INFER_STATE_INTS
