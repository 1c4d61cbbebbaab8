"""``plenoxel`` -- dense-volume ("tensors" model) lookup + SH colour head on libpvd_hip.so (HIP backend only).

In the reference this is torch code inside the model (NeRFNetwork.compute_plenoxel_fea and the
``tensors`` branch of forward, distill_mutual/network.py:311-322, 383-409: a 3-D F.grid_sample over a
channel-major [1,28,128,128,128] parameter, clamp, trunc_exp, SH dot product, sigmoid); here it is one
forward and one backward kernel over a channels-last volume."""
from pvd_hip import plenoxel_backend as _backend

from .volume import make_plenoxel_ops, to_channels_last_3d_param, is_channels_last_3d

plenoxel_features, plenoxel_head = make_plenoxel_ops(_backend, device_type="cuda")


def infer_image(model, rays_o, rays_d, nears, fars, dt_gamma, max_steps):
    """(weights_sum, depth, image) of the eval branch's round loop (renderer.py:450-543) for a frozen Plenoxel model, as ONE persistent
    launch (pvd_infer_image_plenoxel): rays [N,3], nears / fars [N]; the accumulators as the loop leaves them (before background
    compositing)."""
    import torch
    vol = model.tensor_volume[0].detach()
    assert is_channels_last_3d(vol), "the Plenoxel volume must be stored channels-last"
    dev, N = rays_o.device, rays_o.shape[0]
    a = model.args
    f32 = dict(dtype=torch.float32, device=dev)
    weights_sum, depth, img = torch.zeros(N, **f32), torch.zeros(N, **f32), torch.zeros(N, 3, **f32)
    workspace = torch.empty(2 * N + 12, dtype=torch.int32, device=dev)
    model._last_infer_workspace = workspace  # (tools/bench_render.py reads the launch's statistics)
    _backend.infer_image_plenoxel(rays_o.float().contiguous(), rays_d.float().contiguous(), nears.float().contiguous(), fars.float().contiguous(),
                                  model.density_bitfield, float(model.bound), float(dt_gamma), int(max_steps), int(model.cascade), int(model.grid_size),
                                  float(model.density_scale), model._aabb(), vol, int(model.plenoxel_degree), a.sigma_clip_min, a.sigma_clip_max,
                                  workspace, weights_sum, depth, img)
    return weights_sum, depth, img
