"""``plenoxel`` -- dense-volume ("tensors" model) lookup + SH colour head on libpvd_hip.so (HIP backend only).

In the reference this is torch code inside the model (NeRFNetwork.compute_plenoxel_fea and the
``tensors`` branch of forward, distill_mutual/network.py:311-322, 383-409: a 3-D F.grid_sample over a
channel-major [1,28,128,128,128] parameter, clamp, trunc_exp, SH dot product, sigmoid); here it is one
forward and one backward kernel over a channels-last volume."""
from pvd_hip import plenoxel_backend as _backend

from .volume import make_plenoxel_ops, to_channels_last_3d_param, is_channels_last_3d

plenoxel_features, plenoxel_head = make_plenoxel_ops(_backend, device_type="cuda")
