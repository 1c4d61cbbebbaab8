"""``vmencoder`` -- TensoRF plane x line feature lookup on libpvd_hip.so (HIP backend only).

In the reference this piece is torch code inside the model (NeRFNetwork.get_sigma_feat /
get_color_feat, distill_mutual/network.py:216-309: twelve F.grid_sample calls over channel-major
tables); here it is one fused forward and one fused backward kernel over channels-last tables."""
from pvd_hip import vmencoder_backend as _backend

from .vm import make_vm_encode, to_channels_last_param

vm_encode = make_vm_encode(_backend, device_type="cuda")


def vm_encode_infer(xyz, aabb_host, *tables, rows_dev=None):
    """vm_encode without autograd state (inference): (sigma_feat [M] f32, color_prod [M,144] f16).  rows_dev: optional DEVICE
    int32 row count -- only the first min(M, rows_dev) rows are computed (the inference rounds of NeRFRenderer)."""
    import torch
    from .vm import is_channels_last
    xyz = xyz.contiguous().float()
    assert all(is_channels_last(t) for t in tables), "VM factors must be stored channels-last"
    res = [tables[0].shape[3], tables[0].shape[2], tables[1].shape[2]]
    M = xyz.shape[0]
    sigma_feat = torch.empty(M, dtype=torch.float32, device=xyz.device)
    color_prod = torch.empty(M, 144, dtype=torch.float16, device=xyz.device)
    _backend.vm_forward(xyz, aabb_host, [t.detach() for t in tables], res, sigma_feat, color_prod, rows_dev=rows_dev)
    return sigma_feat, color_prod
