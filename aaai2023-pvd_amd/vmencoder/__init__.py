"""``vmencoder`` -- TensoRF plane x line feature lookup on libpvd_hip.so (HIP backend only).

In the reference this piece is torch code inside the model (NeRFNetwork.get_sigma_feat /
get_color_feat, distill_mutual/network.py:216-309: twelve F.grid_sample calls over channel-major
tables); here it is one fused forward and one fused backward kernel over channels-last tables."""
from pvd_hip import vmencoder_backend as _backend

from .vm import make_vm_encode, to_channels_last_param

vm_encode = make_vm_encode(_backend, device_type="cuda")
