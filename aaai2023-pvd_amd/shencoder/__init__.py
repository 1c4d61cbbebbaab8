"""Drop-in for the reference's ``shencoder`` package (shencoder/__init__.py:1), HIP backend only."""
from pvd_hip import shencoder_backend as _backend

from .sphere_harmonics import SHEncoderBase, make_sh_encode

sh_encode = make_sh_encode(_backend, device_type="cuda")


class SHEncoder(SHEncoderBase):
    """reference: shencoder.sphere_harmonics.SHEncoder (sphere_harmonics.py:67-95)"""

    _sh_encode = staticmethod(sh_encode)
