"""Test helper: the harness operator set (pvd.ops) bound to the CPU oracle instead of the HIP
library, so renderer / network / trainer logic can be exercised without a GPU.  TEST ONLY."""
import types

import oracle_backend as ob
from gridencoder.grid import GridEncoderBase, make_grid_encode
from raymarching.raymarching import make_ops
from shencoder.sphere_harmonics import SHEncoderBase, make_sh_encode

_rm = make_ops(ob.raymarching_backend, device_type="cpu")
_ge = make_grid_encode(ob.gridencoder_backend, device_type="cpu")
_sh = make_sh_encode(ob.shencoder_backend, device_type="cpu")


class OracleGridEncoder(GridEncoderBase):
    _grid_encode = staticmethod(_ge)


class OracleSHEncoder(SHEncoderBase):
    _sh_encode = staticmethod(_sh)


def oracle_ops():
    return types.SimpleNamespace(raymarching=_rm, GridEncoder=OracleGridEncoder, SHEncoder=OracleSHEncoder,
                                 device_type="cpu", name="oracle")
