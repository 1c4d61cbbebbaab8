"""Drop-in for the reference's ``gridencoder`` package (gridencoder/__init__.py:1), HIP backend only."""
from pvd_hip import gridencoder_backend as _backend

from .grid import GridEncoderBase, make_grid_encode

grid_encode = make_grid_encode(_backend, device_type="cuda")


class GridEncoder(GridEncoderBase):
    """reference: gridencoder.grid.GridEncoder (grid.py:142-232)"""

    _grid_encode = staticmethod(grid_encode)
