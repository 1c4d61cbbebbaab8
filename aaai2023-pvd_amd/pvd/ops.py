"""Operator set used by the harness.  ``hip_ops()`` binds the product's HIP packages; the
test-suite builds the same namespace around the CPU oracle (tests/oracle_ops.py).  The product
never constructs anything but the HIP set."""
import types


def hip_ops():
    import gridencoder
    import raymarching
    import shencoder
    import vmencoder

    return types.SimpleNamespace(raymarching=raymarching, GridEncoder=gridencoder.GridEncoder, SHEncoder=shencoder.SHEncoder,
                                 vm_encode=vmencoder.vm_encode, device_type="cuda", name="hip")
