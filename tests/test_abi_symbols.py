"""The C-ABI library loads and exports every symbol include/pvd_hip.h declares (no compute calls:
this runs without a GPU; hipcc cross-compiles gfx950 on CPU)."""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def declared_symbols():
    src = open(os.path.join(REPO, "include", "pvd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pvd_[a-zA-Z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    assert len(syms) == 75, syms
    for must in ("pvd_march_rays_train", "pvd_composite_rays_train_forward", "pvd_composite_rays_train_backward",
                 "pvd_grid_encode_forward", "pvd_grid_encode_backward", "pvd_sh_encode_forward", "pvd_near_far_from_aabb"):
        assert must in syms


def test_library_exports_every_declared_symbol(hip_lib_built):
    lib = ctypes.CDLL(hip_lib_built)
    for s in declared_symbols():
        assert hasattr(lib, s), "libpvd_hip.so does not export %s" % s
    lib.pvd_abi_version.restype = ctypes.c_int
    assert lib.pvd_abi_version() == 6
    lib.pvd_status_string.restype = ctypes.c_char_p
    assert lib.pvd_status_string(-2) and lib.pvd_status_string(0) == b"ok"


def test_binding_lists_the_same_entry_points(hip_lib_built):
    import pvd_hip
    assert sorted(pvd_hip.ENTRY_POINTS) == declared_symbols()


def test_product_fails_loudly_without_gpu(hip_lib_built):
    """No CPU fallback: CPU tensors are rejected by the binding, and the public operators try to
    move inputs to the GPU like the reference does (raymarching.py:35-38) and raise without one."""
    import pytest
    import torch
    import pvd_hip
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    x = torch.zeros(4, 3)
    with pytest.raises(pvd_hip.PvdHipError):
        pvd_hip.near_far_from_aabb(x, x, torch.zeros(6), 4, 0.2, torch.zeros(4), torch.zeros(4))
    import raymarching
    with pytest.raises(Exception):
        raymarching.near_far_from_aabb(x, x, torch.zeros(6), 0.2)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "aaai2023-pvd_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "pvd_oracle" not in txt, os.path.join(root, f)


def test_the_binding_settles_the_hardware_queues_before_the_runtime_starts():
    """pvd_hip/__init__.py: GPU_MAX_HW_QUEUES = 2 unless the caller exported a value (the HIP runtime reads it at its first
    call); graphs with parallel chains are only recorded under the validated setting (forked_graphs_ok), with an override."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(os.path.dirname(here), "aaai2023-pvd_amd")
    code = ("import os, sys; sys.path.insert(0, %r); import torch, pvd_hip; "
            "print(os.environ.get('GPU_MAX_HW_QUEUES'), pvd_hip.HW_QUEUES, pvd_hip.HW_QUEUES_SOURCE.split()[0], pvd_hip.forked_graphs_ok())" % pkg)

    def run(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "PVD_FORKED_GRAPHS")}
        e.update(env)
        p = subprocess.run([sys.executable, "-c", code], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        return p.stdout.decode().split()
    assert run() == ["2", "2", "package", "True"]
    assert run(GPU_MAX_HW_QUEUES="4") == ["4", "4", "caller", "False"]
    assert run(GPU_MAX_HW_QUEUES="4", PVD_FORKED_GRAPHS="1") == ["4", "4", "caller", "True"]
    assert run(PVD_FORKED_GRAPHS="0") == ["2", "2", "package", "False"]


def test_span_records_read_as_microseconds(hip_lib_built):
    """pvd_hip.fused_span_us: {start, end} records of pvd_hash_head_forward_fused_span in ticks of the device's 100 MHz counter ->
    microseconds; a record nobody wrote (FUSED_SPAN_INIT: unsigned ~0 / 0) reads as NaN.  (Host-side helper: no GPU involved.)"""
    import numpy as np
    import torch
    import pvd_hip
    rec = torch.tensor([list(pvd_hip.FUSED_SPAN_INIT), [1000, 3350], [2 ** 40, 2 ** 40 + 5]], dtype=torch.int64)
    us = pvd_hip.fused_span_us(rec)
    assert np.isnan(us[0]) and us[1] == 23.5 and abs(us[2] - 0.05) < 1e-12
    assert np.isnan(pvd_hip.fused_span_us(rec[0]))  # a single record
