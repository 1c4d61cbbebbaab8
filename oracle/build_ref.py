#!/usr/bin/env python3
"""oracle/_ref -- the REFERENCE's own kernels, built for gfx950.  TEST INFRASTRUCTURE ONLY (the `-m gpu` tests use the result as a second
checker next to the CPU oracle, and tests/golden/make_golden_ref_kernels.py turns its outputs into fixtures that pin the oracle).

The reference builds its three CUDA extensions with `torch.utils.cpp_extension.load` (raymarching/backend.py, gridencoder/backend.py,
shencoder/backend.py).  On PyTorch-ROCm that very call translates the `.cu` sources to HIP (torch's own hipify step) and compiles them
with hipcc -- no source of the reference is edited, no header or library is stood in for.  This recipe does what those backend.py files
do, for the sources where they lie under /root/reference:

    raymarching/src/{raymarching.cu, bindings.cpp, pcg32.h, raymarching.h}   -> oracle/_ref/_raymarching_ref.so    builds
    shencoder/src/{shencoder.cu, bindings.cpp, shencoder.h}                  -> oracle/_ref/_shencoder_ref.so      builds
    gridencoder/src/{gridencoder.cu, ...}                                    -> UNBUILDABLE here: kernel_grid_backward calls
        atomicAdd(__half2*, __half2) (gridencoder.cu:297-304), which CUDA provides and HIP does not (ROCm 7.2 spells it
        unsafeAtomicAdd); supplying the overload would be a stand-in, so the grid encoder stays pinned by the CPU oracle alone.

torch's build writes its translated copies next to the sources it is given, so the sources are first copied to a scratch directory
under $TMPDIR (never into the repository, never into /root/reference, which is read-only); only the two `.so` files are kept, in
oracle/_ref/ (git-ignored, travels to the GPU box with the tree).  Needs no GPU (hipcc cross-compiles); ~1 min per extension.

    python oracle/build_ref.py [--force]"""
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REFERENCE = os.environ.get("PVD_REFERENCE", "/root/reference")
MODULES = {  # name -> (directory under the reference, translation unit)
    "_raymarching_ref": ("raymarching/src", "raymarching.cu"),
    "_shencoder_ref": ("shencoder/src", "shencoder.cu"),
}


def available():
    """the modules whose .so exists in oracle/_ref"""
    return [m for m in MODULES if os.path.exists(os.path.join(OUT, m + ".so"))]


def build(force=False, verbose=False):
    if not os.path.isdir(REFERENCE):
        return available()  # (the GPU box: only the prebuilt files travel)
    os.makedirs(OUT, exist_ok=True)
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    os.environ.setdefault("MAX_JOBS", "4")
    from torch.utils.cpp_extension import load
    for name, (sub, cu) in MODULES.items():
        src_dir = os.path.join(REFERENCE, sub)
        dst = os.path.join(OUT, name + ".so")
        newest = max(os.path.getmtime(os.path.join(src_dir, f)) for f in os.listdir(src_dir))
        if not force and os.path.exists(dst) and os.path.getmtime(dst) >= newest:
            continue
        scratch = tempfile.mkdtemp(prefix="pvd_ref_")
        try:
            for f in os.listdir(src_dir):
                shutil.copy(os.path.join(src_dir, f), os.path.join(scratch, f))
            # what the reference's backend.py passes, minus the nvcc-only spellings (-std=c++14 predates this torch; the -U__CUDA_NO_HALF_*
            # switches are -U__HIP_NO_HALF_* on this platform)
            built = load(name=name, sources=[os.path.join(scratch, cu), os.path.join(scratch, "bindings.cpp")], extra_cflags=["-O3", "-std=c++17"],
                         extra_cuda_cflags=["-O3", "-std=c++17", "-U__HIP_NO_HALF_OPERATORS__", "-U__HIP_NO_HALF_CONVERSIONS__", "-U__HIP_NO_HALF2_OPERATORS__"],
                         build_directory=scratch, verbose=verbose, is_python_module=False)
            shutil.copy(built, dst)
        finally:
            shutil.rmtree(scratch, ignore_errors=True)
    return available()


def load_module(name):
    """import oracle/_ref/<name>.so (a torch extension module: the reference's pybind11 bindings, bindings.cpp)"""
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)
    path = os.path.join(OUT, name + ".so")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    got = build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print("oracle/_ref:", ", ".join(got) if got else "nothing built")
