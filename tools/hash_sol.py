#!/usr/bin/env python3
"""Speed-of-light study of the hash-grid lookup (tools/probes/hash_sol.hip): gather-only kernels on the product kernel's
address stream, one thing varied at a time, next to the product kernel itself (pvd_hash_head_forward_fused) on the same
samples.  Times are per launch inside HIP graphs of 20 launches (5 replays), HIP events on the launch stream.

    python tools/hash_sol.py                 # the table
    python tools/hash_sol.py --pmc NAME      # 30 eager launches of one row, for a rocprofv3 --pmc pass
"""
import argparse
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tools")]
import numpy as np
import torch

import fusedhead
import pvd_hip
from bench_grid_levels import samples
from pvd.config import PVDConfig
from pvd.ops import hip_ops
from pvd.workload import make_model

ap = argparse.ArgumentParser()
ap.add_argument("--pmc", default=None)
ap.add_argument("--rows", default=None, help="only rows whose name contains this substring")
args = ap.parse_args()

dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(REPO, "tools", "probes", "libhash_sol.so"))
m = make_model(hip_ops(), PVDConfig(model_type="hash"), "hash", True, dev).eval()
m.encoder.embeddings.data.uniform_(-0.3, 0.3)
enc = m.encoder
emb16 = enc.embeddings.detach().half().contiguous()
x01 = samples()  # [M, 3] in [0, 1], ray order
M = x01.shape[0]
d = torch.randn_like(x01)
d = d / d.norm(dim=-1, keepdim=True)
offs = enc.offsets.cpu().numpy().astype(np.int32)
S = float(np.log2(enc.per_level_scale))
scales = (np.exp2(np.arange(14, dtype=np.float32) * np.float32(S)) * np.float32(enc.base_resolution) - np.float32(1)).astype(np.float32)
out = torch.zeros(2 * M, dtype=torch.int32, device=dev)
ALG = 516 * M  # algorithmic bytes of the lookup per launch


def morton_sorted(x):
    """samples sorted by the Morton code of their 32^3 cell: neighbours in the array are neighbours in space"""
    c = (x.clamp(0, 1 - 1e-6) * 32).long()

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v

    code = spread(c[:, 0]) | (spread(c[:, 1]) << 1) | (spread(c[:, 2]) << 2)
    return x[torch.argsort(code)].contiguous()


x_sorted = morton_sorted(x01)
stream_src = torch.empty(ALG // 4 + 16, dtype=torch.int32, device=dev).random_()


def p(t):
    return ctypes.c_void_p(t.data_ptr())


def gather(variant, x=None, row_mask=0xFFFFFFFF, perm=0, blocks=0):
    x = x01 if x is None else x
    s = torch.cuda.current_stream().cuda_stream
    rc = lib.sol_gather(ctypes.c_int(variant), p(x), p(emb16), offs.ctypes.data_as(ctypes.c_void_p), scales.ctypes.data_as(ctypes.c_void_p),
                        ctypes.c_uint32(M), p(out), ctypes.c_uint32(row_mask), ctypes.c_uint32(perm), ctypes.c_uint32(blocks), ctypes.c_void_p(s))
    assert rc == 0, rc


def split(blocks, row_mask=0xFFFFFFFF):
    rc = lib.sol_split(p(x01), p(emb16), offs.ctypes.data_as(ctypes.c_void_p), scales.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(M), p(out),
                       ctypes.c_uint32(row_mask), ctypes.c_uint32(blocks), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc


stage = torch.zeros(28 * 2 * M, dtype=torch.int32, device=dev)  # 7 levels x 4 y-z corners x (2 lanes per sample): 224 B/sample


def split7(what, blocks, row_mask=0xFFFFFFFF):
    rc = lib.sol_split7(ctypes.c_int(what), p(x01), p(emb16), offs.ctypes.data_as(ctypes.c_void_p), scales.ctypes.data_as(ctypes.c_void_p),
                        ctypes.c_uint32(M), p(out), p(stage), ctypes.c_uint32(row_mask), ctypes.c_uint32(blocks),
                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc


def split7_pipeline(blocks):
    split7(1, blocks)
    split7(2, 0)


def stream(blocks):
    rc = lib.sol_stream(p(stream_src), ctypes.c_size_t(ALG), p(out), ctypes.c_uint32(blocks), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc


def empty(blocks):
    assert lib.sol_empty(p(out), ctypes.c_uint32(blocks), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0


xin = (x01 * 2 - 1).contiguous()
xin_sorted = (x_sorted * 2 - 1).contiguous()


def product(x=None):
    fusedhead.hash_head_infer(m, xin if x is None else x, d)


ALL_HIT = 0x3FFF  # 16 k rows = 64 KB per level: every level resident in every L2 (and mostly in L1)
ROWS = [
    ("empty kernel, 726 workgroups", lambda: empty(726), None),
    ("stream 516 B/sample, 726 wg", lambda: stream(726), ALG),
    ("stream 516 B/sample, 2048 wg", lambda: stream(2048), ALG),
    ("stream 516 B/sample, 8192 wg", lambda: stream(8192), ALG),
    ("product: lookup + head (fused), default", product, ALG),
    ("product (k_hash_fwd_fused<14>)", lambda: product(), ALG),
    ("gather G=7 (product's structure)", lambda: gather(0), ALG),
    ("gather G=14 (one round trip)", lambda: gather(1), ALG),
    ("gather G=4", lambda: gather(2), ALG),
    ("gather G=2", lambda: gather(3), ALG),
    ("gather G=1 (level by level)", lambda: gather(4), ALG),
    ("gather G=7, all-hit tables", lambda: gather(0, row_mask=ALL_HIT), ALG),
    ("gather G=14, all-hit tables", lambda: gather(1, row_mask=ALL_HIT), ALG),
    ("gather levels 0-9 only", lambda: gather(5), None),
    ("gather levels 10-13 only", lambda: gather(6), None),
    ("gather levels 10-13 only, all-hit", lambda: gather(6, row_mask=ALL_HIT), None),
    ("split map: levels 10-13, pair-owned rows, 2904 wg", lambda: split(2904), None),
    ("split map: levels 10-13, pair-owned rows, 1456 wg", lambda: split(1456), None),
    ("split map: levels 10-13, pair-owned rows, 728 wg", lambda: split(728), None),
    ("split map, all-hit tables, 2904 wg", lambda: split(2904, row_mask=ALL_HIT), None),
    ("gather levels 0-6 only", lambda: gather(7), None),
    ("gather levels 7-13 only", lambda: gather(8), None),
    ("gather levels 7-13 only, all-hit", lambda: gather(8, row_mask=ALL_HIT), None),
    ("split map: levels 7-13, pair-owned rows, 2904 wg", lambda: split7(0, 2904), None),
    ("split map: levels 7-13, pair-owned rows, 1456 wg", lambda: split7(0, 1456), None),
    ("split map 7-13 + staging writes (224 B/sample), 2904 wg", lambda: split7(1, 2904), None),
    ("split map 7-13 + staging writes (224 B/sample), 1456 wg", lambda: split7(1, 1456), None),
    ("blend side: gather 0-6 + read staged 7-13", lambda: split7(2, 0), None),
    ("SPLIT PIPELINE: split 7-13 + staging, then blend side", lambda: split7_pipeline(1456), ALG),
    ("gather G=7, 64-sample workgroups", lambda: gather(9), ALG),
    ("gather G=14, 64-sample workgroups", lambda: gather(15), ALG),
    ("gather G=7, 256-sample workgroups", lambda: gather(10), ALG),
    ("gather G=7 pipelined, 256 wg", lambda: gather(11, blocks=256), ALG),
    ("gather G=7 pipelined, 512 wg", lambda: gather(11, blocks=512), ALG),
    ("gather G=4 pipelined, 512 wg", lambda: gather(12, blocks=512), ALG),
    ("gather G=2 pipelined, 726 wg", lambda: gather(13, blocks=726), ALG),
    ("gather G=7 pipelined 64-smp, 1024 wg", lambda: gather(14, blocks=1024), ALG),
    ("gather G=14, nontemporal loads", lambda: gather(16), ALG),
    ("gather G=14, nontemporal levels 7-13", lambda: gather(17), ALG),
    ("gather G=14, sc1 loads", lambda: gather(18), ALG),
    ("gather G=14, sc1 levels 7-13", lambda: gather(19), ALG),
    ("gather G=14, sc1 levels 10-13", lambda: gather(20), ALG),
    ("gather G=7, Morton-sorted samples", lambda: gather(0, x=x_sorted), ALG),
    ("gather G=7, sorted + XCD-contiguous", lambda: gather(0, x=x_sorted, perm=1), ALG),
    ("gather G=7, unsorted + XCD-contiguous", lambda: gather(0, perm=1), ALG),
    ("product on Morton-sorted samples", lambda: product(xin_sorted), ALG),
]


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 100 * 1e3
        best = us if best is None else min(best, us)
    return best


if args.pmc:
    fn = dict((r[0], r[1]) for r in ROWS)[args.pmc]
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    print("samples_per_launch", M)
else:
    print("samples per launch: %d   algorithmic bytes (516 B/sample): %.2f MB" % (M, ALG / 1e6))
    print("%-56s %9s %10s %8s" % ("row", "us/launch", "GB/s @516", "of 8TB/s"))
    for name, fn, nbytes in ROWS:
        if args.rows and args.rows not in name:
            continue
        us = timed(fn)
        if nbytes:
            print("%-56s %9.2f %10.0f %8.3f" % (name, us, nbytes / us / 1e3, nbytes / us / 1e3 / 8000.0))
        else:
            print("%-56s %9.2f" % (name, us))
