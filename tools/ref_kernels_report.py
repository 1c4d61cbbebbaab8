#!/usr/bin/env python3
"""Parity report against the REFERENCE'S OWN KERNELS (tests/golden/reference_kernels.npz, written on an MI355X by
tests/golden/make_golden_ref_kernels.py from oracle/_ref): the CPU oracle and -- with a GPU -- libpvd_hip.so on the fixture's inputs.
Prints, per piece, how many values are bit-identical and the largest difference.   python tools/ref_kernels_report.py [--no-hip]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import oracle  # noqa: E402

G = dict(np.load(os.path.join(REPO, "tests", "golden", "reference_kernels.npz")))
use_hip = "--no-hip" not in sys.argv
if use_hip:
    import torch
    use_hip = torch.cuda.is_available()
if use_hip:
    import raymarching as RM
    import shencoder  # noqa: F401
    import pvd_hip
    dev = torch.device("cuda:0")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731


def line(name, ref, got, who):
    ref, got = np.asarray(ref), np.asarray(got)
    if ref.shape != got.shape:
        print("%-34s %-7s SHAPE %s vs %s" % (name, who, ref.shape, got.shape))
        return
    same = (ref.view(np.uint8) == got.view(np.uint8)).reshape(ref.shape + (-1,)).all(-1) if ref.dtype.kind == "f" else (ref == got)
    n = ref.size
    finite = np.isfinite(ref.astype(np.float64)) & np.isfinite(got.astype(np.float64))
    diff = np.abs(ref.astype(np.float64) - got.astype(np.float64))
    md = float(diff[finite].max()) if finite.any() else 0.0
    print("%-34s %-7s %8d of %8d identical (%.4f%%), max |diff| %.3g" % (name, who, int(same.sum()), n, 100.0 * same.sum() / max(n, 1), md))


def cat_by_ray(rays, xyzs, dirs, deltas, N):
    counts = np.zeros(N, np.int32)
    parts = {}
    for idx, off, num in rays:
        counts[idx] = num
        parts[int(idx)] = (xyzs[off:off + num], dirs[off:off + num], deltas[off:off + num])
    order = [parts[i] for i in range(N) if i in parts]
    c = lambda k: np.concatenate([p[k] for p in order]) if order else np.zeros((0, 3 if k < 2 else 2), np.float32)  # noqa: E731
    return counts, c(0), c(1), c(2)


# ---- near / far, polar, Morton, packbits
n, f = oracle.near_far_from_aabb(G["nf_o"], G["nf_d"], G["nf_aabb"], 0.2)
line("near_far_from_aabb nears", G["nf_nears"], n, "oracle"); line("near_far_from_aabb fars", G["nf_fars"], f, "oracle")
line("polar_from_ray", G["polar"], oracle.polar_from_ray(G["nf_o"], G["nf_d"], 2.0), "oracle")
line("morton3D", G["mo_idx"], oracle.morton3D(G["mo_coords"]), "oracle"); line("morton3D_invert", G["mo_back"], oracle.morton3D_invert(G["mo_idx"]), "oracle")
line("packbits", G["pb_bits"], oracle.packbits(G["pb_grid"], 10.0), "oracle")
if use_hip:
    o, d = T(G["nf_o"]), T(G["nf_d"])
    nn, ff = RM.near_far_from_aabb(o, d, T(G["nf_aabb"]), 0.2)
    line("near_far_from_aabb nears", G["nf_nears"], nn.cpu().numpy(), "hip"); line("near_far_from_aabb fars", G["nf_fars"], ff.cpu().numpy(), "hip")
    line("polar_from_ray", G["polar"], RM.polar_from_ray(o, d, 2.0).cpu().numpy(), "hip")
    line("morton3D", G["mo_idx"], RM.morton3D(T(G["mo_coords"])).cpu().numpy(), "hip")
    line("morton3D_invert", G["mo_back"], RM.morton3D_invert(T(G["mo_idx"])).cpu().numpy(), "hip")
    line("packbits", G["pb_bits"], RM.packbits(T(G["pb_grid"]).view(1, -1), 10.0).cpu().numpy().reshape(-1), "hip")

# ---- march_rays_train
for tag in ("a", "b"):
    bound, C, dtg = float(G["m%s_cfg" % tag][0]), int(G["m%s_cfg" % tag][1]), float(G["m%s_cfg" % tag][2])
    o, d, bits, nears, fars = (G["m%s_%s" % (tag, k)] for k in ("o", "d", "bits", "nears", "fars"))
    for perturb in (0, 1):
        N = o.shape[0]
        cnt = G["m%s%d_counts" % (tag, perturb)]
        ref = [cnt, G["m%s%d_xyzs" % (tag, perturb)], np.repeat(d, cnt, axis=0), G["m%s%d_deltas" % (tag, perturb)]]  # (dirs: checked by the generator, not stored)
        M = int(ref[0].sum()) + 128
        x, dd, dl, rays, counter = oracle.march_rays_train(o, d, bits, bound, C, 128, nears, fars, M, perturb=bool(perturb), dt_gamma=dtg)
        got = cat_by_ray(rays, x, dd, dl, N)
        for k, nm in enumerate(("counts", "xyzs", "dirs", "deltas")):
            line("march_train %s p%d %s" % (tag, perturb, nm), ref[k], got[k], "oracle")
        if use_hip:
            xh, dh, lh, rh = RM.march_rays_train(T(o), T(d), bound, T(bits), C, 128, T(nears), T(fars), None, M, bool(perturb), -1, False, dtg, 1024)
            goth = cat_by_ray(rh.cpu().numpy(), xh.cpu().numpy(), dh.cpu().numpy(), lh.cpu().numpy(), N)
            for k, nm in enumerate(("counts", "xyzs", "dirs", "deltas")):
                line("march_train %s p%d %s" % (tag, perturb, nm), ref[k], goth[k], "hip")

# ---- compositing
ws, dep, img = oracle.composite_rays_train_forward(G["cp_sig"], G["cp_rgb"], G["cp_deltas"], G["cp_rays"])
gs, gr = oracle.composite_rays_train_backward(G["cp_gws"], G["cp_gimg"], G["cp_sig"], G["cp_rgb"], G["cp_deltas"], G["cp_rays"], G["cp_ws"], G["cp_image"])
for nm, r, g in (("ws", G["cp_ws"], ws), ("depth", G["cp_depth"], dep), ("image", G["cp_image"], img), ("grad_sigmas", G["cp_gsig"], gs), ("grad_rgbs", G["cp_grgb"], gr)):
    line("composite_train %s" % nm, r, g, "oracle")
if use_hip:
    sig, rgb = T(G["cp_sig"]).requires_grad_(True), T(G["cp_rgb"]).requires_grad_(True)
    w_, d_, i_ = RM.composite_rays_train(sig, rgb, T(G["cp_deltas"]), T(G["cp_rays"]))
    (w_ * T(G["cp_gws"])).sum().add((i_ * T(G["cp_gimg"])).sum()).backward()
    for nm, r, g in (("ws", G["cp_ws"], w_), ("depth", G["cp_depth"], d_), ("image", G["cp_image"], i_), ("grad_sigmas", G["cp_gsig"], sig.grad), ("grad_rgbs", G["cp_grgb"], rgb.grad)):
        line("composite_train %s" % nm, r, g.detach().cpu().numpy(), "hip")

# ---- spherical harmonics
NS = G["sh_dirs"].shape[0]
for deg in range(1, 9):
    out, dy = oracle.sh_encode_forward(G["sh_dirs"], deg, True)
    gi = oracle.sh_encode_backward(G["sh%d_g" % deg], G["sh_dirs"], deg, G["sh%d_dy" % deg])
    line("sh deg %d values" % deg, G["sh%d_out" % deg], out, "oracle"); line("sh deg %d dy_dx" % deg, G["sh%d_dy" % deg], dy, "oracle")
    line("sh deg %d grad_inputs" % deg, G["sh%d_gi" % deg], gi, "oracle")
    if use_hip:
        oh = torch.empty(NS, deg * deg, device=dev)
        dyh = torch.empty(NS, 3 * deg * deg, device=dev)
        pvd_hip.sh_encode_forward(T(G["sh_dirs"]), oh, NS, 3, deg, True, dyh)
        gih = torch.zeros(NS, 3, device=dev)
        pvd_hip.sh_encode_backward(T(G["sh%d_g" % deg]), T(G["sh_dirs"]), NS, 3, deg, dyh, gih)
        line("sh deg %d values" % deg, G["sh%d_out" % deg], oh.cpu().numpy(), "hip"); line("sh deg %d dy_dx" % deg, G["sh%d_dy" % deg], dyh.cpu().numpy(), "hip")
        line("sh deg %d grad_inputs" % deg, G["sh%d_gi" % deg], gih.cpu().numpy(), "hip")

# ---- inference trio
alive = np.arange(1024, dtype=np.int32)
for perturb in (0, 1):
    x, dd, dl = oracle.march_rays(1024, 4, alive, G["inf_nears"].copy(), G["inf_o"], G["inf_d"], 1.0, G["inf_bits"], 1, 128, G["inf_nears"], G["inf_fars"], perturb=perturb)
    line("march_rays p%d xyzs" % perturb, G["inf%d_xyzs" % perturb], x, "oracle"); line("march_rays p%d deltas" % perturb, G["inf%d_deltas" % perturb], dl, "oracle")
    if use_hip:
        xh, dh, lh = RM.march_rays(1024, 4, T(alive), T(G["inf_nears"].copy()), T(G["inf_o"]), T(G["inf_d"]), 1.0, T(G["inf_bits"]), 1, 128, T(G["inf_nears"]), T(G["inf_fars"]),
                                   -1, perturb, 0, 1024)
        line("march_rays p%d xyzs" % perturb, G["inf%d_xyzs" % perturb], xh.cpu().numpy(), "hip"); line("march_rays p%d deltas" % perturb, G["inf%d_deltas" % perturb], lh.cpu().numpy(), "hip")
rt, ws, dep, img = G["inf_nears"].copy(), np.zeros(1024, np.float32), np.zeros(1024, np.float32), np.zeros((1024, 3), np.float32)
al = alive.copy()
oracle.composite_rays(1024, 4, al, rt, G["inf_sig"], G["inf_rgb"], G["inf1_deltas"], ws, dep, img)
for nm, r, g in (("ws", G["inf_ws"], ws), ("depth", G["inf_depth"], dep), ("image", G["inf_image"], img), ("rays_t", G["inf_t_after"], rt), ("rays_alive", G["inf_alive_after"], al)):
    line("composite_rays %s" % nm, r, g, "oracle")
ca, ct, k = oracle.compact_rays(1024, G["inf_alive_after"], G["inf_t_after"])
order = np.argsort(ca[:k])
line("compact_rays alive (sorted)", G["inf_compact_alive"], ca[:k][order], "oracle"); line("compact_rays t (sorted)", G["inf_compact_t"], ct[:k][order], "oracle")
if use_hip:
    rt, ws, dep, img = T(G["inf_nears"].copy()), torch.zeros(1024, device=dev), torch.zeros(1024, device=dev), torch.zeros(1024, 3, device=dev)
    al = T(alive.copy())
    RM.composite_rays(1024, 4, al, rt, T(G["inf_sig"]), T(G["inf_rgb"]), T(G["inf1_deltas"]), ws, dep, img)
    for nm, r, g in (("ws", G["inf_ws"], ws), ("depth", G["inf_depth"], dep), ("image", G["inf_image"], img), ("rays_t", G["inf_t_after"], rt), ("rays_alive", G["inf_alive_after"], al)):
        line("composite_rays %s" % nm, r, g.cpu().numpy(), "hip")
