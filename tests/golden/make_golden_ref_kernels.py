#!/usr/bin/env python3
"""Golden vectors from the REFERENCE'S OWN KERNELS (oracle/_ref: raymarching.cu and shencoder.cu of /root/reference, built for gfx950 by
oracle/build_ref.py with PyTorch-ROCm's torch.utils.cpp_extension.load -- the recipe the reference's backend.py files use).

Runs where those kernels can run: on an MI355X box (`gpurun -- python tests/golden/make_golden_ref_kernels.py`), with oracle/_ref built
beforehand in the build container.  Writes
    tests/golden/reference_kernels.npz   seeded inputs + the reference kernels' outputs (what tests/test_oracle_ref_kernels.py pins the CPU
                                         oracle with, on the CPU, and tests/test_hip_reference_kernels.py the HIP kernels with)
and prints a parity report (reference vs CPU oracle vs libpvd_hip.so on the same inputs).

Order-free forms: the reference reserves sample slots and `rays` rows with atomicAdd (raymarching.cu:408-409), so WHICH row / offset a
ray gets depends on the run; what is stored per ray is its id, its sample count and its samples, re-ordered by ray id."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in (REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle.build_ref import load_module  # noqa: E402

DEV = torch.device("cuda:0")


def t(a, dtype=None):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV) if dtype is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


def rays_for(rs, n, radius=3.2):
    """n rays from a sphere of cameras towards a jittered target; a few misses and axis-parallel directions among them"""
    o = rs.randn(n, 3).astype(np.float32)
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * np.float32(radius)
    target = (rs.rand(n, 3).astype(np.float32) - 0.5) * np.float32(1.2)
    d = target - o
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    d[:4] = np.array([[1, 0, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)  # axis-parallel (division by zero in the slab test)
    o[:4] = np.array([[-3, 0.1, 0.2], [0.3, 3, -0.1], [0.0, 0.0, -3], [5, 5, 3]], np.float32)  # the last one misses the box
    d[4:8] = -d[4:8]  # looking away: misses
    return np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d.astype(np.float32))


def bitfield_for(cascade, bound, scale):
    from pvd.scene import ChairScene, packbits_torch
    grid = ChairScene(thicken=0.08, scale=scale).density_grid(128, bound, cascade, device="cpu")
    return packbits_torch(grid, 10.0).numpy()


def by_ray(rays, xyzs, dirs, deltas, N):
    """the reference's (row order, offsets) -> per ray id: count, and the samples concatenated in ray-id order"""
    rays = rays.cpu().numpy()
    xyzs, dirs, deltas = xyzs.cpu().numpy(), dirs.cpu().numpy(), deltas.cpu().numpy()
    counts = np.zeros(N, np.int32)
    parts = {}
    for idx, off, num in rays:
        counts[idx] = num
        parts[int(idx)] = (xyzs[off:off + num], dirs[off:off + num], deltas[off:off + num])
    order = [parts[i] for i in range(N) if i in parts]
    cat = lambda k: np.concatenate([p[k] for p in order]) if order else np.zeros((0, 3 if k < 2 else 2), np.float32)  # noqa: E731
    return counts, cat(0), cat(1), cat(2)


def main():
    rm = load_module("_raymarching_ref")
    sh = load_module("_shencoder_ref")
    G = {}
    rs = np.random.RandomState(0)

    # ---- F1 near / far, polar
    o, d = rays_for(rs, 256)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = torch.empty(256, device=DEV), torch.empty(256, device=DEV)
    rm.near_far_from_aabb(t(o), t(d), t(aabb), 256, 0.2, nears, fars)
    coords = torch.empty(256, 2, device=DEV)
    rm.polar_from_ray(t(o), t(d), 2.0, 256, coords)
    G.update(nf_o=o, nf_d=d, nf_aabb=aabb, nf_nears=nears.cpu().numpy(), nf_fars=fars.cpu().numpy(), polar=coords.cpu().numpy())

    # ---- F2 Morton, packbits
    c3 = rs.randint(0, 1024, size=(512, 3)).astype(np.int32)
    idx = torch.empty(512, dtype=torch.int32, device=DEV)
    rm.morton3D(t(c3), 512, idx)
    back = torch.empty(512, 3, dtype=torch.int32, device=DEV)
    rm.morton3D_invert(idx, 512, back)
    dens = (rs.rand(4096).astype(np.float32) * 20.0)
    dens[::7] = 10.0  # exactly the threshold
    bits = torch.empty(512, dtype=torch.uint8, device=DEV)
    rm.packbits(t(dens), 512, 10.0, bits)
    G.update(mo_coords=c3, mo_idx=idx.cpu().numpy(), mo_back=back.cpu().numpy(), pb_grid=dens, pb_bits=bits.cpu().numpy())

    # ---- F3 march_rays_train: {bound 1, one cascade, constant step} and {bound 2, two cascades, dt_gamma 1/256}, perturb 0 / 1
    NM = 128  # rays per marching case (the samples are the bulk of the fixture)
    for tag, bound, C, dtg, scale in (("a", 1.0, 1, 0.0, 1.0), ("b", 2.0, 2, 1.0 / 256, 1.9)):
        bf = bitfield_for(C, bound, scale)
        o, d = rays_for(np.random.RandomState(11), NM, radius=3.2)
        ab = np.array([-bound] * 3 + [bound] * 3, np.float32)
        nears, fars = torch.empty(NM, device=DEV), torch.empty(NM, device=DEV)
        rm.near_far_from_aabb(t(o), t(d), t(ab), NM, 0.2, nears, fars)
        G.update({"m%s_o" % tag: o, "m%s_d" % tag: d, "m%s_bits" % tag: bf, "m%s_nears" % tag: nears.cpu().numpy(), "m%s_fars" % tag: fars.cpu().numpy(),
                  "m%s_cfg" % tag: np.array([bound, C, dtg], np.float64)})
        for perturb in (0, 1):
            M = NM * 1024
            xyzs, dirs, deltas = torch.zeros(M, 3, device=DEV), torch.zeros(M, 3, device=DEV), torch.zeros(M, 2, device=DEV)
            rays = torch.empty(NM, 3, dtype=torch.int32, device=DEV)
            counter = torch.zeros(2, dtype=torch.int32, device=DEV)
            rm.march_rays_train(t(o), t(d), t(bf), bound, dtg, 1024, NM, C, 128, M, nears, fars, xyzs, dirs, deltas, rays, counter, perturb)
            torch.cuda.synchronize()
            cnt, x, dd, dl = by_ray(rays, xyzs, dirs, deltas, NM)
            assert int(counter[0]) == int(cnt.sum()) and int(counter[1]) == NM
            # `dirs` is the ray's direction repeated for each of its samples (raymarching.cu:453-455): checked here, not stored
            assert np.array_equal(dd, np.repeat(d, cnt, axis=0))
            G.update({"m%s%d_counts" % (tag, perturb): cnt, "m%s%d_xyzs" % (tag, perturb): x, "m%s%d_deltas" % (tag, perturb): dl})
            if tag == "a" and perturb == 1:  # ---- F4 compositing on these samples, rays table = prefix sum in ray order
                n = int(cnt.sum())
                offs = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int32)
                table = np.stack([np.arange(NM, dtype=np.int32), offs, cnt], 1)
                r2 = np.random.RandomState(5)
                sig = np.exp(r2.uniform(-2, 7, size=n)).astype(np.float32)
                rgb = r2.rand(n, 3).astype(np.float32)
                ws, dep, img = torch.empty(NM, device=DEV), torch.empty(NM, device=DEV), torch.empty(NM, 3, device=DEV)
                rm.composite_rays_train_forward(t(sig), t(rgb), t(dl), t(table), n, NM, ws, dep, img)
                gws, gimg = r2.randn(NM).astype(np.float32), r2.randn(NM, 3).astype(np.float32)
                gs, gr = torch.zeros(n, device=DEV), torch.zeros(n, 3, device=DEV)
                rm.composite_rays_train_backward(t(gws), t(gimg), t(sig), t(rgb), t(dl), t(table), ws, img, n, NM, gs, gr)
                G.update(cp_sig=sig, cp_rgb=rgb, cp_deltas=dl, cp_rays=table, cp_ws=ws.cpu().numpy(), cp_depth=dep.cpu().numpy(), cp_image=img.cpu().numpy(),
                         cp_gws=gws, cp_gimg=gimg, cp_gsig=gs.cpu().numpy(), cp_grgb=gr.cpu().numpy())

    # ---- F6 spherical harmonics, degrees 1-8, values, dy_dx, input gradient
    NS = 96
    dirs = np.random.RandomState(3).randn(NS, 3).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    G["sh_dirs"] = dirs
    for deg in range(1, 9):
        out = torch.empty(NS, deg * deg, device=DEV)
        dy = torch.empty(NS, 3 * deg * deg, device=DEV)
        sh.sh_encode_forward(t(dirs), out, NS, 3, deg, True, dy)
        g = np.random.RandomState(40 + deg).randn(NS, deg * deg).astype(np.float32)
        gi = torch.zeros(NS, 3, device=DEV)
        sh.sh_encode_backward(t(g), t(dirs), NS, 3, deg, dy, gi)
        G.update({"sh%d_out" % deg: out.cpu().numpy(), "sh%d_dy" % deg: dy.cpu().numpy(), "sh%d_g" % deg: g, "sh%d_gi" % deg: gi.cpu().numpy()})

    # ---- F7 the inference trio: one march of 1024 rays x 4 steps, composite, compact
    bf = bitfield_for(1, 1.0, 1.0)
    o, d = rays_for(np.random.RandomState(21), 1024)
    ab = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = torch.empty(1024, device=DEV), torch.empty(1024, device=DEV)
    rm.near_far_from_aabb(t(o), t(d), t(ab), 1024, 0.2, nears, fars)
    n_alive, n_step = 1024, 4
    alive = torch.arange(1024, dtype=torch.int32, device=DEV)
    rays_t = nears.clone()
    for perturb in (0, 1):
        xyzs, dirs_, deltas = torch.zeros(n_alive * n_step, 3, device=DEV), torch.zeros(n_alive * n_step, 3, device=DEV), torch.zeros(n_alive * n_step, 2, device=DEV)
        rm.march_rays(n_alive, n_step, alive, rays_t, t(o), t(d), 1.0, 0.0, 1024, 1, 128, t(bf), nears, fars, xyzs, dirs_, deltas, perturb)
        G.update({"inf%d_xyzs" % perturb: xyzs.cpu().numpy(), "inf%d_deltas" % perturb: deltas.cpu().numpy()})
    r3 = np.random.RandomState(8)
    sig = np.exp(r3.uniform(-2, 5, size=n_alive * n_step)).astype(np.float32)
    rgb = r3.rand(n_alive * n_step, 3).astype(np.float32)
    ws, dep, img = torch.zeros(1024, device=DEV), torch.zeros(1024, device=DEV), torch.zeros(1024, 3, device=DEV)
    t_after = rays_t.clone()
    alive_after = alive.clone()
    rm.composite_rays(n_alive, n_step, alive_after, t_after, t(sig), t(rgb), deltas, ws, dep, img)
    new_alive, new_t = torch.zeros_like(alive_after), torch.zeros_like(t_after)
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    rm.compact_rays(n_alive, new_alive, alive_after, new_t, t_after, counter)
    k = int(counter[0])
    order = torch.argsort(new_alive[:k])
    G.update(inf_o=o, inf_d=d, inf_bits=bf, inf_nears=nears.cpu().numpy(), inf_fars=fars.cpu().numpy(), inf_sig=sig, inf_rgb=rgb,
             inf_ws=ws.cpu().numpy(), inf_depth=dep.cpu().numpy(), inf_image=img.cpu().numpy(), inf_alive_after=alive_after.cpu().numpy(),
             inf_t_after=t_after.cpu().numpy(), inf_compact_alive=new_alive[:k][order].cpu().numpy(), inf_compact_t=new_t[:k][order].cpu().numpy())

    out = os.path.join(os.environ.get("PVD_GOLDEN_OUT", HERE), "reference_kernels.npz")
    np.savez_compressed(out, **G)
    print("wrote %s: %d arrays, %.1f KB" % (out, len(G), os.path.getsize(out) / 1024))


if __name__ == "__main__":
    main()
