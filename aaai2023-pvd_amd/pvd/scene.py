"""Deterministic synthetic stand-in for "Synthetic-NeRF chair" (no dataset ships with the
container): an analytic chair-like density/colour field in [-0.6, 0.6]^3,
Blender-style cameras (800x800, fx = fy = 1111.1) on the radius-3.2 sphere the reference's
random-pose generator uses (distill_mutual/utils.py:53-132: pose_spherical(theta, phi, 4) then
nerf_matrix_to_ngp(scale=0.8)), and the reference's ray generation (utils.py:324-404).
"""
import math

import numpy as np
import torch

# (centre, half-size) boxes: seat, back rest, four legs
_CHAIR_BOXES = [
    ((0.0, 0.0, 0.0), (0.30, 0.30, 0.04)),
    ((0.0, -0.27, 0.30), (0.30, 0.04, 0.28)),
    ((-0.25, -0.25, -0.30), (0.04, 0.04, 0.28)),
    ((0.25, -0.25, -0.30), (0.04, 0.04, 0.28)),
    ((-0.25, 0.25, -0.30), (0.04, 0.04, 0.28)),
    ((0.25, 0.25, -0.30), (0.04, 0.04, 0.28)),
]


class ChairScene:
    """Union of boxes, sigma = `sigma_in` inside, smooth view-dependent albedo.

    `thicken` inflates every box (world units): 0.0 / 0.08 (default) / 0.2 give the occupancy sweep
    (about 1 % / 5 % / 15 % occupied cells of a 128^3 grid at bound 1)."""

    def __init__(self, sigma_in=50.0, thicken=0.08, scale=1.0):
        self.sigma_in = float(sigma_in)
        self.thicken = float(thicken)
        self.scale = float(scale)  # > 1: the object outgrows [-1, 1]^3 (scenes with bound > 1: the outer cascades get samples)

    def inside(self, x):
        if self.scale != 1.0:
            x = x / self.scale
        m = torch.zeros(x.shape[:-1], dtype=torch.bool, device=x.device)
        for c, h in _CHAIR_BOXES:
            c = torch.tensor(c, device=x.device, dtype=x.dtype)
            h = torch.tensor(h, device=x.device, dtype=x.dtype) + self.thicken
            m |= ((x - c).abs() <= h).all(dim=-1)
        return m

    def sigma(self, x):
        return self.inside(x).to(x.dtype) * self.sigma_in

    def color(self, x, d):
        base = 0.5 + 0.5 * torch.sin(x * torch.tensor([5.0, 7.0, 9.0], device=x.device, dtype=x.dtype) + 1.0)
        spec = (0.5 + 0.5 * d[..., 2:3]) * 0.2
        return (0.8 * base + spec).clamp(0, 1)

    def density_grid(self, grid_size=128, bound=1.0, cascade=1, device="cpu"):
        """[cascade, H^3] densities at cell centres in the Morton order the marcher indexes with
        (index = morton3D(nx, ny, nz), raymarching.cu:381)."""
        H = grid_size
        ax = torch.arange(H, device=device)
        xx, yy, zz = torch.meshgrid(ax, ax, ax, indexing="ij")
        coords = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1)
        idx = morton3D_torch(coords)
        grid = torch.zeros(cascade, H ** 3, device=device)
        for cas in range(cascade):
            b = min(2 ** cas, bound)
            centres = ((coords.float() + 0.5) / H * 2 - 1) * b
            # a cell is occupied if any of its corners / centre is inside (conservative)
            s = self.sigma(centres)
            half = b / H
            for ox in (-1, 1):
                for oy in (-1, 1):
                    for oz in (-1, 1):
                        off = torch.tensor([ox, oy, oz], device=device, dtype=torch.float32) * half
                        s = torch.maximum(s, self.sigma(centres + off))
            grid[cas, idx] = s
        return grid


def morton3D_torch(coords):
    """10-bit x 3 interleave in torch (host-side helper for building synthetic grids)."""
    def spread(v):
        v = v.long() & 0x3FF
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(coords[..., 0]) | (spread(coords[..., 1]) << 1) | (spread(coords[..., 2]) << 2)


def packbits_torch(grid, thresh):
    """bit i of byte n <=> grid[8n+i] > thresh (host-side restatement of raymarching.cu:283-288)."""
    g = (grid.reshape(-1, 8) > thresh).to(torch.uint8)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=grid.device)
    return (g * w).sum(-1).to(torch.uint8)


# ---------------------------------------------------------------------- cameras
def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=np.float32)


def _rot_y_neg(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1]], dtype=np.float32)


def pose_spherical(theta_deg, phi_deg, radius):
    """Camera on a sphere looking at the origin, NeRF-Blender convention (utils.py:69-98)."""
    t = np.eye(4, dtype=np.float32)
    t[2, 3] = radius
    c2w = _rot_y_neg(theta_deg / 180.0 * math.pi) @ _rot_x(phi_deg / 180.0 * math.pi) @ t
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float32)
    return flip @ c2w


def nerf_matrix_to_ngp(pose, scale=0.8):
    """Axis permutation + translation scale of the reference loader (utils.py:53-66)."""
    p = pose
    return np.array([[p[1, 0], -p[1, 1], -p[1, 2], p[1, 3] * scale],
                     [p[2, 0], -p[2, 1], -p[2, 2], p[2, 3] * scale],
                     [p[0, 0], -p[0, 1], -p[0, 2], p[0, 3] * scale],
                     [0, 0, 0, 1]], dtype=np.float32)


def synthetic_poses(rng, scale=0.8):
    """One epoch of random poses with the reference's elevation schedule (utils.py:106-132):
    1 + sum_{a<80} ((90-a)//15 + 1) = 312 cameras, theta ~ U(-180,180), phi ~ U(-a, min(5-a, 0)), r = 4."""
    def one(ph):
        theta = -180 + rng.rand() * 360
        lo, hi = -ph, (5 - ph if (5 - ph) <= 0 else 0)
        phi = lo + rng.rand() * (hi - lo)
        return nerf_matrix_to_ngp(pose_spherical(theta, phi, 4.0), scale)
    poses = [one(8)]
    for a in range(80):
        poses += [one(a) for _ in range((90 - a) // 15 + 1)]
    return np.stack(poses)


def tank_poses(rng, scale=0.8):
    """Tanks&Temples-style epoch (utils.py:136-150, --data_type tank): elevations 5..19 only, radius ~ U(3, 4) drawn after
    the two angles of a camera; the first camera is the synthetic generator's (phi from 8, r = 4).  88 - 1 = 87 cameras."""
    def one(ph, rand_radius):
        theta = -180 + rng.rand() * 360
        lo, hi = -ph, (5 - ph if (5 - ph) <= 0 else 0)
        phi = lo + rng.rand() * (hi - lo)
        radius = rng.uniform(3, 4) if rand_radius else 4.0
        return nerf_matrix_to_ngp(pose_spherical(theta, phi, radius), scale)
    poses = [one(8, False)]
    for a in range(5, 20):
        poses += [one(a, True) for _ in range((90 - a) // 15 + 1)]
    return np.stack(poses)


def llff_poses(rng, original_poses, gen_num=30):
    """LLFF-style epoch (utils.py:152-188, --data_type llff): `gen_num` camera centres uniform in the bounding box of the training
    cameras' translations (x, then y, then z drawn as three vectors), every camera looking at the origin with up = -y
    (right = f x up, up = right x f, columns [right, up, forward]) and the sign of element [0, 0] flipped afterwards."""
    t = np.asarray(original_poses)[:, :3, 3]
    hi, lo = t.max(axis=0) + 1e-6, t.min(axis=0) - 1e-6
    xs = rng.uniform(low=lo[0], high=hi[0], size=gen_num)
    ys = rng.uniform(low=lo[1], high=hi[1], size=gen_num)
    zs = rng.uniform(low=lo[2], high=hi[2], size=gen_num)
    centers = torch.from_numpy(np.stack([xs, ys, zs], axis=1).astype(np.float32))

    def normalize(v):
        return v / (torch.norm(v, dim=-1, keepdim=True) + 1e-10)

    fwd = -normalize(centers)
    up = torch.tensor([0.0, -1.0, 0.0]).unsqueeze(0).repeat(gen_num, 1)
    right = normalize(torch.cross(fwd, up, dim=-1))
    up = normalize(torch.cross(right, fwd, dim=-1))
    poses = torch.eye(4, dtype=torch.float32).unsqueeze(0).repeat(gen_num, 1, 1)
    poses[:, :3, :3] = torch.stack((right, up, fwd), dim=-1)
    poses[:, :3, 3] = centers
    poses[:, 0, 0] = -poses[:, 0, 0]
    return poses.numpy()


def forward_facing_train_poses(rng, n=20):
    """Stand-in for an LLFF capture's training cameras (there is no dataset offline): n cameras on a small patch in front of the
    scene, as get_rand_poses' llff branch only uses their translations' bounding box."""
    poses = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    poses[:, :3, 3] = rng.uniform([-0.9, -0.5, 2.4], [0.9, 0.5, 3.0], size=(n, 3)).astype(np.float32)
    return poses


def rand_poses(data_type, rng, original_poses=None, scale=0.8):
    """One epoch of random distillation cameras for --data_type synthetic | llff | tank (get_rand_poses, utils.py:100-197)."""
    if data_type == "synthetic":
        return synthetic_poses(rng, scale)
    if data_type == "tank":
        return tank_poses(rng, scale)
    if data_type == "llff":
        assert original_poses is not None, "the llff generator samples inside the training cameras' bounding box"
        return llff_poses(rng, original_poses)
    raise ValueError("illegal data_type %r" % (data_type,))


BLENDER_INTRINSICS = (1111.1, 1111.1, 400.0, 400.0)  # fx, fy, cx, cy for 800x800, camera_angle_x ~ 0.6911


@torch.no_grad()
def sample_pixels_by_error(error_map, N, H, W, generator=None):
    """reference: the `error_map` branch of get_rays, distill_mutual/utils.py:357-381 (--error_map, main_distill_mutual.py:155).
    error_map [B, 128*128]: per-image sampling weights on a fixed 128 x 128 grid.  N cells drawn without replacement by weight, each
    mapped to a pixel of the H x W image with a uniform jitter inside the cell.  Returns (inds [B,N], inds_coarse [B,N]); the draws
    are the reference's, in its order (multinomial, rand for x, rand for y), so a seeded generator reproduces its choice."""
    device = error_map.device
    B = error_map.shape[0]
    inds_coarse = torch.multinomial(error_map, N, replacement=False, generator=generator)  # [B, N] in [0, 128*128)
    inds_x, inds_y = torch.div(inds_coarse, 128, rounding_mode="floor"), inds_coarse % 128
    sx, sy = H / 128, W / 128
    inds_x = (inds_x * sx + torch.rand(B, N, device=device, generator=generator) * sx).long().clamp(max=H - 1)
    inds_y = (inds_y * sy + torch.rand(B, N, device=device, generator=generator) * sy).long().clamp(max=W - 1)
    return inds_x * W + inds_y, inds_coarse


def update_error_map(error_map, inds_coarse, error):
    """reference: train_step's EMA of the per-ray error into the sampled cells, utils.py:1120-1129 (loss_type L2 only):
    error_map[b, inds_coarse] = 0.1 * old + 0.9 * error, in place.  error [B,N] in [0, 1]."""
    ema = 0.1 * error_map.gather(1, inds_coarse) + 0.9 * error.detach().to(error_map.device)
    error_map.scatter_(1, inds_coarse, ema)
    return error_map


def get_rays(poses, intrinsics, H, W, N=-1, generator=None, inds=None, error_map=None):
    """reference: get_rays, distill_mutual/utils.py:324-404.
    poses [B,4,4] cam2world -> rays_o, rays_d [B,N,3], inds [B,N] (+ inds_coarse [B,N] when sampling by error_map [B, 128*128])."""
    device = poses.device
    B = poses.shape[0]
    fx, fy, cx, cy = intrinsics
    inds_coarse = None
    if N > 0:
        N = min(N, H * W)
        if inds is None and error_map is not None:
            inds, inds_coarse = sample_pixels_by_error(error_map.to(device), N, H, W, generator)
        elif inds is None:
            inds = torch.randint(0, H * W, size=[N], device=device, generator=generator)  # may duplicate
        inds = inds.expand([B, N])
    else:
        inds = torch.arange(H * W, device=device).expand([B, H * W])
    # pixel (i, j) of flat index k = j * W + i, sampled at the pixel centre
    i = (inds % W).float() + 0.5
    j = torch.div(inds, W, rounding_mode="floor").float() + 0.5
    zs = torch.ones_like(i)
    dirs = torch.stack(((i - cx) / fx * zs, (j - cy) / fy * zs, zs), dim=-1)
    dirs = dirs / torch.norm(dirs, dim=-1, keepdim=True)
    rays_d = dirs @ poses[:, :3, :3].transpose(-1, -2)
    rays_o = poses[..., :3, 3][..., None, :].expand_as(rays_d)
    out = {"rays_o": rays_o, "rays_d": rays_d, "inds": inds}
    if inds_coarse is not None:
        out["inds_coarse"] = inds_coarse  # needed when the error map is updated
    return out
