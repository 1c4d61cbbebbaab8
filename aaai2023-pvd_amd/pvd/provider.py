"""Blender-format scene reader: the data side of the reference's ``NeRFDataset`` (distill_mutual/provider.py:123-326;
the reference supports only ``mode == "blender"`` -- LLFF and Tanks&Temples scenes are used through converted
``transforms_*.json`` files).  numpy + PIL only (cv2 is not in this image); no dataset ships with the container, the
tests generate a tiny scene on the fly.

    scene = BlenderScene(root, "train", scale=0.8, device="cuda")
    batch = scene.batch(index, num_rays=4096)                # rays_o, rays_d [1,N,3], images [1,N,3|4], inds
    gt, bg = training_target(batch["images"])                 # random background where the image has alpha (utils.py:987-995)
"""
import glob
import json
import os

import numpy as np
import torch

from .scene import get_rays, nerf_matrix_to_ngp


def _read_image(path):
    """[H, W, 3|4] uint8 in RGB(A) order (the reference reads BGR(A) with cv2 and converts, provider.py:205-213)."""
    from PIL import Image
    im = Image.open(path)
    if im.mode not in ("RGB", "RGBA"):
        im = im.convert("RGBA" if "A" in im.getbands() else "RGB")
    return im


class BlenderScene:
    def __init__(self, root_path, split="train", downscale=1, scale=0.8, device="cpu", num_rays=4096, preload=True, fp16=False, error_map=False):
        """split: train / val / test, or `trainval` (both files), or `all` (every *.json in the directory)."""
        self.root_path, self.split, self.downscale, self.scale = root_path, split, downscale, scale
        self.device = torch.device(device)
        self.training = split in ("train", "all", "trainval")
        self.num_rays = num_rays if self.training else -1
        transform = self._read_transforms(root_path, split)
        H = int(transform["h"]) // downscale if "h" in transform else None
        W = int(transform["w"]) // downscale if "w" in transform else None

        poses, images = [], []
        for f in transform["frames"]:
            path = os.path.join(root_path, f["file_path"])
            if path[-4:].lower() not in (".png", ".jpg"):
                path += ".png"  # Blender scenes name frames without the extension (provider.py:194-199)
            if not os.path.exists(path):
                continue  # silently skipped, as in the reference (:200-201)
            im = _read_image(path)
            if H is None or W is None:
                H, W = im.size[1] // downscale, im.size[0] // downscale
            if im.size != (W, H):
                # area average of every channel on its own, the counterpart of cv2.INTER_AREA (:215-218); PIL would
                # premultiply the colours of an RGBA image by alpha if it were resized as a whole
                from PIL import Image
                im = Image.merge(im.mode, [band.resize((W, H), Image.BOX) for band in im.split()])
            images.append(np.asarray(im, dtype=np.float32) / 255.0)
            poses.append(nerf_matrix_to_ngp(np.array(f["transform_matrix"], dtype=np.float32), scale=scale))
        if not poses:
            raise FileNotFoundError("no frame of %s exists under %s" % (split, root_path))
        self.H, self.W = H, W
        self.poses = torch.from_numpy(np.stack(poses))             # [N, 4, 4]
        self.images = torch.from_numpy(np.stack(images))           # [N, H, W, 3|4]
        # --error_map (provider.py:232-237): per-image sampling weights on a fixed 128 x 128 grid, all ones to begin with
        self.error_map = torch.ones([self.images.shape[0], 128 * 128], dtype=torch.float) if (self.training and error_map) else None
        self.radius = self.poses[:, :3, 3].norm(dim=-1).mean(0).item()

        # intrinsics (provider.py:244-276)
        if "fl_x" in transform or "fl_y" in transform:
            fl_x = (transform["fl_x"] if "fl_x" in transform else transform["fl_y"]) / downscale
            fl_y = (transform["fl_y"] if "fl_y" in transform else transform["fl_x"]) / downscale
        elif "camera_angle_x" in transform or "camera_angle_y" in transform:
            fl_x = W / (2 * np.tan(transform["camera_angle_x"] / 2)) if "camera_angle_x" in transform else None
            fl_y = H / (2 * np.tan(transform["camera_angle_y"] / 2)) if "camera_angle_y" in transform else None
            fl_x = fl_y if fl_x is None else fl_x
            fl_y = fl_x if fl_y is None else fl_y
        else:
            raise RuntimeError("Failed to load focal length, please check the transforms.json!")
        # the reference defaults cx to H/2 and cy to W/2 (:273-274; the same for its square images) -- kept
        cx = (transform["cx"] / downscale) if "cx" in transform else (H / 2)
        cy = (transform["cy"] / downscale) if "cy" in transform else (W / 2)
        self.intrinsics = np.array([fl_x, fl_y, cx, cy])

        if preload:
            self.poses = self.poses.to(self.device)
            self.images = self.images.to(torch.half if fp16 else torch.float).to(self.device)
            if self.error_map is not None:
                self.error_map = self.error_map.to(self.device)

    @staticmethod
    def _read_transforms(root, split):
        def load(name):
            with open(os.path.join(root, name), "r") as f:
                return json.load(f)
        if split == "all":
            transform = None
            for p in sorted(glob.glob(os.path.join(root, "*.json"))):
                t = load(os.path.basename(p))
                if transform is None:
                    transform = t
                else:
                    transform["frames"].extend(t["frames"])
            if transform is None:
                raise FileNotFoundError("no transforms json under %s" % root)
            return transform
        if split == "trainval":
            transform = load("transforms_train.json")
            transform["frames"].extend(load("transforms_val.json")["frames"])
            return transform
        return load("transforms_%s.json" % split)

    def __len__(self):
        return self.poses.shape[0]

    def batch(self, index, num_rays=None, generator=None):
        """reference: collate, provider.py:278-308.  index: list of frame ids (length 1 in the reference's loader)."""
        idx = torch.as_tensor(index, dtype=torch.long, device=self.poses.device)
        poses = self.poses[idx].to(self.device)
        n = self.num_rays if num_rays is None else num_rays
        emap = None if self.error_map is None else self.error_map[idx.to(self.error_map.device)]  # provider.py:289
        rays = get_rays(poses, tuple(float(v) for v in self.intrinsics), self.H, self.W, n, generator=generator, error_map=emap)
        out = {"H": self.H, "W": self.W, "rays_o": rays["rays_o"], "rays_d": rays["rays_d"], "inds": rays["inds"]}
        if emap is not None:  # what the trainer needs to update the map (provider.py:309-312)
            out["index"], out["inds_coarse"] = index, rays["inds_coarse"]
        images = self.images[idx].to(self.device)
        if self.training:
            B, C = images.shape[0], images.shape[-1]
            images = torch.gather(images.view(B, -1, C), 1, torch.stack(C * [rays["inds"]], -1))  # [B, N, 3|4]
        out["images"] = images
        return out


    def update_error(self, batch, error):
        """reference: the error-map update at the end of train_step (distill_mutual/utils.py:1120-1129, --error_map): the per-ray error
        [B, N] of the batch `self.batch()` returned goes into the frames' maps as an EMA at the cells the rays were drawn from,
        and the updated rows are PUT BACK (indexing with a list copies)."""
        if self.error_map is None or "inds_coarse" not in batch:
            return
        from .scene import update_error_map
        idx = torch.as_tensor(batch["index"], dtype=torch.long, device=self.error_map.device)
        rows = update_error_map(self.error_map[idx], batch["inds_coarse"].to(self.error_map.device), error.reshape(len(idx), -1))
        self.error_map[idx] = rows


def training_target(images, generator=None, bg_radius=-1):
    """Ground-truth pixels and the background they were blended over (reference train_step, utils.py:980-1001):
    RGB images train against a white background; RGBA images against a random colour per ray, blended by alpha."""
    C = images.shape[-1]
    if C == 3:
        return images, 1
    if bg_radius > 0:  # a background model supplies the colour; the target is still blended by alpha, over white
        # (bg_color = 1 and `gt = rgb * a + bg * (1 - a)` whenever C == 4: just_train_tea/utils.py:778-788, distill utils.py:1383)
        return images[..., :3] * images[..., 3:] + (1 - images[..., 3:]), 1
    bg = torch.rand(images.shape[:-1] + (3,), dtype=images.dtype, device=images.device, generator=generator)
    gt = images[..., :3] * images[..., 3:] + bg * (1 - images[..., 3:])
    return gt, bg
