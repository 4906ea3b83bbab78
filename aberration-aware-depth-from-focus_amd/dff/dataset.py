"""RGB-D datasets that feed the focal-stack renderer (reference: dff/dataset.py), read with PIL + torch instead of
cv2 / torchvision / skimage (not installed here; none of them is on the hot path).

Same classes, constructor arguments and item layout as the reference for the file sets whose formats need nothing beyond
PNG/JPEG decoding: `Middlebury` (dff/dataset.py:170-200: `im0.png` + 16-bit `depth.png` in millimetres per scene directory)
and `Matterport3D` (:17-52: `undistorted_color_images/*.jpg` + `render_depth/*.png`, 0.25 mm units).  Items are
`[aif_img [3,H,W] float32 in [0,1], depth [1,H,W] float32 in metres]`, resized to `resize=(H, W)`:
  * the image by the reference's `transforms.Resize(resize, antialias=True)` on a tensor = bilinear interpolation with
    antialiasing, `align_corners=False` (torch.nn.functional.interpolate);
  * Middlebury's depth first by `cv.resize(depth / 1000, (W, H))` = plain bilinear at pixel centres without antialiasing,
    then the (identity-size) Resize - restated with the same interpolate call, `antialias=False`.
`FlyingThings3D` (:55-110: `disp.exr` read by dff/exr.py, `AiF.png`, optionally `fs_num` slices of the pre-rendered focal
stack whose file names are the focus disparities) and `RealWorld` (:207-246: captured stacks, focus distance in the file
name) follow the same rules; their focal-stack images keep OpenCV's B, G, R channel order, as in the reference, which
converts only the all-in-focus image to RGB.
"""
import os
from glob import glob

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset


def _read_rgb01(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.float64) / 255.0          # cv.cvtColor(cv.imread(..), BGR2RGB) / 255.


def _read_raw(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im)                                                    # cv.imread(path, -1): 16-bit PNGs stay 16-bit


def _read_bgr01(path):
    """cv.imread(path).astype(np.float32) / 255.: 8-bit, B G R order, float32."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.float32)[..., ::-1] / np.float32(255.0)


def _cv_resize(a, hw):
    """cv.resize(a, (W, H)) with the default INTER_LINEAR on an HxW[xC] float array: bilinear at pixel centres, no
    antialiasing, computed in the array's own precision."""
    t = torch.from_numpy(np.ascontiguousarray(a))
    t = t[None, None] if t.dim() == 2 else t.permute(2, 0, 1)[None]
    if tuple(t.shape[-2:]) != tuple(hw):
        t = F.interpolate(t, size=tuple(hw), mode="bilinear", align_corners=False, antialias=False)
    t = t[0]
    return (t[0] if a.ndim == 2 else t.permute(1, 2, 0)).numpy()


def _to_tensor(a):
    """transforms.ToTensor on a float32 HxW[xC] array: CHW, no rescaling."""
    t = torch.from_numpy(np.ascontiguousarray(a.astype("float32")))
    return t.unsqueeze(0) if t.dim() == 2 else t.permute(2, 0, 1).contiguous()


def _resize(t, size, antialias):
    if size is None or tuple(t.shape[-2:]) == tuple(size):
        return t
    return F.interpolate(t.unsqueeze(0), size=tuple(size), mode="bilinear", align_corners=False, antialias=antialias).squeeze(0)


def AutoAgument(img, depth):
    """Colour jitter, flips and a random rotation with np.random in the reference's call order (dff/dataset.py:245-285)."""
    from scipy.ndimage import rotate
    if np.random.rand() > 0.5:
        contrast = np.random.rand()
        brightness = np.random.rand()
        img = np.clip((0.5 + contrast * (img - 0.5)) + brightness, 0.0, 1.0)
    if np.random.rand() > 0.5:
        img, depth = np.flip(img, 1), np.flip(depth, 1)
    if np.random.rand() > 0.5:
        img, depth = np.flip(img, 0), np.flip(depth, 0)
    if np.random.rand() > 0.5:
        degree = np.random.randint(0, 180)
        if len(img.shape) == 4:
            img = np.array(img)
            for i in range(img.shape[-1]):
                img[..., i] = rotate(img[..., i], degree, reshape=False)
        else:
            img = rotate(img, degree, reshape=False)
        depth = rotate(depth, degree, reshape=False)
        depth[depth < 0] = 0
    return img, depth


class Middlebury(Dataset):
    """Middlebury 2014 / 2021 scenes: `<dir>/<scene>/im0.png` and `<dir>/<scene>/depth.png` (uint16, mm)."""

    def __init__(self, dataset_dir, resize=None, train=False):
        self.dataset_dir = dataset_dir
        self.scenes = sorted(scene.split("/")[-1] for scene in glob(f"{dataset_dir}/*"))
        self.resize = resize
        self.train = train

    def __len__(self):
        return len(self.scenes)

    def __getitem__(self, index):
        scene = self.scenes[index]
        aif = _read_rgb01(f"{self.dataset_dir}/{scene}/im0.png")
        depth = _read_raw(f"{self.dataset_dir}/{scene}/depth.png") / 1000                 # mm -> m, float64 like the reference
        aif_t = _resize(_to_tensor(aif), self.resize, antialias=True)
        depth_t = _resize(torch.from_numpy(depth)[None], self.resize, antialias=False).float()     # cv.resize on float64, then float32
        return [aif_t, depth_t]


class Matterport3D(Dataset):
    def __init__(self, rgb_path, depth_path, resize=None, train=True):
        self.rgb_path, self.depth_path, self.resize, self.train = rgb_path, depth_path, resize, train
        self.scenes = [scene.split("/")[-1] for scene in glob(f"{rgb_path}/*")]
        self.imgs, self.depths = [], []
        for scene in self.scenes:
            self.imgs += sorted(glob(f"{rgb_path}/{scene}/undistorted_color_images/*.jpg"))
            self.depths += sorted(glob(f"{depth_path}/{scene}/render_depth/*.png"))

    def __len__(self):
        return len(self.imgs)

    def __getitem__(self, idx):
        aif = _read_rgb01(self.imgs[idx])
        depth = _read_raw(self.depths[idx]) / 4000                                          # 0.25 mm units -> m
        if self.train:
            aif, depth = AutoAgument(aif, depth)
        return [_resize(_to_tensor(aif), self.resize, antialias=True), _resize(_to_tensor(depth), self.resize, antialias=True)]


class FlyingThings3D(Dataset):
    """`<dir>/<scene>/disp.exr` (disparity; depth = disp / 20), `AiF.png`, and the pre-rendered focal stack `<focus
    disparity>.png`.  fs_num = 0: [aif_img [3,H,W], depth [1,H,W]]; fs_num > 0: [focal_stack [S,3,H,W] (B G R), depth
    [1,H,W], focal_dists [S]] with the slices drawn by `random.sample` like the reference."""
    DEPTH_FACTOR = 20

    def __init__(self, dataset_dir, resize=None, train=True, fs_num=0):
        self.dataset_dir = dataset_dir
        self.scenes = [scene.split("/")[-1] for scene in glob(f"{dataset_dir}/*")]
        self.resize, self.fs_num, self.train = resize, fs_num, train

    def __len__(self):
        return len(self.scenes)

    def __getitem__(self, index):
        import random
        from .exr import read_exr
        d, scene, hw = self.dataset_dir, self.scenes[index], tuple(self.resize)
        depth = _cv_resize(read_exr(f"{d}/{scene}/disp.exr") / self.DEPTH_FACTOR, hw)
        if self.fs_num > 0:
            full = sorted(glob(f"{d}/{scene}/*.png"))[:-1]                              # the last one is AiF.png
            chosen = random.sample(full, self.fs_num)
            dists = [float(name.split("/")[-1][:-4]) / self.DEPTH_FACTOR for name in chosen]
            stack = np.stack([_cv_resize(_read_bgr01(name), hw) for name in chosen], axis=-1)      # [H,W,3,S]
            if self.train:
                stack, depth = AutoAgument(stack, depth)
            stack = torch.from_numpy(np.transpose(stack, (3, 2, 0, 1)).astype("float32"))
            return [stack, torch.from_numpy(np.ascontiguousarray(depth).astype("float32")).unsqueeze(0),
                    torch.from_numpy(np.stack(dists, axis=-1))]
        aif = _read_rgb01(f"{d}/{scene}/AiF.png")
        if self.train:
            aif, depth = AutoAgument(aif, depth)
        return [_resize(_to_tensor(aif), self.resize, antialias=True), _resize(_to_tensor(depth), self.resize, antialias=True)]


class RealWorld(Dataset):
    """Captured focal stacks: `<dir>/<scene>/{align/*.png, *.JPG, *.png}` named `<x>_dist<mm>_...`; optional Blender depth
    `depth/depth.png` (16 bit: 500 mm + 3000 mm * v / 65535).  Items: [focal_stack [S,3,H,W] (B G R), depth [1,H,W] m (zeros
    without `depth`), focal_dists [S] m]."""

    def __init__(self, dataset_dir, resize=None, depth=False):
        self.dataset_dir = dataset_dir
        self.scenes = sorted(scene.split("/")[-1] for scene in glob(f"{dataset_dir}/*"))
        self.resize, self.depth = resize, depth

    def __len__(self):
        return len(self.scenes)

    def __getitem__(self, index):
        d, scene, hw = self.dataset_dir, self.scenes[index], tuple(self.resize)
        names = sorted(glob(f"{d}/{scene}/align/*.png")) + sorted(glob(f"{d}/{scene}/*.JPG")) + sorted(glob(f"{d}/{scene}/*.png"))
        dists = [float(name.split("/")[-1].split("_")[1][4:]) / 1000 for name in names]
        stack = np.stack([_cv_resize(_read_bgr01(name), hw) for name in names], axis=-1)
        stack = torch.from_numpy(np.transpose(stack, (3, 2, 0, 1)).astype("float32"))
        if self.depth:
            depth = _cv_resize(_read_raw(f"{d}/{scene}/depth/depth.png").astype(np.float64), hw)
            depth = torch.from_numpy(((depth / 65535 * 3000 + 500) / 1000).astype("float32")).unsqueeze(0)
        else:
            depth = torch.zeros_like(stack[0, 0, :, :].unsqueeze(0))
        return [stack, depth, torch.from_numpy(np.stack(dists, axis=-1))]
