"""ORACLE (test infrastructure): image-space PSF application, torch CPU fp32.

Follows deeplens/render_psf.py (uniform PSF :12-28, PSF grid :31-73, per-pixel PSF
:76-107, tiled per-pixel variant without halo :110-127).  Same torch ops in the same
order so results are bit-comparable with the reference.
"""
import numpy as np
import torch
import torch.nn.functional as F


def render_psf(img, psf):
    ks = psf.shape[-1]
    pad = int(ks / 2)
    w = torch.flip(psf, [1, 2]).unsqueeze(1)
    x = F.pad(img, (pad, pad, pad, pad), mode="reflect")
    return F.conv2d(x, w, groups=img.shape[1], padding=0, bias=None)


def patch_bounds(n, grid):
    """Python-float patch bounds int(i/grid*n), render_psf.py:65-66."""
    return [int(i / grid * n) for i in range(grid + 1)]


def render_psf_map(img, psf_map, grid):
    assert len(img.shape) == 4, "Input image should be [B, C, H, W]"
    Cp, Hp, Wp = psf_map.shape
    assert Hp % grid == 0 and Wp % grid == 0, "PSF map size should be divisible by grid"
    ks = int(Hp / grid)
    assert ks % 2 == 1, "PSF kernel size should be odd"
    B, C, H, W = img.shape
    assert C == Cp, "PSF map should have the same channel as image"
    pad = int((ks - 1) / 2)
    x = F.pad(img, (pad, pad, pad, pad), mode="reflect")
    out = torch.zeros_like(img)
    hb, wb = patch_bounds(H, grid), patch_bounds(W, grid)
    for i in range(grid):
        for j in range(grid):
            w = torch.flip(psf_map[:, i * ks:(i + 1) * ks, j * ks:(j + 1) * ks], [1, 2]).unsqueeze(1)
            patch = x[:, :, hb[i]:hb[i + 1] + 2 * pad, wb[j]:wb[j + 1] + 2 * pad]
            out[:, :, hb[i]:hb[i + 1], wb[j]:wb[j + 1]] = F.conv2d(patch, w, groups=C, padding="valid", bias=None)
    return out


def local_psf_render(inp, psf, kernel_size=11):
    if len(inp.shape) < 4:
        inp = inp.unsqueeze(0)
    b, c, h, w = inp.shape
    pad = int((kernel_size - 1) / 2)
    x = F.pad(inp, pad=(pad, pad, pad, pad), mode="replicate")
    kernels = psf.reshape(-1, kernel_size, kernel_size)
    kernels_rgb = torch.stack(c * [kernels], 1)
    unf = F.unfold(x, (kernel_size, kernel_size))
    x1 = unf.view(b, c, -1, h * w)
    x2 = kernels_rgb.view(b, h * w, c, -1).permute(0, 2, 3, 1)
    y = (x1 * x2).sum(2)
    return F.fold(y, (h, w), (1, 1))


def local_psf_render_high_res(inp, psf, patch_size=(320, 480), kernel_size=11):
    B, C, H, W = inp.shape
    out = torch.zeros_like(inp)
    for pi in range(int(np.ceil(H / patch_size[0]))):
        for pj in range(int(np.ceil(W / patch_size[1]))):
            i0, i1 = pi * patch_size[0], min((pi + 1) * patch_size[0], H)
            j0, j1 = pj * patch_size[1], min((pj + 1) * patch_size[1], W)
            out[:, :, i0:i1, j0:j1] = local_psf_render(inp[:, :, i0:i1, j0:j1], psf[:, i0:i1, j0:j1, :, :], kernel_size)
    return out


def render_psf_map_closed_form(img, psf_map, grid):
    """Independent closed form, float64, for cross-checking the above."""
    img = img.double().numpy()
    pm = psf_map.double().numpy()
    B, C, H, W = img.shape
    ks = pm.shape[1] // grid
    p = ks // 2
    x = np.pad(img, ((0, 0), (0, 0), (p, p), (p, p)), mode="reflect")
    hb, wb = patch_bounds(H, grid), patch_bounds(W, grid)
    out = np.zeros_like(img)
    for i in range(grid):
        for j in range(grid):
            k = pm[:, i * ks:(i + 1) * ks, j * ks:(j + 1) * ks]
            for u in range(ks):
                for v in range(ks):
                    out[:, :, hb[i]:hb[i + 1], wb[j]:wb[j + 1]] += (
                        x[:, :, hb[i] + u:hb[i + 1] + u, wb[j] + v:wb[j + 1] + v] * k[None, :, ks - 1 - u, ks - 1 - v, None, None])
    return out
