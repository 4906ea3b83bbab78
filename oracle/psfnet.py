"""ORACLE (test infrastructure): PSFNet MLP, per-pixel-PSF render (mode M2), thin-lens
baseline and focal-stack assembly (modes M1 and M2), torch CPU fp32.

Follows deeplens/psfnet_arch.py:24-47 (MLP), deeplens/psfnet.py:375-450 (pred/render/
depth2z), deeplens/psfnet.py:489-570 (ThinLens), 2_aber_aware_dff_aif.py:104-114 and
dff/utils.py:4-50 (stack assembly, `linear` focus rule).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .conv import local_psf_render, render_psf_map

DMIN, DMAX = 200, 20000          # deeplens/psfnet.py:11-12


def mlp_forward(sd, x):
    """psfnet_arch.py:24-47: 4->64->256->(8x)256->ks^2, ReLU, Sigmoid, L1-normalise."""
    n_lin = len([k for k in sd if k.endswith(".weight")])
    for i in range(n_lin):
        x = F.linear(x, sd[f"net.{2 * i}.weight"], sd[f"net.{2 * i}.bias"])
        x = torch.relu(x) if i < n_lin - 1 else torch.sigmoid(x)
    return F.normalize(x, p=1, dim=-1)


def depth2z(depth, d_min=-DMIN, d_max=-DMAX):                      # psfnet.py:447-450
    return torch.clamp((depth - d_min) / (d_max - d_min), min=0, max=1)


def psfnet_render(sd, img, depth, foc_dist, ks=11):
    """psfnet.py:393-441, 4-D branch (img [N,C,H,W], depth [N,1,H,W] mm<0, foc_dist [N])
    and 3-D branch (img [C,H,W], depth [H,W], scalar foc_dist)."""
    if len(img.shape) == 3:
        H, W = depth.shape
        z = depth2z(depth)
        x, y = torch.meshgrid(torch.linspace(-1, 1, W), torch.linspace(1, -1, H), indexing="xy")
        foc_z = depth2z(torch.full_like(depth, foc_dist))
        o = torch.stack((x, y, z, foc_z), -1)
    else:
        N, C, H, W = img.shape
        z = depth2z(depth).squeeze(1)
        x, y = torch.meshgrid(torch.linspace(-1, 1, W), torch.linspace(1, -1, H), indexing="xy")
        x, y = x.unsqueeze(0).repeat(N, 1, 1), y.unsqueeze(0).repeat(N, 1, 1)
        foc_z = depth2z(foc_dist.unsqueeze(-1).unsqueeze(-1).repeat(1, H, W))
        o = torch.stack((x, y, z, foc_z), -1).float()
    psf = mlp_forward(sd, o)
    psf = psf.reshape(*psf.shape[:-1], ks, ks)
    return local_psf_render(img, psf, ks)


def thinlens_coc(depth, foc_dist, foc_len, fnum, ps):              # psfnet.py:503-512
    if (depth < 0).any():
        depth, foc_dist = -depth, -foc_dist
    depth = torch.clamp(depth, DMIN, DMAX)
    coc = foc_len / fnum * torch.abs(depth - foc_dist) / depth * foc_len / (foc_dist - foc_len)
    return torch.clamp(coc / ps, min=0.1)


def thinlens_render(img, depth, foc_dist, foc_len, fnum, ks, sensor_size, sensor_res):
    """psfnet.py:549-570 (4-D branch): Gaussian PSF with a hard disc mask, then a3."""
    N, C, H, W = img.shape
    ps = sensor_size[0] / sensor_res[0]
    fd = foc_dist.unsqueeze(-1).unsqueeze(-1).unsqueeze(-1).repeat(1, 1, H, W)
    x, y = torch.meshgrid(torch.linspace(-ks / 2 + 1 / 2, ks / 2 - 1 / 2, ks),
                          torch.linspace(ks / 2 - 1 / 2, -ks / 2 + 1 / 2, ks), indexing="xy")
    coc = thinlens_coc(depth, fd, foc_len, fnum, ps)
    rad = coc.squeeze(1).unsqueeze(-1).unsqueeze(-1).repeat(1, 1, 1, ks, ks) / 2
    psf = torch.exp(-(x ** 2 + y ** 2) / 2 / rad ** 2) / (2 * np.pi * rad ** 2)
    psf = psf * (x ** 2 + y ** 2 < rad ** 2)
    psf = psf / psf.sum((-1, -2)).unsqueeze(-1).unsqueeze(-1)
    return local_psf_render(img, psf, ks)


def select_focus_dist_linear(depth, num):
    """dff/utils.py:4-50, mode='linear': depth [B,1,H,W] (>0 valid) -> [B,num] sorted."""
    assert num > 3, "Focal stack size is too small"
    B = depth.shape[0]
    dmax = torch.amax(depth, dim=(1, 2, 3))
    dmin = torch.zeros_like(dmax)
    for i in range(B):
        d0 = depth[i]
        dmin[i] = torch.min(d0[d0 > 0])
    f = torch.stack([dmin + i * (dmax - dmin) / (num - 1) for i in range(num)], dim=1)
    return torch.sort(f, dim=-1)[0]


def focal_stack_m2(sd, img, depth_m, n_stack, ks=11):
    """2_aber_aware_dff_aif.py:104-114: depth in metres (>0) -> [B,C,S,H,W]."""
    fds = select_focus_dist_linear(depth_m, n_stack)
    sl = [psfnet_render(sd, img, -depth_m * 1e3, -fds[:, i] * 1e3, ks) for i in range(n_stack)]
    return torch.stack(sl, dim=2)


def focal_stack_m1(lens, img, depth_plane_mm, focus_mm, grid=11, ks=11, spp=2048):
    """Mode M1: per slice refocus(f) -> psf_map(depth plane) ->
    render_psf_map; RNG order per slice = refocus draws then psf_map draws
    (SURVEY.md Appendix B).  Returns ([B,C,S,H,W], [S,3,g*ks,g*ks])."""
    sl, maps = [], []
    for f in focus_mm:
        lens.refocus(float(f))
        pm = lens.psf_map(depth=depth_plane_mm, grid=grid, ks=ks, spp=spp)
        maps.append(pm)
        sl.append(render_psf_map(img, pm, grid))
    return torch.stack(sl, dim=2), torch.stack(maps)


def depth_layers(depth_mm, layers):
    """Oracle twin of aadff.focal_stack.depth_layers (the layered mode is a build extension: SURVEY.md §8d)."""
    d = -depth_mm
    valid = d > 0
    dmin = torch.where(valid, d, torch.full_like(d, float("inf"))).amin()
    dmax = d.amax()
    width = torch.clamp((dmax - dmin) / layers, min=1e-6)
    idx = torch.clamp(((d - dmin) / width).floor().long(), 0, layers - 1)
    idx = torch.where(valid, idx, torch.full_like(idx, layers - 1))
    centres = -(dmin + (torch.arange(layers, dtype=torch.float32) + 0.5) * width)
    return idx, centres


def focal_stack_m1_layered(lens, img, depth_mm, focus_mm, layers=4, grid=11, ks=11, spp=2048):
    """M1-layered with reference primitives only: per focus distance refocus, then per layer psf_map and
    render_psf_map, composited by the per-pixel layer index.  Same host-RNG order as the HIP renderer."""
    idx, centres = depth_layers(depth_mm, layers)
    slices = []
    for f in focus_mm:
        lens.refocus(float(f))
        acc = torch.zeros_like(img)
        for l in range(layers):
            pm = lens.psf_map(depth=float(centres[l]), grid=grid, ks=ks, spp=spp)
            acc = acc + (idx == l).to(img.dtype) * render_psf_map(img, pm, grid)
        slices.append(acc)
    return torch.stack(slices, dim=2)


FOC_D_ARR = np.array([-500, -600, -700, -800, -900, -1000, -1250, -1500, -1750, -2000,
                      -2500, -3000, -4000, -5000, -6000, -8000, -10000, -12000, -15000, -20000])    # psfnet.py:34-38


def training_data(lens, bs, spp, ks=11, d_min=-DMIN, d_max=-DMAX):
    """psfnet.py:135-170 get_training_data: one focus distance (np.random.choice over the 20-entry table), refocus,
    x, y ~ U(-1,1), z ~ clamped Gaussian around foc_z, ray-traced PSFs at one wavelength.  RNG order: np choice,
    refocus draws, rand x, rand y, randn z, psf draws.  Returns (inp [bs,4], psf [bs,ks*ks])."""
    foc_z_arr = (FOC_D_ARR - d_min) / (d_max - d_min)
    foc_z = np.random.choice(foc_z_arr)
    lens.refocus(foc_z * (d_max - d_min) + d_min)
    x = (torch.rand(bs) - 0.5) * 2
    y = (torch.rand(bs) - 0.5) * 2
    zg = torch.clamp(torch.randn(bs), min=-3, max=3)
    z = torch.zeros_like(zg)
    z[zg > 0] = (1 - foc_z) * zg[zg > 0] / 3 + foc_z
    z[zg < 0] = foc_z * zg[zg < 0] / 3 + foc_z
    inp = torch.stack((x, y, z, torch.full_like(x, foc_z)), dim=-1)
    points = torch.stack((x, y, z * (d_max - d_min) + d_min), dim=-1)
    psf = lens.psf(points=points, ks=ks, spp=spp)
    return inp, psf.view(bs, -1)
