"""Image-space PSF application on MI355X (HIP kernels in csrc/conv.hip).

Each function validates like the reference, then calls its torch custom op (torch.ops.aadff.*, aadff/ops.py), which
packs pointers for the C ABI.  Same four functions, argument order and assertion messages as the reference's
deeplens/render_psf.py (:12 render_psf, :31 render_psf_map, :76 local_psf_render, :110
local_psf_render_high_res).  Inputs on any device are made contiguous fp32 on the GPU,
outputs come back on the input's device.  Forward only: the reference never
back-propagates through these, so tensors that require grad are refused.
"""
import numpy as np
import torch

from aadff import _abi
from aadff import ops as _ops      # noqa: F401  (registers torch.ops.aadff.*)


def _prep(t, what):
    if t.requires_grad and torch.is_grad_enabled():
        raise RuntimeError(f"{what}: forward-only HIP op, call it under torch.no_grad() or detach the input")
    _abi.require_gpu()
    dev = t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device())
    return dev


def render_psf(img, psf):
    """One PSF [C,ks,ks] for the whole image [B,C,H,W]: flip + reflect pad + depthwise conv."""
    dev = _prep(img, "render_psf")
    C_, ks, ks2 = psf.shape
    assert len(img.shape) == 4, "Input image should be [B, C, H, W]"
    B, C, H, W = img.shape
    assert C == C_, "PSF map should have the same channel as image"
    assert ks == ks2 and ks % 2 == 1, "PSF kernel size should be odd"
    if img.numel() == 0:
        return torch.empty_like(img, dtype=torch.float32)
    return torch.ops.aadff.render_psf(_abi.f32c(img, dev), _abi.f32c(psf, dev)).to(img.device)


def render_psf_map(img, psf_map, grid):
    """Different PSF per image patch: img [B,3,H,W], psf_map [3,grid*ks,grid*ks]."""
    if torch.is_tensor(img):
        assert len(img.shape) == 4, "Input image should be [B, C, H, W]"
    else:
        img = torch.tensor((img / 255.).astype(np.float32)).permute(2, 0, 1).unsqueeze(0)
    Cpsf, Hpsf, Wpsf = psf_map.shape
    assert Hpsf % grid == 0 and Wpsf % grid == 0, "PSF map size should be divisible by grid"
    ks = int(Hpsf / grid)
    assert ks % 2 == 1, "PSF kernel size should be odd"
    B, C, H, W = img.shape
    assert C == Cpsf, "PSF map should have the same channel as image"
    if img.numel() == 0:                      # empty batch: the reference's conv2d loop returns an empty tensor too
        return torch.empty_like(img, dtype=torch.float32)
    dev = _prep(img, "render_psf_map")
    x, p = _abi.f32c(img, dev), _abi.f32c(psf_map, dev)
    if torch.compiler.is_compiling():
        return torch.ops.aadff.render_psf_map(x, p, grid).to(img.device)
    # eager calls skip the custom-op dispatcher (~35 us of Python per call, more than twice the kernel): the same ABI entry directly
    out = torch.empty_like(x)
    with _abi.on_device(dev):
        _abi.call("aadff_render_psf_map", _abi.ptr(x), _abi.ptr(p), _abi.ptr(out), B, C, H, W, grid, ks, _abi.stream_ptr(dev))
    return out.to(img.device)


def render_psf_map_stack(img, psf_maps, grid):
    """Stack-fused form (new): img [B,C,H,W], psf_maps [S,C,grid*ks,grid*ks] -> [B,C,S,H,W],
    equal to torch.stack([render_psf_map(img, m, grid) for m in psf_maps], dim=2) with the
    image tile staged once for all S slices."""
    assert len(img.shape) == 4, "Input image should be [B, C, H, W]"
    S, Cpsf, Hpsf, Wpsf = psf_maps.shape
    assert Hpsf % grid == 0 and Wpsf % grid == 0, "PSF map size should be divisible by grid"
    ks = int(Hpsf / grid)
    assert ks % 2 == 1, "PSF kernel size should be odd"
    B, C, H, W = img.shape
    assert C == Cpsf, "PSF map should have the same channel as image"
    if img.numel() == 0 or S == 0:
        return torch.empty((B, C, S, H, W), dtype=torch.float32, device=img.device)
    dev = _prep(img, "render_psf_map_stack")
    return torch.ops.aadff.render_psf_map_stack(_abi.f32c(img, dev), _abi.f32c(psf_maps, dev), grid).to(img.device)


def local_psf_render(input, psf, kernel_size=11):
    """Per-pixel PSF [B,H,W,ks,ks] (same for every channel), replicate padding, no flip."""
    if len(input.shape) < 4:
        input = input.unsqueeze(0)
    b, c, h, w = input.shape
    if input.numel() == 0:
        return torch.empty_like(input, dtype=torch.float32)
    dev = _prep(input, "local_psf_render")
    x = _abi.f32c(input, dev)
    p = _abi.f32c(psf, dev).reshape(-1, h, w, kernel_size, kernel_size)
    assert p.shape[0] == b, "psf should be [B, H, W, ks, ks]"
    return torch.ops.aadff.local_psf_render(x, p, kernel_size).to(input.device)


def local_psf_render_high_res(input, psf, patch_size=[320, 480], kernel_size=11):
    """Tiled variant WITHOUT halo: every tile replicate-pads itself, so seams appear at
    tile borders exactly as in the reference (render_psf.py:110-127)."""
    B, C, H, W = input.shape
    out = torch.zeros_like(input)
    for pi in range(int(np.ceil(H / patch_size[0]))):
        for pj in range(int(np.ceil(W / patch_size[1]))):
            i0, i1 = pi * patch_size[0], min((pi + 1) * patch_size[0], H)
            j0, j1 = pj * patch_size[1], min((pj + 1) * patch_size[1], W)
            out[:, :, i0:i1, j0:j1] = local_psf_render(input[:, :, i0:i1, j0:j1], psf[:, i0:i1, j0:j1, :, :],
                                                      kernel_size=kernel_size)
    return out
