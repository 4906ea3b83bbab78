"""torch custom ops (`torch.ops.aadff.*`) over the C ABI of include/aadff.h.

`north_star` asks for the HIP kernels "through PyTorch-ROCm custom ops": each op below is a thin
`torch.library.custom_op` wrapper (schema + fake/meta shape function) whose implementation packs pointers and calls
libaadff.so on the current HIP stream.  With them the path is visible to FakeTensor / `torch.compile` tracing and
`torch.library.opcheck`; no autograd formula is registered — the reference never back-propagates through these
functions, so a backward through them raises torch's "not differentiable" error.

The deeplens mirror (deeplens/render_psf.py, deeplens/psfnet.py) calls these ops; the multi-launch planners
(aadff/focal_stack.py, aadff/training.py) keep calling the ABI directly because they pass raw offsets into pinned rings.
"""
import ctypes as C
from typing import List

import torch
from torch.library import custom_op

from . import _abi


def _st(t):
    return _abi.stream_ptr(t.device)


# ---------------------------------------------------------------- image space (deeplens/render_psf.py:12-107)
@custom_op("aadff::render_psf_map", mutates_args=(), device_types="cuda")
def render_psf_map(img: torch.Tensor, psf_map: torch.Tensor, grid: int) -> torch.Tensor:
    B, Cn, H, W = img.shape
    ks = psf_map.shape[1] // grid
    x, p = img.contiguous().float(), psf_map.contiguous().float()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _abi.call("aadff_render_psf_map", _abi.ptr(x), _abi.ptr(p), _abi.ptr(out), B, Cn, H, W, grid, ks, _st(x))
    return out


@render_psf_map.register_fake
def _(img, psf_map, grid):
    return torch.empty_like(img, dtype=torch.float32, memory_format=torch.contiguous_format)


@custom_op("aadff::render_psf_map_stack", mutates_args=(), device_types="cuda")
def render_psf_map_stack(img: torch.Tensor, psf_maps: torch.Tensor, grid: int) -> torch.Tensor:
    B, Cn, H, W = img.shape
    S, ks = psf_maps.shape[0], psf_maps.shape[2] // grid
    x, p = img.contiguous().float(), psf_maps.contiguous().float()
    out = torch.empty((B, Cn, S, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _abi.call("aadff_render_psf_map_stack", _abi.ptr(x), _abi.ptr(p), _abi.ptr(out), B, Cn, S, H, W, grid, ks, _st(x))
    return out


@render_psf_map_stack.register_fake
def _(img, psf_maps, grid):
    B, Cn, H, W = img.shape
    return img.new_empty((B, Cn, psf_maps.shape[0], H, W), dtype=torch.float32)


@custom_op("aadff::render_psf", mutates_args=(), device_types="cuda")
def render_psf(img: torch.Tensor, psf: torch.Tensor) -> torch.Tensor:
    B, Cn, H, W = img.shape
    x, p = img.contiguous().float(), psf.contiguous().float()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _abi.call("aadff_render_psf", _abi.ptr(x), _abi.ptr(p), _abi.ptr(out), B, Cn, H, W, psf.shape[-1], _st(x))
    return out


@render_psf.register_fake
def _(img, psf):
    return torch.empty_like(img, dtype=torch.float32, memory_format=torch.contiguous_format)


@custom_op("aadff::local_psf_render", mutates_args=(), device_types="cuda")
def local_psf_render(img: torch.Tensor, psf: torch.Tensor, ks: int) -> torch.Tensor:
    B, Cn, H, W = img.shape
    x, p = img.contiguous().float(), psf.contiguous().float()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _abi.call("aadff_local_psf_render", _abi.ptr(x), _abi.ptr(p), _abi.ptr(out), B, Cn, H, W, ks, _st(x))
    return out


@local_psf_render.register_fake
def _(img, psf, ks):
    return torch.empty_like(img, dtype=torch.float32, memory_format=torch.contiguous_format)


@custom_op("aadff::thinlens_render", mutates_args=(), device_types="cuda")
def thinlens_render(img: torch.Tensor, depth: torch.Tensor, foc_dist: torch.Tensor, ks: int, foc_len: float, fnum: float,
                    pixel_size: float, d_min: float, d_max: float) -> torch.Tensor:
    N, Cn, H, W = img.shape
    x, d, fd = img.contiguous().float(), depth.contiguous().float().reshape(N, 1, H, W), foc_dist.contiguous().float().reshape(N)
    neg = (d < 0).any().to(torch.int32).reshape(1)              # the reference's whole-tensor sign test, kept on the device
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _abi.call("aadff_thinlens_render", _abi.ptr(x), _abi.ptr(d), _abi.ptr(fd), _abi.ptr(neg), _abi.ptr(out), N, Cn, H, W, ks,
                  C.c_float(foc_len / fnum), C.c_float(foc_len), C.c_float(1.0 / pixel_size), C.c_float(d_min), C.c_float(d_max), _st(x))
    return out


@thinlens_render.register_fake
def _(img, depth, foc_dist, ks, foc_len, fnum, pixel_size, d_min, d_max):
    return torch.empty_like(img, dtype=torch.float32, memory_format=torch.contiguous_format)


# ---------------------------------------------------------------- PSF network (deeplens/psfnet.py:375-441)
def _ints(v):
    return (C.c_int * len(v))(*v)


@custom_op("aadff::psfnet_forward", mutates_args=("flags",), device_types="cuda")
def psfnet_forward(inp: torch.Tensor, wpack: torch.Tensor, bias: torch.Tensor, in_features: List[int], out_features: List[int],
                   flags: torch.Tensor, precision: int = 0) -> torch.Tensor:
    x = inp.contiguous().float().reshape(-1, 4)
    out = torch.empty((x.shape[0], out_features[-1]), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _abi.call("aadff_psfnet_forward", _abi.ptr(x), x.shape[0], _abi.ptr(wpack), _abi.ptr(bias), len(in_features), _ints(in_features),
                  _ints(out_features), 0, _abi.ptr(out), None, None, 0, 0, 0, 0, 0, int(precision), _abi.ptr(flags), _st(x))
    return out


@psfnet_forward.register_fake
def _(inp, wpack, bias, in_features, out_features, flags, precision=0):
    return inp.new_empty((inp.numel() // 4, out_features[-1]), dtype=torch.float32)


@custom_op("aadff::psfnet_render_rgbd", mutates_args=("flags",), device_types="cuda")
def psfnet_render_rgbd(img: torch.Tensor, depth: torch.Tensor, xs: torch.Tensor, ys: torch.Tensor, foc_z: torch.Tensor, d_min: float,
                       inv_range: float, wpack: torch.Tensor, bias: torch.Tensor, in_features: List[int], out_features: List[int],
                       ks: int, flags: torch.Tensor, precision: int = 0) -> torch.Tensor:
    N, Cn, H, W = img.shape
    S = foc_z.numel() // N
    x, d, fz = img.contiguous().float(), depth.contiguous().float(), foc_z.contiguous().float()
    out = torch.empty((N, Cn, S, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _abi.call("aadff_psfnet_render_rgbd", _abi.ptr(d), _abi.ptr(xs.contiguous().float()), _abi.ptr(ys.contiguous().float()), _abi.ptr(fz),
                  C.c_float(d_min), C.c_float(inv_range), N, S, _abi.ptr(wpack), _abi.ptr(bias), len(in_features), _ints(in_features),
                  _ints(out_features), _abi.ptr(x), _abi.ptr(out), Cn, H, W, ks, int(precision), _abi.ptr(flags), _st(x))
    return out


@psfnet_render_rgbd.register_fake
def _(img, depth, xs, ys, foc_z, d_min, inv_range, wpack, bias, in_features, out_features, ks, flags, precision=0):
    N, Cn, H, W = img.shape
    return img.new_empty((N, Cn, foc_z.numel() // N, H, W), dtype=torch.float32)


# ---------------------------------------------------------------- ray trace -> PSFs (deeplens/optics.py:888-1026)
_LC_FIELDS = [f for f, _ in _abi.LensConst._fields_]


def lens_const_to_list(lc):
    return [float(getattr(lc, f)) for f in _LC_FIELDS]


def lens_const_from_list(v):
    lc = _abi.LensConst()
    for f, x in zip(_LC_FIELDS, v):
        setattr(lc, f, int(x) if f == "n_surf" else float(x))
    return lc


@custom_op("aadff::psf_points", mutates_args=("flags",), device_types="cuda")
def psf_points(points: torch.Tensor, surf_main: torch.Tensor, surf_chief: torch.Tensor, lens_const: List[float], states: torch.Tensor,
               u_main: torch.Tensor, u_chief: torch.Tensor, ks: int, centre: bool, map_layout: bool, flags: torch.Tensor) -> torch.Tensor:
    """points [S,N,3] normalised field points; surf_* / states: the packed byte tensors of deeplens.optics
    (aadff_surface_t tables per wavelength, aadff_lens_state_t[S]); lens_const: the fields of aadff_lens_const_t in
    declaration order (n_surf first); u_main [S,L,2,spp], u_chief [S,L,2,spp_chief] raw uniforms (u_chief may be empty
    when centre is False) -> [S,N,L,ks,ks] or, with map_layout, [S,L,g*ks,g*ks]."""
    S, N = points.shape[0], points.shape[1]
    L, spp, spc = u_main.shape[1], u_main.shape[3], (u_chief.shape[3] if u_chief.numel() else 0)
    lc = lens_const_from_list(lens_const)
    g = int(round(N ** 0.5))
    out = torch.empty((S, L, g * ks, g * ks) if map_layout else (S, N, L, ks, ks), dtype=torch.float32, device=points.device)
    pts, um, uc = points.contiguous().float(), u_main.contiguous().float(), u_chief.contiguous().float()
    with torch.cuda.device(pts.device):
        _abi.call("aadff_psf_points", _abi.ptr(pts), S, N, L, _abi.ptr(surf_main), _abi.ptr(surf_chief), lc, _abi.ptr(states),
                  _abi.ptr(um), spp, 2 * L * spp, 2 * spp, _abi.ptr(uc) if spc else None, spc, 2 * L * spc, 2 * spc, ks, int(centre), int(map_layout),
                  _abi.ptr(out), None, _abi.ptr(flags), _st(pts))
    return out


@custom_op("aadff::psf_points_block", mutates_args=("flags",), device_types="cuda")
def psf_points_block(points: torch.Tensor, surf_main: torch.Tensor, surf_chief: torch.Tensor, lens_const: List[float], states: torch.Tensor,
                     u_block: torch.Tensor, n_wave: int, spp: int, spp_chief: int, ks: int, map_layout: bool, flags: torch.Tensor) -> torch.Tensor:
    """psf_points for ONE focus state with the uniforms as the flat block the host generator produced, in the reference's draw order
    (SURVEY.md Appendix B): per wavelength [main theta spp | main r spp | chief theta spp_chief | chief r spp_chief]; spp_chief = 0:
    no chief rays (center=False).  points [N,3] -> [N,L,ks,ks] or, with map_layout, [L,g*ks,g*ks].  No re-layout copy: the kernel
    takes the rows through strides."""
    N, L = points.shape[0], n_wave
    per_l = 2 * spp + 2 * spp_chief
    lc = lens_const_from_list(lens_const)
    g = int(round(N ** 0.5))
    out = torch.empty((L, g * ks, g * ks) if map_layout else (N, L, ks, ks), dtype=torch.float32, device=points.device)
    base = u_block.data_ptr()
    with torch.cuda.device(points.device):
        _abi.call("aadff_psf_points", _abi.ptr(points), 1, N, L, _abi.ptr(surf_main), _abi.ptr(surf_chief), lc, _abi.ptr(states),
                  C.c_void_p(base), spp, L * per_l, per_l, C.c_void_p(base + 8 * spp) if spp_chief else None, spp_chief, L * per_l, per_l, ks,
                  int(spp_chief > 0), int(map_layout), _abi.ptr(out), None, _abi.ptr(flags), _st(points))
    return out


@psf_points_block.register_fake
def _(points, surf_main, surf_chief, lens_const, states, u_block, n_wave, spp, spp_chief, ks, map_layout, flags):
    N = points.shape[0]
    g = int(round(N ** 0.5))
    return points.new_empty((n_wave, g * ks, g * ks) if map_layout else (N, n_wave, ks, ks), dtype=torch.float32)


@psf_points.register_fake
def _(points, surf_main, surf_chief, lens_const, states, u_main, u_chief, ks, centre, map_layout, flags):
    S, N, L = points.shape[0], points.shape[1], u_main.shape[1]
    g = int(round(N ** 0.5))
    return points.new_empty((S, L, g * ks, g * ks) if map_layout else (S, N, L, ks, ks), dtype=torch.float32)
