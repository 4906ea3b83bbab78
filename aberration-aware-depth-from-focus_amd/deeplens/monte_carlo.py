"""Ray hits -> PSF histogram on MI355X (csrc/trace.hip: psf_splat_kernel).

Reference: deeplens/monte_carlo.py (forward_integral :9-57, assign_points_to_pixels
:60-121).  Only the incoherent, interpolated form used by the PSF path is provided.
"""
import torch

from aadff import _abi
from .basics import EPSILON


def _splat(o, ra, centre, ps, ks):
    """o [spp,N,3], ra [spp,N], centre [N,2] -> un-normalised [N,ks,ks] on o's device."""
    _abi.require_gpu()
    dev = o.device if o.is_cuda else torch.device("cuda", torch.cuda.current_device())
    o_, ra_, c_ = _abi.f32c(o, dev), _abi.f32c(ra, dev), _abi.f32c(centre, dev)
    spp, N = ra_.shape
    raw = torch.empty((N, ks, ks), dtype=torch.float32, device=dev)
    nrm = torch.empty_like(raw)
    with torch.cuda.device(dev):
        _abi.call("aadff_psf_splat", _abi.ptr(o_), _abi.ptr(ra_), _abi.ptr(c_), spp, N, float(ps), ks,
                  _abi.ptr(raw), _abi.ptr(nrm), _abi.stream_ptr(dev))
    return raw.to(o.device)


def forward_integral(ray, ps, ks, pointc_ref=None, interpolate=False):
    """PSF [N,ks,ks] (or [ks,ks]) of ray hits on the sensor; `pointc_ref` [N,2] is the PSF
    centre, default = validity-weighted centroid (reference :27-33)."""
    single = len(ray.o.shape) == 2
    o = ray.o.unsqueeze(1) if single else ray.o
    ra = ray.ra.unsqueeze(1) if single else ray.ra
    if pointc_ref is None:
        pts = -o[..., :2]
        pointc_ref = (pts * ra.unsqueeze(-1)).sum(0) / ra.unsqueeze(-1).sum(0).add(EPSILON)
    centre = pointc_ref.reshape(-1, 2).to(o.device)
    psf = _splat(o, ra, centre, ps, ks)
    return psf[0] if single else psf


def assign_points_to_pixels(points, ks, x_range, y_range, ra, interpolate=True, coherent=False, phase=None, d=None,
                            obliq=None, wvln=0.589):
    """Bilinear splat of already-centred points [spp,2] into [ks,ks] (reference :60-121)."""
    if coherent or not interpolate:
        raise NotImplementedError("only the incoherent, interpolated splat is on the hot path")
    ps = (x_range[1] - x_range[0]) / (ks - 1)
    # the kernel applies flip + centre itself: feed it o = -points with a zero centre
    o = torch.cat((-points, torch.zeros_like(points[:, :1])), dim=-1).unsqueeze(1)
    return _splat(o, ra.reshape(-1, 1), torch.zeros(1, 2, device=points.device), ps, ks)[0]
