"""Lens surfaces for the MI355X build.

`Aspheric` keeps the reference's constructor and attributes (deeplens/surfaces.py:281-331:
r float, d/c/k one-element tensors, ai list, mat1/mat2) but owns no arithmetic: it packs
itself into the `aadff_surface_t` record of include/aadff.h and the Newton intersection /
Snell refraction of deeplens/surfaces.py:391-830 run in csrc/trace.hip.
"""
import ctypes as C
import math

import numpy as np
import torch

from aadff import _abi
from .basics import DEVICE, DeepObj, Material, Ray

f32 = np.float32


class Surface(DeepObj):
    def __init__(self, r, d, mat1, mat2, is_square=False, device=DEVICE):
        self.d = d.to(device) if torch.is_tensor(d) else torch.tensor([d]).float().to(device)
        self.r = float(r)
        if is_square:
            raise NotImplementedError("square apertures are outside the focal-stack hot path")
        self.is_square = False
        self.mat1, self.mat2 = Material(mat1), Material(mat2)
        self.device = device
        self.NEWTONS_MAXITER = 10
        self.NEWTONS_TOLERANCE_TIGHT = 10e-6
        self.NEWTONS_TOLERANCE_LOOSE = 50e-6
        self.NEWTONS_STEP_BOUND = 5

    def surface_sample(self, N=1000):
        """Uniform points on the aperture disc, host RNG (reference: surfaces.py:188-199)."""
        theta = torch.rand(N) * 2 * np.pi
        r = torch.sqrt(torch.rand(N) * self.r ** 2)
        x2, y2 = r * torch.cos(theta), r * torch.sin(theta)
        return torch.stack((x2, y2, torch.full_like(x2, self.d.item())), 1).to(self.device)


class Aspheric(Surface):
    """Plane / stop (c == 0), sphere (no ai, k == 0) or even asphere (reference: surfaces.py:281)."""

    def __init__(self, r, d, c=0., k=0., ai=None, mat1=None, mat2=None, is_square=False, device=DEVICE,
                 diff=False, square=False):
        # The reference passes `is_square` to Surface.__init__ (which only stores h, w) and then OVERWRITES the attribute with
        # `square` (surfaces.py:303,330): Aspheric(is_square=True) ends up with a ROUND aperture, Aspheric(square=True) a square one.
        Surface.__init__(self, r, d, mat1, mat2, False, device)
        if is_square:
            self.h = self.w = r * np.sqrt(2)
        if square:
            raise NotImplementedError("square apertures are outside the focal-stack hot path")
        self.is_square = square
        self.c = torch.Tensor([c]).to(device)
        self.k = torch.Tensor([k]).to(device)
        if ai is not None:
            if len(ai) > _abi.MAX_AI:
                raise ValueError(f"at most {_abi.MAX_AI} even-asphere coefficients are supported, got {len(ai)}")
            self.ai = torch.Tensor(np.array(ai)).to(device)
            self.ai_degree = len(ai)
            for i, a in enumerate(ai):
                setattr(self, f"ai{2 * i + 2}", torch.Tensor([a]).to(device))
            if self.ai_degree == 4:
                # reference wart kept for parity (deeplens/surfaces.py:313): a 4-coefficient asphere evaluates its r^8 term
                # with the r^6 coefficient (`self.ai8 = torch.Tensor([ai[2]])`); `self.ai` (what write_lens_json stores) is untouched
                self.ai8 = torch.Tensor([ai[2]]).to(device)
        else:
            self.ai, self.ai_degree = None, 0

    # ---- packing for the HIP kernels -------------------------------------------------
    def kind(self):
        if float(self.c.item()) == 0.0:
            return _abi.SURF_STOP
        if self.ai is None and float(self.k.item()) == 0.0:
            return _abi.SURF_SPHERIC
        return _abi.SURF_ASPHERIC

    def pack(self, wvln):
        """aadff_surface_t for one wavelength.  Python-float expressions of the reference
        (r**2, eta**2, n1/n2) are evaluated in float64 and rounded once; expressions the
        reference evaluates on fp32 tensors ((1-eps)/c^2/(1+k), (j+1)*a_j) in fp32."""
        s = _abi.Surface()
        d, c, k = f32(self.d.item()), f32(self.c.item()), f32(self.k.item())
        s.d, s.c, s.k, s.r = d, c, k, self.r
        s.r2 = self.r ** 2
        s.kind = self.kind()
        s.k_gt_m1 = int(k > -1)
        with np.errstate(divide="ignore"):
            s.r2_shape = f32(1.0 - 1e-9) / (c * c) / (f32(1) + k) if (c != 0 and k > -1) else f32(np.inf)
        n1, n2 = self.mat1.ior(wvln), self.mat2.ior(wvln)
        ef, eb = n1 / n2, n2 / n1
        s.eta_fwd, s.eta_fwd2, s.eta_bwd, s.eta_bwd2 = ef, ef ** 2, eb, eb ** 2
        s.cos2_min_fwd, s.cos2_min_bwd = max(0.1, 1.0 - 1.0 / ef ** 2), max(0.1, 1.0 - 1.0 / eb ** 2)
        is_stop = s.kind == _abi.SURF_STOP
        s.refract_fwd = int(not (is_stop and ef == 1))
        s.refract_bwd = int(not (is_stop and eb == 1))
        s.n_ai = self.ai_degree
        for i in range(self.ai_degree):
            a = float(getattr(self, f"ai{2 * i + 2}").item())     # the coefficients the reference's sag reads (ai8 wart of degree 4)
            s.ai[i] = a
            s.dai[i] = float(f32(i + 1) * f32(a))
        s.newton_step_tol = self.newton_step_tol()
        return s

    def newton_step_tol(self):
        """Length [mm] of a Newton update below which the fused kernels go straight to the strict step (aadff.h:
        aadff_surface_t::newton_step_tol).  An update of length t leaves a residual of about kappa t^2 / (2 |f'|) with
        kappa = max |d^2 sag / d r^2| over the aperture and |f'| >= 0.7: tol = sqrt(4e-6 / kappa) keeps it <= 3e-6 mm, 3x
        under the strict step's 1e-5 test; capped at 10 um (kappa <= 0.04 / mm: both shipped 50 mm lenses)."""
        if self.kind() != _abi.SURF_ASPHERIC:
            return 0.0
        c, k = float(self.c.item()), float(self.k.item())
        rr = np.linspace(0.0, float(self.r), 257)
        q = 1.0 - (1.0 + k) * c * c * rr * rr
        rr, q = rr[q > 1e-6], q[q > 1e-6]
        curv = c / q ** 1.5                                   # second r-derivative of the conic sag
        for j in range(self.ai_degree):                       # + sum a_j (2j+2)(2j+1) r^(2j)
            n = 2 * (j + 1)
            curv = curv + float(getattr(self, f"ai{n}").item()) * n * (n - 1) * rr ** (n - 2)
        kappa = float(np.abs(curv).max()) if len(rr) else abs(c)
        return float(min(1e-2, np.sqrt(4e-6 / max(kappa, 1e-12))))

    def ray_reaction(self, ray):
        """Intersect + refract one Ray bundle at this surface (reference: surfaces.py:391-520).
        The travel direction is the batch-wide sign test of surfaces.py:399."""
        forward = bool((ray.d * ray.ra.unsqueeze(-1))[..., 2].sum() > 0)
        return trace_ray_object(ray, [self], 0, 1, forward, None)

    def surf_dict(self):
        kind = self.kind()
        out = {"type": ["Stop", "Spheric", "Aspheric"][kind], "r": self.r, "c": self.c.item(), "d": self.d.item()}
        if kind == _abi.SURF_ASPHERIC:
            out["k"] = self.k.item()
            out["ai"] = [float(a) for a in self.ai.tolist()] if self.ai is not None else []
        out["mat1"], out["mat2"] = self.mat1.name, self.mat2.name
        return out


def pack_table(surfaces, wvlns, device):
    """[len(wvlns)][n_surf] aadff_surface_t records as one device byte tensor."""
    n = len(surfaces)
    if n > _abi.MAX_SURF:
        raise ValueError(f"at most {_abi.MAX_SURF} surfaces are supported, got {n}")
    arr = (_abi.Surface * (n * len(wvlns)))()
    for li, w in enumerate(wvlns):
        for i, s in enumerate(surfaces):
            arr[li * n + i] = s.pack(w)
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(device)


def trace_ray_object(ray, surfaces, first, last, forward, state_dev, table=None):
    """Run aadff_trace_rays on a Ray (any leading shape); returns a NEW Ray."""
    dev = ray.o.device
    if dev.type != "cuda":
        _abi.require_gpu()
        dev = torch.device("cuda", torch.cuda.current_device())
    o = _abi.f32c(ray.o, dev).reshape(-1, 3)
    d = _abi.f32c(ray.d, dev).reshape(-1, 3)
    ra = _abi.f32c(ray.ra, dev).reshape(-1)
    if table is None:
        table = pack_table(surfaces, [ray.wvln], dev)
    oo, do, rao = torch.empty_like(o), torch.empty_like(d), torch.empty_like(ra)
    with torch.cuda.device(dev):
        _abi.call("aadff_trace_rays", _abi.ptr(o), _abi.ptr(d), _abi.ptr(ra), _abi.ptr(oo), _abi.ptr(do), _abi.ptr(rao),
                  o.shape[0], _abi.ptr(table), first, last, int(forward), _abi.ptr(state_dev), None,
                  _abi.stream_ptr(dev))
    out = Ray.__new__(Ray)
    out.wvln, out.coherent, out.device = ray.wvln, False, ray.o.device
    shape = ray.o.shape
    out.o = oo.reshape(shape).to(ray.o.device)
    out.d = do.reshape(shape).to(ray.o.device)
    out.ra = rao.reshape(shape[:-1]).to(ray.o.device)
    return out
