"""Constants, the Ray container and glass dispersion for the MI355X build.

Drop-in for the names callers import from the reference's deeplens/basics.py
(constants :18-35, DeepObj :164, Ray :215, Material :298).  Rays are plain holders of
device tensors; all arithmetic on them happens in the HIP kernels behind
include/aadff.h.  Star-import re-exports (torch, nn, np, ...) are part of the API the
reference scripts rely on.
"""
import copy
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as nnF

DEVICE = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")

DEFAULT_WAVE = 0.589
WAVE_RGB = [0.656, 0.589, 0.486]
WAVE_SPEC = [0.400 + 0.020 * i for i in range(16)]
DEPTH = -20000
GEO_SPP = 2048
MINT, MAXT = 1e-5, 1e5
DELTA = 1e-6
EPSILON = 1e-9

# (n_d, V_d) of the media the shipped lenses use by name; any other glass is given as an
# "n/V" string (the form both reference lens files use).  Catalogue glasses are optional:
# register more with Material.register(name, n, V).
_UNIT_MEDIA = ("vacuum", "air", "occluder")
_CATALOGUE = {"bk7": (1.5168, 64.17), "n-bk7": (1.5168, 64.17), "pmma": (1.491756, 58.00)}


class DeepObj:
    """Minimal base: device moves + clone (reference: deeplens/basics.py:164-212)."""

    def to(self, device=DEVICE):
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        for key, val in list(vars(self).items()):
            if torch.is_tensor(val):
                setattr(self, key, val.to(device))
            elif isinstance(val, (nn.Module, DeepObj)):
                val.to(device)
            elif isinstance(val, (list, tuple)):
                moved = [v.to(device) if (torch.is_tensor(v) or isinstance(v, DeepObj)) else v for v in val]
                if isinstance(val, list):
                    val[:] = moved
        return self

    def clone(self):
        return copy.deepcopy(self)

    def __call__(self, inp):
        return self.forward(inp)


class Ray(DeepObj):
    """A bundle of rays sharing one wavelength: o,d [...,3], ra [...] (1 = alive).

    Same constructor as the reference (deeplens/basics.py:216-244): `d` is
    L2-normalised on construction, `wvln` > 10 is read as nanometres."""

    def __init__(self, o, d, wvln=DEFAULT_WAVE, normalized=True, ra=None, en=None, obliq=None, opl=None,
                 coherent=False, device=DEVICE):
        self.o = o if torch.is_tensor(o) else torch.tensor(o).type(torch.float32)
        self.d = d if torch.is_tensor(d) else torch.tensor(d).type(torch.float32)
        self.wvln = wvln if wvln < 10 else wvln * 1e-3
        if coherent:
            raise NotImplementedError("coherent tracing is outside the focal-stack hot path")
        self.coherent = False
        self.ra = ra if ra is not None else torch.full(self.o.shape[:-1], 1.0, dtype=torch.float32)
        self.to(device)
        self.o = self.o.float()
        self.d = nnF.normalize(self.d.float(), p=2, dim=-1)

    def propagate_to(self, z, n=1):
        t = (z - self.o[..., 2]) / self.d[..., 2]
        self.o = self.o + self.d * t[..., None]
        return self

    prop_to = propagate_to

    def project_to(self, z):
        t = (z - self.o[..., 2]) / self.d[..., 2]
        return self.o[..., 0:2] + self.d[..., 0:2] * t[..., None]

    def clone(self, device=None):
        return copy.deepcopy(self).to(self.device if device is None else device)


class Material:
    """Refractive index n(lambda) in float64 on the host (deeplens/basics.py:298-379).

    "n/V" strings and catalogue names use Cauchy's n = A + B/lambda_nm^2 with (A,B) from
    (n_d, V_d); air/vacuum/occluder are exactly 1."""

    def __init__(self, name=None):
        self.name = "vacuum" if name is None else name.lower()
        if self.name in _UNIT_MEDIA:
            self.n, self.V = 1.0, math.inf
            self.dispersion = "unit"
        elif self.name in _CATALOGUE:
            self.n, self.V = _CATALOGUE[self.name]
            self.dispersion = "naive"
        else:
            parts = self.name.split("/")
            if len(parts) != 2:
                raise KeyError(f"unknown glass {name!r}: use an 'n/V' string or Material.register()")
            self.n, self.V = float(parts[0]), float(parts[1])
            self.dispersion = "naive"
        self.A, self.B = self.nV_to_AB(self.n, self.V)
        self.glassname = self.name

    @staticmethod
    def register(name, n, V):
        _CATALOGUE[name.lower()] = (float(n), float(V))

    @staticmethod
    def nV_to_AB(n, V):
        lam_c, lam_d, lam_f = 656.3, 589.3, 486.1
        B = (n - 1) / V / (1.0 / lam_f ** 2 - 1.0 / lam_c ** 2)
        A = n - B * (1.0 / lam_d ** 2)
        return A, B

    def ior(self, wvln):
        wv = wvln if wvln < 10 else wvln * 1e-3
        if self.dispersion == "unit":
            return 1.0
        return self.A + self.B / (wv * 1e3) ** 2
