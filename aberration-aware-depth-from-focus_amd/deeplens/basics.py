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

_CAT = None


def _catalogue():
    """Glass tables of the reference (deeplens/basics.py:40-160) as data: {"material": name -> [n_d, V_d], "sellmeier":
    name -> [k1,l1,k2,l2,k3,l3], "schott": name -> [a0..a5], "glass_name": name -> display name}."""
    global _CAT
    if _CAT is None:
        import json
        import os
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "glass_catalogue.json")) as f:
            _CAT = json.load(f)
        for v in _CAT["material"].values():
            if v[1] == "inf":
                v[1] = math.inf
    return _CAT


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

    Three dispersion branches, chosen by the (lower-cased) name exactly as the reference does (:298-313):
    a name with Sellmeier coefficients (incl. air / vacuum / occluder, whose coefficients are all zero -> n = 1) uses the
    Sellmeier equation (:325-327); a name with Schott coefficients (the plastics) the Schott polynomial in nested form
    (:329-332); anything else - catalogue names that only have (n_d, V_d), and "n/V" strings - Cauchy's
    n = A + B / lambda_nm^2 (:334-335).  (A, B) are computed for every glass (find_aperture reads A, optics.py:190-198).
    The coefficient tables are data (glass_catalogue.json, emitted by tests/golden/make_golden.py)."""

    def __init__(self, name=None):
        self.name = "vacuum" if name is None else name.lower()
        cat = _catalogue()
        self.A, self.B = self._lookup_material()
        if self.name in cat["sellmeier"]:
            self.dispersion = "sellmeier"
            self.k1, self.l1, self.k2, self.l2, self.k3, self.l3 = cat["sellmeier"][self.name]
            self.glassname = self.name
        elif self.name in cat["schott"]:
            self.dispersion = "schott"
            self.a0, self.a1, self.a2, self.a3, self.a4, self.a5 = cat["schott"][self.name]
            self.glassname = cat["glass_name"][self.name]
        else:
            self.dispersion = "naive"
            self.glassname = self.name

    @staticmethod
    def register(name, n, V):
        """Add an (n_d, V_d) glass under a name of its own (Cauchy branch)."""
        _catalogue()["material"][name.lower()] = [float(n), float(V)]

    def load_sellmeier_param(self, params=None):
        """Set the Sellmeier coefficients k1, l1, k2, l2, k3, l3 by hand (deeplens/basics.py:339-347)."""
        self.k1, self.l1, self.k2, self.l2, self.k3, self.l3 = (0, 0, 0, 0, 0, 0) if params is None else params

    @staticmethod
    def nV_to_AB(n, V):
        lam_c, lam_d, lam_f = 656.3, 589.3, 486.1
        B = (n - 1) / V / (1.0 / lam_f ** 2 - 1.0 / lam_c ** 2)
        A = n - B * (1.0 / lam_d ** 2)
        return A, B

    def _lookup_material(self):
        """(A, B) from the catalogue's (n_d, V_d) or from an "n/V" string (deeplens/basics.py:363-379)."""
        hit = _catalogue()["material"].get(self.name)
        if hit is not None:
            n, V = hit
        else:
            parts = self.name.split("/")
            try:
                n, V = float(parts[0]), float(parts[1])
            except (ValueError, IndexError):
                raise ValueError(f"unknown glass {self.name!r}: not in the catalogue and not an 'n/V' string") from None
        self.n, self.V = n, V
        return self.nV_to_AB(n, V)

    def ior(self, wvln):
        wv = wvln if wvln < 10 else wvln * 1e-3          # [um]
        if self.dispersion == "sellmeier":
            w2 = wv ** 2
            return np.sqrt(1 + self.k1 * w2 / (w2 - self.l1) + self.k2 * w2 / (w2 - self.l2) + self.k3 * w2 / (w2 - self.l3))
        if self.dispersion == "schott":
            ws = wv ** 2
            return np.sqrt(self.a0 + self.a1 * ws + (self.a2 + (self.a3 + (self.a4 + self.a5 / ws) / ws) / ws) / ws)
        return self.A + self.B / (wv * 1e3) ** 2
