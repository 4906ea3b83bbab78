"""ORACLE (test infrastructure): refractive index of the lens glasses, float64 on host.

Follows deeplens/basics.py:298-379: the dispersion branch is chosen by table membership of the lower-cased name -
Sellmeier (:325-327; "air" / "vacuum" / "occluder" have all-zero coefficients, i.e. n = 1), Schott (:329-332), otherwise
the Cauchy form n = A + B/(1000*lambda)^2 with (A, B) from nV_to_AB (:353-361) of the table's (n_d, V_d) or of an "n/V"
string (:363-379).  The tables are data: oracle/glass_catalogue.json (emitted by tests/golden/make_golden.py from the
imported reference); pinned by fixture G14 (every catalogue name at six wavelengths) and G1.
"""
import json
import math
import os

import numpy as np

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "glass_catalogue.json")) as _f:
    _T = json.load(_f)


class Glass:
    def __init__(self, name=None):
        self.name = "vacuum" if name is None else name.lower()
        nv = _T["material"].get(self.name)
        if nv is None:
            parts = self.name.split("/")
            nv = [float(parts[0]), float(parts[1])]
        n, V = nv[0], (math.inf if nv[1] == "inf" else nv[1])
        self.n, self.V = n, V
        inv2 = lambda a: 1.0 / a ** 2
        lambdas = [656.3, 589.3, 486.1]
        self.B = (n - 1) / V / (inv2(lambdas[2]) - inv2(lambdas[0]))
        self.A = n - self.B * inv2(lambdas[1])
        if self.name in _T["sellmeier"]:
            self.dispersion, self.coef = "sellmeier", _T["sellmeier"][self.name]
        elif self.name in _T["schott"]:
            self.dispersion, self.coef = "schott", _T["schott"][self.name]
        else:
            self.dispersion, self.coef = "naive", None

    def ior(self, wvln):
        wv = wvln if wvln < 10 else wvln * 1e-3
        if self.dispersion == "sellmeier":
            k1, l1, k2, l2, k3, l3 = self.coef
            return np.sqrt(1 + k1 * wv ** 2 / (wv ** 2 - l1) + k2 * wv ** 2 / (wv ** 2 - l2) + k3 * wv ** 2 / (wv ** 2 - l3))
        if self.dispersion == "schott":
            a0, a1, a2, a3, a4, a5 = self.coef
            ws = wv ** 2
            return np.sqrt(a0 + a1 * ws + (a2 + (a3 + (a4 + a5 / ws) / ws) / ws) / ws)
        return self.A + self.B / (wv * 1e3) ** 2
