"""ORACLE (test infrastructure): refractive index of the lens glasses, float64 on host.

Follows deeplens/basics.py:298-379.  Both shipped lenses use "n/V" strings, which the
reference resolves to the Cauchy form n = A + B/(1000*lambda)^2 with (A,B) from
nV_to_AB (:353-361); "air" resolves through the Sellmeier branch with all-zero
coefficients, i.e. n = 1 (:333-336, SELLMEIER_TABLE :103-106).
"""
import math

import numpy as np

_UNIT_INDEX = ("vacuum", "air", "occluder")


class Glass:
    def __init__(self, name=None):
        self.name = "vacuum" if name is None else name.lower()
        if self.name in _UNIT_INDEX:
            n, V = 1.0, math.inf
            self.dispersion = "sellmeier0"
        else:
            parts = self.name.split("/")
            if len(parts) != 2:
                raise KeyError(f"oracle covers n/V glasses and air only, got {name!r}")
            n, V = float(parts[0]), float(parts[1])
            self.dispersion = "cauchy"
        self.n, self.V = n, V
        inv2 = lambda a: 1.0 / a ** 2
        lambdas = [656.3, 589.3, 486.1]
        self.B = (n - 1) / V / (inv2(lambdas[2]) - inv2(lambdas[0]))
        self.A = n - self.B * inv2(lambdas[1])

    def ior(self, wvln):
        wv = wvln if wvln < 10 else wvln * 1e-3
        if self.dispersion == "sellmeier0":
            return np.sqrt(1 + 0.0 * wv ** 2 / (wv ** 2 - 0.0) + 0.0 * wv ** 2 / (wv ** 2 - 0.0) + 0.0 * wv ** 2 / (wv ** 2 - 0.0))
        return self.A + self.B / (wv * 1e3) ** 2
