"""ORACLE (test infrastructure, not product code): the reference's ray-surface arithmetic as a SCALAR program.

oracle/lens.py restates the reference with the same torch tensor ops, so it is bit-equal by construction.  This module
restates the same computation one float32 operation at a time (numpy float32 arrays, every `*`, `+`, `/`, `sqrt`
individually rounded, in the order the reference's tensor expressions evaluate them) - the specification the HIP kernels
of `parity="strict"` follow (csrc/trace.hip, namespace strict).  tests/test_oracle_golden.py checks it bit for bit
against oracle/lens.py and the G2 fixture.  What the tensor ops hide and this file makes explicit:

* `F.normalize` / `vector_norm` accumulate x0*x0 -> fma(x1,x1,.) -> fma(x2,x2,.) (ATen's norm kernel contracts; measured:
  100 % match with the fused form, 89 % with separate rounding), then THREE IEEE divisions by max(norm, 1e-12);
* `torch.sum(d * n, -1)` is ((p0 + p1) + p2) with separately rounded products;
* the Newton loop runs a BATCH-WIDE number of iterations (`while (|ft| > 5e-5).any()`, deeplens/surfaces.py:547): every
  ray of the batch takes the same number of updates, converged or not;
* Python-float scalars (r^2, eta, eta^2, tolerances) enter tensor ops rounded to float32 once.

Citations: deeplens/surfaces.py:391-520 (ray_reaction), :523-586 (newton), :589-630 (normal), :633-679 (refract),
:724-743 (validity), :787-830 (sag, dsag/dr2).
"""
import numpy as np

f32 = np.float32
MAXT, EPS = f32(1e5), f32(1e-9)
TOL_LOOSE, TOL_TIGHT, STEP_BOUND, MAXITER = f32(50e-6), f32(10e-6), f32(5), 10


def fma(a, b, c):
    """fused multiply-add of float32 arrays (product exact in float64; the double rounding of the sum is a 2^-29 event)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def normalize3(x, y, z):
    n2 = fma(z, z, fma(y, y, x * x))
    den = np.maximum(np.sqrt(n2), f32(1e-12))
    return x / den, y / den, z / den


class ScalarSurface:
    """Per-surface constants as the reference holds them: d, c, k, a_j float32 tensors; r a Python float."""

    def __init__(self, d, c, k, r, ai, eta_fwd, eta_bwd):
        self.d, self.c, self.k, self.r = f32(d), f32(c), f32(k), float(r)
        self.ai = None if ai is None else [f32(a) for a in ai]
        self.r2 = f32(self.r ** 2)
        self.eta = {True: float(eta_fwd), False: float(eta_bwd)}
        self.flat = self.c == 0
        self.spheric = (self.ai is None) and self.k == 0
        if not self.flat and self.k > -1:
            self.r2_shape = f32(1 - 1e-9) / (self.c * self.c) / (f32(1) + self.k)

    def _a(self, r2):                                  # (1 + k) * r2 * c**2
        return (f32(1) + self.k) * r2 * (self.c * self.c)

    def sag(self, r2):
        z = r2 * self.c / (f32(1) + np.sqrt(f32(1) - self._a(r2)))
        if self.ai is not None:
            for j, a in enumerate(self.ai):
                z = z + (a * r2 if j == 0 else a * _pow(r2, j + 1))
        return z

    def dsag(self, r2):
        sf = np.sqrt(f32(1) - self._a(r2))
        g = (f32(1) + sf + self._a(r2) / f32(2) / sf) * self.c / ((f32(1) + sf) * (f32(1) + sf))
        if self.ai is not None:
            for j, a in enumerate(self.ai):
                if j == 0:
                    g = g + a
                elif j == 1:
                    g = g + f32(2) * a * r2
                else:
                    g = g + f32(j + 1) * a * _pow(r2, j)
        return g

    def valid_strict(self, x, y):
        q = x * x + y * y
        return (q < self.r2) & (q < self.r2_shape) if self.k > -1 else (q < self.r2)

    def valid_loose(self, x, y):
        q = x * x + y * y
        return (q < self.r2_shape) if self.k > -1 else (q > 0)


def _pow(x, n):
    """torch.pow(tensor, int n) for n >= 2 on CPU: n == 2 is x*x, n == 3 is x*x*x; higher integer powers go through the
    vectorised pow (Sleef) - the reference only reaches n <= 6 through r2**j; checked against torch in the tests."""
    if n == 2:
        return x * x
    if n == 3:
        return (x * x) * x
    import torch
    return torch.pow(torch.from_numpy(np.ascontiguousarray(x)), n).numpy()


def newton(s, o, d, ra, n_iter=None):
    """(valid, t, iterations).  `n_iter` None: the batch-wide count of the reference's while loop over THIS batch."""
    ox, oy, oz = o
    dx, dy, dz = d
    t0 = (s.d - oz) / dz
    t = t0
    ft = np.full_like(oz, MAXT)
    alive = ra > 0

    def residual(t, mask_fn):
        px, py, pz = ox + dx * t, oy + dy * t, oz + dz * t
        m = (mask_fn(px, py) & alive).astype(f32)
        xm, ym = px * m, py * m
        r2 = xm * xm + ym * ym
        ft = s.sag(r2) + s.d - pz
        dr2dt = f32(2) * ((dx * dx + dy * dy) * t + (dx * ox + dy * oy))
        dfdt = s.dsag(r2) * dr2dt - dz
        return ft, dfdt

    it = 0
    with np.errstate(all="ignore"):
        while (it < n_iter) if n_iter is not None else ((np.abs(ft) > TOL_LOOSE).any() and it < MAXITER):
            it += 1
            ft, dfdt = residual(t, s.valid_loose)
            t = t - np.clip(ft / (dfdt + EPS), -STEP_BOUND, STEP_BOUND)
        t1 = t - t0
        t = t0 + t1
        ft, dfdt = residual(t, s.valid_strict)
        t = t - np.clip(ft / (dfdt + EPS), -STEP_BOUND, STEP_BOUND)
        px, py = ox + dx * t, oy + dy * t
        valid = s.valid_strict(px, py) & (np.abs(ft) < TOL_TIGHT) & alive & (t > 0)
    return valid, t, it


def refract(s, p, d, ra, forward):
    x, y, z = p
    dx, dy, dz = d
    eta = s.eta[forward]
    if s.flat:
        nx, ny, nz = np.zeros_like(x), np.zeros_like(y), np.full_like(z, -1)
    elif s.spheric:
        R = f32(1) / s.c
        if s.c > 0:
            nx, ny, nz = f32(2) * x, f32(2) * y, f32(2) * z - f32(2) * (s.d + R)
        else:
            nx, ny, nz = f32(-2) * x, f32(-2) * y, f32(-2) * z + f32(2) * (s.d + R)
    else:
        v = (ra > 0).astype(f32)
        xv, yv = x * v, y * v
        g = s.dsag(xv * xv + yv * yv)
        nx, ny, nz = g * f32(2) * xv, g * f32(2) * yv, np.full_like(x, -1)
    nx, ny, nz = normalize3(nx, ny, nz)
    if forward:
        nx, ny, nz = -nx, -ny, -nz
    cosi = (dx * nx + dy * ny) + dz * nz
    e, e2 = f32(eta), f32(eta ** 2)
    c2 = cosi * cosi
    valid = (c2 > f32(0.1)) & (e2 * (f32(1) - c2) < f32(1)) & (ra > 0)
    with np.errstate(invalid="ignore"):
        sr = np.sqrt(f32(1) - e2 * (f32(1) - c2) * valid.astype(f32))
    ndx = sr * nx + e * (dx - cosi * nx)
    ndy = sr * ny + e * (dy - cosi * ny)
    ndz = sr * nz + e * (dz - cosi * nz)
    return (np.where(valid, ndx, dx), np.where(valid, ndy, dy), np.where(valid, ndz, dz)), ra * valid.astype(f32)


def react(s, o, d, ra, forward, n_iter=None):
    """One surface interaction; returns (o, d, ra, newton iterations)."""
    ox, oy, oz = o
    dx, dy, dz = d
    it = 0
    if s.flat:
        t = (s.d - oz) / dz
        px, py, pz = ox + t * dx, oy + t * dy, oz + t * dz
        valid = (np.sqrt(px * px + py * py) <= f32(s.r)) & (ra > 0)
    else:
        nvalid, t, it = newton(s, o, d, ra, n_iter)
        px, py, pz = ox + t * dx, oy + t * dy, oz + t * dz
        valid = ((px * px + py * py <= s.r2) & (t >= 0) & (ra > 0)) if s.spheric else nvalid
    p = (np.where(valid, px, ox), np.where(valid, py, oy), np.where(valid, pz, oz))
    ra = ra * valid.astype(f32)
    if s.flat and s.eta[forward] == 1:
        return p, d, ra, it
    d, ra = refract(s, p, d, ra, forward)
    return p, d, ra, it


def surfaces_from_oracle(lens, wvln):
    """ScalarSurface list from an oracle.lens.OracleLens (constants only)."""
    out = []
    for s in lens.surfaces:
        n1, n2 = s.mat1.ior(wvln), s.mat2.ior(wvln)
        out.append(ScalarSurface(s.d.item(), s.c.item(), s.k.item(), s.r, None if s.ai is None else [a.item() for a in s.ai],
                                 n1 / n2, n2 / n1))
    return out
