"""ORACLE (test infrastructure): ray hits -> ks x ks PSF histogram.

Follows deeplens/monte_carlo.py:9-121 (incoherent, interpolated branch only; `obliq`
is computed by the reference but unused for incoherent PSFs, :42,:51).
"""
import torch


def splat_one(points, ks, lo, hi, ra):
    """monte_carlo.py:60-121 for one point source: points [spp,2] (x,y), ra [spp]."""
    norm = torch.zeros_like(points)
    norm[:, 0] = (points[:, 1] - hi) / (lo - hi)          # row: +y at the top
    norm[:, 1] = (points[:, 0] - lo) / (hi - lo)          # col
    f = norm * (ks - 1)
    w_b = f[..., 0] - f[..., 0].floor()
    w_r = f[..., 1] - f[..., 1].floor()
    tl = f.floor().long()
    tr = torch.stack((f[:, 0], f[:, 1] + 1), dim=-1).floor().long()
    bl = torch.stack((f[:, 0] + 1, f[:, 1]), dim=-1).floor().long()
    br = tl + 1
    grid = torch.zeros(ks, ks)
    grid.index_put_(tuple(tl.t()), (1 - w_b) * (1 - w_r) * ra, accumulate=True)
    grid.index_put_(tuple(tr.t()), (1 - w_b) * w_r * ra, accumulate=True)
    grid.index_put_(tuple(bl.t()), w_b * (1 - w_r) * ra, accumulate=True)
    grid.index_put_(tuple(br.t()), w_b * w_r * ra, accumulate=True)
    return grid


def forward_integral(ray, ps, ks, pointc_ref):
    """monte_carlo.py:9-57 with an explicit reference centre; ray.o is [spp,N,3]."""
    pts = -ray.o[..., :2]
    lo, hi = (-ks / 2 + 0.5) * ps, (ks / 2 - 0.5) * ps
    shift = pts - pointc_ref
    ra = ray.ra * (shift[..., 0].abs() < (hi - 0.01 * ps)) * (shift[..., 1].abs() < (hi - 0.01 * ps))
    shift = shift * ra.unsqueeze(-1)
    return torch.stack([splat_one(shift[:, i, :], ks, lo, hi, ra[:, i]) for i in range(ray.o.shape[1])], dim=0)
