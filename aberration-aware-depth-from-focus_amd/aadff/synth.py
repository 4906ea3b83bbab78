"""Seeded synthetic inputs shared by tests, bench.py and the golden-vector generator.

Pure numpy (PCG64) so the same arrays are produced in the build container and on the
GPU box without touching torch's RNG (measurement plan: SURVEY.md §8d).
"""
import numpy as np


def _box5(a):
    """5x5 box filter, edge-replicated, fixed summation order (float32)."""
    p = np.pad(a, ((0, 0), (2, 2), (2, 2)), mode="edge")
    h, w = a.shape[-2:]
    acc = np.zeros_like(a)
    for dy in range(5):
        for dx in range(5):
            acc += p[:, dy:dy + h, dx:dx + w]
    return (acc * np.float32(1.0 / 25.0)).astype(np.float32)


def synth_rgb(h, w, seed=1234):
    """All-in-focus RGB image [3,h,w] float32 in [0,1]: box-filtered PCG64 noise plus
    a few hard edges so blur differences between focus distances are visible."""
    rng = np.random.Generator(np.random.PCG64(seed))
    a = _box5(rng.random((3, h, w), dtype=np.float32))
    yy, xx = np.mgrid[0:h, 0:w]
    bars = (((xx // max(w // 32, 1)) + (yy // max(h // 32, 1))) % 2).astype(np.float32)
    a = 0.6 * a + 0.4 * bars[None] * np.array([1.0, 0.8, 0.6], np.float32)[:, None, None]
    return np.ascontiguousarray(a.astype(np.float32))


def synth_depth_mm(h, w, seed=5678, dmin=500.0, dmax=5000.0, planes=12):
    """Piecewise-planar depth map [h,w] float32, POSITIVE millimetres in [dmin,dmax]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cy = rng.random(planes) * h
    cx = rng.random(planes) * w
    d0 = dmin + rng.random(planes) * (dmax - dmin)
    gy = (rng.random(planes) - 0.5) * (dmax - dmin) / h
    gx = (rng.random(planes) - 0.5) * (dmax - dmin) / w
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    dist = (yy[None] - cy[:, None, None]) ** 2 + (xx[None] - cx[:, None, None]) ** 2
    idx = dist.argmin(0)
    d = d0[idx] + gy[idx] * (yy - cy[idx]) + gx[idx] * (xx - cx[idx])
    return np.ascontiguousarray(np.clip(d, dmin, dmax).astype(np.float32))


MLP_LAYERS = [(4, 64), (64, 256)] + [(256, 256)] * 8 + [(256, 121)]


def mlp_state_dict(seed=4321, ks=11):
    """Procedural PSFNet-MLP weights with the reference's checkpoint key layout
    (`net.{0,2,...,20}.{weight,bias}`, deeplens/psfnet_arch.py:24-41) as numpy arrays."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    layers = list(MLP_LAYERS)
    layers[-1] = (256, ks * ks)
    for i, (fi, fo) in enumerate(layers):
        b = np.sqrt(6.0 / fi)
        sd[f"net.{2 * i}.weight"] = ((rng.random((fo, fi), dtype=np.float32) * 2 - 1) * b).astype(np.float32)
        sd[f"net.{2 * i}.bias"] = ((rng.random(fo, dtype=np.float32) * 2 - 1) * 0.1).astype(np.float32)
    return sd
