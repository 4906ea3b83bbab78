#!/usr/bin/env python3
"""Where does the fp32 floor of a needle-PSF slice come from?  (CPU, oracle only; VERDICT r2 item 1 background.)

Slice 1 of the bench stack (focus -989.9 mm, depth plane -1500 mm... needle PSFs), seed 0.  The float32 oracle reproduces the
reference bit for bit.  Each variant perturbs ONE ingredient at the level any other float32 evaluation order would and
reports the rel-L2 distance of the PSF map / rendered slice to the unperturbed result:
  centre   chief-ray centre summed in float64 instead of torch's float32 cascade sum (same hits)
  dsensor  d_sensor moved by one float32 ulp
  hits     main-ray sensor hits from a float64 trace of the same float32 draws (centre and d_sensor unperturbed)
  all      everything in float64 (= G13's floor)
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np
import torch

from aadff.synth import synth_depth_mm, synth_rgb
from oracle import conv as oconv
from oracle import lens as ol
from oracle.splat import forward_integral

K = int(os.environ.get("SLICE", "1"))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


class Probe(ol.OracleLens):
    mode = "none"

    def psf_center(self, pobj):
        ray = self.trace2sensor(self.sample_from_points(pobj, spp=ol.GEO_SPP, shrink_pupil=True))
        if self.mode == "centre":
            c = ((ray.o.double() * ray.ra.double().unsqueeze(-1)).sum(0) / ray.ra.double().unsqueeze(-1).sum(0).add(ol.EPSILON)).float()
        else:
            c = (ray.o * ray.ra.unsqueeze(-1)).sum(0) / ray.ra.unsqueeze(-1).sum(0).add(ol.EPSILON)
        return -c[..., :2]

    def psf(self, points, ks=31, wvln=ol.DEFAULT_WAVE, spp=ol.GEO_SPP, center=True):
        pobj = self.object_points(points)
        rays = self.sample_from_points(pobj, spp=spp, wvln=wvln)
        if self.mode == "hits":
            torch.set_default_dtype(torch.float64)
            l64 = ol.OracleLens(self.path, sensor_res=self.sensor_res)
            l64.d_sensor, l64.hfov = self.d_sensor, self.hfov
            r64 = l64.trace2sensor(ol.Rays(rays.o.double(), rays.d.double(), rays.ra.double(), wvln, normalize=False))
            torch.set_default_dtype(torch.float32)
            ray = ol.Rays(r64.o.float(), r64.d.float(), r64.ra.float(), wvln, normalize=False)
        else:
            ray = self.trace2sensor(rays)
        ref = self.psf_center(pobj)
        psf = forward_integral(ray, ps=self.pixel_size, ks=ks, pointc_ref=ref)
        return psf / psf.sum(-1).sum(-1).unsqueeze(-1).unsqueeze(-1)


def main():
    H = W = 1024
    img = torch.from_numpy(synth_rgb(H, W, seed=1234))[None]
    depth = synth_depth_mm(H, W, seed=5678)
    dbar, fds = -float(depth.mean()), -np.linspace(depth.min(), depth.max(), 10)
    lp = os.path.join(REPO, "lenses", "rf50mm", "lens.json")
    g9 = np.load(os.path.join(REPO, "tests", "golden", "g9_stack_m1_1024.npz"))
    lens = Probe(lp, sensor_res=(H, W))
    lens.path = lp
    torch.manual_seed(0)
    for k in range(K):
        lens.refocus(float(fds[k]))
        lens.psf_map(depth=dbar, grid=11, ks=11, spp=2048)
    lens.refocus(float(fds[K]))
    st = torch.get_rng_state()
    base = lens.psf_map(depth=dbar, grid=11, ks=11, spp=2048)
    assert np.abs(base.numpy() - g9["psf_maps"][K]).max() <= 1e-6
    im0 = oconv.render_psf_map(img, base, 11)[0].numpy()
    d0 = lens.d_sensor
    for mode in ("centre", "dsensor", "hits"):
        torch.set_rng_state(st)
        lens.mode = mode
        lens.d_sensor = float(np.nextafter(np.float32(d0), np.float32(1e9))) if mode == "dsensor" else d0
        pm = lens.psf_map(depth=dbar, grid=11, ks=11, spp=2048)
        im = oconv.render_psf_map(img, pm, 11)[0].numpy()
        # per-grid-point PSF error, to see whether a few points dominate
        a = pm.numpy().reshape(3, 11, 11, 11, 11).transpose(0, 1, 3, 2, 4).reshape(3, 121, 121)
        b = base.numpy().reshape(3, 11, 11, 11, 11).transpose(0, 1, 3, 2, 4).reshape(3, 121, 121)
        per = np.sqrt(((a - b) ** 2).sum(-1)) / np.sqrt((b ** 2).sum(-1))
        print(f"slice {K} {mode:8s}: PSF rel-L2 {rel(pm.numpy(), base.numpy()):.3e}  image rel-L2 {rel(im, im0):.3e}   "
              f"per-point PSF err: median {np.median(per):.2e} p90 {np.percentile(per, 90):.2e} max {per.max():.2e}", flush=True)


if __name__ == "__main__":
    main()
