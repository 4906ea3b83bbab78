#!/usr/bin/env python3
"""The reference's OWN loops through the drop-in API, timed: no StackPlan / StackPipeline, only the calls a reference
script makes.
  M1  for f in focus: lens.refocus(f); pm = lens.psf_map(depth, 11, 11, 2048); outs.append(render_psf_map(img, pm, 11)); torch.stack(outs, 2)
      (deeplens/optics.py:779-783 per slice, 2_aber_aware_dff_aif.py:104-114 for the stack)
  M2  for i in range(S): outs.append(psfnet.render(aif, depth, foc[:, i])); torch.stack(outs, 2)       (2_aber_aware_dff_aif.py:104-114)
`measure()` is what bench.py's `dropin_api` block calls."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np
import torch


def m1_loop(lens, img, dbar, fds, grid, ks, spp):
    from deeplens.render_psf import render_psf_map
    outs = []
    for f in fds:
        lens.refocus(f)
        pm = lens.psf_map(depth=dbar, grid=grid, ks=ks, spp=spp)
        outs.append(render_psf_map(img, pm, grid=grid))
    return torch.stack(outs, dim=2)


def measure(lens, img, dbar, fds, grid=11, ks=11, spp=2048, reps=20, psfnet=None, depth_map=None):
    dev = img.device
    out = {}
    for _ in range(3):
        m1_loop(lens, img, dbar, fds, grid, ks, spp)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        st = m1_loop(lens, img, dbar, fds, grid, ks, spp)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    t = (time.perf_counter() - t0) / reps
    lat = []
    for _ in range(max(5, reps // 2)):
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        m1_loop(lens, img, dbar, fds, grid, ks, spp)
        torch.cuda.synchronize(dev)
        lat.append(time.perf_counter() - t1)
    S, H, W = len(fds), img.shape[-2], img.shape[-1]
    out["m1_loop"] = {"ms_per_stack": round(t * 1e3, 4), "host_ms_per_stack": round(t_host / reps * 1e3, 4),
                      "latency_ms_p50": round(float(np.median(lat)) * 1e3, 4), "value": round(S * H * W / 1e6 / t, 1), "unit": "MP/s", "stacks": reps,
                      "loop": "for f in focus: lens.refocus(f); pm = lens.psf_map(depth, 11, 11, 2048); render_psf_map(img, pm, 11); torch.stack(.., 2)"}
    lens.check_flags()
    if psfnet is not None:
        from aadff.focal_stack import select_focus_dist
        fd = select_focus_dist(depth_map, S)                        # [B,S] metres, the 'linear' rule of dff/utils.py:4-50

        def m2_loop():
            return torch.stack([psfnet.render(img, -depth_map * 1e3, -fd[:, i] * 1e3) for i in range(S)], dim=2)
        for _ in range(2):
            m2_loop()
        torch.cuda.synchronize(dev)
        n2 = max(3, reps // 4)
        t0 = time.perf_counter()
        for _ in range(n2):
            m2_loop()
        torch.cuda.synchronize(dev)
        t2 = (time.perf_counter() - t0) / n2
        out["m2_loop"] = {"ms_per_stack": round(t2 * 1e3, 3), "value": round(S * H * W / 1e6 / t2, 1), "unit": "MP/s", "stacks": n2,
                          "loop": "for i in range(S): lens.render(aif, depth, foc_dist[:, i]); torch.stack(.., 2)   (PSFNet surrogate, fused MLP + gather kernel)"}
    return out


if __name__ == "__main__":
    from aadff.synth import mlp_state_dict, synth_depth_mm, synth_rgb
    from deeplens.optics import Lensgroup
    from deeplens.psfnet import PSFNet
    H = W = 1024
    S = 10
    dev = torch.device("cuda:0")
    depth = synth_depth_mm(H, W, seed=5678)
    dbar, fds = -float(depth.mean()), [float(f) for f in -np.linspace(depth.min(), depth.max(), S)]
    img = torch.from_numpy(synth_rgb(H, W, seed=1234))[None].to(dev)
    lens = Lensgroup(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(H, W), device=dev)
    net = PSFNet(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(H, W), kernel_size=11, device=dev)
    net.psfnet.load_state_dict({k: torch.from_numpy(v) for k, v in mlp_state_dict(seed=4321).items()})
    depth_m = (torch.from_numpy(depth)[None, None] / 1e3).to(dev)
    torch.manual_seed(0)
    print(measure(lens, img, dbar, fds, reps=int(sys.argv[1]) if len(sys.argv) > 1 else 20, psfnet=net, depth_map=depth_m))
