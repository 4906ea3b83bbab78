#!/usr/bin/env python3
"""local_psf_render timing (gather kernel alone): coalesced kernel vs AADFF_LOCAL_PSF=stream."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import torch, importlib
from aadff.synth import synth_rgb
rp = importlib.import_module("deeplens.render_psf")
dev = "cuda:0"
for H, W, ks in ((1024, 1024, 11), (480, 640, 11), (1024, 1024, 5)):
    img = torch.from_numpy(synth_rgb(H, W))[None].to(dev)
    psf = torch.rand(1, H, W, ks, ks, device=dev); psf /= psf.sum((-1, -2), keepdim=True)
    def t(n=20):
        rp.local_psf_render(img, psf, ks); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): rp.local_psf_render(img, psf, ks)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    tg = t()
    bpp = ks * ks * 4 + 24
    print(f"{H}x{W} ks {ks}: {tg*1e3:8.1f} us = {bpp*H*W/(tg*1e-3)/1e12:5.2f} TB/s ({bpp*H*W/(tg*1e-3)/8e12*100:.1f}% of 8 TB/s)")

# thin-lens baseline: PSF evaluated in the gather kernel (28 B/pixel) vs the tensor form (build [N,H,W,ks,ks] + local_psf_render)
from deeplens.psfnet import ThinLens
from aadff.synth import synth_depth_mm
for H, W in ((1024, 1024), (480, 640)):
    img = torch.from_numpy(synth_rgb(H, W))[None].to(dev)
    depth = -torch.from_numpy(synth_depth_mm(H, W))[None, None].to(dev)
    fd = torch.tensor([-1500.0], device=dev)
    thin = ThinLens(foc_len=50.0, fnum=1.8, kernel_size=11, sensor_size=[24.0, 24.0 * W / H], sensor_res=(H, W))
    for name, fn in (("in-kernel PSF", lambda: thin.render(img, depth, fd)), ("tensor form  ", lambda: thin.render_psf_tensor(img, depth, fd))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"ThinLens.render {H}x{W} {name}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us")
