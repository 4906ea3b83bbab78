#!/usr/bin/env python3
"""Print the actual parity errors (not just pass/fail) of the HIP path vs the golden vectors."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root (uses the oracle: lives under tests/)
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np, torch, importlib
from aadff import _abi
from aadff.focal_stack import render_focal_stack_m1
from aadff.synth import synth_rgb
from deeplens.optics import Lensgroup
from oracle import conv as oconv
rp = importlib.import_module("deeplens.render_psf")
G = os.path.join(REPO, "tests/golden"); DEV = "cuda:0"
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64)))
lp = lambda n: os.path.join(REPO, "lenses", n, "lens.json")

g = np.load(f"{G}/g2_g3_trace_splat.npz")
lens = Lensgroup(lp("rf50mm"), sensor_res=(1024, 1024), device=DEV)
torch.manual_seed(0); lens.refocus(-2000.0)
print(f"refocus d_sensor rel err {abs(lens.d_sensor - float(g['d_sensor'])) / float(g['d_sensor']):.2e}  hfov rel err {abs(lens.hfov - float(g['hfov'])) / float(g['hfov']):.2e}")
pobj, ut, ur = tt(g["points_obj"]).to(DEV), tt(g["u_theta"]).to(DEV), tt(g["u_r"]).to(DEV)
pz, pr = lens.entrance_pupil()
o = torch.zeros((256, 121, 3), device=DEV); d = torch.zeros_like(o); ra = torch.zeros((256, 121), device=DEV)
_abi.call("aadff_trace_points", _abi.ptr(pobj), 121, _abi.ptr(ut), _abi.ptr(ur), 256, float(pz), float(pr), _abi.ptr(lens._table([0.589])),
          12, _abi.ptr(lens._state_device()), _abi.ptr(o), _abi.ptr(d), _abi.ptr(ra), _abi.stream_ptr(torch.device(DEV)))
rah, want = ra.cpu().numpy() > 0, g["sensor_ra"] > 0
both = rah & want
err = np.abs(o[..., :2].cpu().numpy() - g["sensor_xy"])[both]
print(f"sensor hits: mask mismatch {(rah != want).mean():.2e}  xy err mean {err.mean():.2e} max {err.max():.2e} mm (fp32-vs-fp64 floor: 5.3e-6 / 4.1e-5)")

g4 = np.load(f"{G}/g4_psf_map.npz")
for name, res, foc, depth, spp in (("rf50mm", (1024, 1024), -2000.0, -1500.0, 2048), ("50mm_f2.8", (256, 256), -1000.0, -1250.0, 512)):
    key = name.replace(".", "_")
    lens = Lensgroup(lp(name), sensor_res=res, device=DEV)
    torch.manual_seed(0); lens.refocus(foc)
    pm = lens.psf_map(depth=depth, grid=11, ks=11, spp=spp)
    img = tt(synth_rgb(256, 256))[None]
    want = oconv.render_psf_map(img, tt(g4[f"{key}_psf_map"]), 11).numpy()
    got = rp.render_psf_map(img.to(DEV), pm, 11).cpu().numpy()
    print(f"psf_map {name:10s}: PSF rel-L2 {rel(pm.cpu().numpy(), g4[f'{key}_psf_map']):.2e} (tol 2e-3)   image rel-L2 {rel(got, want):.2e} (tol 1e-4)")

g8 = np.load(f"{G}/g8_stack_m1.npz")
lens = Lensgroup(lp("rf50mm"), sensor_res=(256, 256), device=DEV)
img = tt(synth_rgb(256, 256))[None].to(DEV)
torch.manual_seed(0)
stack, maps = render_focal_stack_m1(lens, img, float(g8["dbar"]), g8["fds"], grid=11, ks=11, spp=2048, return_maps=True)
s = stack[0].cpu().numpy()
print(f"M1 stack 256^2: maps rel-L2 {rel(maps.cpu().numpy(), g8['psf_maps']):.2e}  image crop rel-L2 {rel(s[:, :, 96:160, 96:160], g8['crop']):.2e} (tol 1e-4)")
