#!/usr/bin/env python3
"""Ablation of the fused PSF kernel: time it with (a) the real lens, (b) no chief pass, (c) lens truncated
to its first k surfaces -> marginal cost per surface, (d) spp scaling."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np, torch
from aadff import _abi
from aadff.focal_stack import StackPlan
from deeplens.optics import Lensgroup
dev = torch.device("cuda:0"); H = W = 1024; S = 10
lens = Lensgroup(os.path.join(REPO, "lenses/rf50mm/lens.json"), sensor_res=(H, W), device=dev)
plan = StackPlan(lens, S, H, W)
torch.manual_seed(0)
u = plan.uniforms(lens.sampler); ub = u.data_ptr()
dep, pts = plan.geometry(list(-np.linspace(500, 5000, S)), -1500.0)
lib = _abi.load_library(); st = _abi.stream_ptr(dev); p = lambda t: C.c_void_p(t.data_ptr())
assert lib.aadff_refocus(p(dep), S, C.c_void_p(ub), 2048, plan.per, p(plan.tab_green), plan.lc, p(plan.states), st) == 0

def run(nsurf=12, centre=1, spp=2048, spp_c=2048, iters=10):
    lc = _abi.LensConst.from_buffer_copy(bytes(plan.lc)); lc.n_surf = nsurf
    def f():
        return lib.aadff_psf_points(p(pts), S, 121, 3, p(plan.tab_rgb), p(plan.tab_green), lc, p(plan.states),
                                    C.c_void_p(ub + 4 * plan.o_main), spp, plan.per, plan.per_l,
                                    C.c_void_p(ub + 4 * plan.o_chief), spp_c, plan.per, plan.per_l, 11, centre, 1,
                                    p(plan.psf_maps), None, p(plan.flags), st)
    # NOTE: with nsurf < 12 the tables of wavelength l>0 are mis-indexed (stride n_surf) - timing only
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): assert f() == 0
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

print(f"full                         {run():8.1f} us")
print(f"no chief pass                {run(centre=0):8.1f} us")
print(f"spp 1024 (+2048 chief)       {run(spp=1024):8.1f} us")
print(f"spp 64, chief 64             {run(spp=64, spp_c=64):8.1f} us")
prev = None
for k in range(0, 13):
    t = run(nsurf=k, centre=0) if k > 0 else run(nsurf=1, centre=0, spp=2048)
    print(f"main pass only, first {k:2d} surfaces: {t:8.1f} us" + (f"   (+{t - prev:6.1f})" if prev is not None else ""))
    prev = t
