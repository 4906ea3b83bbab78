#!/usr/bin/env python3
"""Kernel A/B harness: time the conv / PSF kernels of several builds of libaadff.so in ONE
process (interleaved rounds, median + min), checking each build's output against the first.
Usage: python tools/kbench.py [--psf] lib1.so lib2.so ...   (default: csrc/libaadff.so)"""
import argparse, ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np, torch
from aadff import _abi
from aadff.focal_stack import StackPlan, stack_uniform_layout
from aadff.synth import synth_rgb
from deeplens.optics import Lensgroup

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="*")
ap.add_argument("--psf", action="store_true")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
libs = a.libs or [_abi.LIB_PATH]
dev = torch.device("cuda:0")
H = W = 1024; S, G, KS = 10, 11, 11
img = torch.from_numpy(synth_rgb(H, W))[None].to(dev)
rng = np.random.Generator(np.random.PCG64(3))
maps = torch.from_numpy(rng.random((S, 3, G * KS, G * KS), dtype=np.float32)).to(dev) / 121
out = torch.empty((1, 3, S, H, W), device=dev)
out1 = torch.empty((1, 3, H, W), device=dev)
L = [_abi.load_library(p) for p in libs]
st = _abi.stream_ptr(dev)
p = lambda t: C.c_void_p(t.data_ptr())

def run(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        rc = fn(); assert rc == 0, rc
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us

cases = {
    "conv_stack10": lambda lib: (lambda: lib.aadff_render_psf_map_stack(p(img), p(maps), p(out), 1, 3, S, H, W, G, KS, st)),
    "conv_single": lambda lib: (lambda: lib.aadff_render_psf_map(p(img), p(maps), p(out1), 1, 3, H, W, G, KS, st)),
}
if a.psf:
    lens = Lensgroup(os.path.join(REPO, "lenses/rf50mm/lens.json"), sensor_res=(H, W), device=dev)
    plan = StackPlan(lens, S, H, W)
    torch.manual_seed(0)
    u = plan.uniforms(lens.sampler)
    dep, pts = plan.geometry(list(-np.linspace(500, 5000, S)), -1500.0)
    lens.refocus(-1500.0)
    ub = u.data_ptr()
    def mk_psf(lib):
        def f():
            rc = lib.aadff_refocus(p(dep), S, C.c_void_p(ub), 2048, plan.per, p(plan.tab_green), plan.lc, p(plan.states), st)
            return rc or lib.aadff_psf_points(p(pts), S, 121, 3, p(plan.tab_rgb), p(plan.tab_green), plan.lc, p(plan.states),
                                              C.c_void_p(ub + 4 * plan.o_main), 2048, plan.per, plan.per_l,
                                              C.c_void_p(ub + 4 * plan.o_chief), 2048, plan.per, plan.per_l, KS, 1, 1,
                                              p(plan.psf_maps), None, p(plan.flags), st)
        return f
    cases["refocus+psf_grid"] = mk_psf

for name, mk in cases.items():
    fns = [mk(lib) for lib in L]
    ref = None
    res = {i: [] for i in range(len(L))}
    for i, fn in enumerate(fns):      # warm + correctness
        run(fn, 2)
        cur = (plan.psf_maps if name.startswith("refocus") else (out if "stack" in name else out1)).clone()
        if ref is None:
            ref = cur
        else:
            d = (cur - ref).abs().max().item()
            if name.startswith("refocus"):
                print(f"   [{os.path.basename(libs[i])}] max |psf_map - first lib| = {d:.2e}")
            else:
                if d > 1e-6: print(f"   [{os.path.basename(libs[i])}] {name}: output differs from first lib by {d:.2e} (ablation build?)")
    for r in range(a.rounds):
        for i, fn in enumerate(fns):
            res[i].append(run(fn, a.iters))
    for i in range(len(L)):
        v = np.array(res[i])
        extra = ""
        if name.startswith("conv"):
            s_ = S if "stack" in name else 1
            extra = f"  roofline_frac={24 * H * W * s_ / (np.median(v) * 1e-6) / 8e12:.3f}  TFLOP/s={726 * H * W * s_ / (np.median(v) * 1e-6) / 1e12:.1f}"
        print(f"{name:18s} {os.path.basename(libs[i]):28s} median {np.median(v):9.1f} us  min {v.min():9.1f} us{extra}", flush=True)
