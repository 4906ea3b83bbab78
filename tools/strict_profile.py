#!/usr/bin/env python3
"""Time a strict-parity stack of the bench workload (aadff/strict_stack.py: fused levels on speculated counts; AADFF_STRICT_FUSED=0: the round-4 per-surface form): wall time per stack and, under
`rocprofv3 --kernel-trace --stats -- python3 tools/strict_profile.py`, the kernels' share of it."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np
import torch

os.environ["AADFF_STRICT_TIMING"] = "1"
try:
    q, per = open("/sys/fs/cgroup/cpu.max").read().split()
    torch.set_num_threads(max(1, min(16, int(q) // int(per)))) if q != "max" else None
except (OSError, ValueError):
    pass

from aadff import strict_stack
from aadff.synth import synth_depth_mm
from deeplens.optics import Lensgroup

H = W = 1024
S = 10
depth = synth_depth_mm(H, W, seed=5678)
dbar, fds = -float(depth.mean()), [float(f) for f in -np.linspace(depth.min(), depth.max(), S)]
lens = Lensgroup(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(H, W), device="cuda:0", parity="strict")
ts = []
render = "--render" in sys.argv                       # the whole stack (PSF maps + convolution) through render_focal_stack_m1
if render:
    from aadff.focal_stack import render_focal_stack_m1
    from aadff.synth import synth_rgb
    img = torch.from_numpy(synth_rgb(H, W, seed=1234))[None].to("cuda:0")
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6):
    torch.manual_seed(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if render:
        render_focal_stack_m1(lens, img, dbar, fds, 11, 11, 2048)
    else:
        strict_stack.strict_psf_maps(lens, dbar, fds, 11, 11, 2048)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print("strict stack: seconds per stack", [round(t, 4) for t in ts])
print("last stack, ms per segment:", lens._strict_timing)
print("count table:", strict_stack.StrictCounts.of(lens).stats)
