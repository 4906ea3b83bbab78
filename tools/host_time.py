#!/usr/bin/env python3
"""Host-side cost of one M1 step: time the Python call path with the GPU work stubbed out by an idle device
(the call only enqueues) -> how far the host is from being the bottleneck."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np, torch
from aadff.focal_stack import StackPlan, render_focal_stack_m1
from aadff.synth import synth_rgb
from deeplens.optics import Lensgroup
dev = torch.device("cuda:0"); H = W = 1024
lens = Lensgroup(os.path.join(REPO, "lenses/rf50mm/lens.json"), sensor_res=(H, W), device=dev)
img = torch.from_numpy(synth_rgb(H, W))[None].to(dev); plan = StackPlan(lens, 10, H, W)
fds = -np.linspace(500, 5000, 10)
for _ in range(20): render_focal_stack_m1(lens, img, -1500., fds, plan=plan, update_lens=False)
torch.cuda.synchronize()
# host-only: rand_into cost
t0 = time.perf_counter()
for i in range(200): lens.sampler.rand_into(plan.u_pin[i % plan.RING])
t_rng = (time.perf_counter() - t0) / 200
# whole call, GPU-bound pace vs host pace: run N steps and sync at the end
t0 = time.perf_counter()
for _ in range(300): render_focal_stack_m1(lens, img, -1500., fds, plan=plan, update_lens=False)
t_enq = (time.perf_counter() - t0) / 300
torch.cuda.synchronize()
t_tot = (time.perf_counter() - t0) / 300
print(f"host RNG fill {t_rng*1e3:.3f} ms | enqueue-side per step {t_enq*1e3:.3f} ms | end-to-end per step {t_tot*1e3:.3f} ms")
