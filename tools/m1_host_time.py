#!/usr/bin/env python3
"""Host time per M1 stack: enqueue 3 stacks right after a device sync (nothing blocks: the queue is empty and the ring
guards passed long ago) and time the enqueue alone; then the same 3 stacks including the device time."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np, torch
from aadff.focal_stack import StackPlan, render_focal_stack_m1
from aadff.synth import synth_rgb
from deeplens.optics import Lensgroup
torch.set_num_threads(4)
dev = "cuda:0"
lens = Lensgroup(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(1024, 1024), device=dev)
img = torch.from_numpy(synth_rgb(1024, 1024))[None].to(dev)
fds = np.linspace(-500, -5000, 10)
plan = StackPlan(lens, 10, 1024, 1024)
torch.manual_seed(0)
for _ in range(40): render_focal_stack_m1(lens, img, -1500.0, fds, plan=plan)
torch.cuda.synchronize()
host, tot = [], []
for rep in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): render_focal_stack_m1(lens, img, -1500.0, fds, plan=plan)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append((t1 - t0) / 3); tot.append((t2 - t0) / 3)
print(f"host enqueue per stack: median {np.median(host) * 1e3:.3f} ms (min {min(host) * 1e3:.3f}); with device: {np.median(tot) * 1e3:.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); torch.cuda.synchronize(); pr.enable()
for _ in range(3): render_focal_stack_m1(lens, img, -1500.0, fds, plan=plan)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
