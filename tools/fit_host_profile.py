#!/usr/bin/env python3
"""cProfile of the host side of the fit loop (plan.next + step) on the GPU box."""
import os, sys, cProfile, pstats, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np, torch
torch.set_num_threads(8)
from deeplens.psfnet import PSFNet, _TrainStep
dev = torch.device("cuda:0")
net = PSFNet(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(512, 512), kernel_size=11, device=dev)
step = _TrainStep(net.psfnet, 1e-4, 10000, 128, 121, dev, True, True)
plan = net._training_plan(128, 2048)
def it():
    inp, psf = plan.next()
    step(inp, psf)
for _ in range(30): it()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): it()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host loop {1e3 * (t1 - t0) / 300:.4f} ms/it, with device {1e3 * (t2 - t0) / 300:.4f} ms/it")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): it()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
