#!/usr/bin/env python3
"""cProfile of the host side of a fused strict stack (bench workload): where the ~2 ms of Python / torch-CPU time per stack go."""
import cProfile
import os
import pstats
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np
import torch

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
for i in range(3):
    strict_stack.strict_psf_maps(lens, dbar, fds, 11, 11, 2048)
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    strict_stack.strict_psf_maps(lens, dbar, fds, 11, 11, 2048)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
