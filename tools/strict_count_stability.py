#!/usr/bin/env python3
"""How stable are the batch-wide Newton counts (deeplens/surfaces.py:547) of a strict stack from seed to seed?  Runs the bench
workload's strict stack in the per-surface form (counting passes = the true counts) for several seeds and prints every
(level, batch, surface) whose count is not the same for all seeds, with the any-bits margin where available."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np
import torch

from aadff import strict_stack
from aadff.synth import synth_depth_mm
from deeplens.optics import Lensgroup

H = W = 1024
S = 10
depth = synth_depth_mm(H, W, seed=5678)
dbar, fds = -float(depth.mean()), [float(f) for f in -np.linspace(depth.min(), depth.max(), S)]
lens = Lensgroup(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(H, W), device="cuda:0", parity="strict")
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seen = {}
for i in range(n_seeds):
    torch.manual_seed(i)
    strict_stack.strict_psf_maps(lens, dbar, fds, 11, 11, 2048, fused=False)
    for k, v in strict_stack.StrictCounts.of(lens).rows.items():
        seen.setdefault(k, []).append(v.copy())
for k, vs in seen.items():
    a = np.stack(vs)                                   # [seeds, B, (2,) MAX_SURF]
    var = (a != a[0]).any(0)
    print(k, "entries that vary:", int(var.sum()), "of", int(np.prod(var.shape[:-1]) * 12))
    for idx in np.argwhere(var):
        print("   ", tuple(int(x) for x in idx), "counts per seed:", a[(slice(None),) + tuple(idx)].tolist())
