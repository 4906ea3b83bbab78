#!/usr/bin/env python3
"""The reference's own slice loop (refocus -> psf_map -> render_psf_map, deeplens/optics.py:779-783 per slice) through a STRICT lens at
the bench workload: per-call fused halves (default) against the per-surface calls of round 3 (AADFF_STRICT_CALLS_FUSED=0)."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd"), os.path.join(REPO, "tools")]
import numpy as np
import torch

from aadff.synth import synth_depth_mm, synth_rgb
from deeplens.optics import Lensgroup
from dropin_bench import m1_loop

H = W = 1024
S = 10
dev = torch.device("cuda:0")
depth = synth_depth_mm(H, W, seed=5678)
dbar, fds = -float(depth.mean()), [float(f) for f in -np.linspace(depth.min(), depth.max(), S)]
img = torch.from_numpy(synth_rgb(H, W, seed=1234))[None].to(dev)
for mode, reps in (("1", 10), ("0", 2)):
    os.environ["AADFF_STRICT_CALLS_FUSED"] = mode
    lens = Lensgroup(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(H, W), device=dev, parity="strict")
    torch.manual_seed(0)
    for _ in range(2):
        m1_loop(lens, img, dbar, fds, 11, 11, 2048)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        m1_loop(lens, img, dbar, fds, 11, 11, 2048)
    torch.cuda.synchronize()
    print(f"strict lens, per-call loop, AADFF_STRICT_CALLS_FUSED={mode}: {(time.perf_counter() - t0) / reps * 1e3:.1f} ms per 10-slice stack", flush=True)
