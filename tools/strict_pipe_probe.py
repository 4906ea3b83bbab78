#!/usr/bin/env python3
"""Strict stacks: the sequential loop against `StrictPipeline` at depths 2 and 3, with and without the high-priority stream for the
short levels and re-launches."""
import gc
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np
import torch

os.environ["AADFF_STRICT_TIMING"] = "1"

torch.set_num_threads(1)
from aadff import strict_stack
from aadff.focal_stack import render_focal_stack_m1
from aadff.synth import synth_depth_mm, synth_rgb
from deeplens.optics import Lensgroup

H = W = 1024
S = 10
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
depth = synth_depth_mm(H, W, seed=5678)
dbar, fds = -float(depth.mean()), [float(f) for f in -np.linspace(depth.min(), depth.max(), S)]
img = torch.from_numpy(synth_rgb(H, W, seed=1234))[None].to("cuda:0")
make = lambda: Lensgroup(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(H, W), device="cuda:0", parity="strict")

lens = make()
for i in range(3):
    render_focal_stack_m1(lens, img, dbar, fds, 11, 11, 2048)
gc.collect(); gc.freeze()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(STEPS):
    render_focal_stack_m1(lens, img, dbar, fds, 11, 11, 2048)
torch.cuda.synchronize()
print(f"sequential: {(time.perf_counter() - t0) / STEPS * 1e3:.3f} ms per stack", flush=True)
print("   segments:", lens._strict_timing, flush=True)

DEPTHS = [int(d) for d in os.environ.get("PROBE_DEPTHS", "2,3").split(",")]
for depth_n, prio in [(d, p) for d in DEPTHS for p in ("0", "1")]:
    os.environ["AADFF_STRICT_PRIO"] = prio
    pipe = strict_stack.StrictPipeline(make, depth=depth_n)
    for l in pipe.lenses:                                    # seed every lens's count table
        render_focal_stack_m1(l, img, dbar, fds, 11, 11, 2048)
    futs = [pipe.submit(img, dbar, fds) for _ in range(4)]
    [f.result() for f in futs]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    futs = [pipe.submit(img, dbar, fds) for _ in range(STEPS)]
    for f in futs:
        f.result()
    torch.cuda.synchronize()
    print(f"pipeline depth {depth_n}, priority stream {prio}: {(time.perf_counter() - t0) / STEPS * 1e3:.3f} ms per stack", flush=True)
    print("   segments:", pipe.lenses[0]._strict_timing, flush=True)
    pipe.close()

# host timeline of a depth-2 pipeline: when each half of each stack starts and ends, and what the stream of lens 0 saw
os.environ["AADFF_STRICT_PIPE_TRACE"] = "1"
os.environ["AADFF_STRICT_PRIO"] = "1"
pipe = strict_stack.StrictPipeline(make, depth=2)
for l in pipe.lenses:
    render_focal_stack_m1(l, img, dbar, fds, 11, 11, 2048)
futs = [pipe.submit(img, dbar, fds) for _ in range(4)]
[f.result() for f in futs]
torch.cuda.synchronize()
pipe.trace.clear()
t0 = time.perf_counter()
futs = [pipe.submit(img, dbar, fds) for _ in range(8)]
[f.result() for f in futs]
torch.cuda.synchronize()
print("timeline [ms]:", " ".join(f"{k}{w}@{(t - t0) * 1e3:.2f}" for k, w, t in pipe.trace))
print("stats:", [strict_stack.StrictCounts.of(l).stats for l in pipe.lenses])

if os.environ.get("PROBE_CPROFILE") == "1":
    import cProfile
    import pstats
    pipe = strict_stack.StrictPipeline(make, depth=3)
    for l in pipe.lenses:
        render_focal_stack_m1(l, img, dbar, fds, 11, 11, 2048)
    futs = [pipe.submit(img, dbar, fds) for _ in range(6)]
    [f.result() for f in futs]
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    futs = [pipe.submit(img, dbar, fds) for _ in range(30)]
    [f.result() for f in futs]
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(45)
