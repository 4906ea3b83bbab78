#!/usr/bin/env python3
"""Edge-exact stacks (Lensgroup(parity="edge")): ms per stack of the sequential loop (with the host segments of one stack) and of
StrictPipeline at several depths, next to the strict and the fast path on the same box.  GPU only.
Usage: python tools/edge_bench.py [steps] ; PROBE_DEPTHS=2,4 ; PROBE_STRICT=0 skips the strict legs."""
import gc
import json
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
from aadff.focal_stack import StackPipeline, render_focal_stack_m1
from aadff.synth import synth_depth_mm, synth_rgb
from deeplens.optics import Lensgroup

H = W = 1024
S = 10
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
depth = synth_depth_mm(H, W, seed=5678)
dbar, fds = -float(depth.mean()), [float(f) for f in -np.linspace(depth.min(), depth.max(), S)]
img = torch.from_numpy(synth_rgb(H, W, seed=1234))[None].to("cuda:0")
LP = os.path.join(REPO, "lenses", "rf50mm", "lens.json")
res = {}


def seq(parity, steps):
    lens = Lensgroup(LP, sensor_res=(H, W), device="cuda:0", parity=parity)
    for i in range(4):
        render_focal_stack_m1(lens, img, dbar, fds, 11, 11, 2048)
    gc.collect()
    torch.cuda.synchronize()
    ts = []
    for i in range(steps):
        t0 = time.perf_counter()
        render_focal_stack_m1(lens, img, dbar, fds, 11, 11, 2048)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print(f"{parity} sequential: mean {ts.mean():.3f} p50 {np.median(ts):.3f} max {ts.max():.3f} ms per stack", flush=True)
    print("   segments:", getattr(lens, "_strict_timing", None), flush=True)
    print("   stats:", strict_stack.StrictCounts.of(lens).stats, flush=True)
    res[f"{parity}_sequential_ms"] = {"mean": round(float(ts.mean()), 4), "p50": round(float(np.median(ts)), 4), "max": round(float(ts.max()), 4)}


def pipe(parity, depth_n, steps):
    make = lambda: Lensgroup(LP, sensor_res=(H, W), device="cuda:0", parity=parity)
    p = strict_stack.StrictPipeline(make, depth=depth_n)
    for l in p.lenses:
        render_focal_stack_m1(l, img, dbar, fds, 11, 11, 2048)
    futs = [p.submit(img, dbar, fds) for _ in range(2 * depth_n)]
    [f.result() for f in futs]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    futs = [p.submit(img, dbar, fds) for _ in range(steps)]
    for f in futs:
        f.result()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{parity} pipeline depth {depth_n}: {ms:.3f} ms per stack", flush=True)
    res[f"{parity}_pipeline_depth{depth_n}_ms"] = round(ms, 4)
    p.close()


seq("edge", STEPS)
if os.environ.get("PROBE_CPROFILE") == "1":
    import cProfile
    import pstats
    lens = Lensgroup(LP, sensor_res=(H, W), device="cuda:0", parity="edge")
    for i in range(4):
        render_focal_stack_m1(lens, img, dbar, fds, 11, 11, 2048)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(STEPS):
        render_focal_stack_m1(lens, img, dbar, fds, 11, 11, 2048)
    torch.cuda.synchronize()
    pr.disable()
    print(f"--- cProfile of {STEPS} sequential edge stacks (cumulative) ---")
    pstats.Stats(pr, stream=sys.stdout).sort_stats("cumulative").print_stats(45)
    print("--- by tottime ---")
    pstats.Stats(pr, stream=sys.stdout).sort_stats("tottime").print_stats(30)
for d in [int(x) for x in os.environ.get("PROBE_DEPTHS", "2,4").split(",")]:
    pipe("edge", d, STEPS)
if os.environ.get("PROBE_STRICT", "1") != "0":
    seq("strict", max(8, STEPS // 2))
    pipe("strict", 4, max(8, STEPS // 2))
# the fast path on the same box
lens = Lensgroup(LP, sensor_res=(H, W), device="cuda:0")
sp = StackPipeline(lens, S, H, W, depth=2)
for i in range(5):
    sp.render(lens, img, dbar, fds, inputs_ready=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(100):
    sp.render(lens, img, dbar, fds, inputs_ready=True)
torch.cuda.synchronize()
res["fast_two_streams_ms"] = round((time.perf_counter() - t0) / 100 * 1e3, 4)
print(f"fast path, two streams: {res['fast_two_streams_ms']:.3f} ms per stack")
print(json.dumps(res))
