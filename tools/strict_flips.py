#!/usr/bin/env python3
"""Which batch-wide Newton counts of the strict psf_map level are mispredicted from stack to stack: 64 bench-sized stacks, every failing
(batch, phase, surface) with its predicted -> true counts.  Round 6: all of them are the CHIEF count at surface 9 (second aspheric) of
8 batches flipping 4 <-> 5 - what aadff_strict_psf_points_alt renders under both counts.
Run with AADFF_STRICT_ALT=0 to see the flips (with the two-variant jobs on there is nothing left to mispredict)."""
import os, sys, collections
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np, torch
from aadff import strict_stack
from aadff.focal_stack import render_focal_stack_m1
from aadff.synth import synth_depth_mm, synth_rgb
from deeplens.optics import Lensgroup
H = W = 1024; S = 10
depth = synth_depth_mm(H, W, seed=5678)
dbar, fds = -float(depth.mean()), [float(f) for f in -np.linspace(depth.min(), depth.max(), S)]
img = torch.from_numpy(synth_rgb(H, W, seed=1234))[None].to("cuda:0")
lens = Lensgroup(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(H, W), device="cuda:0", parity="strict")
log = []
orig = strict_stack.check_counts
def spy(any_bits, pred, curved, order):
    ok, fix = orig(any_bits, pred, curved, order)
    p = np.asarray(pred)
    if p.ndim == 3 and p.shape[0] == 30 and p.shape[1] == 2:
        bad = np.argwhere(~ok)
        log.append([(int(b), int(ph), [(int(s), int(p[b, ph, s]), int(fix[b, ph, s])) for s in np.nonzero(p[b, ph] != fix[b, ph])[0]]) for b, ph in bad])
    return ok, fix
strict_stack.check_counts = spy
torch.manual_seed(0)
for i in range(64):
    render_focal_stack_m1(lens, img, dbar, fds, 11, 11, 2048)
torch.cuda.synchronize()
first = log[4:]
flat = [(b, ph, s, n0, n1) for st in first for (b, ph, lst) in st for (s, n0, n1) in lst]
print("stacks", len(first), "level-3 launches with a failing batch:", sum(1 for st in first if st), "failing (batch, phase) per launch: mean %.2f" % np.mean([len(st) for st in first]))
c = collections.Counter((b, ph, s) for b, ph, s, _, _ in flat)
print("distinct (batch, phase 0=chief 1=main, surface) that ever failed:", len(c))
for k, v in c.most_common(30):
    vals = collections.Counter((n0, n1) for b, ph, s, n0, n1 in flat if (b, ph, s) == k)
    print("  batch %2d (slice %d, lambda %d) phase %d surface %2d: %2d times, (predicted -> true): %s" % (k[0], k[0] // 3, k[0] % 3, k[1], k[2], v, dict(vals)))
print("chief failures", sum(1 for x in flat if x[1] == 0), "main failures", sum(1 for x in flat if x[1] == 1))
