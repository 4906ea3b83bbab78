#!/usr/bin/env python3
"""Many M1 focal stacks back to back with two in flight on two HIP streams (aadff.focal_stack.StackPipeline).

    python examples/stack_pipeline_two_streams.py [n_stacks]

The pupil samples are drawn on the host in call order (the reference's RNG stream), so every stack equals the one a
single render_focal_stack_m1 call would give; only the GPU schedule changes: the VALU-bound PSF-grid kernel of stack i+1
runs beside the LDS/MFMA/HBM-bound convolution of stack i.  A stack's output belongs to its slot and is complete when
the returned event has fired; the slot is reused two calls later, so consume (or copy) it before that."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
from aadff.focal_stack import StackPipeline          # noqa: E402
from aadff.synth import synth_rgb                    # noqa: E402
from deeplens.optics import Lensgroup                # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
H = W = 1024
lens = Lensgroup(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(H, W), device=dev)
img = torch.from_numpy(synth_rgb(H, W))[None].to(dev)
focus = -np.linspace(500.0, 5000.0, 10)
means = torch.zeros(n, device=dev)
for depth in (1, 2):
    pipe = StackPipeline(lens, 10, H, W, depth=depth)
    torch.manual_seed(0)
    for i in range(20):                                                       # warm-up
        pipe.render(lens, img, -1500.0, focus)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        out, done = pipe.render(lens, img, -1500.0, focus)                    # out: [1, 3, 10, H, W] of this slot
        if done is not None:
            torch.cuda.current_stream().wait_event(done)                      # consumer: ordered behind the slot's stream
        means[i] = out.mean()                                                 # ... anything that reads the stack
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pipe.check_flags()
    print(f"{depth} stream(s): {n} stacks in {dt * 1e3:.1f} ms = {10 * H * W / 1e6 * n / dt:.0f} MP/s (mean of last stack {float(means[-1]):.6f})")
