#!/usr/bin/env python3
"""Where the latency of ONE M1 stack goes (host call -> device idle): wall time of every host segment of
render_focal_stack_m1 (seeding, the MT19937 draws into the pinned block, each C-ABI launch call, the guard/flag
bookkeeping) and of the final synchronise, median over `--n` lone stacks, next to the kernels' own durations."""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np
import torch

from aadff import _abi, focal_stack as fs, sampling
from aadff.synth import synth_depth_mm, synth_rgb
from deeplens.optics import Lensgroup


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=60)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    H = W = 1024
    S = 10
    lens = Lensgroup(os.path.join(REPO, "lenses/rf50mm/lens.json"), sensor_res=(H, W), device=dev)
    img = torch.from_numpy(synth_rgb(H, W, seed=1234))[None].to(dev)
    depth = synth_depth_mm(H, W, seed=5678)
    dbar, fds = -float(depth.mean()), -np.linspace(depth.min(), depth.max(), S)
    plan = fs.StackPlan(lens, S, H, W)
    seg = {}
    now = time.perf_counter

    def wrap(name, fn):
        def w(*args, **kw):
            t = now()
            r = fn(*args, **kw)
            seg.setdefault(name, []).append(now() - t)
            return r
        return w

    orig_call = _abi.call
    _abi.call = lambda name, *args: wrap("launch " + name, orig_call)(name, *args)
    fs._abi.call = _abi.call
    sampling.HostSampler.rand_into = wrap("draws (MT19937 -> pinned block)", sampling.HostSampler.rand_into)
    plan.staged = wrap("guard event / flag mirror", plan.staged)
    for i in range(30):
        torch.manual_seed(i)
        fs.render_focal_stack_m1(lens, img, dbar, fds, plan=plan, update_lens=False)
    torch.cuda.synchronize()
    seg.clear()
    tot, seed, body, sync = [], [], [], []
    for i in range(a.n):
        torch.cuda.synchronize()
        t0 = now()
        torch.manual_seed(i)
        t1 = now()
        fs.render_focal_stack_m1(lens, img, dbar, fds, plan=plan, update_lens=False)
        t2 = now()
        torch.cuda.synchronize()
        t3 = now()
        tot.append(t3 - t0); seed.append(t1 - t0); body.append(t2 - t1); sync.append(t3 - t2)
    med = lambda v: float(np.median(v)) * 1e6
    print(f"latency (seed + host call + wait for the device), median of {a.n}: {med(tot):7.1f} us")
    print(f"  torch.manual_seed                                        {med(seed):7.1f} us")
    print(f"  render_focal_stack_m1 (host, returns after enqueue) {med(body):7.1f} us")
    for k, v in seg.items():
        per = len(v) / a.n
        print(f"    {k:<50s} {med(v):7.1f} us x {per:.2f} per stack")
    print(f"    other Python in the call                           {med(body) - sum(med(v) * len(v) / a.n for v in seg.values()):7.1f} us")
    print(f"  torch.cuda.synchronize (kernels still running)      {med(sync):7.1f} us")


if __name__ == "__main__":
    main()
