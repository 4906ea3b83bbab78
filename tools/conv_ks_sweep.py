#!/usr/bin/env python3
"""Patch-PSF convolution over the kernel sizes the reference uses: ks 7 / 11 / 15 / 21 / 31 / 51 at grid 7 and 11,
1024^2, one slice (`render_psf_map` itself) and a 10-slice stack: us per launch (back to back, median of 5 x 20), useful TFLOP/s
(2 ks^2 flop per pixel and channel), fraction of the HBM roofline on 24 B/pixel/slice, for the default (matrix-core) path and with
AADFF_CONV_PATH=valu (packed-FMA kernel for ks 13 / 15 / 21, generic LDS kernel otherwise)."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np
import torch

from aadff import _abi

lib = _abi.load_library()
dev = torch.device("cuda:0")
H = W = 1024
p = lambda t: C.c_void_p(t.data_ptr())
st = _abi.stream_ptr(dev)
print(f"{'ks':>3} {'grid':>4} {'S':>2} | {'default us':>10} {'TFLOP/s':>8} {'roofline':>8} | {'valu us':>9} {'TFLOP/s':>8} | speed-up")
for ks in (7, 11, 15, 21, 31, 51):
    for G in (7, 11):
        for S in (1, 10):
            img = torch.rand(1, 3, H, W, device=dev)
            maps = torch.rand(S, 3, G * ks, G * ks, device=dev) / (ks * ks)
            out = torch.empty(1, 3, S, H, W, device=dev)
            res = {}
            for path in ("", "valu"):
                if path:
                    os.environ["AADFF_CONV_PATH"] = path
                else:
                    os.environ.pop("AADFF_CONV_PATH", None)
                f = lambda: lib.aadff_render_psf_map_stack(p(img), p(maps), p(out), 1, 3, S, H, W, G, ks, st)
                reps = 20 if ks <= 21 else 5
                for _ in range(3):
                    f()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ts = []
                for r in range(5):
                    e0.record()
                    for _ in range(reps):
                        f()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / reps * 1e3)
                res[path or "default"] = float(np.median(ts))
            os.environ.pop("AADFF_CONV_PATH", None)
            fl = 2.0 * ks * ks * 3 * H * W * S
            d, v = res["default"], res["valu"]
            print(f"{ks:3d} {G:4d} {S:2d} | {d:10.1f} {fl / d / 1e6:8.1f} {24.0 * H * W * S / (d * 1e-6) / 8e12:8.3f} | {v:9.1f} {fl / v / 1e6:8.1f} | {v / d:5.2f}x", flush=True)
