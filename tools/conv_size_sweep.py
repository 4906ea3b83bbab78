#!/usr/bin/env python3
"""Lone-slice render_psf_map (block-GEMM kernel) over image sizes: is the launch a fixed chain of latencies or a rate?
python tools/conv_size_sweep.py"""
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
st = _abi.stream_ptr(dev)
p = lambda t: C.c_void_p(t.data_ptr())
G, KS = 11, 11
for n in (128, 256, 512, 768, 1024, 1536, 2048, 4096):
    img = torch.rand(1, 3, n, n, device=dev)
    maps = torch.rand(3, G * KS, G * KS, device=dev) / 121
    out = torch.empty(1, 3, n, n, device=dev)
    f = lambda: lib.aadff_render_psf_map(p(img), p(maps), p(out), 1, 3, n, n, G, KS, st)
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    ts = []
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    t = float(np.median(ts))
    print(f"{n:5d}^2: {t:7.1f} us   {24 * n * n / (t * 1e-6) / 1e12:5.2f} TB/s algorithmic ({24 * n * n / (t * 1e-6) / 8e12:.3f} of 8 TB/s)   workgroups {3 * 121 * -(-(-(-n // G)) // 24) * -(-(-(-n // G)) // 96)}", flush=True)
