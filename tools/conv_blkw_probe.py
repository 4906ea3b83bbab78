#!/usr/bin/env python3
"""ks 13 .. 21: the block-GEMM kernel (AADFF_CONV_BLKW=1) against the wide Toeplitz kernel (=0) and the packed-FMA kernel
(AADFF_CONV_PATH=valu): max |difference| to the packed-FMA result on ragged shapes, and us per launch at 1024^2."""
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
p = lambda t: C.c_void_p(t.data_ptr())
st = _abi.stream_ptr(dev)


def run(img, maps, S, G, ks, path):
    os.environ.pop("AADFF_CONV_PATH", None)
    os.environ.pop("AADFF_CONV_BLKW", None)
    if path == "valu":
        os.environ["AADFF_CONV_PATH"] = "valu"
    else:
        os.environ["AADFF_CONV_BLKW"] = "1" if path == "blkw" else "0"
    B, Cn, H, W = img.shape
    out = torch.full((B, Cn, S, H, W), float("nan"), device=dev)
    rc = lib.aadff_render_psf_map_stack(p(img), p(maps), p(out), B, Cn, S, H, W, G, ks, st)
    assert rc == 0, lib.aadff_last_error()
    return out


torch.manual_seed(0)
worst = 0.0
for ks in (13, 15, 17, 19, 21):
    for (B, Cn, H, W, G, S) in ((1, 3, 1024, 1024, 7, 1), (2, 1, 257, 389, 3, 2), (1, 3, 64, 48, 1, 1), (1, 2, 100, 333, 5, 3), (1, 1, 31, 23, 1, 1)):
        if ks // 2 >= min(H, W):
            continue
        img = torch.randn(B, Cn, H, W, device=dev) * 3
        maps = torch.rand(S, Cn, G * ks, G * ks, device=dev) / (ks * ks)
        ref = run(img, maps, S, G, ks, "valu").double()
        for path in ("blkw", "toep"):
            got = run(img, maps, S, G, ks, path).double()
            assert not torch.isnan(got).any(), (ks, path, "unwritten outputs")
            d = float((got - ref).abs().max())
            worst = max(worst, d) if path == "blkw" else worst
            print(f"ks {ks} {(B, Cn, H, W, G, S)} {path}: max |d| vs packed-FMA {d:.2e}", flush=True)
print("worst blkw", worst)
H = W = 1024
print(f"{'ks':>3} {'grid':>4} {'S':>2} | {'blkw us':>8} {'toeplitz us':>11} {'valu us':>8} | valu/blkw  toeplitz/blkw")
for ks in (13, 15, 17, 19, 21):
    for G in (7, 11):
        for S in (1, 10):
            img = torch.rand(1, 3, H, W, device=dev)
            maps = torch.rand(S, 3, G * ks, G * ks, device=dev) / (ks * ks)
            out = torch.empty(1, 3, S, H, W, device=dev)
            res = {}
            for path in ("blkw", "toep", "valu"):
                os.environ.pop("AADFF_CONV_PATH", None)
                os.environ.pop("AADFF_CONV_BLKW", None)
                if path == "valu":
                    os.environ["AADFF_CONV_PATH"] = "valu"
                else:
                    os.environ["AADFF_CONV_BLKW"] = "1" if path == "blkw" else "0"
                f = lambda: lib.aadff_render_psf_map_stack(p(img), p(maps), p(out), 1, 3, S, H, W, G, ks, st)
                for _ in range(3):
                    f()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ts = []
                for r in range(5):
                    e0.record()
                    for _ in range(20):
                        f()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
                res[path] = float(np.median(ts))
            print(f"{ks:3d} {G:4d} {S:2d} | {res['blkw']:8.1f} {res['toep']:11.1f} {res['valu']:8.1f} | {res['valu'] / res['blkw']:6.2f}x {res['toep'] / res['blkw']:6.2f}x", flush=True)
