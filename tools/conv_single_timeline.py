#!/usr/bin/env python3
"""Per-workgroup timeline of a LONE-SLICE render_psf_map launch (block-GEMM kernel, or the Toeplitz MFMA kernel with
AADFF_CONV_PATH=toeplitz; instrumentation build
csrc/libaadff_sbtrace.so): every workgroup stamps the 100 MHz real-time counter at its start (0), when its image loads have
arrived (1), when the tile is in LDS (2), when its tap rows are in LDS (3), after its last MFMA (4) and when its stores have
retired (5).  Prints when workgroups start and how long each phase takes inside the full launch.

    python tools/conv_single_timeline.py [--json out]"""
import argparse
import ctypes as C
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np
import torch

from aadff import _abi
from aadff.synth import synth_rgb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(os.path.dirname(_abi.LIB_PATH), "libaadff_sbtrace.so"))
    ap.add_argument("--json", default=None)
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--ks", type=int, default=11)        # 13 .. 21: conv_psf_map_blkw_kernel (stamp 3 = T fragments built)
    ap.add_argument("--grid", type=int, default=11)
    a = ap.parse_args()
    lib = _abi.load_library(a.lib)
    dev = torch.device("cuda:0")
    H = W = 1024
    G, KS = a.grid, a.ks
    img = torch.from_numpy(synth_rgb(H, W))[None].to(dev)
    rng = np.random.Generator(np.random.PCG64(3))
    maps = torch.from_numpy(rng.random((3, G * KS, G * KS), dtype=np.float32)).to(dev) / (KS * KS)
    out = torch.empty((1, 3, H, W), device=dev)
    n_wg = 16384                                      # upper bound: Toeplitz form 3 267 workgroups (32 x 32 tiles), block-GEMM form 1 452 (24 x 96 bands)
    buf = torch.zeros(n_wg * 8, dtype=torch.int64, device=dev)
    lib.aadff_sb_trace_buffer.argtypes = [C.c_void_p]
    st = _abi.stream_ptr(dev)
    call = lambda: lib.aadff_render_psf_map(C.c_void_p(img.data_ptr()), C.c_void_p(maps.data_ptr()), C.c_void_p(out.data_ptr()), 1, 3, H, W, G, KS, st)
    assert lib.aadff_sb_trace_buffer(None) == 0
    for _ in range(30):
        assert call() == 0
    torch.cuda.synchronize()
    res = []
    for r in range(a.runs):
        buf.zero_()
        assert lib.aadff_sb_trace_buffer(C.c_void_p(buf.data_ptr())) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            call()
        e0.record()
        assert call() == 0
        e1.record()
        torch.cuda.synchronize()
        t = buf.cpu().numpy().reshape(n_wg, 8).astype(np.int64)
        t = t[t[:, 5] > 0]
        t0 = t[:, 0].min()
        us = lambda col: (t[:, col] - t0) / 100.0
        q = lambda v: [round(float(np.percentile(v, p)), 2) for p in (10, 50, 90)]
        res.append({"event_us": round(e0.elapsed_time(e1) * 1e3, 2), "workgroups": int(len(t)), "span_us": round(float(us(5).max()), 2),
                    "start_us_p10_50_90": q(us(0)), "start_max_us": round(float(us(0).max()), 2),
                    "loads_us": q(us(1) - us(0)), "tile_to_lds_us": q(us(2) - us(1)), "taps_us": q(np.maximum(us(3), us(2)) - us(2)),
                    "mfma_us": q(us(4) - np.maximum(us(3), us(2))), "stores_us": q(us(5) - us(4)), "life_us": q(us(5) - us(0)), "end_us_p10_50_90": q(us(5))})
    assert lib.aadff_sb_trace_buffer(None) == 0
    res.sort(key=lambda d: d["event_us"])
    med = res[len(res) // 2]
    print(json.dumps(med, indent=1))
    if a.json:
        json.dump({"median_run": med, "runs": res}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
