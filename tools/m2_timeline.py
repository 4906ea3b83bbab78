#!/usr/bin/env python3
"""Per-workgroup phase timeline of the fused PSF-network kernel (instrumentation build csrc/libaadff_pntrace.so): for the first
4096 workgroups of a 1024^2 slice - start, input stage, and per layer: k-loop, barrier wait, write-back, barrier wait; epilogue.
Prints the median microseconds per phase summed over the layers, and the life of a workgroup.

    python tools/m2_timeline.py [--json out]"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    lib = _abi.load_library(os.path.join(os.path.dirname(_abi.LIB_PATH), "libaadff_pntrace.so"))
    _abi._lib = lib
    from aadff.focal_stack import render_focal_stack_m2
    from aadff.synth import mlp_state_dict, synth_depth_mm, synth_rgb
    from deeplens.psfnet import PSFNet
    dev = torch.device("cuda:0")
    H = W = 1024
    net = PSFNet(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(H, W), kernel_size=11, device=dev)
    net.psfnet.load_state_dict({k: torch.from_numpy(v) for k, v in mlp_state_dict(seed=4321).items()})
    img = torch.from_numpy(synth_rgb(H, W))[None].to(dev)
    depth_m = (torch.from_numpy(synth_depth_mm(H, W))[None, None] / 1e3).to(dev)
    lib.aadff_pn_trace_buffer.argtypes = [C.c_void_p]
    for _ in range(2):
        render_focal_stack_m2(net, img, depth_m, 4)
    torch.cuda.synchronize()
    buf = torch.zeros(4096 * 64, dtype=torch.int64, device=dev)
    assert lib.aadff_pn_trace_buffer(C.c_void_p(buf.data_ptr())) == 0
    render_focal_stack_m2(net, img, depth_m, 4)
    torch.cuda.synchronize()
    assert lib.aadff_pn_trace_buffer(None) == 0
    t = buf.cpu().numpy().reshape(4096, 64).astype(np.int64)
    t = t[t[:, 60] > 0]
    us = lambda col: (t[:, col] - t[:, 0]) / 100.0
    L = 11
    k = sum(us(2 + 4 * l) - (us(1) if l == 0 else us(5 + 4 * (l - 1))) for l in range(L))
    b1 = sum(us(3 + 4 * l) - us(2 + 4 * l) for l in range(L))
    wb = sum(us(4 + 4 * l) - us(3 + 4 * l) for l in range(L))
    b2 = sum(us(5 + 4 * l) - us(4 + 4 * l) for l in range(L))
    ep = us(60) - us(5 + 4 * (L - 1))
    q = lambda v: [round(float(np.percentile(v, p)), 2) for p in (10, 50, 90)]
    per_layer_k = [round(float(np.median(us(2 + 4 * l) - (us(1) if l == 0 else us(5 + 4 * (l - 1))))), 2) for l in range(L)]
    res = {"workgroups": int(len(t)), "input_stage_us": q(us(1)), "k_loops_us_sum": q(k), "k_loop_us_per_layer_median": per_layer_k,
           "barrier_after_k_loop_us_sum": q(b1), "write_back_us_sum": q(wb), "barrier_after_write_back_us_sum": q(b2), "epilogue_us": q(ep),
           "life_us": q(us(60)),
           "mfma_us_of_a_workgroup_alone_on_its_cu": round(13488 * 16 / 4 / 2.4e3, 2)}
    print(json.dumps(res, indent=1))
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
