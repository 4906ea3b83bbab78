#!/usr/bin/env python3
"""A/B of the stack convolution's producer / consumer form (AADFF_CONV_PC=1) against the default: bit-equality of the
10-slice 1024^2 bench launch and of a ragged case, and the launch time of both (the switch is read once per process, so
each arm runs in a child process).  python tools/conv_pc_check.py"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, ctypes as C
sys.path[:0] = [%r, %r]
import numpy as np, torch
from aadff import _abi
from aadff.synth import synth_rgb
lib = _abi.load_library(); dev = torch.device("cuda:0"); st = _abi.stream_ptr(dev); p = lambda t: C.c_void_p(t.data_ptr())
res = {}
for name, (H, W, S, G) in {"bench": (1024, 1024, 10, 11), "ragged": (201, 333, 7, 3)}.items():
    rng = np.random.Generator(np.random.PCG64(3))
    img = torch.from_numpy(rng.random((1, 3, H, W), dtype=np.float32)).to(dev) * 3 - 1
    maps = torch.from_numpy(rng.random((S, 3, G * 11, G * 11), dtype=np.float32)).to(dev) / 121
    out = torch.full((1, 3, S, H, W), float("nan"), device=dev)
    f = lambda: lib.aadff_render_psf_map_stack(p(img), p(maps), p(out), 1, 3, S, H, W, G, 11, st)
    assert f() == 0
    torch.cuda.synchronize()
    res[name] = out.cpu().numpy().copy()
    if name == "bench":
        for _ in range(20): f()
        ts = []
        for r in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        print("launch us: median %%.1f min %%.1f" %% (np.median(ts), min(ts)), flush=True)
np.savez(sys.argv[1], **res)
''' % (REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd"))


def main():
    import numpy as np
    outs = {}
    for pc in ("0", "1"):
        path = f"/tmp/conv_pc_{pc}.npz"
        env = dict(os.environ, AADFF_CONV_PC=pc)
        print(f"AADFF_CONV_PC={pc}", flush=True)
        subprocess.run([sys.executable, "-c", CHILD, path], env=env, check=True)
        outs[pc] = np.load(path)
    for k in outs["0"].files:
        d = np.abs(outs["0"][k] - outs["1"][k])
        print(f"{k}: max |default - producer/consumer| = {np.nanmax(d):.3e}, NaNs {int(np.isnan(outs['1'][k]).sum())}")


if __name__ == "__main__":
    main()
