import os, sys, ctypes as C
sys.path[:0] = ['/root/repo', '/root/repo/aberration-aware-depth-from-focus_amd']
import numpy as np, torch
from aadff import _abi
lib = _abi.load_library()
dev = torch.device("cuda:0")
H = W = 1024
for ks in (9, 11):
    for S in (1, 2, 3, 10):
        G = 11
        img = torch.rand(1, 3, H, W, device=dev)
        maps = torch.rand(S, 3, G * ks, G * ks, device=dev) / (ks * ks)
        out = torch.empty(1, 3, S, H, W, device=dev)
        st = _abi.stream_ptr(dev)
        p = lambda t: C.c_void_p(t.data_ptr())
        res = {}
        for path in ("", "toeplitz"):
            if S > 4 and ks == 11 and path == "":
                pass
            if path: os.environ["AADFF_CONV_PATH"] = path
            else: os.environ.pop("AADFF_CONV_PATH", None)
            f = lambda: lib.aadff_render_psf_map_stack(p(img), p(maps), p(out), 1, 3, S, H, W, G, ks, st)
            for _ in range(5): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for r in range(5):
                e0.record()
                for _ in range(20): f()
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 20 * 1e3)
            res[path or "default"] = round(float(np.median(ts)), 1)
        print(f"ks {ks} S {S}: {res}", flush=True)
os.environ.pop("AADFF_CONV_PATH", None)
