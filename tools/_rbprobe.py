import ctypes as C, os, sys
sys.path[:0] = ["/root/repo", "/root/repo/aberration-aware-depth-from-focus_amd"]
import numpy as np, torch
from aadff import _abi
lib = _abi.load_library(); dev = torch.device("cuda:0"); p = lambda t: C.c_void_p(t.data_ptr()); st = _abi.stream_ptr(dev)
H = W = 1024
for ks in (3, 5, 7):
    for G in (7, 11, 5):
      for S in (1, 10):
        img = torch.rand(1, 3, H, W, device=dev); maps = torch.rand(S, 3, G * ks, G * ks, device=dev) / (ks * ks); out = torch.empty(1, 3, S, H, W, device=dev)
        row = []
        ref = None
        for rb in ("", "24", "32", "48"):
            os.environ.pop("AADFF_CONV_BLKW", None); os.environ.pop("AADFF_CONV_BLKW_RB", None)
            if rb:
                os.environ["AADFF_CONV_BLKW"] = "1"; os.environ["AADFF_CONV_BLKW_RB"] = rb
            f = lambda: lib.aadff_render_psf_map_stack(p(img), p(maps), p(out), 1, 3, S, H, W, G, ks, st)
            for _ in range(3): f()
            torch.cuda.synchronize()
            if ref is None: ref = out.clone()
            else: assert (out - ref).abs().max().item() < 1e-5
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
            for r in range(5):
                e0.record()
                for _ in range(20): f()
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
            row.append(float(np.median(ts)))
        print(f"ks {ks} grid {G} S {S}: default / blkw RB 24 / 32 / 48 = " + " / ".join(f"{v:.1f}" for v in row), flush=True)
