#!/usr/bin/env python3
"""A strict trace kernel on IDENTICAL device inputs, launched again and again beside a load on a second stream: does any ray come out
different from a quiet launch?  FORMS=fused (aadff_trace_rays_strict_fused, packed float32) , batched (one launch pair per surface,
scalar); LOADS=none,conv,conv1,agg0..3,matmul (see tools/concurrency_isa_probe.py; conv1 = the lone-slice convolution); LAST=k traces
surfaces [0, k) only.  Round 6, before the fix (profiles/r06_concurrency_probe_grid.txt): fused beside conv 31/200, conv1 198/200,
agg0 200/200 - always odd rays (the second ray of a lane), only from the first aspheric surface on; batched 0.  After building the
strict units with -fno-slp-vectorize: 0 of 200 everywhere.

    python tools/concurrency_kernel_probe.py [launches]"""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd"), os.path.join(REPO, "tests")]
import numpy as np, torch
from aadff import _abi, strict_stack as ss
from deeplens.optics import Lensgroup
from test_gpu_margins import _psf_level_inputs
DEV = "cuda:0"; dev = torch.device(DEV)
lens = Lensgroup(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(512, 512), device=DEV, parity="strict")
S, grid, spp = 4, 5, 1024
a = _psf_level_inputs(lens, S, grid, spp, seed=11)
N, B, n_surf = a["N"], a["B"], len(lens.surfaces)
n = spp * N
o0, d0, r0 = (torch.empty(B, n, 3, device=dev), torch.empty(B, n, 3, device=dev), torch.empty(B, n, device=dev))
flag = torch.zeros(1, dtype=torch.int32, device=dev)
scratch = ss._trace(o0, d0, r0, n, B, a["tabs"], len(a["wv"]), n_surf, a["bt_main"], True, flag, dev, a["points"], a["pset"], a["pm"], N, a["zs"])
cnt = ss._masks_to_counts(scratch, B)
pred = torch.from_numpy(np.ascontiguousarray(cnt, dtype=np.int32)).to(dev)
LAST = int(os.environ.get("LAST", n_surf))
def fused():
    o1, d1, r1 = torch.empty_like(o0), torch.empty_like(d0), torch.empty_like(r0)
    bits = torch.empty((B, 2, _abi.MAX_SURF), dtype=torch.int32, device=dev)
    _abi.call("aadff_trace_rays_strict_fused", _abi.ptr(o1), _abi.ptr(d1), _abi.ptr(r1), n, B, _abi.ptr(a["tab_dev"]), len(a["wv"]), n_surf,
              _abi.ptr(a["bt_main"]), _abi.ptr(a["points"]), _abi.ptr(a["pset"]), _abi.ptr(a["pm"]), N, 0, LAST, 1, _abi.ptr(a["zs"]),
              _abi.ptr(pred), _abi.ptr(bits), 0, 0, None, None, None, _abi.stream_ptr(dev))
    return o1, d1, r1
def batched():
    o1, d1, r1 = torch.empty_like(o0), torch.empty_like(d0), torch.empty_like(r0)
    ss._trace(o1, d1, r1, n, B, a["tabs"], len(a["wv"]), n_surf, a["bt_main"], True, flag, dev, a["points"], a["pset"], a["pm"], N, a["zs"])
    return o1, d1, r1
img_big = torch.rand(1, 3, 1024, 1024, device=dev)
psf = torch.rand(10, 3, 121, 121, device=dev); psf /= psf.sum()
out_big = torch.empty(1, 3, 10, 1024, 1024, device=dev); out1 = torch.empty(1, 3, 1024, 1024, device=dev)
side = torch.cuda.Stream()
lib = C.CDLL(os.path.join(REPO, "tools", "conc_victims.so"))
lib.aggressor_launch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
agg_out = torch.empty(8192 * 256, device=dev)
A = torch.rand(4096, 4096, device=dev, dtype=torch.float16)
NIT = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for name, fn in (("fused", fused), ("batched", batched)):
    if name not in os.environ.get("FORMS", "fused,batched"): continue
    for kind in os.environ.get("LOADS", "none,conv").split(","):
        ref = fn(); torch.cuda.synchronize()
        bad = 0; shown = 0
        for it in range(NIT):
            if kind.startswith("agg"):
                k = int(kind[3:])
                for _ in range(4): lib.aggressor_launch(k, agg_out.data_ptr(), 8192, [5000, 1500, 10000, 5000][k], C.c_void_p(side.cuda_stream))
            if kind == "matmul":
                with torch.cuda.stream(side):
                    for _ in range(4): mm = A @ A
            if kind == "conv1":
                with torch.cuda.stream(side):
                    for _ in range(30): _abi.call("aadff_render_psf_map", _abi.ptr(img_big), _abi.ptr(psf[0]), _abi.ptr(out1), 1, 3, 1024, 1024, 11, 11, C.c_void_p(side.cuda_stream))
            if kind == "conv":
                with torch.cuda.stream(side):
                    for _ in range(6): _abi.call("aadff_render_psf_map_stack", _abi.ptr(img_big), _abi.ptr(psf), _abi.ptr(out_big), 1, 3, 10, 1024, 1024, 11, 11, C.c_void_p(side.cuda_stream))
            got = fn(); torch.cuda.synchronize()
            dif = (got[0].view(torch.int32) != ref[0].view(torch.int32)).any(-1) | (got[1].view(torch.int32) != ref[1].view(torch.int32)).any(-1) | (got[2].view(torch.int32) != ref[2].view(torch.int32))
            nd = int(dif.sum())
            if nd:
                bad += 1
                if shown < int(os.environ.get('SHOW', 2)):
                    shown += 1
                    idx = torch.nonzero(dif)
                    rays = [(int(i[0]), int(i[1])) for i in idx[:12]]
                    b0, r_0 = rays[0]
                    print(f"  {name}/{kind} it {it}: {nd} rays differ; first (batch, ray [= sample * N + point]):", rays,
                          "| ray", r_0, "o ref", ref[0][b0, r_0].tolist(), "got", got[0][b0, r_0].tolist(), "ra", float(ref[2][b0, r_0]), float(got[2][b0, r_0]), flush=True)
        print(f"{name} trace (surfaces [0, {LAST})) beside load {kind}: {bad} of {NIT} launches differ from a quiet launch", flush=True)
