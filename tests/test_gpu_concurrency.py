"""Strict kernels beside a busy second stream.  Round 6 found the strict (packed float32) trace kernels giving other results for a few
rays - always the second ray of a lane - whenever an MFMA kernel ran on another stream, our own convolution included; the cause is one
instruction form (v_pk_mul_f32 / v_pk_add_f32 with op_sel on the second source, produced by the SLP vectoriser), removed by building
the strict translation units with -fno-slp-vectorize (csrc/Makefile, tools/check_isa.py).  Before the fix this test failed in 198 of
200 launches (profiles/r06_concurrency_probe_grid.txt)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from aadff import _abi, strict_stack as ss                 # noqa: E402
from deeplens.optics import Lensgroup                     # noqa: E402
from test_gpu_margins import _psf_level_inputs, lp        # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_strict_trace_is_bit_stable_beside_the_convolution(repo_root):
    dev = torch.device(DEV)
    lens = Lensgroup(lp(repo_root), sensor_res=(512, 512), device=DEV, parity="strict")
    S, grid, spp = 4, 5, 1024
    a = _psf_level_inputs(lens, S, grid, spp, seed=11)
    N, B, n_surf = a["N"], a["B"], len(lens.surfaces)
    n = spp * N
    o0, d0, r0 = (torch.empty(B, n, 3, device=dev), torch.empty(B, n, 3, device=dev), torch.empty(B, n, device=dev))
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = ss._trace(o0, d0, r0, n, B, a["tabs"], len(a["wv"]), n_surf, a["bt_main"], True, flag, dev, a["points"], a["pset"], a["pm"], N, a["zs"])
    pred = torch.from_numpy(np.ascontiguousarray(ss._masks_to_counts(scratch, B), dtype=np.int32)).to(dev)

    def fused():
        o1, d1, r1 = torch.empty_like(o0), torch.empty_like(d0), torch.empty_like(r0)
        bits = torch.empty((B, 2, _abi.MAX_SURF), dtype=torch.int32, device=dev)
        _abi.call("aadff_trace_rays_strict_fused", _abi.ptr(o1), _abi.ptr(d1), _abi.ptr(r1), n, B, _abi.ptr(a["tab_dev"]), len(a["wv"]), n_surf,
                  _abi.ptr(a["bt_main"]), _abi.ptr(a["points"]), _abi.ptr(a["pset"]), _abi.ptr(a["pm"]), N, 0, n_surf, 1, _abi.ptr(a["zs"]),
                  _abi.ptr(pred), _abi.ptr(bits), 0, 0, None, None, None, _abi.stream_ptr(dev))
        return torch.cat([o1.view(torch.int32).reshape(-1), d1.view(torch.int32).reshape(-1), r1.view(torch.int32).reshape(-1), bits.reshape(-1)])

    def batched():
        o1, d1, r1 = torch.empty_like(o0), torch.empty_like(d0), torch.empty_like(r0)
        ss._trace(o1, d1, r1, n, B, a["tabs"], len(a["wv"]), n_surf, a["bt_main"], True, flag, dev, a["points"], a["pset"], a["pm"], N, a["zs"])
        return torch.cat([o1.view(torch.int32).reshape(-1), d1.view(torch.int32).reshape(-1), r1.view(torch.int32).reshape(-1)])

    img = torch.rand(1, 3, 1024, 1024, device=dev)
    psf = torch.rand(3, 121, 121, device=dev)
    psf /= psf.sum()
    out = torch.empty(1, 3, 1024, 1024, device=dev)
    side = torch.cuda.Stream()
    for name, fn, launches in (("fused", fused, 60), ("per-surface", batched, 20)):
        ref = fn()
        torch.cuda.synchronize()
        differing = 0
        for _ in range(launches):
            with torch.cuda.stream(side):                 # the lone-slice MFMA convolution, back to back, while the trace runs
                for _ in range(30):
                    _abi.call("aadff_render_psf_map", _abi.ptr(img), _abi.ptr(psf), _abi.ptr(out), 1, 3, 1024, 1024, 11, 11, C.c_void_p(side.cuda_stream))
            got = fn()
            torch.cuda.synchronize()
            differing += int(not torch.equal(got, ref))
        assert differing == 0, f"{name} strict trace: {differing} of {launches} launches beside the convolution differ from a quiet launch"
