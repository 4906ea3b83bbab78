#!/usr/bin/env python3
"""Eager (no graph, default stream) runs of the fused fit step for rocprofv3 --kernel-trace --stats."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np, torch
from aadff.mlp_fit import FusedFit
from aadff.synth import mlp_state_dict
from deeplens.psfnet_arch import MLP
dev = torch.device("cuda:0")
HL, BS, HID = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
net = MLP(4, 121, HID, HL).to(dev)
rng = np.random.Generator(np.random.PCG64(1))
inp = torch.from_numpy(rng.random((BS, 4), dtype=np.float32)).to(dev)
psf = torch.from_numpy(rng.random((BS, 121), dtype=np.float32)).to(dev); psf /= psf.sum(-1, keepdim=True)
fit = FusedFit(net, 1e-3, 1000, BS, dev)
fit.inp.copy_(inp); fit.psf.copy_(psf)
for _ in range(20):
    fit._body()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import aadff._abi as _abi, ctypes as C
st = _abi.stream_ptr(dev)
e0.record()
for _ in range(50):
    _abi.call("aadff_fit_chain", C.byref(fit.net_desc), None, _abi.ptr(fit.scal), C.c_float(1e-3), 1000, C.c_float(0.9), C.c_float(0.999), C.c_float(0.01), st)
e1.record(); torch.cuda.synchronize()
print(f"hidden layers {HL} width {HID} batch {BS}: chain + dW = {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call (back to back)")
