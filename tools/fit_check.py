#!/usr/bin/env python3
"""Fused bf16 fit step (aadff/mlp_fit.py) against torch autograd on the same batch: gradients, predictions, a few steps."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np, torch, copy
from aadff.mlp_fit import FusedFit, supported
from aadff.synth import mlp_state_dict
from deeplens.psfnet_arch import MLP
dev = torch.device("cuda:0")
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
net = MLP(4, 121, 256, 8).to(dev)
net.load_state_dict({k: torch.from_numpy(v) for k, v in mlp_state_dict(seed=4321).items()})
ref = copy.deepcopy(net)
rng = np.random.Generator(np.random.PCG64(1))
inp = torch.from_numpy(rng.random((128, 4), dtype=np.float32)).to(dev); inp[:, :2] = inp[:, :2] * 2 - 1
psf = torch.from_numpy(rng.random((128, 121), dtype=np.float32)).to(dev); psf /= psf.sum(-1, keepdim=True)
assert supported(net, 128)
fit = FusedFit(net, 1e-3, 100, 128, dev)
grad, fpred = fit.gradients(inp, psf)
with torch.autocast("cuda", dtype=torch.bfloat16):
    pred = ref(inp)
loss = torch.nn.functional.mse_loss(pred.float(), psf); loss.backward()
print("pred rel", rel(fpred, pred.float()))
lin = [m for m in ref.net if isinstance(m, torch.nn.Linear)]
for l, m in enumerate(lin):
    gw = grad[fit.w_off[l]:fit.w_off[l] + m.weight.numel()].view_as(m.weight)
    gb = grad[fit.b_off[l]:fit.b_off[l] + m.bias.numel()]
    print(f"layer {l}: dW rel {rel(gw, m.weight.grad):.3e}  db rel {rel(gb, m.bias.grad):.3e}")
# a few optimisation steps vs torch AdamW (fp32 reference, no autocast)
ref2 = copy.deepcopy(ref); ref2.zero_grad()
opt = torch.optim.AdamW(ref2.parameters(), 1e-3); sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=100, eta_min=0)
for it in range(10):
    p = fit(inp, psf)
    pr = ref2(inp); opt.zero_grad(); l2 = torch.nn.functional.mse_loss(pr, psf); l2.backward(); opt.step(); sch.step()
    if it in (0, 4, 9):
        torch.cuda.synchronize()
        print(f"step {it}: loss fused {float(((p - psf) ** 2).mean()):.6e} torch-fp32 {float(l2):.6e}")
print("weights rel after 10 steps:", [round(rel(a.detach(), b.detach()), 4) for a, b in zip(net.parameters(), ref2.parameters())][:6])
import time, collections
import aadff._abi as _abi
evs = []; orig = _abi.call
def timed(name, *a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(name, *a); e1.record()
    evs.append((name, (), e0, e1)); return r
for rep in range(3):
    evs.clear(); _abi.call = timed; fit._body(); _abi.call = orig; torch.cuda.synchronize()
for n, dims, e0, e1 in evs[:6]: print(f'{n:24s} {e0.elapsed_time(e1) * 1e3:8.1f} us (eager, includes host launch time)')
for _ in range(5): fit(inp, psf)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): fit(inp, psf)
torch.cuda.synchronize(); print(f"fused step: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
