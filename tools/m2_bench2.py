#!/usr/bin/env python3
"""M2 timing: PSFNet.render at 1024^2, fused kernel vs torch fp32 MLP + gather vs bf16."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np, torch
from aadff.synth import mlp_state_dict, synth_depth_mm, synth_rgb
from deeplens.psfnet import PSFNet
dev = "cuda:0"; H = W = 1024
net = PSFNet(os.path.join(REPO, "lenses/rf50mm/lens.json"), sensor_res=(H, W), kernel_size=11, device=dev)
net.psfnet.load_state_dict({k: torch.from_numpy(v) for k, v in mlp_state_dict(seed=4321).items()})
img = torch.from_numpy(synth_rgb(H, W))[None].to(dev)
depth = -torch.from_numpy(synth_depth_mm(H, W))[None, None].to(dev)
fd = torch.tensor([-1500.0], device=dev)
outs = {}
for mode in ("fp32", "torch", "bf16"):
    net.mlp_precision = mode
    for _ in range(2): o = net.render(img, depth, fd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): o = net.render(img, depth, fd)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    outs[mode] = o
    print(f"{mode:6s} {dt*1e3:8.3f} ms  {H*W/1e6/dt:8.1f} MP/s", flush=True)
ref = outs["torch"].double()
for m in ("fp32", "bf16"):
    print(m, "rel-L2 vs torch fp32:", float((outs[m].double() - ref).norm() / ref.norm()))
