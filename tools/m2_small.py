#!/usr/bin/env python3
"""A small M2 workload for counter passes that are slow on the full stack (L2 / L1 counters replay the kernel): PSFNet.render_stack on a
512 x 512 image, 4 slices - 1/10 of the bench launch, same kernel, same per-pixel traffic."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import torch

from aadff.focal_stack import render_focal_stack_m2
from aadff.synth import mlp_state_dict, synth_depth_mm, synth_rgb
from deeplens.psfnet import PSFNet

H = W = 512
dev = torch.device("cuda:0")
net = PSFNet(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(H, W), kernel_size=11, device=dev)
net.psfnet.load_state_dict({k: torch.from_numpy(v) for k, v in mlp_state_dict(seed=4321).items()})
img = torch.from_numpy(synth_rgb(H, W, seed=1234))[None].to(dev)
depth_m = (torch.from_numpy(synth_depth_mm(H, W, seed=5678))[None, None] / 1e3).to(dev)
for _ in range(3):
    render_focal_stack_m2(net, img, depth_m, 4)
torch.cuda.synchronize()
print("pixels per launch", 4 * H * W)
