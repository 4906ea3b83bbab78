#!/usr/bin/env python3
"""The call sequence of the reference's 0_warm_up.py (load lens + PSFNet, `analysis()`, one `render`, save PNGs) and of
BASELINE.json configs[0] (rf50mm, 256x256 all-in-focus + depth, 5-slice focal stack), on seeded synthetic RGB-D input and
procedural MLP weights: the reference's image, depth map and checkpoint are files this repository does not ship.

    PYTHONPATH=aberration-aware-depth-from-focus_amd python examples/0_warm_up_synthetic.py [out_dir]
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]

import torch
from deeplens.psfnet import PSFNet
from deeplens.utils import save_image
from aadff.focal_stack import render_focal_stack_m1, render_focal_stack_m2
from aadff.synth import mlp_state_dict, synth_depth_mm, synth_rgb

if __name__ == "__main__":
    out_dir = sys.argv[1] if len(sys.argv) > 1 else "."
    os.makedirs(out_dir, exist_ok=True)

    # Load lens and PSFNet (0_warm_up.py:9-11)
    psfnet = PSFNet(filename=os.path.join(REPO, "lenses/rf50mm/lens.json"), sensor_res=(480, 640), kernel_size=11)
    psfnet.psfnet.load_state_dict({k: torch.from_numpy(v) for k, v in mlp_state_dict().items()})      # stands in for load_net(ckpt)
    psfnet.analysis()

    # Read image (0_warm_up.py:14-17): synthetic RGB in [0,1], depth in metres
    img = torch.from_numpy(synth_rgb(480, 640))[None]
    depth = torch.from_numpy(synth_depth_mm(480, 640))[None, None] / 1e3

    # Render an image (0_warm_up.py:20-24)
    depth_mm = -depth * 1e3
    focus_dist = torch.tensor([-2400.0])
    defocused_img = psfnet.render(img.to(psfnet.device), depth_mm.to(psfnet.device), focus_dist.to(psfnet.device))
    save_image(defocused_img, os.path.join(out_dir, "aberrated_defocused_img.png"))
    save_image(img, os.path.join(out_dir, "all_in_focus_img.png"))

    # BASELINE.json configs[0]: 256x256, 5-slice stacks (per-pixel PSFs from the network, and the ray-traced PSF-grid path)
    net256 = PSFNet(filename=os.path.join(REPO, "lenses/rf50mm/lens.json"), sensor_res=(256, 256), kernel_size=11)
    net256.psfnet.load_state_dict(psfnet.psfnet.state_dict())
    img256 = torch.from_numpy(synth_rgb(256, 256))[None].to(net256.device)
    d256 = (torch.from_numpy(synth_depth_mm(256, 256))[None, None] / 1e3).to(net256.device)
    stack_m2, fds = render_focal_stack_m2(net256, img256, d256, 5)
    torch.manual_seed(0)
    stack_m1 = render_focal_stack_m1(net256, img256, -float(d256.mean()) * 1e3, (-fds[0] * 1e3).tolist(), grid=11, ks=11, spp=2048)
    for i in range(5):
        save_image(stack_m2[:, :, i], os.path.join(out_dir, f"stack_m2_{i}.png"))
        save_image(stack_m1[:, :, i], os.path.join(out_dir, f"stack_m1_{i}.png"))
    print(f"rendered {tuple(defocused_img.shape)} and two {tuple(stack_m2.shape)} stacks into {out_dir}; focus distances [m]: "
          f"{[round(float(f), 3) for f in fds[0]]}")
