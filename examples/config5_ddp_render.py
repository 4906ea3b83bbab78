#!/usr/bin/env python3
"""BASELINE.json config 5 in miniature: every rank renders the focal stacks of ITS OWN mini-batch on the fly and feeds a
depth-from-focus consumer wrapped in DistributedDataParallel (the loop of 2_aber_aware_dff_aif.py:95-130 with
`nn.DataParallel` replaced by one process per GPU).  The renderer needs no collective — stacks never leave the rank that
made them; the only exchange is DDP's gradient all-reduce of the consumer (RCCL over xGMI on a GPU node).

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/config5_ddp_render.py

`train(render_stack, ...)` takes the renderer as a callable so that the call pattern can be exercised on CPU with gloo
(tests/test_dist_gloo.py); on a GPU box `hip_render_stack()` returns PSFNet.render_stack (one fused launch per stack).
"""
import os
import sys

import torch
import torch.distributed as dist
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]


class TinyDFF(nn.Module):
    """Stand-in for the DFF consumer (AiFDepthNet / DFVNet are out of scope): [B,3,S,H,W] -> depth [B,1,H,W]."""

    def __init__(self, s):
        super().__init__()
        self.net = nn.Sequential(nn.Conv3d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv3d(8, 1, 3, padding=1))
        self.head = nn.Conv2d(s, 1, 1)

    def forward(self, stack):
        return self.head(self.net(stack).squeeze(1))


def train(render_stack, device, steps=2, n_stack=5, hw=(32, 32), batch=2):
    """`render_stack(img [B,3,H,W], depth_m [B,1,H,W], n_stack) -> ([B,3,S,H,W], fds [B,S])` runs rank-locally."""
    from aadff.dist import init_from_env
    rank, world = init_from_env(backend="nccl" if device.type == "cuda" else "gloo", device=device if device.type == "cuda" else None)
    torch.manual_seed(0)                                  # same initial weights everywhere
    net = TinyDFF(n_stack).to(device)
    ddp = nn.parallel.DistributedDataParallel(net, device_ids=[device.index] if device.type == "cuda" else None) if world > 1 else net
    opt = torch.optim.Adam(ddp.parameters(), 1e-3)
    g = torch.Generator().manual_seed(1000 + rank)        # every rank sees different scenes
    for _ in range(steps):
        img = torch.rand((batch, 3) + tuple(hw), generator=g).to(device)
        depth = (torch.rand((batch, 1) + tuple(hw), generator=g) * 4 + 0.5).to(device)      # metres
        with torch.no_grad():
            stack, fds = render_stack(img, depth, n_stack)                                   # rank-local, no collective
        loss = nn.functional.l1_loss(ddp(stack), depth)
        opt.zero_grad()
        loss.backward()                                   # DDP all-reduces the consumer's gradients here
        opt.step()
    return rank, world, net, float(loss)


def hip_render_stack(device, hw):
    from aadff.focal_stack import render_focal_stack_m2
    from aadff.synth import mlp_state_dict
    from deeplens.psfnet import PSFNet
    lens = PSFNet(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=hw, kernel_size=11, device=device)
    lens.psfnet.load_state_dict({k: torch.from_numpy(v) for k, v in mlp_state_dict().items()})
    return lambda img, depth_m, n: render_focal_stack_m2(lens, img, depth_m, n)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--save", default=None, help="directory: every rank writes its consumer's parameters there (tests compare them)")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--hw", type=int, nargs=2, default=(64, 64), help="image size; configs/aber_aware_dff_dfv.yml:21 is 480 640")
    ap.add_argument("--n-stack", type=int, default=5, help="focal-stack size; configs/aber_aware_dff_dfv.yml:20 is 8")
    ap.add_argument("--batch", type=int, default=2, help="mini-batch per rank; configs/aber_aware_dff_dfv.yml:19 is 2")
    a = ap.parse_args()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    hw = tuple(a.hw)
    rank, world, net, loss = train(hip_render_stack(dev, hw), dev, steps=a.steps, n_stack=a.n_stack, hw=hw, batch=a.batch)
    print(f"rank {rank}/{world}: loss {loss:.4f}", flush=True)
    if a.save:
        torch.save({"loss": loss, "params": [p.detach().cpu() for p in net.parameters()]}, os.path.join(a.save, f"config5_rank{rank}.pt"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
