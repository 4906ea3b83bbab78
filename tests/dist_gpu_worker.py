#!/usr/bin/env python3
"""Rank process of tests/test_gpu_dist.py (not collected by pytest): (scene, slice) units of a small M1 workload through
the HIP renderer, sharded in blocks of units dealt round-robin (block 1: u = r (mod world)) and all-gathered (gloo when the ranks are emulated on one GPU, RCCL otherwise).
Run without WORLD_SIZE it renders everything on one rank and also writes the plain per-scene stacks."""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--scenes", type=int, default=4)
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--slices", type=int, default=10)
    ap.add_argument("--grid", type=int, default=5)
    ap.add_argument("--spp", type=int, default=512)
    ap.add_argument("--block", type=int, default=None, help="units per block of the partition (default: aadff.dist.scene_block; 1 = u = r mod world)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the scene groups of a rank alternate over")
    ap.add_argument("--tag", default="")
    ap.add_argument("--check-inproc", action="store_true",
                    help="full-size runs: compare every gathered unit with the plain per-scene stack ON THE DEVICE in every rank "
                         "and write only a summary (the full set is 2 GB at 16 scenes x 10 x 1024^2)")
    a = ap.parse_args()
    import torch.distributed as dist
    from aadff import dist as adist
    from aadff.focal_stack import SceneUnitRenderer, render_focal_stack_m1, render_scenes_sharded
    from aadff.synth import synth_depth_mm, synth_rgb
    from deeplens.optics import Lensgroup
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rank, world = adist.init_from_env(backend="nccl", device=dev)
    H = W = a.res
    S, GRID, KS, SPP = a.slices, a.grid, 11, a.spp
    lens = Lensgroup(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(H, W), device=dev)
    scenes = []
    for sc in range(a.scenes):
        depth = synth_depth_mm(H, W, seed=900 + sc)
        scenes.append((torch.from_numpy(synth_rgb(H, W, seed=800 + sc))[None].to(dev), -float(depth.mean()),
                       -np.linspace(depth.min(), depth.max(), S)))
    rend = SceneUnitRenderer(lens, scenes, S, GRID, KS, SPP, streams=a.streams)
    full, mine = render_scenes_sharded(rend, gather=True, block=a.block)
    block = a.block or adist.scene_block(a.scenes * S, S, world)
    assert mine == adist.shard_units(a.scenes * S, rank, world, block) and all((u // block) % world == rank for u in mine)
    torch.cuda.synchronize(dev)
    if adist.grouped():                                # the overlapped form (side stream + event) must deliver the same bytes
        side = torch.cuda.Stream(dev)
        full2, mine2, done = render_scenes_sharded(rend, gather=True, stream=side, block=a.block)
        torch.cuda.current_stream(dev).wait_event(done)
        torch.cuda.synchronize(dev)
        assert mine2 == mine and full2.shape == full.shape
        d = float((full2 - full).abs().max())
        assert d <= 5e-6, f"rank {rank}: overlapped gather differs from the blocking one by {d}"
        # GatherRing (bench.py --gather): two output slots, gather of step i beside the render of step i+1
        ring = adist.GatherRing(lambda: torch.empty((2, 3, H, W), device=dev), world, slots=2, device=dev)
        got = []
        for i in range(5):
            k, buf = ring.acquire()
            buf.fill_(float(10 * i + rank))
            g, e = ring.submit(k)
            got.append((g, e, i))
            if len(got) == 2:                          # consume one step late, like a pipelined consumer
                g0, e0, i0 = got.pop(0)
                e0.synchronize()
                want = torch.tensor([10.0 * i0 + r for r in range(world)], device=dev)
                assert torch.equal(g0[:, 0, 0, 0, 0], want), (g0[:, 0, 0, 0, 0], want)
        ring.drain()
    rend.check_flags()
    if a.check_inproc:
        import json
        worst, mean = 0.0, 0.0
        for sc, (img, dbar, fds) in enumerate(scenes):
            torch.manual_seed(sc)
            st = render_focal_stack_m1(lens, img, dbar, fds, GRID, KS, SPP)[0].permute(1, 0, 2, 3)      # [S,3,H,W]
            worst = max(worst, float((full[sc * S:(sc + 1) * S] - st).abs().max()))
            mean += float(st.abs().mean()) / len(scenes)
        json.dump({"world": world, "rank": rank, "units": int(full.shape[0]), "shape": list(full.shape), "worst_abs_diff": worst,
                   "mean_abs_pixel": mean}, open(os.path.join(a.out, f"check_w{world}_r{rank}.json"), "w"))
        assert worst <= 5e-6, f"rank {rank}: gathered units differ from the plain per-scene stacks by {worst}"
    elif rank == 0:
        np.save(os.path.join(a.out, f"full_w{world}{a.tag}.npy"), full.cpu().numpy())
    if world == 1 and not a.check_inproc:
        stacks = []
        for sc, (img, dbar, fds) in enumerate(scenes):
            torch.manual_seed(sc)
            stacks.append(render_focal_stack_m1(lens, img, dbar, fds, GRID, KS, SPP)[0].permute(1, 0, 2, 3).cpu().numpy().copy())   # [S,3,H,W]
        np.save(os.path.join(a.out, "plain_stacks.npy"), np.concatenate(stacks, 0))
    if adist.grouped():
        t = adist.all_reduce_max(float(rank + 1))          # device tensor under RCCL, host tensor under gloo
        assert t == float(world), t
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
