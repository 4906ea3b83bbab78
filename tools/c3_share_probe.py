#!/usr/bin/env python3
"""What ONE rank of an N-rank config-3 job (BASELINE.json configs[2]: 16 scenes x 10 slices = 160 units) has to do,
measured on one GPU: rank 0's share of the partition is rendered into a local buffer (no gather, no peers) for
N = 1, 2, 4, 8 and compared with the 1-rank time / N that `expected_scaling` would otherwise assume.

    python tools/c3_share_probe.py [--partition interleaved|blocked] [--streams K] [--steps 10]

(interleaved = SURVEY.md 8e's u = r mod N; blocked = aadff.dist.scene_block, the default of render_scenes_sharded)

Per N: wall ms per step, host ms per step (time until every launch of the step is queued), the ideal t1/N and the ratio.
The share of an 8-rank job in the interleaved partition (u = r mod 8) is 1-2 slices of EVERY scene: 16 small launch triples
and 16 scenes' worth of host draws per step; the blocked partition gives the rank 2 whole scenes."""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--partition", default="interleaved")
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--scenes", type=int, default=16)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--profile", action="store_true", help="cProfile of the timed loop (stderr): where the host spends a step")
    args = ap.parse_args()
    from aadff import dist as adist
    from aadff.focal_stack import SceneUnitRenderer
    from aadff.synth import synth_depth_mm, synth_rgb
    from deeplens.optics import Lensgroup
    H = W = args.res
    S = 10
    dev = torch.device("cuda", 0)
    lens = Lensgroup(os.path.join(REPO, "lenses", "rf50mm", "lens.json"), sensor_res=(H, W), device=dev)
    scenes = []
    for sc in range(args.scenes):
        depth = synth_depth_mm(H, W, seed=5678 + sc)
        scenes.append((torch.from_numpy(synth_rgb(H, W, seed=1234 + sc))[None].to(dev), -float(depth.mean()),
                       -np.linspace(depth.min(), depth.max(), S)))
    kw = {}
    if args.streams > 1:
        kw["streams"] = args.streams
    rend = SceneUnitRenderer(lens, scenes, S, 11, 11, 2048, **kw)
    n = rend.n_units()
    rows, t1 = [], None
    for world in [int(w) for w in args.worlds.split(",")]:
        block = adist.scene_block(n, S, world) if args.partition == "blocked" else 1
        mine = adist.shard_units(n, 0, world, block)
        local = torch.empty((len(mine), 3, H, W), dtype=torch.float32, device=dev)
        for _ in range(2):
            rend.render(mine, out=local)
        torch.cuda.synchronize(dev)
        host = 0.0
        if args.profile:
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            h0 = time.perf_counter()
            rend.render(mine, out=local)
            host += time.perf_counter() - h0
        torch.cuda.synchronize(dev)
        wall = (time.perf_counter() - t0) / args.steps * 1e3
        if args.profile:
            import pstats
            prof.disable()
            pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(14)
        rend.check_flags()
        if world == 1:
            t1 = wall
        ideal = (t1 / world) if t1 else None
        rows.append({"world": world, "units_of_rank0": len(mine), "scene_groups": len({u // S for u in mine}),
                     "wall_ms": round(wall, 3), "host_ms": round(host / args.steps * 1e3, 3),
                     "ideal_ms": None if ideal is None else round(ideal, 3),
                     "efficiency": None if ideal is None else round(ideal / wall, 3),
                     "checksum": float(local.double().sum())})
        del local
    print(json.dumps({"partition": args.partition, "streams": args.streams, "scenes": args.scenes, "res": args.res, "rows": rows}))


if __name__ == "__main__":
    main()
