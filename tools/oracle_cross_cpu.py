#!/usr/bin/env python3
"""How far apart are two runs of the REFERENCE FORMULATION on two different host CPUs?  The oracle (torch CPU, float32,
the reference's operation order) is run for the bench workload's PSF maps, cases k = 0..N-1 (torch.manual_seed(k), scene k
of tools/parity_seeds.py), once on each machine:

    python tools/oracle_cross_cpu.py --save tools/_oracle_maps_<name>.npz        # machine A (e.g. the build container)
    python tools/oracle_cross_cpu.py --compare tools/_oracle_maps_<name>.npz     # machine B (e.g. the GPU box's host)

--compare prints, per case and slice, the rel-L2 between the two machines' PSF maps, and - when a GPU is present - the
distance of the HIP fast / strict PSF maps to EACH of the two.  torch's CPU kernels for sqrt / sin / cos / atan2 are not
correctly rounded and differ between the AVX2 and AVX-512 code paths; every ulp upstream of the window-edge test of
deeplens/monte_carlo.py:37 re-draws which rays fall inside the ks x ks window."""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch


def scene(k):
    from aadff.synth import synth_depth_mm
    depth = synth_depth_mm(1024, 1024, seed=5678 + k)
    return -float(depth.mean()), -np.linspace(depth.min(), depth.max(), 10)


def oracle_maps(k, olens):
    dbar, fds = scene(k)
    torch.manual_seed(k)
    maps = []
    for f in fds:
        olens.refocus(float(f))
        maps.append(olens.psf_map(depth=dbar, grid=11, ks=11, spp=2048).numpy())
    return np.stack(maps)                                     # [S,3,121,121]


def rel(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--save", default=None)
    ap.add_argument("--compare", default=None)
    ap.add_argument("--cases", type=int, default=5)
    a = ap.parse_args()
    import bench
    from oracle.lens import OracleLens
    torch.set_num_threads(bench.usable_cpus())
    lens_path = os.path.join(REPO, "lenses", "rf50mm", "lens.json")
    olens = OracleLens(lens_path, sensor_res=(1024, 1024))
    here = {f"case{k}": oracle_maps(k, olens) for k in range(a.cases)}
    if a.save:
        np.savez_compressed(a.save, cpu=np.array(bench.cpu_model()), **here)
        print(json.dumps({"saved": a.save, "cpu": bench.cpu_model(), "cases": a.cases}))
        return
    other = np.load(a.compare)
    res = {"this_cpu": bench.cpu_model(), "other_cpu": str(other["cpu"]), "rows": []}
    gpu = torch.cuda.is_available()
    if gpu:
        from aadff.focal_stack import render_focal_stack_m1
        from aadff.synth import synth_rgb
        from deeplens.optics import Lensgroup
        dev = torch.device("cuda", 0)
        lenses = {"fast": Lensgroup(lens_path, sensor_res=(1024, 1024), device=dev),
                  "strict": Lensgroup(lens_path, sensor_res=(1024, 1024), device=dev, parity="strict")}
        import importlib
        rp = importlib.import_module("deeplens.render_psf")

        def images(maps, img):                                # the same (deterministic) HIP convolution for every set of maps
            return np.stack([rp.render_psf_map(img, torch.from_numpy(np.ascontiguousarray(m)).to(dev), 11)[0].cpu().numpy() for m in maps], 1)
    for k in range(a.cases):
        A, B = here[f"case{k}"], other[f"case{k}"]
        row = {"case": k, "oracle_here_vs_oracle_there_per_slice": [float(f"{rel(A[s], B[s]):.2e}") for s in range(10)],
               "oracle_here_vs_oracle_there": float(f"{rel(A, B):.3e}")}
        if gpu:
            dbar, fds = scene(k)
            img = torch.from_numpy(synth_rgb(1024, 1024, seed=1234 + k))[None].to(dev)
            IA, IB = images(A, img), images(B, img)
            row["images_oracle_here_vs_there"] = float(f"{rel(IA, IB):.3e}")
            row["images_oracle_here_vs_there_per_slice"] = [float(f"{rel(IA[:, s], IB[:, s]):.2e}") for s in range(10)]
            for name, lens in lenses.items():
                torch.manual_seed(k)
                _, maps = render_focal_stack_m1(lens, img, dbar, fds, 11, 11, 2048, return_maps=True)
                M = maps.cpu().numpy()
                IM = images(M, img)
                row[name] = {"vs_oracle_here": float(f"{rel(M, A):.3e}"), "vs_oracle_there": float(f"{rel(M, B):.3e}"),
                             "images_vs_oracle_here": float(f"{rel(IM, IA):.3e}"), "images_vs_oracle_there": float(f"{rel(IM, IB):.3e}"),
                             "images_worst_slice_vs_here": float(f"{max(rel(IM[:, s], IA[:, s]) for s in range(10)):.3e}"),
                             "images_worst_slice_vs_there": float(f"{max(rel(IM[:, s], IB[:, s]) for s in range(10)):.3e}"),
                             "per_slice_vs_here": [float(f"{rel(M[s], A[s]):.2e}") for s in range(10)],
                             "per_slice_vs_there": [float(f"{rel(M[s], B[s]):.2e}") for s in range(10)]}
        res["rows"].append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
