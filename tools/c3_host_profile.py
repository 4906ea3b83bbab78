import os, sys, time, cProfile, pstats
REPO = "/root/repo"
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np, torch
from aadff.focal_stack import SceneUnitRenderer, render_scenes_sharded
from aadff.synth import synth_depth_mm, synth_rgb
from deeplens.optics import Lensgroup
dev = torch.device("cuda:0"); H = W = 1024; S = 10
lens = Lensgroup(os.path.join(REPO, "lenses/rf50mm/lens.json"), sensor_res=(H, W), device=dev)
scenes = []
for sc in range(4):
    depth = synth_depth_mm(H, W, seed=5678 + sc)
    scenes.append((torch.from_numpy(synth_rgb(H, W, seed=1234 + sc))[None].to(dev), -float(depth.mean()), -np.linspace(depth.min(), depth.max(), S)))
rend = SceneUnitRenderer(lens, scenes, S)
for _ in range(2): render_scenes_sharded(rend)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): render_scenes_sharded(rend)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("no profiler: host", (t1 - t0) / 5 * 1e3, "ms/step; sync tail", (t2 - t1) * 1e3)
import aadff.focal_stack as fs
T = {}
def timed(name, f):
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[name] = T.get(name, 0) + time.perf_counter() - t; return r
    return g
fs.render_focal_stack_m1 = timed("render_focal_stack_m1", fs.render_focal_stack_m1)
rend.seed_scene = timed("seed", rend.seed_scene)
orig_zeros = torch.zeros
torch.zeros = timed("zeros", torch.zeros)
fs.StackPlan.geometry = timed("geometry", fs.StackPlan.geometry)
fs.StackPlan.uniforms_host = timed("uniforms_host", fs.StackPlan.uniforms_host)
fs.PresetSampler.rand_into = timed("preset.rand_into", fs.PresetSampler.rand_into)
torch.cuda.Event.synchronize = timed("event.sync", torch.cuda.Event.synchronize)
fs.StackPlan._poll_mirror = timed("poll", fs.StackPlan._poll_mirror)
fs.StackPlan.staged = timed("staged", fs.StackPlan.staged)
t0 = time.perf_counter()
for _ in range(5): render_scenes_sharded(rend)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("timed: host", (t1 - t0) / 5 * 1e3, "ms/step", {k: round(v / 5 * 1e3, 3) for k, v in T.items()})
