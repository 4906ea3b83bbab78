import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np, torch
from aadff.focal_stack import StackPipeline, StackPlan, render_focal_stack_m1
from aadff.synth import synth_rgb
from deeplens.optics import Lensgroup
dev = torch.device("cuda:0"); H = W = 1024
lens = Lensgroup(os.path.join(REPO, "lenses/rf50mm/lens.json"), sensor_res=(H, W), device=dev)
img = torch.from_numpy(synth_rgb(H, W))[None].to(dev); plan = StackPlan(lens, 10, H, W)
STREAMS = int(sys.argv[1]) if len(sys.argv) > 1 else 1        # 2: StackPipeline, two stacks in flight on two streams
pipe = StackPipeline(lens, 10, H, W, depth=STREAMS) if STREAMS > 1 else None
fds = -np.linspace(500, 5000, 10)
t0 = time.perf_counter()
chk = None
for i in range(20000):
    torch.manual_seed(i % 50)
    if pipe is None:
        out = render_focal_stack_m1(lens, img, -1500., fds, plan=plan, update_lens=False)
    else:
        out, done = pipe.render(lens, img, -1500., fds, inputs_ready=True)
    if i % 50 == 7:
        if pipe is not None:
            torch.cuda.current_stream().wait_event(done)
        s = float(out.sum())
        if chk is None: chk = s
        assert abs(s - chk) <= 1e-3 * abs(chk), (i, s, chk)
torch.cuda.synchronize()
plan.check_flags() if pipe is None else pipe.check_flags()
print(f"{STREAMS} stream(s): 20000 stacks in {time.perf_counter() - t0:.2f} s, checksum stable ({chk:.4f}), flags clean")
