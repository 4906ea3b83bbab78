import os,sys,time,cProfile,pstats
REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[REPO,os.path.join(REPO,"aberration-aware-depth-from-focus_amd")]
import numpy as np, torch
from aadff.focal_stack import StackPlan, render_focal_stack_m1
from aadff.synth import synth_rgb
from deeplens.optics import Lensgroup
dev=torch.device("cuda:0"); H=W=1024
lens=Lensgroup(os.path.join(REPO,"lenses/rf50mm/lens.json"),sensor_res=(H,W),device=dev)
img=torch.from_numpy(synth_rgb(H,W))[None].to(dev); plan=StackPlan(lens,10,H,W)
fds=-np.linspace(500,5000,10)
for _ in range(5): render_focal_stack_m1(lens,img,-1500.,fds,plan=plan,update_lens=False)
torch.cuda.synchronize()
pr=cProfile.Profile(); pr.enable()
for _ in range(200): render_focal_stack_m1(lens,img,-1500.,fds,plan=plan,update_lens=False)
pr.disable(); torch.cuda.synchronize()
st=pstats.Stats(pr); st.sort_stats("cumulative").print_stats(22)
