import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aberration-aware-depth-from-focus_amd")]
import numpy as np, torch
from aadff import strict_stack, _abi
from aadff.focal_stack import render_focal_stack_m1, StackPlan
from aadff.synth import synth_depth_mm, synth_rgb
from deeplens.optics import Lensgroup
from deeplens.basics import GEO_SPP, WAVE_RGB
DEV = "cuda:0"; dev = torch.device(DEV)
H = W = 256
S, grid, spp, ks = 4, 5, 512, 11
N, L, MS = grid * grid, 3, _abi.MAX_SURF
B = S * L
depth = synth_depth_mm(H, W, seed=5678)
dbar, fds = -float(depth.mean()), [float(f) for f in -np.linspace(depth.min(), depth.max(), S)]
img = torch.from_numpy(synth_rgb(H, W, seed=3))[None].to(DEV)
LP = os.path.join(REPO, "lenses", "rf50mm", "lens.json")
lens = Lensgroup(LP, sensor_res=(H, W), device=DEV, parity="strict")
torch.manual_seed(5)
for _ in range(3):
    render_focal_stack_m1(lens, img, dbar, fds, grid, 11, spp)
torch.cuda.synchronize()
st = lens._table_cache["strict-stage"]
counts = strict_stack.StrictCounts.of(lens)
rows = counts.rows[("psf", B, N, spp)]                      # the truth of the last stack: its inputs are still in the stage's device blocks
par = st.d_par[2]
par[B + S * N * 3:].copy_(torch.from_numpy(rows.reshape(-1)).to(dev))
wv = list(WAVE_RGB)
tab = lens._table(wv)
n_surf = len(lens.surfaces)
def launch(maps, centre, res):
    _abi.call("aadff_strict_psf_points", strict_stack._ptr_at(par, B), N, B, None, _abi.ptr(st.pset), _abi.ptr(tab), len(wv), n_surf, _abi.ptr(st.bt_main), _abi.ptr(st.bt_green),
              strict_stack._ptr_at(par, 0), strict_stack._ptr_at(st.d_pupil, st.n_pf), spp, strict_stack._ptr_at(st.d_pupil, st.n_pf + st.n_pm), GEO_SPP,
              strict_stack._ptr_at(par, B + S * N * 3), float(lens.pixel_size), ks, grid, _abi.ptr(maps), _abi.ptr(centre), strict_stack._ptr_at(res, 0),
              strict_stack._ptr_at(res, B * 4 * MS), _abi.stream_ptr(dev))
def run():
    maps = torch.empty((S, L, grid * ks, grid * ks), device=dev); centre = torch.empty((B, N, 2), device=dev); res = torch.zeros(B * 4 * MS + B, dtype=torch.int32, device=dev)
    launch(maps, centre, res)
    return maps, centre, res
ref = run(); torch.cuda.synchronize()
bg = Lensgroup(LP, sensor_res=(1024, 1024), device=DEV)
side = torch.cuda.Stream()
img_big = torch.rand(1, 3, 1024, 1024, device=DEV)
plan = StackPlan(bg, 10, 1024, 1024)
big0 = torch.rand(64 * 1024 * 1024, device=dev); A = torch.rand(4096, 4096, device=dev)
render_focal_stack_m1(bg, img_big, -3000.0, -np.linspace(500, 5000, 10), 11, 11, 2048, plan=plan, update_lens=False); torch.cuda.synchronize()
uu = torch.rand(10 * plan.per, device=dev)
gdep, gpts = plan.geometry(list(-np.linspace(500, 5000, 10)), -3000.0)
bad = badc = badbits = 0
NIT = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for it in range(NIT):
    with torch.cuda.stream(side):
        LOAD = os.environ.get("LOAD", "stack")
        if LOAD == "stack":
            for _ in range(3): render_focal_stack_m1(bg, img_big, -3000.0, -np.linspace(500, 5000, 10), 11, 11, 2048, plan=plan, update_lens=False)
        elif LOAD == "torch":
            for _ in range(40): big = torch.sin(big0) * 1.0001
        elif LOAD == "matmul":
            for _ in range(6): mm = A @ A
        elif LOAD == "conv":
            for _ in range(12): _abi.call("aadff_render_psf_map_stack", _abi.ptr(img_big), _abi.ptr(plan.psf_maps), _abi.ptr(plan.out), 1, 3, 10, 1024, 1024, 11, 11, _abi.stream_ptr(dev))
        elif LOAD == "conv1":
            for _ in range(40): _abi.call("aadff_render_psf_map", _abi.ptr(img_big), _abi.ptr(plan.psf_maps[0]), _abi.ptr(plan.out[:, :, 0].contiguous()), 1, 3, 1024, 1024, 11, 11, _abi.stream_ptr(dev))
        elif LOAD == "psf":
            ub = uu.data_ptr()
            for _ in range(2):
                _abi.call("aadff_psf_points", _abi.ptr(gpts), 10, 121, 3, _abi.ptr(plan.tab_rgb), _abi.ptr(plan.tab_green), plan.lc, _abi.ptr(plan.states), C.c_void_p(ub + 4 * plan.o_main), 2048, plan.per, plan.per_l,
                          C.c_void_p(ub + 4 * plan.o_chief), 2048, plan.per, plan.per_l, 11, 1, 1, _abi.ptr(plan.psf_maps), None, _abi.ptr(plan.flags), _abi.stream_ptr(dev))
        elif LOAD == "none":
            pass
    got = run()
    torch.cuda.synchronize()
    d = float((got[0] - ref[0]).abs().max() / ref[0].max())
    dc = float((got[1] - ref[1]).abs().max())
    db = int((got[2][:B * 4 * MS] != ref[2][:B * 4 * MS]).sum())
    if d > 2e-6 or dc > 0 or db:
        bad += d > 2e-6; badc += dc > 0; badbits += db > 0
        if bad + badc + badbits <= 3:
            dm = (got[0] - ref[0]).abs().amax(dim=(2, 3)) / ref[0].max()
            wd = torch.nonzero(got[2][:B * 4 * MS].view(B, 2, 2, MS) != ref[2][:B * 4 * MS].view(B, 2, 2, MS))
            print("   any/nan words differing (job, phase, any|nan, surface, ref, got):", [(int(i[0]), int(i[1]), int(i[2]), int(i[3]), hex(int(ref[2][:B*4*MS].view(B,2,2,MS)[tuple(i)])), hex(int(got[2][:B*4*MS].view(B,2,2,MS)[tuple(i)]))) for i in wd[:12]], flush=True)
            cd = torch.nonzero((got[1] != ref[1]).any(-1))
            print("   centres differing (batch, point):", len(cd), [(int(i[0]), int(i[1]), float(ref[1][tuple(i)][0]), float(got[1][tuple(i)][0] - ref[1][tuple(i)][0])) for i in cd[:10]], flush=True)
            print("it", it, "maps max|d|/max %.1e" % d, "centres max|d| %.1e" % dc, "any-bit words differing", db, "batches (slice, lambda) off:", [(int(i[0]), int(i[1])) for i in torch.nonzero(dm > 2e-6)][:6], flush=True)
print("strict psf_map kernel alone, identical inputs, beside a busy stream:", bad, "launches with other PSFs,", badc, "with other centres,", badbits, "with other any-bits, of", NIT, flush=True)
