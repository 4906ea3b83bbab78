"""Parity MARGINS of the HIP path (run with `-m gpu`): the same comparisons as tests/test_gpu_parity.py, but every measured
error is recorded next to its tolerance and printed in the terminal summary (tests/conftest.py), so a shortcut in a kernel
that eats into the budget shows up as a number, not only when it finally fails.

Tolerances (SURVEY.md 8c / BASELINE.json north_star): rendered image <= 1e-4 rel-L2 (fp32-vs-fp64 floor 4.7e-5), PSF
<= 2e-3 rel-L2 (floor 5.2e-4), validity mismatches <= 1e-4 of rays, focus scalars <= 1e-5 relative."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from aadff import _abi                                   # noqa: E402
from aadff.focal_stack import StackPlan, render_focal_stack_m1   # noqa: E402
from aadff.synth import mlp_state_dict, synth_depth_mm, synth_rgb   # noqa: E402
from deeplens.optics import Lensgroup                     # noqa: E402
from deeplens.psfnet import PSFNet, _TrainStep            # noqa: E402
from oracle import conv as oconv                          # noqa: E402

rp = importlib.import_module("deeplens.render_psf")
DEV = "cuda:0"


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def tt(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def lp(repo_root, name="rf50mm"):
    return os.path.join(repo_root, "lenses", name, "lens.json")


def test_margin_refocus_and_sensor_hits(golden_dir, repo_root, margin):
    g = np.load(os.path.join(golden_dir, "g2_g3_trace_splat.npz"))
    lens = Lensgroup(lp(repo_root), sensor_res=(1024, 1024), device=DEV)
    torch.manual_seed(0)
    lens.refocus(-2000.0)
    margin("refocus(-2000) d_sensor, relative", abs(lens.d_sensor - float(g["d_sensor"])) / float(g["d_sensor"]), 1e-5)
    margin("refocus(-2000) hfov, relative", abs(lens.hfov - float(g["hfov"])) / float(g["hfov"]), 1e-5)
    pobj, ut, ur = tt(g["points_obj"]).to(DEV), tt(g["u_theta"]).to(DEV), tt(g["u_r"]).to(DEV)
    pz, pr = lens.entrance_pupil()
    o = torch.zeros((256, 121, 3), device=DEV)
    d = torch.zeros_like(o)
    ra = torch.zeros((256, 121), device=DEV)
    _abi.call("aadff_trace_points", _abi.ptr(pobj), 121, _abi.ptr(ut), _abi.ptr(ur), 256, float(pz), float(pr),
              _abi.ptr(lens._table([0.589])), 12, _abi.ptr(lens._state_device()), _abi.ptr(o), _abi.ptr(d), _abi.ptr(ra),
              _abi.stream_ptr(torch.device(DEV)))
    rah, want = ra.cpu().numpy() > 0, g["sensor_ra"] > 0
    both = rah & want
    err = np.abs(o[..., :2].cpu().numpy() - g["sensor_xy"])[both]
    margin("sensor hits: validity-mask mismatch fraction", (rah != want).mean(), 1e-4)
    margin("sensor hits: mean |dxy| [mm] (fp32-vs-fp64 floor 5.3e-6)", err.mean(), 2e-5)
    margin("sensor hits: max |dxy| [mm] (floor 4.1e-5)", err.max(), 5e-4)


@pytest.mark.parametrize("name,res,foc,depth,spp", [("rf50mm", (1024, 1024), -2000.0, -1500.0, 2048),
                                                    ("50mm_f2.8", (256, 256), -1000.0, -1250.0, 512)])
def test_margin_psf_map_and_rendered_image(golden_dir, repo_root, margin, name, res, foc, depth, spp):
    g4 = np.load(os.path.join(golden_dir, "g4_psf_map.npz"))
    key = name.replace(".", "_")
    lens = Lensgroup(lp(repo_root, name), sensor_res=res, device=DEV)
    torch.manual_seed(0)
    lens.refocus(foc)
    pm = lens.psf_map(depth=depth, grid=11, ks=11, spp=spp)
    img = tt(synth_rgb(256, 256))[None]
    want = oconv.render_psf_map(img, tt(g4[f"{key}_psf_map"]), 11).numpy()
    got = rp.render_psf_map(img.to(DEV), pm, 11).cpu().numpy()
    margin(f"psf_map {name}: PSF rel-L2 vs reference (floor 5.2e-4)", rel(pm.cpu().numpy(), g4[f"{key}_psf_map"]), 2e-3)
    margin(f"psf_map {name}: rendered image rel-L2 (floor 4.7e-5)", rel(got, want), 1e-4)


def test_margin_stack_256(golden_dir, repo_root, margin):
    g8 = np.load(os.path.join(golden_dir, "g8_stack_m1.npz"))
    lens = Lensgroup(lp(repo_root), sensor_res=(256, 256), device=DEV)
    img = tt(synth_rgb(256, 256))[None].to(DEV)
    torch.manual_seed(0)
    stack, maps = render_focal_stack_m1(lens, img, float(g8["dbar"]), g8["fds"], grid=11, ks=11, spp=2048, return_maps=True)
    s = stack[0].cpu().numpy()
    margin("M1 stack 256^2 x 5: PSF maps rel-L2", rel(maps.cpu().numpy(), g8["psf_maps"]), 2e-3)
    margin("M1 stack 256^2 x 5: image crop rel-L2", rel(s[:, :, 96:160, 96:160], g8["crop"]), 1e-4)


def test_margin_bench_config_stack_plan_staged(golden_dir, repo_root, margin):
    """THE bench workload through THE bench path (BASELINE.json configs[1]: 1024^2, 10 focus distances, grid 11, ks 11,
    spp 2048; reused StackPlan, pinned ring, staged upload riding on the refocus / PSF launches, slice-batched convolution)
    against the G9 fixture generated from the reference with the same seed."""
    g = np.load(os.path.join(golden_dir, "g9_stack_m1_1024.npz"))
    H = W = 1024
    S = 10
    lens = Lensgroup(lp(repo_root), sensor_res=(H, W), device=DEV)
    img = tt(synth_rgb(H, W, seed=1234))[None].to(DEV)
    depth = synth_depth_mm(H, W, seed=5678)
    dbar, fds = -float(depth.mean()), -np.linspace(depth.min(), depth.max(), S)
    assert dbar == pytest.approx(float(g["dbar"]), abs=1e-9) and fds == pytest.approx(g["fds"], abs=1e-9)
    plan = StackPlan(lens, S, H, W, 1, 3, 11, 11, 2048)
    for i in (3, 2, 1, 0, 0):                         # several steps on the ring first; the last two are seed 0
        torch.manual_seed(i)
        out = render_focal_stack_m1(lens, img, dbar, fds, 11, 11, 2048, plan=plan, update_lens=False)
    plan.check_flags()
    s = out[0].cpu().numpy()                          # [3,S,H,W]
    maps = plan.psf_maps.cpu().numpy()
    st = np.frombuffer(plan.states.cpu().numpy().tobytes(), dtype=np.float32).reshape(S, 8)
    margin("bench config: d_sensor per slice, worst relative", np.abs(st[:, 0] / g["d_sensor"] - 1).max(), 1e-5)
    # The fp32 noise of this algorithm is far from uniform over the stack: slices focused near the depth plane have needle
    # PSFs (G13, tests/golden/make_floor.py: the reference's own fp32 result is 5e-3 / 1.4e-4 away from the float64
    # evaluation of the same draws there, 1e-4 / 5e-6 elsewhere).  So: the STACK is held to the north-star budget, every
    # slice to max(budget, 2 x that slice's floor), and on the stored truth slices the HIP result must be no further
    # from the float64 truth than 1.5 x the reference's own distance.
    fl = np.load(os.path.join(golden_dir, "g13_fp32_floor.npz"))
    psf_err = [rel(maps[k], g["psf_maps"][k]) for k in range(S)]
    margin("bench config: PSF maps rel-L2 vs reference, whole stack", rel(maps, g["psf_maps"]), 2e-3)
    for k in range(S):
        margin(f"bench config: slice {k} PSF map rel-L2 (floor {fl['psf_floor'][k]:.1e})", psf_err[k], max(2e-3, 2 * fl["psf_floor"][k]))
    crops = {"seam": s[:, :, 61:125, 154:218], "centre": s[:, :, 480:544, 480:544], "corner": s[:, :, 960:1024, 960:1024]}
    for k, v in crops.items():
        margin(f"bench config: 64x64 crop '{k}' rel-L2, whole stack", rel(v, g[f"crop_{k}"]), 1e-4)
        for j in range(S):
            margin(f"bench config: crop '{k}' slice {j} (image floor {fl['img_floor'][j]:.1e})", rel(v[:, j], g[f"crop_{k}"][:, j]),
                   max(1e-4, 2 * fl["img_floor"][j]))
    bm = s.astype(np.float64).reshape(3, S, 64, 16, 64, 16).mean((3, 5))
    margin("bench config: 16x16 block means of the whole stack, rel-L2", rel(bm, g["block_means"]), 1e-4)
    margin("bench config: per-slice per-channel sums, worst relative", np.abs(s.astype(np.float64).sum((2, 3)).T / g["sums"] - 1).max(), 1e-5)
    for i, k in enumerate(fl["truth_slices"]):
        ours, ref = rel(maps[k], fl["truth_maps"][i]), rel(g["psf_maps"][k], fl["truth_maps"][i])
        margin(f"bench config: slice {k} PSF distance to float64 truth, HIP / reference ({ref:.1e})", ours / ref, 1.5)


def test_margin_bench_stack_literal_op_order_build(golden_dir, repo_root, margin, monkeypatch):
    """the same stack from the build that traces in the reference's OWN formulation
    (csrc/libaadff_literal.so: one ray per lane, Newton from the vertex plane on every curved surface incl. spheres, IEEE
    division / sqrt, libm sin/cos, literal d sag/d r^2 and normal, no fma contraction; deeplens/surfaces.py:456-487,523-586),
    slice by slice against the reference (G9), next to the shipped build's numbers above.  Per slice: distance of the
    whole rendered slice to the image rendered from G9's PSF map by the same convolution.  The literal order does NOT
    bring the two wide-PSF slices under 1e-4 (measured 1.36e-4 / 7.8e-5 against 1.53e-4 / 1.08e-4): what separates any
    float32 implementation from the reference there is not the formulation but which side of the 11x11 window's hard
    edge (deeplens/monte_carlo.py:37) a few border rays fall on, and that follows the last bit of every sqrt / sin / cos
    the reference takes from MKL's vector maths (tools/floor_sources.py, oracle/scalar_trace.py)."""
    path = os.path.join(os.path.dirname(_abi.LIB_PATH), "libaadff_literal.so")
    if not os.path.exists(path):
        pytest.skip("csrc/libaadff_literal.so not built (make -C csrc libaadff_literal.so)")
    g = np.load(os.path.join(golden_dir, "g9_stack_m1_1024.npz"))
    fl = np.load(os.path.join(golden_dir, "g13_fp32_floor.npz"))
    H = W = 1024
    S = 10
    img = tt(synth_rgb(H, W, seed=1234))[None].to(DEV)
    depth = synth_depth_mm(H, W, seed=5678)
    dbar, fds = -float(depth.mean()), -np.linspace(depth.min(), depth.max(), S)
    ref_imgs = [rp.render_psf_map(img, tt(g["psf_maps"][k]).to(DEV), 11)[0].cpu().numpy().astype(np.float64) for k in range(S)]
    worst = {}
    for label, lib in (("shipped", None), ("literal", _abi.load_library(path))):
        if lib is not None:
            monkeypatch.setattr(_abi, "_lib", lib)
        lens = Lensgroup(lp(repo_root), sensor_res=(H, W), device=DEV)
        plan = StackPlan(lens, S, H, W, 1, 3, 11, 11, 2048)
        torch.manual_seed(0)
        out = render_focal_stack_m1(lens, img, dbar, fds, 11, 11, 2048, plan=plan, update_lens=False)
        plan.check_flags()
        s = out[0].cpu().numpy().astype(np.float64)
        per = [float(np.linalg.norm(s[:, k] - ref_imgs[k]) / np.linalg.norm(ref_imgs[k])) for k in range(S)]
        num = sum(float(((s[:, k] - ref_imgs[k]) ** 2).sum()) for k in range(S))
        den = sum(float((ref_imgs[k] ** 2).sum()) for k in range(S))
        margin(f"{label} build: whole stack, full-image rel-L2 vs reference PSFs", np.sqrt(num / den), 1e-4)
        for k in range(S):
            margin(f"{label} build: slice {k} full-image rel-L2 (fp32 floor {fl['img_floor'][k]:.1e})", per[k], max(1e-4, 2 * fl["img_floor"][k]))
        worst[label] = max(per)
    monkeypatch.undo()


def test_margin_bench_stack_strict_parity_mode(golden_dir, repo_root, margin):
    """Lensgroup(parity="strict"): the bench workload with every trace in the reference's own float32 operation order
    (aadff_trace_rays_strict: one launch pair per surface, batch-wide Newton iteration counts, IEEE division / sqrt, no fma
    contraction) and the reference's host-side arithmetic done by the same torch / numpy calls.  EVERY slice is held to the
    north-star 1e-4 with NO floor widening (the two wide-PSF slices, 1.5e-4 / 1.1e-4 in the fast build, come out at
    <= 7e-5; slices whose d_sensor / hfov / pupil reproduce to the bit at < 1e-6).  10 ms per stack (three batched traces)."""
    import time
    g = np.load(os.path.join(golden_dir, "g9_stack_m1_1024.npz"))
    fl = np.load(os.path.join(golden_dir, "g13_fp32_floor.npz"))
    H = W = 1024
    S = 10
    lens = Lensgroup(lp(repo_root), sensor_res=(H, W), device=DEV, parity="strict")
    img = tt(synth_rgb(H, W, seed=1234))[None].to(DEV)
    depth = synth_depth_mm(H, W, seed=5678)
    dbar, fds = -float(depth.mean()), -np.linspace(depth.min(), depth.max(), S)
    t0 = time.perf_counter()
    torch.manual_seed(0)
    out, maps = render_focal_stack_m1(lens, img, dbar, fds, 11, 11, 2048, return_maps=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s = out[0].cpu().numpy().astype(np.float64)
    num = den = 0.0
    for k in range(S):
        ref = rp.render_psf_map(img, tt(g["psf_maps"][k]).to(DEV), 11)[0].cpu().numpy().astype(np.float64)
        dlt = s[:, k] - ref
        num += float((dlt * dlt).sum())
        den += float((ref * ref).sum())
        margin(f"strict mode: slice {k} full-image rel-L2 vs reference PSFs (no floor widening)", np.sqrt((dlt * dlt).sum() / (ref * ref).sum()), 1e-4)
        # PSF tolerance 2e-3, or that slice's own fp32-vs-fp64 floor where the floor is above it (G13: slices 1 and 2, 5.0e-3 / 3.3e-3)
        margin(f"strict mode: slice {k} PSF map rel-L2 (fp32 floor {fl['psf_floor'][k]:.1e})", rel(maps[k].cpu().numpy(), g["psf_maps"][k]),
               max(2e-3, float(fl["psf_floor"][k])))
    margin("strict mode: whole stack, full-image rel-L2", np.sqrt(num / den), 1e-4)
    crops = {"seam": s[:, :, 61:125, 154:218], "centre": s[:, :, 480:544, 480:544], "corner": s[:, :, 960:1024, 960:1024]}
    for k, v in crops.items():
        margin(f"strict mode: 64x64 crop '{k}' rel-L2 vs the reference's pixels, whole stack", rel(v, g[f"crop_{k}"]), 1e-4)
    margin("strict mode: seconds per 10-slice stack (informative)", dt, 120.0)


@pytest.mark.parametrize("case", [1, 2, 3, 4])
def test_margin_more_seeds_and_scenes(golden_dir, repo_root, margin, case):
    """the reference draws new pupil samples in every call (deeplens/optics.py:480-481, :1006-1026), so fixture
    G9 / bench.py pin ONE realisation of the Monte-Carlo noise.  G9b holds the reference's own output for four further (generator
    seed k, scene k) pairs of the bench workload (tests/golden/make_golden.py G9B), G13b the fp32-vs-fp64 floor of each
    (make_floor.py g13b).  Per case, the whole rendered stack against the image the reference's PSF maps give (same HIP
    convolution, <= 2e-6 abs from F.conv2d; the reference's own 64 x 64 crops and block means check that reconstruction):
      * parity="strict": <= 1e-4 per stack on EVERY case, no widening;
      * fast path: <= max(1e-4, 2 x that case's whole-stack floor); per slice max(1e-4, 2 x that slice's floor).
    Case 2 is the one that read 1.5e-4 in round 3's probe (tools/parity_seeds.py): its floor - the float32 reference against
    float64 on the same draws - is 1.5e-4 for the stack and 4.5e-4 on slice 0 (focus -500 mm against a plane at -2.7 m: every
    blur disc is cropped by the 11 x 11 window, the border-ray flips of DESIGN.md section 2)."""
    g = np.load(os.path.join(golden_dir, f"g9b_case{case}.npz"))
    fl = np.load(os.path.join(golden_dir, "g13b_fp32_floor_cases.npz"))
    stack_floor, slice_floor, psf_floor = float(fl[f"stack_img_floor_{case}"]), fl[f"img_floor_{case}"], fl[f"psf_floor_{case}"]
    H = W = 1024
    S = 10
    img = tt(synth_rgb(H, W, seed=1234 + case))[None].to(DEV)
    depth = synth_depth_mm(H, W, seed=5678 + case)
    dbar, fds = -float(depth.mean()), -np.linspace(depth.min(), depth.max(), S)
    assert dbar == float(g["dbar"]) and np.array_equal(fds, g["fds"])
    ref = torch.stack([rp.render_psf_map(img, tt(g["psf_maps"][k]).to(DEV), 11)[0] for k in range(S)], 1).double()      # [3,S,H,W]
    refn = ref.cpu().numpy()
    # the reconstruction IS the reference's stack: its own crops (fp32 conv2d on the CPU) and block means
    for name, sl in (("seam", (slice(61, 125), slice(154, 218))), ("centre", (slice(480, 544), slice(480, 544))), ("corner", (slice(960, 1024), slice(960, 1024)))):
        assert np.abs(refn[:, :, sl[0], sl[1]] - g[f"crop_{name}"]).max() <= 2e-6, name
    assert rel(refn.reshape(3, S, 64, 16, 64, 16).mean((3, 5)), g["block_means"]) <= 1e-6
    den = float((ref * ref).sum())
    den_k = (ref * ref).sum((0, 2, 3)).cpu().numpy()
    for label, kw in (("fast", {}), ("strict", {"parity": "strict"})):
        lens = Lensgroup(lp(repo_root), sensor_res=(H, W), device=DEV, **kw)
        torch.manual_seed(case)
        out, maps = render_focal_stack_m1(lens, img, dbar, fds, 11, 11, 2048, return_maps=True)
        d = out[0].double() - ref
        per = np.sqrt((d * d).sum((0, 2, 3)).cpu().numpy() / den_k)
        whole = float(np.sqrt(float((d * d).sum()) / den))
        if label == "strict":
            margin(f"case {case} strict: whole stack rel-L2 vs the reference (no widening)", whole, 1e-4)
            margin(f"case {case} strict: worst slice rel-L2 (slice {int(per.argmax())}; informative, floor {slice_floor[per.argmax()]:.1e})", per.max(), max(1e-4, 2 * slice_floor[per.argmax()]))
            assert lens.d_sensor == pytest.approx(float(g["d_sensor"][-1]), rel=2e-7) and lens.hfov == pytest.approx(float(g["hfov"][-1]), rel=2e-7)
        else:
            margin(f"case {case} fast: whole stack rel-L2 vs the reference (stack floor {stack_floor:.1e})", whole, max(1e-4, 2 * stack_floor))
            k = int(np.argmax(per / np.maximum(1e-4, 2 * slice_floor)))
            margin(f"case {case} fast: slice {k} rel-L2, the one nearest its budget (slice floor {slice_floor[k]:.1e})", per[k], max(1e-4, 2 * slice_floor[k]))
            assert all(per[i] <= max(1e-4, 2 * slice_floor[i]) for i in range(S)), per
        margin(f"case {case} {label}: PSF maps rel-L2, whole stack (worst slice floor {psf_floor.max():.1e})", rel(maps.cpu().numpy(), g["psf_maps"]),
               max(2e-3, 2 * float(psf_floor.max())))


def test_strict_centroid_has_torch_cpu_sum_bits():
    """aadff_strict_centroid: -(sum_s o_xy ra) / (sum_s ra + 1e-9) in ATen's CPU summation order (psf_center, deeplens/optics.py:902-904),
    bit for bit against torch on the host - the shapes of psf_center and of a training batch, a ragged ray count, dead points."""
    dev = torch.device(DEV)
    for spp, N, B in ((2048, 121, 3), (4096, 128, 1), (1000, 9, 2)):
        g = torch.Generator().manual_seed(spp + N)
        o = (torch.randn(B, spp, N, 3, generator=g) * 5).contiguous()
        ra = (torch.rand(B, spp, N, generator=g) > 0.25).float()
        ra[0, :, 0] = 0                                                        # a point without a valid ray: 0 / 1e-9
        want = torch.stack([-((o[b] * ra[b].unsqueeze(-1)).sum(0) / ra[b].unsqueeze(-1).sum(0).add(1e-9))[..., :2] for b in range(B)])
        od, rd = o.to(dev), ra.to(dev)
        c = torch.empty((B, N, 2), device=dev)
        av = torch.zeros(B, dtype=torch.int32, device=dev)
        _abi.call("aadff_strict_centroid", _abi.ptr(od), _abi.ptr(rd), spp, N, B, _abi.ptr(c), _abi.ptr(av), _abi.stream_ptr(dev))
        assert torch.equal(c.cpu(), want), (spp, N, B, float((c.cpu() - want).abs().max()))
        assert av.cpu().tolist() == [1] * B


@pytest.mark.parametrize("res,S,grid,spp", [((256, 256), 4, 5, 512), ((1024, 1024), 10, 11, 2048)])
def test_strict_stack_batched_equals_call_by_call(repo_root, margin, monkeypatch, res, S, grid, spp):
    """Round 4: a strict-parity stack as THREE batched traces (aadff/strict_stack.py: `aadff_trace_rays_strict_batched` - one launch
    per surface for all Newton batches of a level, every batch with its own iteration counts; rays built on the device; chief-ray
    centres by `aadff_strict_centroid`) against the reference's loop run call by call (AADFF_STRICT_BATCHED=0: 72 single traces per
    slice, host-side ray construction and centroid): identical d_sensor / hfov per slice, PSF maps equal to the float atomics of
    the histogram kernel, and the time per stack of both."""
    import time
    from aadff import strict_stack
    H, W = res
    depth = synth_depth_mm(H, W, seed=5678)
    dbar, fds = -float(depth.mean()), [float(f) for f in -np.linspace(depth.min(), depth.max(), S)]
    out = {}
    for label in ("loop", "batched"):
        lens = Lensgroup(lp(repo_root), sensor_res=(H, W), device=DEV, parity="strict")
        torch.manual_seed(3)
        if label == "batched":
            strict_stack.strict_psf_maps(lens, dbar, fds, grid, 11, spp, fused=False)          # warm the allocator
            torch.manual_seed(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        maps = strict_stack.strict_psf_maps_loop(lens, dbar, fds, grid, 11, spp) if label == "loop" else strict_stack.strict_psf_maps(lens, dbar, fds, grid, 11, spp, fused=False)
        torch.cuda.synchronize()
        out[label] = (maps.cpu().numpy(), lens.d_sensor, lens.hfov, lens.foclen, lens.fnum, time.perf_counter() - t0, torch.rand(1).item())
    a, b = out["loop"], out["batched"]
    assert a[1:5] == b[1:5], (a[1:5], b[1:5])                                  # the lens is left in the same state, to the bit
    assert a[6] == b[6]                                                        # and the host generator at the same position
    margin(f"strict stack {H}x{W} S={S}: batched vs call-by-call PSF maps, max |d| / max", np.abs(a[0] - b[0]).max() / np.abs(a[0]).max(), 2e-6)
    print(f"strict stack {H}x{W} S={S}: {b[5]:.4f} s batched, {a[5]:.3f} s call by call")       # informative (no timing assertion)


def _psf_level_inputs(lens, S, grid, spp, seed):
    """Inputs of a strict level-3 call (object points, pupil points, tables) for S focus states, as aadff/strict_stack.py builds them."""
    from aadff import strict_stack as ss
    from deeplens.basics import DEFAULT_WAVE, GEO_SPP, WAVE_RGB
    L, N = len(WAVE_RGB), grid * grid
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(seed)
    wv = list(WAVE_RGB) + ([] if DEFAULT_WAVE in WAVE_RGB else [DEFAULT_WAVE])
    enp_z, enp_r = lens.entrance_pupil()
    pts = lens.point_source_grid(depth=-3000.0, grid=grid, quater=False).reshape(-1, 3).float()
    pobj = []
    for k in range(S):
        scale = -pts[:, 2] * np.tan(lens.hfov * (1 + 0.01 * k)) / lens.r_last
        p = pts.clone()
        p[..., 0] = pts[..., 0] * scale * lens.sensor_size[1] / 2
        p[..., 1] = pts[..., 1] * scale * lens.sensor_size[0] / 2
        pobj.append(p)
    B = S * L
    return dict(points=torch.stack(pobj).to(dev).contiguous(), N=N, B=B, wv=wv, tabs=ss._tables(lens, wv), tab_dev=lens._table(wv),
                pm=ss._pupil_points(torch.rand(B, spp, generator=g), torch.rand(B, spp, generator=g), enp_r, enp_z).to(dev).contiguous(),
                pc=ss._pupil_points(torch.rand(B, GEO_SPP, generator=g), torch.rand(B, GEO_SPP, generator=g), enp_r * 0.5, enp_z).to(dev).contiguous(),
                pset=torch.arange(S, dtype=torch.int32).repeat_interleave(L).to(dev),
                zs=torch.tensor([lens.d_sensor + 0.01 * k for k in range(S)], dtype=torch.float32).repeat_interleave(L).to(dev),
                bt_main=torch.arange(L, dtype=torch.int32).repeat(S).to(dev),
                bt_chief=torch.full((B,), wv.index(DEFAULT_WAVE), dtype=torch.int32, device=dev))


def test_strict_fused_trace_is_bit_equal_to_the_per_surface_form(repo_root):
    """Round 5: `aadff_trace_rays_strict_fused` (one launch, every surface in registers, predicted batch-wide Newton counts, iterates
    not evaluated past the point where they become periodic) against `aadff_trace_rays_strict_batched` (one launch pair per surface,
    counting passes) on a psf_map level: with the counts the per-surface form found, every ray's o / d / ra is BIT-identical, the
    any-bits the fused launch reports reproduce those counts, and `prediction_holds` accepts them; with one count off by one (too
    low, too high) the check rejects exactly that batch."""
    from aadff import strict_stack as ss
    dev = torch.device(DEV)
    lens = Lensgroup(lp(repo_root), sensor_res=(512, 512), device=DEV, parity="strict")
    S, grid, spp = 2, 5, 256
    a = _psf_level_inputs(lens, S, grid, spp, seed=11)
    N, B, n_surf = a["N"], a["B"], len(lens.surfaces)
    n = spp * N
    o0, d0, r0 = (torch.empty(B, n, 3, device=dev), torch.empty(B, n, 3, device=dev), torch.empty(B, n, device=dev))
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = ss._trace(o0, d0, r0, n, B, a["tabs"], len(a["wv"]), n_surf, a["bt_main"], True, flag, dev, a["points"], a["pset"], a["pm"], N, a["zs"])
    cnt = ss._masks_to_counts(scratch, B)
    curved = ss._curved(lens)
    assert cnt[:, curved].max() == 10 and cnt[:, curved].min() >= 1

    def fused(pred):
        o1, d1, r1 = torch.empty_like(o0), torch.empty_like(d0), torch.empty_like(r0)
        bits = torch.empty((B, 2, _abi.MAX_SURF), dtype=torch.int32, device=dev)
        _abi.call("aadff_trace_rays_strict_fused", _abi.ptr(o1), _abi.ptr(d1), _abi.ptr(r1), n, B, _abi.ptr(a["tab_dev"]), len(a["wv"]), n_surf,
                  _abi.ptr(a["bt_main"]), _abi.ptr(a["points"]), _abi.ptr(a["pset"]), _abi.ptr(a["pm"]), N, 0, n_surf, 1, _abi.ptr(a["zs"]),
                  _abi.ptr(torch.from_numpy(np.ascontiguousarray(pred, dtype=np.int32)).to(dev)), _abi.ptr(bits), 0, 0, None, None, None, _abi.stream_ptr(dev))
        return o1, d1, r1, bits.cpu().numpy().view(np.uint32)

    o1, d1, r1, hb = fused(cnt)
    for x, y, name in ((o0, o1, "o"), (d0, d1, "d"), (r0, r1, "ra")):
        assert torch.equal(x.view(torch.int32), y.view(torch.int32)), f"{name}: {int((x.view(torch.int32) != y.view(torch.int32)).sum())} words differ"
    assert ss.prediction_holds(hb[:, 0], cnt, curved).all()
    full = scratch[:B * _abi.MAX_SURF].cpu().numpy().view(np.uint32).reshape(B, _abi.MAX_SURF)
    ran = (np.uint32(1) << cnt.astype(np.uint32)) - np.uint32(1)
    assert np.array_equal(hb[:, 0] & ran, full & ran)                          # the bits of the iterations that ran are the counting pass's
    assert not hb[:, 1].any()
    # mispredictions: batch 1 one iteration short at a surface that needs ten, batch 2 one too many at a surface that needs fewer
    s10 = int(np.nonzero(curved & (cnt[1] == 10))[0][0])
    few = np.nonzero(curved & (cnt[2] < 10))[0]
    bad = cnt.copy()
    bad[1, s10] = 9
    bad[2, few[0]] += 1
    _, _, _, hb2 = fused(bad)
    assert ss.prediction_holds(hb2[:, 0], bad, curved).tolist() == [b not in (1, 2) for b in range(B)]


@pytest.mark.parametrize("res,S,grid,spp", [((256, 256), 4, 5, 512), ((1024, 1024), 10, 11, 2048)])
def test_strict_stack_fused_equals_batched(repo_root, margin, res, S, grid, spp):
    """Round 5: a strict stack with every level in ONE launch on speculated Newton counts (aadff/strict_stack.py, csrc/strict_fused.hip)
    against the round-4 form (one launch pair per surface): identical d_sensor / hfov / foclen / fnum and host generator position,
    PSF maps equal to the float atomics of the histogram; the first call seeds the count table, the second runs fused with no replay;
    a POISONED table (one count wrong per level) is detected, the batches are replayed, the rows corrected, and the maps are the same."""
    import time
    from aadff import strict_stack as ss
    H, W = res
    depth = synth_depth_mm(H, W, seed=5678)
    dbar, fds = -float(depth.mean()), [float(f) for f in -np.linspace(depth.min(), depth.max(), S)]
    out = {}
    for label in ("batched", "fused"):
        lens = Lensgroup(lp(repo_root), sensor_res=(H, W), device=DEV, parity="strict")
        torch.manual_seed(3)
        ss.strict_psf_maps(lens, dbar, fds, grid, 11, spp, fused=label == "fused")          # seeds the table / warms the allocator
        torch.manual_seed(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        maps = ss.strict_psf_maps(lens, dbar, fds, grid, 11, spp, fused=label == "fused")
        torch.cuda.synchronize()
        out[label] = (maps.cpu().numpy(), lens.d_sensor, lens.hfov, lens.foclen, lens.fnum, time.perf_counter() - t0, torch.rand(1).item())
        if label == "fused":
            counts = ss.StrictCounts.of(lens)
            want_stats = {"seeded": 1, "fused": 3, "replayed_batches": 0, "fused_replays": 0, "per_surface_replays": 0}
            assert {k: counts.stats[k] for k in want_stats} == want_stats and counts.stats.get("native_fallbacks", 0) == 0, counts.stats
            good = {k: v.copy() for k, v in counts.rows.items()}
            curved = ss._curved(lens)
            for k, v in counts.rows.items():                                     # poison: one curved surface of one batch per level
                v.reshape(-1, _abi.MAX_SURF)[1, np.nonzero(curved)[0][2]] = 7
            torch.manual_seed(3)
            maps2 = ss.strict_psf_maps(lens, dbar, fds, grid, 11, spp, fused=True)
            assert counts.stats["fused_replays"] + counts.stats["per_surface_replays"] >= 3 and counts.stats["replayed_batches"] >= 1, counts.stats
            for k in good:
                assert np.array_equal(counts.rows[k], good[k]), k                # rows repaired
            assert (lens.d_sensor, lens.hfov) == out[label][1:3]
            margin(f"strict stack {H}x{W} S={S}: fused after a poisoned table vs fused, max |d| / max",
                   np.abs(maps2.cpu().numpy() - out[label][0]).max() / np.abs(out[label][0]).max(), 2e-6)
    a, b = out["batched"], out["fused"]
    assert a[1:5] == b[1:5], (a[1:5], b[1:5])
    assert a[6] == b[6]
    margin(f"strict stack {H}x{W} S={S}: fused vs per-surface PSF maps, max |d| / max", np.abs(a[0] - b[0]).max() / np.abs(a[0]).max(), 2e-6)
    print(f"strict stack {H}x{W} S={S}: {b[5] * 1e3:.2f} ms fused, {a[5] * 1e3:.2f} ms per-surface form")


def test_strict_lens_through_the_sharded_unit_renderer(repo_root):
    """a strict-parity lens through SceneUnitRenderer (its direct path hands `dest` to render_focal_stack_m1, which the
    strict branch used to ignore: uninitialised units).  Every unit - whole scenes and a rank's share of odd slices, written by the
    strided convolution into the caller's buffer - equals the plain strict stack of its scene, and update_lens=False leaves the
    lens where it was."""
    from aadff.focal_stack import SceneUnitRenderer
    H = W = 128
    S, GRID, SPP = 4, 3, 256
    lens = Lensgroup(lp(repo_root), sensor_res=(H, W), device=DEV, parity="strict")
    scenes = []
    for sc in range(2):
        depth = synth_depth_mm(H, W, seed=900 + sc)
        scenes.append((tt(synth_rgb(H, W, seed=800 + sc))[None].to(DEV), -float(depth.mean()), -np.linspace(depth.min(), depth.max(), S)))
    plain = []
    for sc, (img, dbar, fds) in enumerate(scenes):
        torch.manual_seed(sc)
        plain.append(render_focal_stack_m1(lens, img, dbar, fds, GRID, 11, SPP)[0].permute(1, 0, 2, 3).clone())      # [S,3,H,W]
    plain = torch.cat(plain)
    before = (lens.d_sensor, lens.hfov)
    rend = SceneUnitRenderer(lens, scenes, S, GRID, 11, SPP)
    out = torch.full((2 * S, 3, H, W), float("nan"), device=DEV)
    rend.render(list(range(2 * S)), out=out)
    assert (out - plain).abs().max().item() <= 2e-6
    odd = [1, 3, 5, 7]
    out2 = torch.full((len(odd), 3, H, W), float("nan"), device=DEV)
    rend.render(odd, out=out2)
    assert (out2 - plain[odd]).abs().max().item() <= 2e-6
    assert (lens.d_sensor, lens.hfov) == before


def test_strict_trace_reproduces_reference_bits(golden_dir, repo_root, margin):
    """The strict tracer against the reference's per-surface ray states (G2) and load / refocus scalars (G1), BIT for bit
    where the reference's arithmetic can be reproduced at all: identical validity at every surface, the first surfaces'
    states equal on every ray, >= 85 % of the rays still bit-equal behind all 12 surfaces (the rest: torch's CPU sqrt is
    MKL's, 0.7 % of its results are not correctly rounded), d_sensor and hfov of the loaded lens equal to the last bit."""
    import json as _json
    g = np.load(os.path.join(golden_dir, "g2_g3_trace_splat.npz"))
    g1 = _json.load(open(os.path.join(golden_dir, "g1_scalars.json")))["rf50mm@1024x1024"]
    from deeplens.basics import Ray
    lens = Lensgroup(lp(repo_root), sensor_res=(1024, 1024), device=DEV, parity="strict")
    assert lens.d_sensor == g1["load"]["d_sensor"] and lens.hfov == g1["load"]["hfov"]
    assert lens.exit_pupil() == pytest.approx(tuple(g1["exit_pupil"]), rel=1e-7) and lens.entrance_pupil() == pytest.approx(tuple(g1["entrance_pupil"]), rel=1e-7)
    exact = 0
    for f, want in g1["refocus"].items():
        torch.manual_seed(0)
        lens.refocus(float(f))
        assert lens.d_sensor == pytest.approx(want["d_sensor"], rel=2e-7) and lens.hfov == pytest.approx(want["hfov"], rel=2e-7)
        exact += int(lens.d_sensor == want["d_sensor"]) + int(lens.hfov == want["hfov"])
    assert exact >= 6, f"only {exact} of 10 refocus scalars reproduce to the bit"
    torch.manual_seed(0)
    lens.refocus(-2000.0)
    ray = Ray(tt(g["ray_o0"]).clone(), tt(g["ray_d0"]).clone(), wvln=0.589, device="cpu")
    for i in range(len(lens.surfaces)):
        ray, _, _ = lens.trace(ray, lens_range=range(i, i + 1))
        ra = ray.ra.numpy()
        assert np.array_equal(ra, g["states_ra"][i]), f"surface {i} validity"
        alive = ra > 0
        same = ((ray.o.numpy() == g["states_o"][i]).all(-1) & (ray.d.numpy() == g["states_d"][i]).all(-1))[alive].mean()
        if i < 2:
            assert same == 1.0, f"surface {i}: {same}"
        assert np.abs(ray.o.numpy() - g["states_o"][i])[alive].max() <= 1e-5
    margin("strict trace: fraction of rays NOT bit-equal to the reference behind all 12 surfaces", 1.0 - same, 0.15)
    # a whole PSF map through the strict path (G4: reference output, seed 0, refocus(-2000), depth -1500, spp 2048)
    g4 = np.load(os.path.join(golden_dir, "g4_psf_map.npz"))
    torch.manual_seed(0)
    lens.refocus(-2000.0)
    pm = lens.psf_map(depth=-1500.0, grid=11, ks=11, spp=2048)
    assert lens.d_sensor == float(g4["rf50mm_d_sensor"])
    margin("strict mode: psf_map rf50mm rel-L2 vs reference (fast path: 6e-4)", rel(pm.cpu().numpy(), g4["rf50mm_psf_map"]), 2e-4)


def test_margin_training_data(golden_dir, repo_root, margin):
    """PSFNet.get_training_data vs the reference's (G10): identical network inputs (host RNG order: np choice, refocus
    draws, rand x, rand y, randn z, psf draws), ray-traced target PSFs within the PSF tolerance."""
    g = np.load(os.path.join(golden_dir, "g10_training_data.npz"))
    net = PSFNet(lp(repo_root), sensor_res=(480, 640), kernel_size=11, device=DEV)
    np.random.seed(0)
    torch.manual_seed(0)
    for i in range(2):
        inp, psf = net.get_training_data(bs=16, spp=256)
        assert np.array_equal(inp.cpu().numpy(), g[f"inp_{i}"]), "network inputs must be bit-identical (host RNG order)"
        margin(f"get_training_data call {i}: d_sensor relative", abs(net.d_sensor / float(g[f'd_sensor_{i}']) - 1), 1e-5)
        margin(f"get_training_data call {i}: target PSFs rel-L2", rel(psf.cpu().numpy(), g[f"psf_{i}"]), 2e-3)


def test_bf16_train_step_runs_and_tracks_fp32(repo_root, margin):
    """Config 4 (1_fit_psfnet.py, bf16): the graph-captured bf16-autocast step that `bench.py --mode fit` times, against
    the same steps in fp32: the loss must fall and the bf16 trajectory must stay close to the fp32 one."""
    dev = torch.device(DEV)
    net = PSFNet(lp(repo_root), sensor_res=(128, 128), kernel_size=11, device=DEV)
    sd = {k: torch.from_numpy(v) for k, v in mlp_state_dict(seed=4321).items()}
    np.random.seed(1)
    torch.manual_seed(1)
    data = [net.get_training_data(bs=64, spp=256) for _ in range(3)]
    data = [(a.to(dev), b.to(dev)) for a, b in data]
    losses = {}
    for bf16 in (False, True):
        net.psfnet.load_state_dict(sd)
        step = _TrainStep(net.psfnet, 1e-3, 100, 64, 121, dev, bf16, True)
        ls = []
        for it in range(12):
            inp, psf = data[it % 3]
            pred = step(inp, psf)
            ls.append(float(((pred.detach().float() - psf) ** 2).mean()))
        torch.cuda.synchronize()
        assert step.fused is not None or step.graph is not None or step.use_graph == "eager-static"
        assert np.isfinite(ls).all() and all(ls[9 + b] < ls[b] for b in range(3)), ls      # batch b: its 4th visit vs its 1st
        losses[bf16] = np.array(ls)
    margin("bf16 train step: |loss_bf16/loss_fp32 - 1| after 12 steps", abs(losses[True][-1] / losses[False][-1] - 1), 0.05)


def test_strict_per_call_api_fused_equals_per_surface(repo_root, margin, monkeypatch):
    """`refocus` / `psf_map` of a strict lens (the reference's own slice loop, deeplens/optics.py:779-783 per slice) run the refocus +
    calc_fov levels and the psf_map level of the fused strict stack for ONE state each (round 5; `AADFF_STRICT_CALLS_FUSED=0`: every
    trace a chain of per-surface launches, round 3): same draws in the same order, d_sensor / hfov / foclen / fnum equal to the last
    bit after every refocus, PSF maps equal to the histograms' float atomics, the generator left at the same position - over focus
    distances whose batch-wide Newton counts differ (the count table's candidate rows and re-launches are exercised)."""
    import time
    H = W = 256
    grid, ks, spp = 5, 11, 512
    depth = synth_depth_mm(H, W, seed=5678)
    dbar, fds = -float(depth.mean()), [float(f) for f in -np.linspace(depth.min(), depth.max(), 6)]
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("AADFF_STRICT_CALLS_FUSED", mode)
        lens = Lensgroup(lp(repo_root), sensor_res=(H, W), device=DEV, parity="strict")
        torch.manual_seed(33)
        outs, t0 = [], time.perf_counter()
        for rep in range(2):                                 # second pass: every (level, shape) has its count table
            for f in fds:
                lens.refocus(f)
                outs.append(torch.tensor([lens.d_sensor, lens.hfov, lens.foclen, lens.fnum], dtype=torch.float64))
                outs.append(lens.psf_map(depth=dbar, grid=grid, ks=ks, spp=spp).cpu().double())
        res[mode] = (outs, torch.rand(1).item(), time.perf_counter() - t0)
    monkeypatch.delenv("AADFF_STRICT_CALLS_FUSED")
    assert res["0"][1] == res["1"][1]
    worst = 0.0
    for a, b in zip(res["1"][0], res["0"][0]):
        if a.numel() == 4:
            assert torch.equal(a, b), (a, b)
        else:
            assert a.shape == b.shape == (3, grid * ks, grid * ks)
            worst = max(worst, float((a - b).abs().max() / b.abs().max()))
    margin("strict per-call API: fused halves vs per-surface calls, PSF maps max |d| / max", worst, 2e-6)
    print(f"strict per-call loop, 12 slices at 256^2 / grid 5 / spp 512: per-surface {res['0'][2]:.3f} s, fused {res['1'][2]:.3f} s")


def test_strict_pipeline_equals_the_sequential_loop(repo_root, margin):
    """aadff.strict_stack.StrictPipeline (round 5): two / four strict stacks in flight (as many lenses / streams, one host thread: at every
    host wait of a stack the host goes on with another one whose awaited launch has finished).  The host draws are taken at submission, so every
    stack equals the one the sequential loop renders from the same generator (PSF histograms are float atomics: 2e-6 of the peak),
    the generator ends where the loop leaves it, and results may be asked for in any order."""
    import time
    from aadff.strict_stack import StrictPipeline
    H = W = 256
    S, grid, spp, n = 4, 5, 512, 6
    depth = synth_depth_mm(H, W, seed=5678)
    dbar, fds = -float(depth.mean()), [float(f) for f in -np.linspace(depth.min(), depth.max(), S)]
    img = tt(synth_rgb(H, W, seed=3))[None].to(DEV)
    make = lambda: Lensgroup(lp(repo_root), sensor_res=(H, W), device=DEV, parity="strict")
    seq_lens = make()
    torch.manual_seed(21)
    want = [render_focal_stack_m1(seq_lens, img, dbar, fds, grid, 11, spp).clone() for _ in range(n)]
    tail = torch.rand(1).item()
    worst = 0.0
    for depth_n in (2, 4):
        pipe = StrictPipeline(make, depth=depth_n)
        torch.manual_seed(21)
        futs = [pipe.submit(img, dbar, fds, grid, 11, spp) for _ in range(n)]
        got = [None] * n
        for k in (1, 0, 2, 5, 4, 3):
            out, ev = futs[k].result()
            ev.synchronize()
            got[k] = out
        pipe.close()
        assert not pipe.pending
        assert torch.rand(1).item() == tail
        worst = max(worst, max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(got, want)))
    margin("strict pipeline (2 and 4 stacks in flight) vs sequential loop, max |d| / max over 6 stacks", worst, 2e-6)
    # a stack that fails (a focus distance in front of the lens: the reference's "sensor position is negative." / no valid ray) raises
    # from ITS handle; the stacks around it are rendered, the pipeline stays usable
    with StrictPipeline(make, depth=2) as pipe:
        torch.manual_seed(21)
        a = pipe.submit(img, dbar, fds, grid, 11, spp)
        bad = pipe.submit(img, dbar, [fds[0], 5.0], grid, 11, spp)
        c = pipe.submit(img, dbar, fds, grid, 11, spp)
        assert c.result()[0].shape == want[0].shape and a.result()[0].shape == want[0].shape
        with pytest.raises((AssertionError, FloatingPointError)):
            bad.result()
        d = pipe.submit(img, dbar, fds, grid, 11, spp)
        assert torch.isfinite(d.result()[0]).all()
