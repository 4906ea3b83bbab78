#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference, which never travels to the
GPU box).  Nothing from the reference is copied: this script calls the reference's
public functions on seeded inputs and stores inputs + outputs as small .npz/.json
fixtures.  The fixtures pin `oracle/` (tests/test_oracle_golden.py); the HIP path is
then compared with the oracle and with these fixtures (tests/test_gpu_*.py).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Import-time stubs (none of them is arithmetic on the hot path):
cv2, lpips, skimage.metrics, torchvision.{utils,transforms.functional}.  The only
stub with behaviour is torchvision.utils.make_grid(t, nrow, padding=0), which with
padding 0 is pure tiling (reference call site: deeplens/optics.py:1025).
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
REF = os.environ.get("AADFF_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))

import numpy as np
import torch


def _install_stubs():
    cv2 = types.ModuleType("cv2")
    sys.modules["cv2"] = cv2
    sys.modules["lpips"] = types.ModuleType("lpips")
    sk = types.ModuleType("skimage")
    skm = types.ModuleType("skimage.metrics")
    skm.peak_signal_noise_ratio = lambda *a, **k: 0.0
    skm.structural_similarity = lambda *a, **k: 0.0
    sk.metrics = skm
    sys.modules["skimage"] = sk
    sys.modules["skimage.metrics"] = skm

    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")

    def make_grid(tensor, nrow=8, padding=0, pad_value=0.0, **kw):
        # torchvision semantics for a 4-D batch with padding=0: tile row-major,
        # nrow images per row; 1-channel inputs are repeated to 3 channels.
        assert padding == 0
        if tensor.dim() == 4 and tensor.shape[1] == 1:
            tensor = torch.cat((tensor, tensor, tensor), 1)
        n, c, h, w = tensor.shape
        xmaps = min(nrow, n)
        ymaps = int(np.ceil(float(n) / xmaps))
        grid = tensor.new_full((c, h * ymaps, w * xmaps), pad_value)
        k = 0
        for yy in range(ymaps):
            for xx in range(xmaps):
                if k >= n:
                    break
                grid[:, yy * h:(yy + 1) * h, xx * w:(xx + 1) * w] = tensor[k]
                k += 1
        return grid

    tvu.make_grid = make_grid
    tvu.save_image = lambda *a, **k: None
    tv.utils = tvu
    tvt.functional = tvf
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.utils"] = tvu
    sys.modules["torchvision.transforms"] = tvt
    sys.modules["torchvision.transforms.functional"] = tvf


_install_stubs()
sys.path.insert(0, REF)
from deeplens.optics import Lensgroup  # noqa: E402
from deeplens.basics import Ray, Material, WAVE_RGB, DEFAULT_WAVE, GEO_SPP  # noqa: E402
from deeplens.monte_carlo import forward_integral  # noqa: E402
import importlib  # noqa: E402
import importlib.util  # noqa: E402
ref_render = importlib.import_module("deeplens.render_psf")  # the package re-exports a same-named function
from deeplens.psfnet import PSFNet, ThinLens  # noqa: E402
from deeplens.psfnet_arch import MLP  # noqa: E402

CPU = torch.device("cpu")
sys.path.insert(0, os.path.join(REPO, "aberration-aware-depth-from-focus_amd"))
from aadff.synth import synth_rgb, synth_depth_mm, mlp_state_dict  # noqa: E402  (shared seeded inputs)


def restate_lens_json(name):
    """Write the lens prescription (data, not code) in the reference's JSON schema,
    keeping only the keys `read_lens_json` consumes (deeplens/optics.py:2045-2070)."""
    src = json.load(open(f"{REF}/lenses/{name}/lens.json"))
    out = {"r_last": src["r_last"], "d_sensor": src["d_sensor"], "surfaces": []}
    for s in src["surfaces"]:
        t = {"type": s["type"], "r": s["r"], "c": s["c"], "d": s["d"],
             "mat1": s["mat1"], "mat2": s["mat2"]}
        if s["type"] == "Aspheric":
            t["k"] = s["k"]
            t["ai"] = s["ai"]
        out["surfaces"].append(t)
    os.makedirs(f"{REPO}/lenses/{name}", exist_ok=True)
    with open(f"{REPO}/lenses/{name}/lens.json", "w") as f:
        json.dump(out, f, indent=1)


def lens_scalars(lens):
    return dict(d_sensor=float(lens.d_sensor), hfov=float(lens.hfov), foclen=float(lens.foclen),
                fnum=float(lens.fnum), aper_idx=int(lens.aper_idx), pixel_size=float(lens.pixel_size),
                sensor_size=[float(v) for v in lens.sensor_size], r_last=float(lens.r_last))


def g1_scalars():
    out = {}
    for name, res in (("rf50mm", (1024, 1024)), ("rf50mm", (480, 640)), ("50mm_f2.8", (1024, 1024))):
        lens = Lensgroup(filename=f"{REF}/lenses/{name}/lens.json", sensor_res=res, device=CPU)
        rec = {"load": lens_scalars(lens)}
        rec["entrance_pupil"] = [float(v) for v in lens.entrance_pupil()]
        rec["exit_pupil"] = [float(v) for v in lens.exit_pupil()]
        rec["entrance_pupil_shrunk"] = [float(v) for v in lens.entrance_pupil(shrink_pupil=True)]
        rec["exit_pupil_shrunk"] = [float(v) for v in lens.exit_pupil(shrink_pupil=True)]
        mats = sorted({s.mat1.name for s in lens.surfaces} | {s.mat2.name for s in lens.surfaces})
        rec["ior"] = {m: [float(Material(m).ior(w)) for w in WAVE_RGB] for m in mats}
        rec["refocus"] = {}
        for f in (-500., -1000., -2000., -5000., -20000.):
            lens = Lensgroup(filename=f"{REF}/lenses/{name}/lens.json", sensor_res=res, device=CPU)
            torch.manual_seed(0)
            lens.refocus(f)
            rec["refocus"][str(int(f))] = lens_scalars(lens)
        out[f"{name}@{res[0]}x{res[1]}"] = rec
    with open(f"{HERE}/g1_scalars.json", "w") as f:
        json.dump(out, f, indent=1)


def g2_g3_trace_and_splat():
    """Per-surface ray states, sensor hits, chief-ray centres and splatted PSFs."""
    lens = Lensgroup(filename=f"{REF}/lenses/rf50mm/lens.json", sensor_res=(1024, 1024), device=CPU)
    torch.manual_seed(0)
    lens.refocus(-2000.)
    ks, spp = 11, 256
    pts = lens.point_source_grid(depth=-1500., grid=11).reshape(-1, 3)
    # object-space points exactly as psf_diff builds them (deeplens/optics.py:953-959)
    scale = lens.calc_scale_pinhole(pts[:, 2])
    pobj = pts.clone()
    pobj[..., 0] = pts[..., 0] * scale * lens.sensor_size[1] / 2
    pobj[..., 1] = pts[..., 1] * scale * lens.sensor_size[0] / 2

    gen_state = torch.get_rng_state()
    u_theta = torch.rand(spp)
    u_r = torch.rand(spp)
    torch.set_rng_state(gen_state)
    ray = lens.sample_from_points(o=pobj, spp=spp, wvln=0.589)
    o0, d0 = ray.o.clone(), ray.d.clone()

    # per-surface states for a 64-ray subset (samples 0..7 x points {0,12,60,61,108,120,5,115})
    psel = [0, 12, 60, 61, 108, 120, 5, 115]
    sub = Ray(o0[:8][:, psel].clone(), d0[:8][:, psel].clone(), wvln=0.589, device=CPU)
    states_o, states_d, states_ra = [], [], []
    for s in lens.surfaces:
        sub = s.ray_reaction(sub)
        states_o.append(sub.o.clone().numpy())
        states_d.append(sub.d.clone().numpy())
        states_ra.append(sub.ra.clone().numpy())

    ray = lens.trace2sensor(ray)
    # chief-ray centres with their own stored draws
    gen_state = torch.get_rng_state()
    c_theta = torch.rand(GEO_SPP)
    c_r = torch.rand(GEO_SPP)
    torch.set_rng_state(gen_state)
    centre = lens.psf_center(pobj)
    psf_raw = forward_integral(ray, ps=lens.pixel_size, ks=ks, pointc_ref=centre)
    psf = psf_raw / psf_raw.sum(-1).sum(-1).unsqueeze(-1).unsqueeze(-1)

    # backward case: the 32 entrance-pupil rays through the front group (deeplens/optics.py:1333-1366)
    M = 32
    aper = lens.surfaces[lens.aper_idx]
    phi = torch.arange(-0.5, 0.5, 1.0 / M)
    bo = torch.tensor([[aper.r, 0, aper.d.item()]]).repeat(M, 1).to(torch.float32)
    bd = torch.stack((torch.sin(phi), torch.zeros_like(phi), -torch.cos(phi)), axis=-1)
    bray = Ray(bo, bd, device=CPU)
    bray, _, _ = lens.trace(bray, lens_range=range(0, lens.aper_idx))

    np.savez_compressed(
        f"{HERE}/g2_g3_trace_splat.npz",
        d_sensor=np.float64(lens.d_sensor), hfov=np.float64(lens.hfov), pixel_size=np.float64(lens.pixel_size),
        points=pts.numpy(), points_obj=pobj.numpy(), u_theta=u_theta.numpy(), u_r=u_r.numpy(),
        ray_o0=o0[:8][:, psel].numpy(), ray_d0=d0[:8][:, psel].numpy(), psel=np.array(psel),
        states_o=np.stack(states_o), states_d=np.stack(states_d), states_ra=np.stack(states_ra),
        sensor_xy=ray.o[..., :2].numpy(), sensor_ra=ray.ra.numpy().astype(np.uint8),
        final_d=ray.d.numpy().astype(np.float32),
        c_theta=c_theta.numpy(), c_r=c_r.numpy(), centre=centre.numpy(),
        psf_raw=psf_raw.numpy(), psf=psf.numpy(),
        back_o=bray.o.numpy(), back_d=bray.d.numpy(), back_ra=bray.ra.numpy(),
    )


def g4_psf_map():
    out = {}
    for name, res, foc, depth, spp in (("rf50mm", (1024, 1024), -2000., -1500., 2048),
                                       ("50mm_f2.8", (256, 256), -1000., -1250., 512)):
        lens = Lensgroup(filename=f"{REF}/lenses/{name}/lens.json", sensor_res=res, device=CPU)
        torch.manual_seed(0)
        lens.refocus(foc)
        st = torch.get_rng_state()
        draws = []
        for _ in WAVE_RGB:                      # RNG order: SURVEY.md Appendix B
            draws += [torch.rand(spp), torch.rand(spp), torch.rand(GEO_SPP), torch.rand(GEO_SPP)]
        torch.set_rng_state(st)
        pm = lens.psf_map(depth=depth, grid=11, ks=11, spp=spp)
        key = name.replace(".", "_")
        out[f"{key}_psf_map"] = pm.numpy()
        out[f"{key}_draws_main"] = np.stack([torch.stack((draws[4 * i], draws[4 * i + 1])).numpy() for i in range(3)])
        out[f"{key}_draws_chief"] = np.stack([torch.stack((draws[4 * i + 2], draws[4 * i + 3])).numpy() for i in range(3)])
        out[f"{key}_d_sensor"] = np.float64(lens.d_sensor)
        out[f"{key}_hfov"] = np.float64(lens.hfov)
    # single-point / list-input forms of psf() (deeplens/optics.py:945-951,980-981)
    lens = Lensgroup(filename=f"{REF}/lenses/rf50mm/lens.json", sensor_res=(480, 640), device=CPU)
    torch.manual_seed(3)
    out["single_point_psf"] = lens.psf([0.3, -0.4, -1200.], ks=11, spp=1024).numpy()
    torch.manual_seed(3)
    out["nocenter_psf"] = lens.psf(torch.tensor([[0.3, -0.4, -1200.], [0., 0., -3000.]]), ks=11, spp=1024, center=False).numpy()
    np.savez_compressed(f"{HERE}/g4_psf_map.npz", **out)


def g5_conv():
    out = {}
    rng = np.random.Generator(np.random.PCG64(99))

    def rnd(*shape):
        return torch.from_numpy(rng.random(shape, dtype=np.float32))

    def psfmap(c, g, ks):
        p = rnd(c, g, g, ks, ks)
        p = p / p.sum((-1, -2), keepdim=True)
        return p.permute(0, 1, 3, 2, 4).reshape(c, g * ks, g * ks).contiguous()

    # small full-tensor cases, incl. H/W not divisible by grid, B>1, ks 3/5/11/21
    for tag, (b, c, h, w, g, ks) in {"a": (2, 3, 50, 50, 5, 11), "b": (1, 3, 37, 53, 4, 5),
                                      "c": (1, 3, 12, 12, 3, 3), "d": (1, 3, 64, 48, 7, 21),
                                      "e": (1, 1, 40, 40, 1, 11)}.items():
        img, pm = rnd(b, c, h, w), psfmap(c, g, ks)
        out[f"map_{tag}_img"], out[f"map_{tag}_psf"], out[f"map_{tag}_grid"] = img.numpy(), pm.numpy(), np.int64(g)
        out[f"map_{tag}_out"] = ref_render.render_psf_map(img, pm, g).numpy()
    for tag, (b, c, h, w, ks) in {"a": (2, 3, 40, 56, 11), "b": (1, 3, 16, 16, 7)}.items():
        img = rnd(b, c, h, w)
        p = rnd(c, ks, ks)
        p = p / p.sum((-1, -2), keepdim=True)
        out[f"uni_{tag}_img"], out[f"uni_{tag}_psf"] = img.numpy(), p.numpy()
        out[f"uni_{tag}_out"] = ref_render.render_psf(img, p).numpy()
    for tag, (b, c, h, w, ks) in {"a": (2, 3, 14, 20, 11), "b": (1, 3, 12, 12, 5), "c": (1, 1, 20, 9, 3)}.items():
        img = rnd(b, c, h, w)
        p = rnd(b, h, w, ks, ks)
        p = p / p.sum((-1, -2), keepdim=True)
        out[f"loc_{tag}_img"], out[f"loc_{tag}_psf"] = img.numpy(), p.numpy()
        out[f"loc_{tag}_out"] = ref_render.local_psf_render(img, p, kernel_size=ks).numpy()
    # 3-D input is auto-unsqueezed (deeplens/render_psf.py:89-90)
    img3 = rnd(3, 12, 14)
    p3 = rnd(1, 12, 14, 5, 5)
    out["loc_3d_img"], out["loc_3d_psf"] = img3.numpy(), p3.numpy()
    out["loc_3d_out"] = ref_render.local_psf_render(img3, p3, kernel_size=5).numpy()
    # tiled variant without halo -> seams (deeplens/render_psf.py:110-127): one 2x2-tile case
    img = rnd(1, 3, 26, 36)
    p = rnd(1, 26, 36, 11, 11)
    p = p / p.sum((-1, -2), keepdim=True)
    out["hr_img"], out["hr_psf"] = img.numpy(), p.numpy()
    out["hr_out"] = ref_render.local_psf_render_high_res(img, p, patch_size=[16, 20], kernel_size=11).numpy()
    np.savez_compressed(f"{HERE}/g5_conv_small.npz", **out)

    # 1024^2 procedural image, grid 11, ks 11: store crops at patch seams + fp64 sums
    img = torch.from_numpy(synth_rgb(1024, 1024))[None]
    pm = psfmap(3, 11, 11)
    res = ref_render.render_psf_map(img, pm, 11)[0].numpy()
    crops = {}
    for (y, x) in ((0, 0), (61, 61), (900, 340), (960, 960), (433, 715)):
        crops[f"{y}_{x}"] = res[:, y:y + 64, x:x + 64]
    np.savez_compressed(f"{HERE}/g5_conv_1024.npz", psf_map=pm.numpy(),
                        sums=res.astype(np.float64).sum((1, 2)), abs_sums=np.abs(res).astype(np.float64).sum((1, 2)),
                        **{f"crop_{k}": v for k, v in crops.items()})


def g6_g7_psfnet():
    sd = mlp_state_dict(seed=4321)
    net = MLP(in_features=4, out_features=121, hidden_features=256, hidden_layers=8)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    rng = np.random.Generator(np.random.PCG64(7))
    x = rng.random((1024, 4), dtype=np.float32)
    x[:, :2] = x[:, :2] * 2 - 1
    with torch.no_grad():
        y = net(torch.from_numpy(x)).numpy()
    out = {"mlp_in": x, "mlp_out": y}

    # G7: PSFNet.render on a 64x64 RGB-D, 5 focus distances; 4-D and 3-D image branches
    lens = PSFNet(filename=f"{REF}/lenses/rf50mm/lens.json", sensor_res=(64, 64), kernel_size=11, device="cpu")
    lens.psfnet.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    img = torch.from_numpy(synth_rgb(64, 64, seed=11))[None]
    depth = -torch.from_numpy(synth_depth_mm(64, 64, seed=12))[None, None]
    fds = [-500., -800., -1500., -3000., -5000.]
    outs = [lens.render(img, depth, torch.tensor([f])).numpy() for f in fds]
    out["render_fds"] = np.array(fds, dtype=np.float32)
    out["render_out"] = np.concatenate(outs, 0)
    out["render3d_out"] = lens.render(img[0], depth[0, 0], -1500.).numpy()
    # batched 4-D branch, two images with different focus distances
    img2 = torch.cat((img, torch.flip(img, [-1])), 0)
    depth2 = torch.cat((depth, torch.flip(depth, [-2])), 0)
    out["render_b2_out"] = lens.render(img2, depth2, torch.tensor([-700., -2500.])).numpy()

    # ThinLens baseline (deeplens/psfnet.py:489-570), 4-D branch
    thin = ThinLens(foc_len=50.0, fnum=1.8, kernel_size=11, sensor_size=[24.0, 24.0], sensor_res=(64, 64))
    out["thin_out"] = thin.render(img, depth, torch.tensor([-1500.])).numpy()
    out["thin_coc"] = thin.coc(depth, torch.full_like(depth, -1500.)).numpy()
    np.savez_compressed(f"{HERE}/g6_g7_psfnet.npz", **out)


def g7b_config5_stack():
    """G7b (round 4): BASELINE.json config 5's render at ITS size - configs/aber_aware_dff_dfv.yml:19-21 (bs 2, n_stack 8,
    res 480x640, ks 11) through the loop of 2_aber_aware_dff_dfv.py:101-107: select_focus_dist(depth, 8, 'linear') then
    PSFNet.render(aif, -depth*1e3, -foc_dist*1e3) per slice, stacked on dim 2.  Procedural weights (aadff.synth.mlp_state_dict) and
    scenes; stored: the focus distances, three 64x64 crops, 16x16 block means and fp64 sums of every (sample, slice)."""
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    spec = importlib.util.spec_from_file_location("ref_dff_utils", f"{REF}/dff/utils.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    H, W, B, S = 480, 640, 2, 8
    lens = PSFNet(filename=f"{REF}/lenses/rf50mm/lens.json", sensor_res=(H, W), kernel_size=11, device="cpu")
    lens.psfnet.load_state_dict({k: torch.from_numpy(v) for k, v in mlp_state_dict(seed=4321).items()})
    aif = torch.from_numpy(np.stack([synth_rgb(H, W, seed=31 + b) for b in range(B)]))
    depth = torch.from_numpy(np.stack([synth_depth_mm(H, W, seed=41 + b) for b in range(B)]))[:, None] / 1e3      # metres
    focus_dists = mod.select_focus_dist(depth, S, mode="linear")
    crops, blocks, sums = {"a": [], "b": [], "c": []}, [], []
    for i in range(S):
        with torch.no_grad():
            sl = lens.render(aif, depth=-depth * 1e3, foc_dist=-focus_dists[:, i] * 1e3).numpy()        # [B,3,H,W]
        crops["a"].append(sl[:, :, 0:64, 0:64])
        crops["b"].append(sl[:, :, 208:272, 288:352])
        crops["c"].append(sl[:, :, 416:480, 576:640])
        blocks.append(sl.astype(np.float64).reshape(B, 3, H // 16, 16, W // 16, 16).mean((3, 5)).astype(np.float32))
        sums.append(sl.astype(np.float64).sum((2, 3)))
        print("G7b slice", i, flush=True)
    np.savez_compressed(f"{HERE}/g7b_config5_stack.npz", focus_dists=focus_dists.numpy(), block_means=np.stack(blocks, 2),
                        sums=np.stack(sums, 2), **{f"crop_{k}": np.stack(v, 2) for k, v in crops.items()})


def g8_focal_stack_m1():
    """Config-0 scale M1 stack: 256^2, 5 slices, seed 0: per slice refocus -> psf_map -> render_psf_map."""
    H = W = 256
    lens = Lensgroup(filename=f"{REF}/lenses/rf50mm/lens.json", sensor_res=(H, W), device=CPU)
    img = torch.from_numpy(synth_rgb(H, W))[None]
    depth = synth_depth_mm(H, W)
    dbar = -float(depth.mean())
    fds = -np.linspace(depth.min(), depth.max(), 5)
    torch.manual_seed(0)
    slices, maps, dsens = [], [], []
    for f in fds:
        lens.refocus(float(f))
        pm = lens.psf_map(depth=dbar, grid=11, ks=11, spp=GEO_SPP)
        slices.append(ref_render.render_psf_map(img, pm, 11))
        maps.append(pm.numpy())
        dsens.append(lens.d_sensor)
    stack = torch.stack(slices, dim=2)[0].numpy()      # [3,S,H,W]
    np.savez_compressed(f"{HERE}/g8_stack_m1.npz", fds=fds, dbar=np.float64(dbar), d_sensor=np.array(dsens),
                        psf_maps=np.stack(maps), centre_f16=stack[:, :, 64:192, 64:192].astype(np.float16),
                        crop=stack[:, :, 96:160, 96:160], sums=stack.astype(np.float64).sum((2, 3)))


def _g9_case(case, out_name):
    """Case k of the bench workload: torch.manual_seed(k), scene = synth_rgb(seed 1234 + k) / synth_depth_mm(seed 5678 + k) (k = 0 is
    bench.py's stack).  Stored: PSF maps, d_sensor / hfov per slice, three 64x64 crops per slice (patch seam, centre, corner), 16x16
    block means of every slice, fp64 sums."""
    H = W = 1024
    lens = Lensgroup(filename=f"{REF}/lenses/rf50mm/lens.json", sensor_res=(H, W), device=CPU)
    img = torch.from_numpy(synth_rgb(H, W, seed=1234 + case))[None]
    depth = synth_depth_mm(H, W, seed=5678 + case)
    dbar = -float(depth.mean())
    fds = -np.linspace(depth.min(), depth.max(), 10)
    torch.manual_seed(case)
    maps, dsens, hfovs, crops, blocks, sums = [], [], [], {"seam": [], "centre": [], "corner": []}, [], []
    for f in fds:
        lens.refocus(float(f))
        pm = lens.psf_map(depth=dbar, grid=11, ks=11, spp=GEO_SPP)
        sl = ref_render.render_psf_map(img, pm, 11)[0].numpy()           # [3,H,W]
        maps.append(pm.numpy())
        dsens.append(lens.d_sensor)
        hfovs.append(lens.hfov)
        crops["seam"].append(sl[:, 61:125, 154:218])                       # patch borders at 93 and 186
        crops["centre"].append(sl[:, 480:544, 480:544])
        crops["corner"].append(sl[:, 960:1024, 960:1024])
        blocks.append(sl.astype(np.float64).reshape(3, 64, 16, 64, 16).mean((2, 4)).astype(np.float32))
        sums.append(sl.astype(np.float64).sum((1, 2)))
    extra = {} if case == 0 else {"hfov": np.array(hfovs), "case": np.int64(case)}       # G9 keeps its round-2 key set (bit-identical file)
    np.savez_compressed(f"{HERE}/{out_name}", fds=fds, dbar=np.float64(dbar), d_sensor=np.array(dsens),
                        psf_maps=np.stack(maps), sums=np.stack(sums), block_means=np.stack(blocks, 1),
                        **{f"crop_{k}": np.stack(v, 1) for k, v in crops.items()}, **extra)


def g9_stack_m1_full():
    """The bench workload itself (BASELINE.json configs[1]): rf50mm, 1024^2, 10 focus distances, grid 11, ks 11,
    spp 2048, seed 0 — same image / depth plane / focus list as bench.py."""
    _g9_case(0, "g9_stack_m1_1024.npz")


G9B_CASES = (1, 2, 3, 4)


def g9b_more_seeds_and_scenes():
    """G9b (round 4): the same workload for four further (generator seed, scene) pairs - the reference draws new pupil samples in
    every call (deeplens/optics.py:480-481), so one stack pins one realisation of the Monte-Carlo noise only."""
    for k in G9B_CASES:
        _g9_case(k, f"g9b_case{k}.npz")
        print("G9b case", k, flush=True)


def g10_training_data():
    """PSFNet.get_training_data (deeplens/psfnet.py:135-170): two consecutive calls after np.random.seed(0);
    torch.manual_seed(0), bs 16, spp 256, sensor 480x640 (the resolution 1_fit_psfnet.py:18 uses)."""
    lens = PSFNet(filename=f"{REF}/lenses/rf50mm/lens.json", sensor_res=(480, 640), kernel_size=11, device="cpu")
    np.random.seed(0)
    torch.manual_seed(0)
    out = {}
    for i in range(2):
        inp, psf = lens.get_training_data(bs=16, spp=256)
        out[f"inp_{i}"], out[f"psf_{i}"], out[f"d_sensor_{i}"] = inp.numpy(), psf.numpy(), np.float64(lens.d_sensor)
    np.savez_compressed(f"{HERE}/g10_training_data.npz", **out)


def g11_ckpt_activation_range():
    """Per-layer max |pre-activation| and max |weight| of the SHIPPED checkpoint (ckpt/rf50mm/PSFNet480x640_ks11.pkl,
    the net 0_warm_up.py:10 loads) over the G7 render inputs and a dense (x, y, z, foc_z) lattice.  Numbers only —
    no weights are stored.  They bound the operands of the fused kernel's fp16 hi/lo split (psfnet.hip)."""
    sd = torch.load(f"{REF}/ckpt/rf50mm/PSFNet480x640_ks11.pkl", map_location="cpu")
    net = MLP(in_features=4, out_features=121, hidden_features=256, hidden_layers=8)
    net.load_state_dict(sd)
    net.eval()
    lens = PSFNet(filename=f"{REF}/lenses/rf50mm/lens.json", sensor_res=(64, 64), kernel_size=11, device="cpu")
    depth = -torch.from_numpy(synth_depth_mm(64, 64, seed=12))
    x, y = torch.meshgrid(torch.linspace(-1, 1, 64), torch.linspace(1, -1, 64), indexing="xy")
    rows = []
    for f in (-500., -800., -1500., -3000., -5000.):
        rows.append(torch.stack((x, y, lens.depth2z(depth), lens.depth2z(torch.full_like(depth, f))), -1).reshape(-1, 4))
    g = torch.linspace(0, 1, 13)
    lat = torch.stack(torch.meshgrid(g * 2 - 1, g * 2 - 1, g, g, indexing="ij"), -1).reshape(-1, 4)
    inp = torch.cat(rows + [lat], 0).float()
    rec = {"n_inputs": int(inp.shape[0]), "layers": []}
    h = inp
    with torch.no_grad():
        for m in net.net:
            h = m(h)
            if isinstance(m, torch.nn.Linear):
                rec["layers"].append({"out_features": m.out_features, "max_abs_preact": float(h.abs().max()),
                                      "max_abs_weight": float(m.weight.abs().max()), "max_abs_bias": float(m.bias.abs().max()),
                                      "min_abs_weight_nonzero": float(m.weight.abs()[m.weight != 0].min())})
        rec["psf_sum_min"] = float(h.sum(-1).min())
        rec["psf_sum_max"] = float(h.sum(-1).max())
    with open(f"{HERE}/g11_ckpt_activation_range.json", "w") as f:
        json.dump(rec, f, indent=1)


def g12_select_focus_dist():
    """dff/utils.py:4-50 select_focus_dist: 'linear' on a batch of 2 and 'importance' (B = 1, np.random.seed(3))."""
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    spec = importlib.util.spec_from_file_location("ref_dff_utils", f"{REF}/dff/utils.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    d = torch.from_numpy(np.stack([synth_depth_mm(32, 48, seed=21), synth_depth_mm(32, 48, seed=22)]))[:, None] / 1e3
    d[0, 0, :4, :4] = 0.0                                                 # invalid pixels
    out = {"depth": d.numpy(), "linear_8": mod.select_focus_dist(d, 8, "linear").numpy()}
    np.random.seed(3)
    out["importance_8"] = mod.select_focus_dist(d[:1], 8, "importance").numpy()
    out["importance_np_state_after"] = np.float64(np.random.rand())
    np.savez_compressed(f"{HERE}/g12_select_focus_dist.npz", **out)



NAMED_LENS = "rf50mm_named"


def g14_glass():
    """Named glasses: every dispersion branch of Material (deeplens/basics.py:298-379).
    Emits (a) glass_catalogue.json (deeplens/ and oracle/) - the reference's four glass TABLES (data: n_d/V_d, Sellmeier and Schott
    coefficients, display names) as Python holds them after import (later duplicate keys win), (b) a variant of the rf50mm
    prescription whose elements name catalogue glasses of all three branches (COC: Schott, N-LAK34: Sellmeier, SF5: table
    n/V), and (c) the fixture: ior of every catalogue name at six wavelengths, and the named lens end to end (load scalars,
    pupils, refocus scalars, one psf_map with its draws)."""
    import deeplens.basics as rb
    cat = {"material": {k: [float(v[0]), ("inf" if np.isinf(v[1]) else float(v[1]))] for k, v in rb.MATERIAL_TABLE.items()},
           "sellmeier": {k: [float(x) for x in v] for k, v in rb.SELLMEIER_TABLE.items()},
           "schott": {k: [float(x) for x in v] for k, v in rb.SCHOTT_TABLE.items()},
           "glass_name": dict(rb.GLASS_NAME)}
    for dst in (f"{REPO}/aberration-aware-depth-from-focus_amd/deeplens/glass_catalogue.json", f"{REPO}/oracle/glass_catalogue.json"):
        with open(dst, "w") as f:                           # one copy for the product's Material, one for the oracle's Glass
            json.dump(cat, f, indent=1)
    src = json.load(open(f"{REPO}/lenses/rf50mm/lens.json"))
    swap = {"1.53110/55.9": "coc", "1.73400/51.5": "n-lak34", "1.67270/32.1": "sf5"}
    for s in src["surfaces"]:
        s["mat1"], s["mat2"] = swap.get(s["mat1"], s["mat1"]), swap.get(s["mat2"], s["mat2"])
    os.makedirs(f"{REPO}/lenses/{NAMED_LENS}", exist_ok=True)
    with open(f"{REPO}/lenses/{NAMED_LENS}/lens.json", "w") as f:
        json.dump(src, f, indent=1)

    waves = [0.656, 0.589, 0.486, 0.4, 0.7, 589.0]          # the last one in nanometres (ior converts > 10)
    out = {"waves": waves, "ior": {}, "named_lens": {}}
    names = sorted(set(rb.MATERIAL_TABLE) | set(rb.SELLMEIER_TABLE) | set(rb.SCHOTT_TABLE))
    for n in names + ["1.5168/64.17", "N-BK7", "PMMA"]:
        try:
            m = Material(n)
        except Exception as e:                              # a name with coefficients but no (n_d, V_d) entry
            out["ior"][n] = {"error": type(e).__name__}
            continue
        out["ior"][n] = {"dispersion": m.dispersion, "A": float(m.A), "B": float(m.B), "glassname": m.glassname,
                         "n": [float(m.ior(w)) for w in waves]}
    path = f"{REPO}/lenses/{NAMED_LENS}/lens.json"
    res = (256, 256)
    lens = Lensgroup(filename=path, sensor_res=res, device=CPU)
    rec = {"load": lens_scalars(lens), "entrance_pupil": [float(v) for v in lens.entrance_pupil()],
           "exit_pupil": [float(v) for v in lens.exit_pupil()], "refocus": {}}
    for fd in (-700., -1500., -6000.):
        lens = Lensgroup(filename=path, sensor_res=res, device=CPU)
        torch.manual_seed(0)
        lens.refocus(fd)
        rec["refocus"][str(int(fd))] = lens_scalars(lens)
    out["named_lens"] = rec
    with open(f"{HERE}/g14_glass.json", "w") as f:
        json.dump(out, f, indent=1)
    lens = Lensgroup(filename=path, sensor_res=res, device=CPU)
    torch.manual_seed(0)
    lens.refocus(-1500.)
    pm = lens.psf_map(depth=-1200., grid=5, ks=11, spp=512)
    np.savez_compressed(f"{HERE}/g14_named_psf_map.npz", psf_map=pm.numpy(), d_sensor=np.float64(lens.d_sensor), hfov=np.float64(lens.hfov))


def g15_ai_degree4():
    """A 4-coefficient asphere: the reference evaluates its r^8 term with the r^6 coefficient (deeplens/surfaces.py:313,
    `self.ai8 = torch.Tensor([ai[2]])`).  Emits lenses/rf50mm_ai4/lens.json (rf50mm with its two aspheres cut to four
    coefficients, the fourth made different from the third so the wart is visible) and the fixture: load / refocus
    scalars, per-surface states of a small ray bundle through the two aspheres, and a PSF map with its d_sensor."""
    src = json.load(open(f"{REPO}/lenses/rf50mm/lens.json"))
    for s in src["surfaces"]:
        if s["type"] == "Aspheric":
            s["ai"] = [s["ai"][0], s["ai"][1], s["ai"][2], s["ai"][3] * 3.0]
    os.makedirs(f"{REPO}/lenses/rf50mm_ai4", exist_ok=True)
    path = f"{REPO}/lenses/rf50mm_ai4/lens.json"
    with open(path, "w") as f:
        json.dump(src, f, indent=1)
    res = (256, 256)
    lens = Lensgroup(filename=path, sensor_res=res, device=CPU)
    out = {"load": lens_scalars(lens), "refocus": {}}
    for fd in (-800., -3000.):
        lens = Lensgroup(filename=path, sensor_res=res, device=CPU)
        torch.manual_seed(0)
        lens.refocus(fd)
        out["refocus"][str(int(fd))] = lens_scalars(lens)
    with open(f"{HERE}/g15_ai4.json", "w") as f:
        json.dump(out, f, indent=1)
    lens = Lensgroup(filename=path, sensor_res=res, device=CPU)
    torch.manual_seed(0)
    lens.refocus(-1500.)
    pts = lens.point_source_grid(depth=-1200., grid=3).reshape(-1, 3)
    scale = lens.calc_scale_pinhole(pts[:, 2])
    pobj = pts.clone()
    pobj[..., 0] = pts[..., 0] * scale * lens.sensor_size[1] / 2
    pobj[..., 1] = pts[..., 1] * scale * lens.sensor_size[0] / 2
    torch.manual_seed(7)
    ray = lens.sample_from_points(o=pobj, spp=16, wvln=0.589)
    o0, d0 = ray.o.clone().numpy(), ray.d.clone().numpy()
    so, sd, sra = [], [], []
    for s in lens.surfaces:
        ray = s.ray_reaction(ray)
        so.append(ray.o.clone().numpy()); sd.append(ray.d.clone().numpy()); sra.append(ray.ra.clone().numpy())
    torch.manual_seed(0)
    pm = lens.psf_map(depth=-1200., grid=5, ks=11, spp=512)
    np.savez_compressed(f"{HERE}/g15_ai4.npz", ray_o0=o0, ray_d0=d0, states_o=np.stack(so), states_d=np.stack(sd), states_ra=np.stack(sra),
                        psf_map=pm.numpy(), d_sensor=np.float64(lens.d_sensor), hfov=np.float64(lens.hfov))

ALL = [("G1", lambda: g1_scalars()), ("G2/G3", lambda: g2_g3_trace_and_splat()), ("G4", lambda: g4_psf_map()),
       ("G5", lambda: g5_conv()), ("G6/G7", lambda: g6_g7_psfnet()), ("G7B", lambda: g7b_config5_stack()), ("G8", lambda: g8_focal_stack_m1()),
       ("G9", lambda: g9_stack_m1_full()), ("G9B", lambda: g9b_more_seeds_and_scenes()), ("G10", lambda: g10_training_data()),
       ("G11", lambda: g11_ckpt_activation_range()), ("G12", lambda: g12_select_focus_dist()), ("G14", lambda: g14_glass()), ("G15", lambda: g15_ai_degree4())]

if __name__ == "__main__":
    want = {a.upper() for a in sys.argv[1:]}            # e.g. `make_golden.py G9 G10`; no arguments = everything
    if not want:
        for n in ("rf50mm", "50mm_f2.8"):
            restate_lens_json(n)
    for name, fn in ALL:
        if not want or any(w in name.split("/") for w in want):
            fn()
            print(name, "done", flush=True)
    os.system(f"ls -la {HERE}")
