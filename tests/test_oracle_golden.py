"""Pin the oracle: every golden vector generated from the imported reference
(tests/golden/make_golden.py) must be reproduced by oracle/ on CPU.

Tolerance: <=1e-6 abs; most cases are bit-equal because the oracle
keeps the reference's torch-op order.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import conv as oconv
from oracle import psfnet as opsf
from oracle.glass import Glass
from oracle.lens import GEO_SPP, WAVE_RGB, OracleLens, Rays
from oracle.splat import forward_integral
from aadff.synth import mlp_state_dict, synth_depth_mm, synth_rgb

ATOL = 1e-6


def lens_path(repo_root, name):
    return os.path.join(repo_root, "lenses", name, "lens.json")


def tt(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# ----------------------------------------------------------------- G1 scalars
@pytest.mark.parametrize("key", ["rf50mm@1024x1024", "rf50mm@480x640", "50mm_f2.8@1024x1024"])
def test_g1_scalars(golden_dir, repo_root, key):
    g = json.load(open(os.path.join(golden_dir, "g1_scalars.json")))[key]
    name, res = key.split("@")
    res = tuple(int(v) for v in res.split("x"))
    lens = OracleLens(lens_path(repo_root, name), sensor_res=res)
    for k in ("d_sensor", "hfov", "foclen", "fnum", "pixel_size", "r_last"):
        assert getattr(lens, k) == pytest.approx(g["load"][k], rel=0, abs=1e-9), k
    assert lens.aper_idx == g["load"]["aper_idx"]
    assert list(lens.sensor_size) == pytest.approx(g["load"]["sensor_size"], abs=1e-12)
    assert list(lens.entrance_pupil()) == pytest.approx(g["entrance_pupil"], abs=1e-9)
    assert list(lens.exit_pupil()) == pytest.approx(g["exit_pupil"], abs=1e-9)
    assert list(lens.entrance_pupil(shrink_pupil=True)) == pytest.approx(g["entrance_pupil_shrunk"], abs=1e-9)
    assert list(lens.exit_pupil(shrink_pupil=True)) == pytest.approx(g["exit_pupil_shrunk"], abs=1e-9)
    for m, vals in g["ior"].items():
        assert [float(Glass(m).ior(w)) for w in WAVE_RGB] == pytest.approx(vals, abs=1e-14), m


@pytest.mark.parametrize("name,res", [("rf50mm", (1024, 1024)), ("50mm_f2.8", (1024, 1024))])
def test_g1_refocus(golden_dir, repo_root, name, res):
    g = json.load(open(os.path.join(golden_dir, "g1_scalars.json")))[f"{name}@{res[0]}x{res[1]}"]
    for f, want in g["refocus"].items():
        lens = OracleLens(lens_path(repo_root, name), sensor_res=res)
        torch.manual_seed(0)
        lens.refocus(float(f))
        for k in ("d_sensor", "hfov", "foclen", "fnum"):
            assert getattr(lens, k) == pytest.approx(want[k], rel=0, abs=1e-9), (f, k)


def test_g14_named_glasses(golden_dir, repo_root):
    """G14: every catalogue glass (all three dispersion branches) bit-equal; a lens that names Schott / Sellmeier / table
    glasses end to end (load scalars, pupils, refocus scalars, PSF map)."""
    g = json.load(open(os.path.join(golden_dir, "g14_glass.json")))
    for name, rec in g["ior"].items():
        gl = Glass(name)
        assert gl.dispersion == rec["dispersion"], name
        assert (gl.A, gl.B) == (rec["A"], rec["B"]), name
        assert [float(gl.ior(w)) for w in g["waves"]] == rec["n"], name
    nl = g["named_lens"]
    path = lens_path(repo_root, "rf50mm_named")
    lens = OracleLens(path, sensor_res=(256, 256))
    for k in ("d_sensor", "hfov", "foclen", "fnum", "pixel_size"):
        assert getattr(lens, k) == pytest.approx(nl["load"][k], rel=0, abs=1e-9), k
    assert lens.aper_idx == nl["load"]["aper_idx"]
    assert list(lens.entrance_pupil()) == pytest.approx(nl["entrance_pupil"], abs=1e-9)
    assert list(lens.exit_pupil()) == pytest.approx(nl["exit_pupil"], abs=1e-9)
    for f, want in nl["refocus"].items():
        lens = OracleLens(path, sensor_res=(256, 256))
        torch.manual_seed(0)
        lens.refocus(float(f))
        for k in ("d_sensor", "hfov", "foclen", "fnum"):
            assert getattr(lens, k) == pytest.approx(want[k], rel=0, abs=1e-9), (f, k)
    gm = np.load(os.path.join(golden_dir, "g14_named_psf_map.npz"))
    lens = OracleLens(path, sensor_res=(256, 256))
    torch.manual_seed(0)
    lens.refocus(-1500.0)
    pm = lens.psf_map(depth=-1200.0, grid=5, ks=11, spp=512)
    assert lens.d_sensor == pytest.approx(float(gm["d_sensor"]), abs=1e-9)
    assert np.abs(pm.numpy() - gm["psf_map"]).max() <= ATOL


def test_g15_four_coefficient_asphere_keeps_the_reference_wart(golden_dir, repo_root):
    """G15: with four even-asphere coefficients the reference evaluates r^8 with the r^6 coefficient (surfaces.py:313);
    the oracle must reproduce that (scalars, per-surface states, PSF map) - and must NOT match if the wart is 'fixed'."""
    g = json.load(open(os.path.join(golden_dir, "g15_ai4.json")))
    z = np.load(os.path.join(golden_dir, "g15_ai4.npz"))
    path = lens_path(repo_root, "rf50mm_ai4")
    lens = OracleLens(path, sensor_res=(256, 256))
    for k in ("d_sensor", "hfov", "foclen", "fnum"):
        assert getattr(lens, k) == pytest.approx(g["load"][k], rel=0, abs=1e-9), k
    for f, want in g["refocus"].items():
        lens = OracleLens(path, sensor_res=(256, 256))
        torch.manual_seed(0)
        lens.refocus(float(f))
        for k in ("d_sensor", "hfov"):
            assert getattr(lens, k) == pytest.approx(want[k], rel=0, abs=1e-9), (f, k)
    lens = OracleLens(path, sensor_res=(256, 256))
    torch.manual_seed(0)
    lens.refocus(-1500.0)
    rays = Rays(tt(z["ray_o0"]).clone(), tt(z["ray_d0"]).clone(), wvln=0.589, normalize=False)     # stored after Ray() normalised them
    for i, s in enumerate(lens.surfaces):
        rays = s.react(rays)
        assert np.array_equal(rays.ra.numpy(), z["states_ra"][i]), f"surface {i} validity"
        assert np.abs(rays.o.numpy() - z["states_o"][i]).max() <= ATOL and np.abs(rays.d.numpy() - z["states_d"][i]).max() <= ATOL, f"surface {i}"
    torch.manual_seed(0)
    pm = lens.psf_map(depth=-1200.0, grid=5, ks=11, spp=512)
    assert np.abs(pm.numpy() - z["psf_map"]).max() <= ATOL
    fixed = OracleLens(path, sensor_res=(256, 256))
    rec = json.load(open(path))["surfaces"][8]
    fixed.surfaces[8].ai[3] = torch.Tensor([rec["ai"][3]])          # what a 'corrected' implementation would use
    torch.manual_seed(0)
    fixed.refocus(-1500.0)
    assert abs(fixed.d_sensor - float(z["d_sensor"])) > 1e-5


def test_scalar_restatement_tracks_the_tensor_oracle(repo_root):
    """oracle/scalar_trace.py - the reference's surface arithmetic one float32 operation at a time - against the tensor
    oracle on the same batch: identical validity everywhere, the first surface's Newton root (10 batch-wide iterations
    from 1.5 m away: where the 1e-4 mm of hit noise is born) bit-equal on every ray, and >= 94 % of the sensor-side ray
    states bit-equal after all 12 surfaces.  The rest differ in the last bit because torch's CPU sqrt is MKL's vector
    sqrt, which is not correctly rounded (0.7 % of arguments): no implementation without that library reproduces the
    reference bit for bit, which bounds what a 'strict' float32 mode could match."""
    from oracle import scalar_trace as st
    lens = OracleLens(lens_path(repo_root, "rf50mm"), sensor_res=(1024, 1024))
    pobj = lens.object_points(lens.point_source_grid(-1500.0, 11).reshape(-1, 3))
    torch.manual_seed(5)
    rays = lens.sample_from_points(pobj, spp=128, wvln=0.589)
    surfs = st.surfaces_from_oracle(lens, 0.589)
    o = tuple(np.ascontiguousarray(rays.o[..., i].numpy()) for i in range(3))
    d = tuple(np.ascontiguousarray(rays.d[..., i].numpy()) for i in range(3))
    ra = rays.ra.numpy().copy()
    its = []
    for i, (so, ss) in enumerate(zip(lens.surfaces, surfs)):
        rays = so.react(rays)
        o, d, ra, it = st.react(ss, o, d, ra, True)
        its.append(it)
        assert np.array_equal(ra, rays.ra.numpy()), f"surface {i} validity"
        alive = ra > 0
        same_o = (np.stack(o, -1) == rays.o.numpy()).all(-1)[alive].mean()
        same_d = (np.stack(d, -1) == rays.d.numpy()).all(-1)[alive].mean()
        if i == 0:
            assert same_o == 1.0 and same_d >= 0.99
        assert np.abs(np.stack(o, -1) - rays.o.numpy())[alive].max() <= 5e-4      # far from bit-equal never (1e-4 mm noise scale)
    assert its[0] == 10 and its[5] == 0 and max(its[1:4]) <= 5
    assert same_o >= 0.94 and same_d >= 0.94


def test_appendix_d_known_answers(repo_root):
    """SURVEY.md Appendix D values measured on the reference (independent of the fixtures)."""
    lens = OracleLens(lens_path(repo_root, "rf50mm"), sensor_res=(1024, 1024))
    assert lens.d_sensor == pytest.approx(59.63294983, abs=1e-7)
    assert lens.hfov == pytest.approx(0.4097871482, abs=1e-9)
    assert lens.foclen == pytest.approx(49.81834174, abs=1e-6)
    assert lens.fnum == pytest.approx(1.86565024, abs=1e-7)
    assert lens.aper_idx == 5
    assert lens.entrance_pupil() == pytest.approx((19.809342636, 13.351468737), abs=1e-8)
    assert lens.exit_pupil() == pytest.approx((12.845902532, 12.972473741), abs=1e-8)
    assert Glass("1.83481/42.7").ior(0.589) == pytest.approx(1.83484003, abs=1e-8)
    assert Glass("1.53110/55.9").ior(0.486) == pytest.approx(1.53783484, abs=1e-8)
    assert Glass("air").ior(0.589) == 1.0


# ----------------------------------------------------------------- G2 trace / G3 splat
@pytest.fixture(scope="module")
def g23(golden_dir):
    return np.load(os.path.join(golden_dir, "g2_g3_trace_splat.npz"))


@pytest.fixture(scope="module")
def lens_foc2000(repo_root):
    lens = OracleLens(lens_path(repo_root, "rf50mm"), sensor_res=(1024, 1024))
    torch.manual_seed(0)
    lens.refocus(-2000.0)
    return lens


def test_g2_setup(g23, lens_foc2000):
    lens = lens_foc2000
    assert lens.d_sensor == pytest.approx(float(g23["d_sensor"]), abs=1e-9)
    assert lens.hfov == pytest.approx(float(g23["hfov"]), abs=1e-12)
    pts = lens.point_source_grid(-1500.0, 11).reshape(-1, 3)
    assert torch.equal(pts, tt(g23["points"]))
    assert torch.allclose(lens.object_points(pts), tt(g23["points_obj"]), atol=0, rtol=0)


def test_g2_per_surface_states(g23, lens_foc2000):
    rays = Rays(tt(g23["ray_o0"]).clone(), tt(g23["ray_d0"]).clone(), wvln=0.589, normalize=True)
    for i, s in enumerate(lens_foc2000.surfaces):
        rays = s.react(rays)
        assert np.array_equal(rays.ra.numpy(), g23["states_ra"][i]), f"surface {i} validity"
        assert np.abs(rays.o.numpy() - g23["states_o"][i]).max() <= ATOL, f"surface {i} o"
        assert np.abs(rays.d.numpy() - g23["states_d"][i]).max() <= ATOL, f"surface {i} d"


def _replay(draws):
    """Make torch.rand return the stored draws in order (RNG order: SURVEY.md App. B)."""
    it = iter(draws)
    orig = torch.rand

    def fake(n, *a, **k):
        v = next(it)
        assert v.shape[0] == n
        return v.clone()
    return orig, fake


def test_g2_sensor_hits_and_g3_psf(g23, lens_foc2000, monkeypatch):
    lens = lens_foc2000
    pobj = tt(g23["points_obj"])
    _, fake = _replay([tt(g23["u_theta"]), tt(g23["u_r"])])
    monkeypatch.setattr(torch, "rand", fake)
    ray = lens.trace2sensor(lens.sample_from_points(pobj, spp=256, wvln=0.589))
    monkeypatch.undo()
    assert np.array_equal(ray.ra.numpy().astype(np.uint8), g23["sensor_ra"])
    alive = g23["sensor_ra"] > 0
    assert np.abs(ray.o[..., :2].numpy() - g23["sensor_xy"])[alive].max() <= ATOL
    assert np.abs(ray.d.numpy() - g23["final_d"])[alive].max() <= ATOL

    _, fake = _replay([tt(g23["c_theta"]), tt(g23["c_r"])])
    monkeypatch.setattr(torch, "rand", fake)
    centre = lens.psf_center(pobj)
    monkeypatch.undo()
    assert np.abs(centre.numpy() - g23["centre"]).max() <= ATOL

    raw = forward_integral(ray, ps=lens.pixel_size, ks=11, pointc_ref=tt(g23["centre"]))
    assert np.abs(raw.numpy() - g23["psf_raw"]).max() <= 1e-5      # sums of up to 256 unit weights
    psf = raw / raw.sum(-1).sum(-1).unsqueeze(-1).unsqueeze(-1)
    assert np.abs(psf.numpy() - g23["psf"]).max() <= ATOL


def test_g2_backward_trace(g23, lens_foc2000):
    lens = lens_foc2000
    M = 32
    aper = lens.surfaces[lens.aper_idx]
    phi = torch.arange(-0.5, 0.5, 1.0 / M)
    o = torch.tensor([[aper.r, 0, aper.d.item()]]).repeat(M, 1).to(torch.float32)
    d = torch.stack((torch.sin(phi), torch.zeros_like(phi), -torch.cos(phi)), axis=-1)
    ray = lens.trace(Rays(o, d), lens_range=range(0, lens.aper_idx))
    assert np.array_equal(ray.ra.numpy(), g23["back_ra"])
    assert np.abs(ray.o.numpy() - g23["back_o"]).max() <= ATOL
    assert np.abs(ray.d.numpy() - g23["back_d"]).max() <= ATOL


# ----------------------------------------------------------------- G4 psf_map
@pytest.mark.parametrize("name,res,foc,depth,spp", [("rf50mm", (1024, 1024), -2000.0, -1500.0, 2048),
                                                   ("50mm_f2.8", (256, 256), -1000.0, -1250.0, 512)])
def test_g4_psf_map_seeded(golden_dir, repo_root, name, res, foc, depth, spp):
    g = np.load(os.path.join(golden_dir, "g4_psf_map.npz"))
    key = name.replace(".", "_")
    lens = OracleLens(lens_path(repo_root, name), sensor_res=res)
    torch.manual_seed(0)
    lens.refocus(foc)
    assert lens.d_sensor == pytest.approx(float(g[f"{key}_d_sensor"]), abs=1e-9)
    pm = lens.psf_map(depth=depth, grid=11, ks=11, spp=spp)
    assert pm.shape == (3, 121, 121)
    assert np.abs(pm.numpy() - g[f"{key}_psf_map"]).max() <= ATOL


def test_g4_psf_map_replayed_draws(golden_dir, repo_root, monkeypatch):
    """Same map from the STORED draws (guards against torch-generator changes)."""
    g = np.load(os.path.join(golden_dir, "g4_psf_map.npz"))
    lens = OracleLens(lens_path(repo_root, "50mm_f2.8"), sensor_res=(256, 256))
    torch.manual_seed(0)
    lens.refocus(-1000.0)
    draws = []
    for i in range(3):
        draws += [tt(g["50mm_f2_8_draws_main"][i, 0]), tt(g["50mm_f2_8_draws_main"][i, 1]),
                  tt(g["50mm_f2_8_draws_chief"][i, 0]), tt(g["50mm_f2_8_draws_chief"][i, 1])]
    _, fake = _replay(draws)
    monkeypatch.setattr(torch, "rand", fake)
    pm = lens.psf_map(depth=-1250.0, grid=11, ks=11, spp=512)
    monkeypatch.undo()
    assert np.abs(pm.numpy() - g["50mm_f2_8_psf_map"]).max() <= ATOL


def test_g4_single_point_and_nocenter(golden_dir, repo_root):
    g = np.load(os.path.join(golden_dir, "g4_psf_map.npz"))
    lens = OracleLens(lens_path(repo_root, "rf50mm"), sensor_res=(480, 640))
    torch.manual_seed(3)
    p = lens.psf([0.3, -0.4, -1200.0], ks=11, spp=1024)
    assert p.shape == (11, 11)
    assert np.abs(p.numpy() - g["single_point_psf"]).max() <= ATOL
    torch.manual_seed(3)
    p = lens.psf(torch.tensor([[0.3, -0.4, -1200.0], [0.0, 0.0, -3000.0]]), ks=11, spp=1024, center=False)
    assert np.abs(p.numpy() - g["nocenter_psf"]).max() <= ATOL


# ----------------------------------------------------------------- G5 conv
@pytest.fixture(scope="module")
def g5(golden_dir):
    return np.load(os.path.join(golden_dir, "g5_conv_small.npz"))


@pytest.mark.parametrize("tag", list("abcde"))
def test_g5_render_psf_map(g5, tag):
    img, pm, grid = tt(g5[f"map_{tag}_img"]), tt(g5[f"map_{tag}_psf"]), int(g5[f"map_{tag}_grid"])
    out = oconv.render_psf_map(img, pm, grid)
    assert np.abs(out.numpy() - g5[f"map_{tag}_out"]).max() <= ATOL
    cf = oconv.render_psf_map_closed_form(img, pm, grid)
    assert np.abs(cf - g5[f"map_{tag}_out"]).max() <= 3e-6


@pytest.mark.parametrize("tag", list("ab"))
def test_g5_render_psf(g5, tag):
    out = oconv.render_psf(tt(g5[f"uni_{tag}_img"]), tt(g5[f"uni_{tag}_psf"]))
    assert np.abs(out.numpy() - g5[f"uni_{tag}_out"]).max() <= ATOL


@pytest.mark.parametrize("tag,ks", [("a", 11), ("b", 5), ("c", 3), ("3d", 5)])
def test_g5_local_psf_render(g5, tag, ks):
    out = oconv.local_psf_render(tt(g5[f"loc_{tag}_img"]), tt(g5[f"loc_{tag}_psf"]), kernel_size=ks)
    assert out.shape == g5[f"loc_{tag}_out"].shape
    assert np.abs(out.numpy() - g5[f"loc_{tag}_out"]).max() <= ATOL


def test_g5_high_res_seams(g5):
    out = oconv.local_psf_render_high_res(tt(g5["hr_img"]), tt(g5["hr_psf"]), patch_size=[16, 20], kernel_size=11)
    assert np.abs(out.numpy() - g5["hr_out"]).max() <= ATOL
    # the seams are real: the untiled render differs at tile borders
    whole = oconv.local_psf_render(tt(g5["hr_img"]), tt(g5["hr_psf"]), kernel_size=11)
    assert np.abs(whole.numpy() - g5["hr_out"]).max() > 1e-3


def test_g5_conv_1024(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_conv_1024.npz"))
    img = tt(synth_rgb(1024, 1024))[None]
    out = oconv.render_psf_map(img, tt(g["psf_map"]), 11)[0].numpy()
    for k in g.files:
        if k.startswith("crop_"):
            y, x = (int(v) for v in k[5:].split("_"))
            assert np.abs(out[:, y:y + 64, x:x + 64] - g[k]).max() <= ATOL, k
    assert out.astype(np.float64).sum((1, 2)) == pytest.approx(g["sums"], rel=1e-9)


# ----------------------------------------------------------------- G6 MLP / G7 M2 render
@pytest.fixture(scope="module")
def g67(golden_dir):
    return np.load(os.path.join(golden_dir, "g6_g7_psfnet.npz"))


@pytest.fixture(scope="module")
def mlp_sd():
    return {k: tt(v) for k, v in mlp_state_dict(seed=4321).items()}


def test_g6_mlp(g67, mlp_sd):
    y = opsf.mlp_forward(mlp_sd, tt(g67["mlp_in"]))
    assert np.abs(y.numpy() - g67["mlp_out"]).max() <= ATOL
    assert y.sum(-1).numpy() == pytest.approx(1.0, abs=1e-5)


def test_g7_psfnet_render(g67, mlp_sd):
    img = tt(synth_rgb(64, 64, seed=11))[None]
    depth = -tt(synth_depth_mm(64, 64, seed=12))[None, None]
    for i, f in enumerate(g67["render_fds"]):
        out = opsf.psfnet_render(mlp_sd, img, depth, torch.tensor([float(f)]))
        assert np.abs(out.numpy() - g67["render_out"][i:i + 1]).max() <= 2e-6, f
    out3 = opsf.psfnet_render(mlp_sd, img[0], depth[0, 0], -1500.0)
    assert np.abs(out3.numpy() - g67["render3d_out"]).max() <= 2e-6
    img2 = torch.cat((img, torch.flip(img, [-1])), 0)
    depth2 = torch.cat((depth, torch.flip(depth, [-2])), 0)
    outb = opsf.psfnet_render(mlp_sd, img2, depth2, torch.tensor([-700.0, -2500.0]))
    assert np.abs(outb.numpy() - g67["render_b2_out"]).max() <= 2e-6


def test_g7_thinlens(g67):
    img = tt(synth_rgb(64, 64, seed=11))[None]
    depth = -tt(synth_depth_mm(64, 64, seed=12))[None, None]
    coc = opsf.thinlens_coc(depth, torch.full_like(depth, -1500.0), 50.0, 1.8, 24.0 / 64)
    assert np.abs(coc.numpy() - g67["thin_coc"]).max() <= 1e-5
    out = opsf.thinlens_render(img, depth, torch.tensor([-1500.0]), 50.0, 1.8, 11, [24.0, 24.0], (64, 64))
    assert np.abs(out.numpy() - g67["thin_out"]).max() <= 2e-6


# ----------------------------------------------------------------- G8 M1 focal stack
def test_g8_focal_stack_m1(golden_dir, repo_root):
    g = np.load(os.path.join(golden_dir, "g8_stack_m1.npz"))
    H = W = 256
    lens = OracleLens(lens_path(repo_root, "rf50mm"), sensor_res=(H, W))
    img = tt(synth_rgb(H, W))[None]
    depth = synth_depth_mm(H, W)
    assert -float(depth.mean()) == pytest.approx(float(g["dbar"]), abs=1e-9)
    fds = -np.linspace(depth.min(), depth.max(), 5)
    assert fds == pytest.approx(g["fds"], abs=1e-9)
    torch.manual_seed(0)
    stack, maps = opsf.focal_stack_m1(lens, img, float(g["dbar"]), fds, grid=11, ks=11, spp=GEO_SPP)
    assert stack.shape == (1, 3, 5, H, W)
    assert np.abs(maps.numpy() - g["psf_maps"]).max() <= ATOL
    s = stack[0].numpy()
    assert np.abs(s[:, :, 96:160, 96:160] - g["crop"]).max() <= ATOL
    assert np.abs(s[:, :, 64:192, 64:192] - g["centre_f16"].astype(np.float32)).max() <= 1e-3
    assert s.astype(np.float64).sum((2, 3)) == pytest.approx(g["sums"], rel=1e-7)


# ----------------------------------------------------------------- G9: the bench workload (first slices)
def test_g9_bench_stack_first_slices(golden_dir, repo_root):
    """BASELINE.json configs[1] as bench.py runs it (1024^2, 10 distances, grid 11, spp 2048, seed 0): the first two
    slices of the reference stack (the RNG stream is sequential, so a prefix is exact; all ten take ~25 s)."""
    g = np.load(os.path.join(golden_dir, "g9_stack_m1_1024.npz"))
    H = W = 1024
    lens = OracleLens(lens_path(repo_root, "rf50mm"), sensor_res=(H, W))
    img = tt(synth_rgb(H, W, seed=1234))[None]
    depth = synth_depth_mm(H, W, seed=5678)
    assert -float(depth.mean()) == pytest.approx(float(g["dbar"]), abs=1e-9)
    fds = -np.linspace(depth.min(), depth.max(), 10)
    assert fds == pytest.approx(g["fds"], abs=1e-9)
    torch.manual_seed(0)
    stack, maps = opsf.focal_stack_m1(lens, img, float(g["dbar"]), fds[:2], grid=11, ks=11, spp=GEO_SPP)
    assert np.abs(maps.numpy() - g["psf_maps"][:2]).max() <= ATOL
    s = stack[0].numpy()                                     # [3,2,H,W]
    assert np.abs(s[:, :, 61:125, 154:218] - g["crop_seam"][:, :2]).max() <= ATOL
    assert np.abs(s[:, :, 480:544, 480:544] - g["crop_centre"][:, :2]).max() <= ATOL
    assert np.abs(s[:, :, 960:1024, 960:1024] - g["crop_corner"][:, :2]).max() <= ATOL
    bm = s.astype(np.float64).reshape(3, 2, 64, 16, 64, 16).mean((3, 5))
    assert np.abs(bm - g["block_means"][:, :2]).max() <= ATOL
    assert s.astype(np.float64).sum((2, 3)).T == pytest.approx(g["sums"][:2], rel=1e-7)


# ----------------------------------------------------------------- G10: PSFNet.get_training_data
def test_g10_training_data(golden_dir, repo_root):
    g = np.load(os.path.join(golden_dir, "g10_training_data.npz"))
    lens = OracleLens(lens_path(repo_root, "rf50mm"), sensor_res=(480, 640))
    np.random.seed(0)
    torch.manual_seed(0)
    for i in range(2):
        inp, psf = opsf.training_data(lens, bs=16, spp=256)
        assert np.array_equal(inp.numpy(), g[f"inp_{i}"])
        assert lens.d_sensor == pytest.approx(float(g[f"d_sensor_{i}"]), abs=1e-6)
        assert np.abs(psf.numpy() - g[f"psf_{i}"]).max() <= ATOL


# ----------------------------------------------------------------- plain-C restatement (oracle/conv_ref.c)
def _c_oracle(repo_root):
    import ctypes
    p = os.path.join(repo_root, "oracle", "_build", "liboracle_conv.so")
    if not os.path.exists(p):
        pytest.skip("oracle/_build/liboracle_conv.so not built (run __graft_entry__.build())")
    return ctypes.CDLL(p)


def test_c_oracle_matches_golden(g5, repo_root):
    import ctypes
    lib = _c_oracle(repo_root)
    fp = ctypes.POINTER(ctypes.c_float)
    for tag in "abcd":
        img, pm, grid = g5[f"map_{tag}_img"], g5[f"map_{tag}_psf"], int(g5[f"map_{tag}_grid"])
        out = np.zeros_like(img)
        B, Cn, H, W = img.shape
        lib.oracle_render_psf_map(img.ctypes.data_as(fp), pm.ctypes.data_as(fp), out.ctypes.data_as(fp), B, Cn, H, W, grid, pm.shape[1] // grid)
        assert np.abs(out - g5[f"map_{tag}_out"]).max() <= 3e-6, tag
    for tag, ks in (("a", 11), ("b", 5), ("c", 3)):
        img, p = g5[f"loc_{tag}_img"], g5[f"loc_{tag}_psf"]
        out = np.zeros_like(img)
        B, Cn, H, W = img.shape
        lib.oracle_local_psf_render(img.ctypes.data_as(fp), np.ascontiguousarray(p).ctypes.data_as(fp), out.ctypes.data_as(fp), B, Cn, H, W, ks)
        assert np.abs(out - g5[f"loc_{tag}_out"]).max() <= 3e-6, tag


def test_aten_sum_order_program_is_torch_bit_for_bit():
    """oracle/aten_sum.py (the summation order `aadff_strict_centroid` follows) against torch's own CPU `sum(0)`, bit for bit:
    the shapes of psf_center (deeplens/optics.py:902-904: [2048, 121, 3]), training batches ([4096, 128, 3]), ragged row counts,
    fewer columns than one vector - at one, three and all threads."""
    from oracle import aten_sum
    keep = torch.get_num_threads()
    try:
        for nt in (1, 3, keep):
            torch.set_num_threads(nt)
            for spp, N in ((2048, 121), (4096, 128), (3001, 121), (512, 9), (2048, 1), (17, 40)):
                g = torch.Generator().manual_seed(spp + N)
                x = torch.randn(spp, N, 3, generator=g) * 3 * (torch.rand(spp, N, 1, generator=g) > 0.3).float()
                assert np.array_equal(x.sum(0).numpy(), aten_sum.sum0(x.numpy())), (nt, spp, N)
    finally:
        torch.set_num_threads(keep)
