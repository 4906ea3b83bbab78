"""CPU: host-side logic of the product package (no kernels run): glass indices, surface
packing, lens loading, RNG call order, focus-distance rule, and the loud failure when
no GPU is present."""
import json
import os

import numpy as np
import pytest
import torch

from aadff import _abi
from aadff.focal_stack import draw_stack_uniforms, select_focus_dist, shard_units
from aadff.sampling import HostSampler
from aadff.synth import synth_depth_mm
from deeplens.basics import WAVE_RGB, Material
from deeplens.optics import Lensgroup
from deeplens.utils import make_grid
from oracle import psfnet as opsf
from oracle.lens import OracleLens


def lens_path(repo_root, name="rf50mm"):
    return os.path.join(repo_root, "lenses", name, "lens.json")


def test_material_ior_matches_reference(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "g1_scalars.json")))
    for key in ("rf50mm@1024x1024", "50mm_f2.8@1024x1024"):
        for m, vals in g[key]["ior"].items():
            assert [float(Material(m).ior(w)) for w in WAVE_RGB] == pytest.approx(vals, abs=1e-14), m
    with pytest.raises(ValueError):                    # the reference fails in float("unobtainium") (basics.py:372-374)
        Material("unobtainium")


def test_material_named_glasses_match_reference(golden_dir):
    """Every catalogue name (Sellmeier / Schott / table n-V branches, deeplens/basics.py:298-336) at six wavelengths,
    bit-equal to the reference (fixture G14), incl. A / B / dispersion / glassname and upper-case names."""
    g = json.load(open(os.path.join(golden_dir, "g14_glass.json")))
    seen = set()
    for name, rec in g["ior"].items():
        m = Material(name)
        seen.add(m.dispersion)
        assert m.dispersion == rec["dispersion"] and m.glassname == rec["glassname"], name
        assert (m.A, m.B) == (rec["A"], rec["B"]), name
        assert [float(m.ior(w)) for w in g["waves"]] == rec["n"], name
    assert seen == {"sellmeier", "schott", "naive"} and len(g["ior"]) >= 40
    bk7 = Material("n-bk7")
    assert abs(bk7.ior(0.589) - (bk7.A + bk7.B / 589.0 ** 2)) > 5e-5      # NOT the Cauchy value (round-2 bug)
    air = Material("air")
    assert air.dispersion == "sellmeier" and float(air.ior(0.5)) == 1.0 and air.A == 1.0
    m = Material("f2")
    m.load_sellmeier_param()
    assert float(m.ior(0.6)) == 1.0


def test_lens_loads_on_cpu_without_kernels(repo_root, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "g1_scalars.json")))["rf50mm@480x640"]["load"]
    lens = Lensgroup(lens_path(repo_root), sensor_res=(480, 640), post_computation=False, device="cpu")
    assert len(lens.surfaces) == 12 and lens.aper_idx == g["aper_idx"]
    assert lens.pixel_size == pytest.approx(g["pixel_size"], abs=1e-15)
    assert list(lens.sensor_size) == pytest.approx(g["sensor_size"], abs=1e-12)
    assert lens.d_sensor == pytest.approx(g["d_sensor"], abs=1e-6)
    kinds = [s.kind() for s in lens.surfaces]
    assert kinds == [1, 1, 1, 1, 1, 0, 1, 1, 2, 2, 1, 1]


def test_surface_packing_follows_reference_rounding(repo_root):
    lens = Lensgroup(lens_path(repo_root), post_computation=False, device="cpu")
    ora = OracleLens(lens_path(repo_root))
    for i, (s, o) in enumerate(zip(lens.surfaces, ora.surfaces)):
        for w in WAVE_RGB:
            p = s.pack(w)
            assert p.d == o.d.item() and p.c == o.c.item() and p.k == o.k.item()
            assert p.r2 == np.float32(o.r ** 2)
            ef = o.mat1.ior(w) / o.mat2.ior(w)
            assert p.eta_fwd == np.float32(ef) and p.eta_fwd2 == np.float32(ef ** 2)
            assert p.eta_bwd == np.float32(1 / ef if False else o.mat2.ior(w) / o.mat1.ior(w))
            if p.kind != 0:
                assert p.r2_shape == ((1 - 1e-9) / o.c ** 2 / (1 + o.k)).item()
            if p.kind == 2:
                assert p.n_ai == 6 and [p.ai[j] for j in range(6)] == [a.item() for a in o.ai]
                assert [p.dai[j] for j in range(6)] == [((j + 1) * a).item() for j, a in enumerate(o.ai)]
    stop = lens.surfaces[5].pack(0.589)
    assert stop.kind == 0 and stop.refract_fwd == 0 and stop.refract_bwd == 0


def test_stack_uniform_order_equals_reference_call_order():
    """draw_stack_uniforms must consume the host generator exactly as the reference's
    per-slice refocus -> psf_map loop does."""
    S, spp = 3, 512
    torch.manual_seed(5)
    uf, um, uc = draw_stack_uniforms(HostSampler(), S, spp)
    torch.manual_seed(5)
    for s in range(S):
        assert torch.equal(uf[s, 0], torch.rand(2048)) and torch.equal(uf[s, 1], torch.rand(2048))
        for l in range(3):
            assert torch.equal(um[s, l, 0], torch.rand(spp)) and torch.equal(um[s, l, 1], torch.rand(spp))
            assert torch.equal(uc[s, l, 0], torch.rand(2048)) and torch.equal(uc[s, l, 1], torch.rand(2048))


def test_select_focus_dist_matches_oracle():
    d = torch.from_numpy(synth_depth_mm(32, 48)).reshape(1, 1, 32, 48) / 1e3
    d = torch.cat((d, d.flip(-1) * 0.5), 0)
    d[0, 0, :4, :4] = 0.0          # invalid pixels are ignored by the min
    assert torch.equal(select_focus_dist(d, 6), opsf.select_focus_dist_linear(d, 6))
    with pytest.raises(AssertionError):
        select_focus_dist(d, 3)


def test_make_grid_is_row_major_tiling():
    t = torch.arange(6 * 1 * 2 * 2, dtype=torch.float32).reshape(6, 1, 2, 2)
    g = make_grid(t, nrow=3, padding=0)
    assert g.shape == (3, 4, 6)
    assert torch.equal(g[0, 0:2, 2:4], t[1, 0]) and torch.equal(g[2, 2:4, 0:2], t[3, 0])


def test_shard_units_partition():
    for world in (1, 2, 8):
        got = sorted(u for r in range(world) for u in shard_units(160, r, world))
        assert got == list(range(160))
    assert shard_units(160, 3, 8)[:3] == [3, 11, 19] and len(shard_units(160, 3, 8)) == 20


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_fails_loudly_without_gpu(repo_root):
    from deeplens.render_psf import local_psf_render, render_psf_map
    img = torch.rand(1, 3, 16, 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        render_psf_map(img, torch.rand(3, 6, 6), 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        local_psf_render(img, torch.rand(1, 16, 16, 3, 3), 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Lensgroup(lens_path(repo_root), device="cpu")      # post_computation needs the trace kernel


def test_reference_assertions_are_kept():
    from deeplens.render_psf import render_psf_map
    with pytest.raises(AssertionError, match="Input image should be"):
        render_psf_map(torch.rand(3, 16, 16), torch.rand(3, 6, 6), 2)
    with pytest.raises(AssertionError, match="divisible by grid"):
        render_psf_map(torch.rand(1, 3, 16, 16), torch.rand(3, 7, 7), 2)
    with pytest.raises(AssertionError, match="should be odd"):
        render_psf_map(torch.rand(1, 3, 16, 16), torch.rand(3, 8, 8), 2)
    with pytest.raises(AssertionError, match="same channel"):
        render_psf_map(torch.rand(1, 1, 16, 16), torch.rand(3, 6, 6), 2)


def test_fast_host_rng_continues_torch_generator_bit_exactly():
    """aadff_host_mt19937_uniform_f32 == torch.rand, draw for draw, and leaves torch's global
    generator in the identical state (also across MT19937 block boundaries and odd sizes)."""
    import aadff.sampling as sm
    s = HostSampler()
    sizes = [1, 7, 623, 624, 625, 2048, 5000, 3]
    torch.manual_seed(123)
    want = [torch.rand(n) for n in sizes]
    tail_want = torch.rand(9)
    state_want = torch.get_rng_state()
    torch.manual_seed(123)
    got = [s.rand_block([n]) for n in sizes]
    assert sm._FAST is True, "fast host RNG path was not taken"
    tail_got = torch.rand(9)
    assert all(torch.equal(a, b) for a, b in zip(want, got))
    assert torch.equal(tail_want, tail_got) and torch.equal(state_want, torch.get_rng_state())
    # interleaving with torch.randn / np-free draws keeps working
    torch.manual_seed(5)
    a1, z1, a2 = torch.rand(10), torch.randn(4), torch.rand(10)
    torch.manual_seed(5)
    b1, z2, b2 = s.rand_block([10]), torch.randn(4), s.rand_block([10])
    assert torch.equal(a1, b1) and torch.equal(z1, z2) and torch.equal(a2, b2)
    lib = _abi.load_library()
    import ctypes as C
    bad = torch.zeros(100, dtype=torch.uint8)
    assert lib.aadff_host_mt19937_uniform_f32(C.c_void_p(bad.data_ptr()), 100, 4, C.c_void_p(torch.empty(4).data_ptr())) == -1


def test_host_rng_discard_and_row_sampler():
    """aadff_host_mt19937_discard leaves the generator where torch.rand(n) would; RowSampler hands a sharded rank exactly
    the rows of the whole-stack block that belong to its slices without producing the others."""
    import aadff.sampling as sm
    from aadff.focal_stack import PresetSampler, RowSampler
    s = HostSampler()
    s.rand_block([8])                       # arms the fast path (self-check)
    assert sm._FAST is True
    for n in (0, 1, 623, 624, 625, 20480, 100000):
        torch.manual_seed(77)
        torch.rand(5)
        if n:
            torch.rand(n)
        want_state, want_next = torch.get_rng_state(), torch.rand(4)
        torch.manual_seed(77)
        torch.rand(5)
        s.skip(n)
        assert torch.equal(torch.get_rng_state(), want_state), n
        assert torch.equal(torch.rand(4), want_next)
    per, S = 2051, 10
    torch.manual_seed(3)
    block = torch.rand(S * per).reshape(S, per)
    for rows in ([0], [9], [3, 4, 5], [1, 9], [0, 2, 4, 6, 8], list(range(10))):
        torch.manual_seed(3)
        out = torch.empty(len(rows) * per)
        RowSampler(s, rows, per).rand_into(out)
        assert torch.equal(out.reshape(len(rows), per), block[rows]), rows
        torch.manual_seed(3)                # consumers that draw call by call get the same rows
        r = RowSampler(s, rows, per)
        got = torch.cat([r.rand(per // 2), r.rand_block([per - per // 2] + [per] * (len(rows) - 1))])
        assert torch.equal(got.reshape(len(rows), per), block[rows])
    with pytest.raises(AssertionError, match="ascend"):
        RowSampler(s, [3, 1], per)


def test_select_focus_dist_matches_reference_goldens(golden_dir):
    """dff/utils.py:4-50: 'linear' on a batch with invalid pixels, 'importance' in the reference's np.random call order
    (and with its num - 2 quirk)."""
    g = np.load(os.path.join(golden_dir, "g12_select_focus_dist.npz"))
    d = torch.from_numpy(g["depth"])
    assert torch.equal(select_focus_dist(d, 8, "linear"), torch.from_numpy(g["linear_8"]))
    np.random.seed(3)
    got = select_focus_dist(d[:1], 8, "importance", center=True)
    assert got.shape == (1, 6) and torch.equal(got, torch.from_numpy(g["importance_8"]))
    assert np.random.rand() == float(g["importance_np_state_after"])          # consumed exactly the reference's draws
    with pytest.raises(NotImplementedError):
        select_focus_dist(d, 8, "nope")


def test_preset_sampler_replays_selected_slices():
    from aadff.focal_stack import PresetSampler, stack_uniform_layout
    per = stack_uniform_layout(64)[0]
    torch.manual_seed(5)
    block = HostSampler().rand_block([4 * per]).reshape(4, per)
    for ps in (PresetSampler(block[[1, 3]].contiguous()), PresetSampler(block, rows=[1, 3])):
        out = torch.empty(2 * per)
        ps.rand_into(out)
        assert torch.equal(out.reshape(2, per), block[[1, 3]])
        with pytest.raises(AssertionError):
            ps.rand(1)
    ps = PresetSampler(block, rows=[2, 0])                     # piecewise consumers fall back to the flattened selection
    assert torch.equal(torch.cat((ps.rand(5), ps.rand_block([per - 5, per]))), block[[2, 0]].reshape(-1))


def test_star_import_surface_of_the_reference_scripts():
    """`from deeplens.psfnet import *` is how 1_fit_psfnet.py / 2_aber_aware_dff_*.py get torch, nn, np, plt, tqdm,
    save_image ... (reference chain psfnet -> optics -> surfaces -> basics, deeplens/optics.py:5-20, basics.py:7-12)."""
    ns = {}
    exec("from deeplens.psfnet import *", ns)
    for name in ("torch", "nn", "np", "plt", "tqdm", "save_image", "make_grid", "random", "datetime", "json", "F", "nnF", "stats",
                 "PSFNet", "ThinLens", "Lensgroup", "Ray", "Material", "render_psf_map", "local_psf_render", "forward_integral",
                 "GEO_SPP", "DEPTH", "WAVE_RGB", "DEFAULT_WAVE", "EPSILON", "set_seed", "set_logger", "MLP"):
        assert name in ns, name
    assert ns["nn"].DataParallel is torch.nn.DataParallel          # 2_aber_aware_dff_aif.py:67 resolves it this way


def test_dff_star_import_surface_of_the_reference_scripts():
    """`from dff import *` (2_aber_aware_dff_aif.py:25, 2_aber_aware_dff_dfv.py:25; reference dff/__init__.py:1-6) binds what the
    scripts use from the rendering side (:57 get_lens, :73 get_dataset, :106 select_focus_dist) and, for the consumer-side names
    (AiFDepthNet :66, the mask_* metrics :194-214), stubs that say where they come from."""
    ns = {}
    exec("from dff import *", ns)
    for name in ("get_lens", "get_dataset", "select_focus_dist", "Matterport3D", "FlyingThings3D", "Middlebury", "RealWorld",
                 "dataset", "factory", "utils", "PSFNet", "ThinLens", "AiFDepthNet", "mask_abs_rel", "mask_psnr", "mask_ssim"):
        assert name in ns, name
    import dff
    assert ns["select_focus_dist"] is select_focus_dist and ns["get_lens"] is dff.factory.get_lens
    if not os.environ.get("AADFF_REFERENCE_ROOT"):
        with pytest.raises(ImportError, match="AADFF_REFERENCE_ROOT"):
            ns["AiFDepthNet"](n_stack=5)
        with pytest.raises(ImportError, match="dff/metrics.py"):
            ns["mask_abs_rel"](None, None, None)


def test_reference_scripts_run_on_this_package_through_the_launcher(tmp_path):
    """a script inside a checkout that has its OWN deeplens/ and dff/ (like the reference's) runs on this package
    through `python -m aadff.run_script`; started directly, CPython resolves the checkout's packages first (the reason for the
    launcher).  The checkout's consumer modules (dff/AiFNet.py, DFV_models/) are still the checkout's."""
    import subprocess
    import sys
    co = tmp_path / "checkout"
    (co / "deeplens").mkdir(parents=True)
    (co / "dff").mkdir()
    (co / "DFV_models").mkdir()
    (co / "configs").mkdir()
    (co / "deeplens" / "__init__.py").write_text("")
    (co / "deeplens" / "psfnet.py").write_text("WHO = 'checkout'\n")
    (co / "deeplens" / "utils.py").write_text("def set_seed(s): pass\ndef set_logger(d): pass\n")
    (co / "dff" / "__init__.py").write_text("WHO_DFF = 'checkout'\n")
    (co / "dff" / "AiFNet.py").write_text("class AiFDepthNet:\n    def __init__(self, n_stack): self.n_stack = n_stack\n")
    (co / "DFV_models" / "__init__.py").write_text("class DFVNet:\n    pass\n")
    (co / "configs" / "c.yml").write_text("n_stack: 5\n")
    (co / "script.py").write_text(
        "import json, os, sys\n"
        "from deeplens.utils import set_seed, set_logger\n"
        "from deeplens.psfnet import *\n"
        "from dff import *\n"
        "from DFV_models import DFVNet\n"
        "import deeplens, dff, DFV_models\n"
        "g = globals()\n"
        "json.dump({'deeplens': deeplens.__file__, 'dff': dff.__file__, 'dfv': DFV_models.__file__, 'who': g.get('WHO'),\n"
        "           'names': [k for k in ('PSFNet', 'get_lens', 'get_dataset', 'select_focus_dist', 'nn', 'AiFDepthNet', 'DFVNet') if k in g],\n"
        "           'net': (AiFDepthNet(n_stack=5).n_stack if 'AiFDepthNet' in g else None), 'cfg': os.path.exists('configs/c.yml'),\n"
        "           'argv': sys.argv[1:]}, open(sys.argv[1], 'w'))\n")
    from conftest import PKG
    env = {k: v for k, v in os.environ.items() if k != "AADFF_REFERENCE_ROOT"}
    env["PYTHONPATH"] = PKG
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    out = tmp_path / "via_launcher.json"
    subprocess.run([sys.executable, "-m", "aadff.run_script", str(co / "script.py"), str(out), "--flag"], check=True, env=env,
                   cwd=str(tmp_path), timeout=300)
    r = json.load(open(out))
    assert r["deeplens"].startswith(PKG) and r["dff"].startswith(PKG) and r["dfv"].startswith(str(co)), r
    assert r["names"] == ["PSFNet", "get_lens", "get_dataset", "select_focus_dist", "nn", "AiFDepthNet", "DFVNet"] and r["who"] is None
    assert r["net"] == 5 and r["cfg"] is True and r["argv"] == [str(out), "--flag"]
    out2 = tmp_path / "direct.json"                      # the recipe INTEGRATION.md used to give: PYTHONPATH alone does not win
    subprocess.run([sys.executable, str(co / "script.py"), str(out2)], check=True, env=env, cwd=str(co), timeout=300)
    r2 = json.load(open(out2))
    assert r2["deeplens"].startswith(str(co)) and r2["who"] == "checkout"


def test_custom_ops_are_registered_with_shape_functions():
    """torch.ops.aadff.* (aadff/ops.py): schemas exist, the fake/meta implementations infer the output shapes, and a CPU
    call fails loudly (no CPU fallback)."""
    from aadff import ops  # noqa: F401
    m = lambda *s: torch.empty(*s, device="meta")
    assert torch.ops.aadff.render_psf_map(m(2, 3, 20, 24), m(3, 33, 33), 3).shape == (2, 3, 20, 24)
    assert torch.ops.aadff.render_psf_map_stack(m(2, 3, 20, 24), m(5, 3, 33, 33), 3).shape == (2, 3, 5, 20, 24)
    assert torch.ops.aadff.render_psf(m(1, 3, 8, 8), m(3, 5, 5)).shape == (1, 3, 8, 8)
    assert torch.ops.aadff.local_psf_render(m(1, 3, 8, 8), m(1, 8, 8, 5, 5), 5).shape == (1, 3, 8, 8)
    assert torch.ops.aadff.thinlens_render(m(2, 3, 8, 8), m(2, 1, 8, 8), m(2), 11, 50.0, 1.8, 0.02, 200.0, 20000.0).shape == (2, 3, 8, 8)
    fl = torch.empty(1, dtype=torch.int32, device="meta")
    assert torch.ops.aadff.psfnet_forward(m(10, 4), m(4), m(4), [4, 64], [64, 121], fl).shape == (10, 121)
    assert torch.ops.aadff.psfnet_forward(m(10, 4), m(4), m(4), [4, 64], [64, 121], fl, 1).shape == (10, 121)
    assert torch.ops.aadff.psfnet_render_rgbd(m(2, 3, 8, 9), m(2, 8, 9), m(9), m(8), m(2, 5), -200.0, -1e-4, m(4), m(4), [4, 64], [64, 121],
                                              11, fl).shape == (2, 3, 5, 8, 9)
    lc = [12.0] + [1.0] * 12
    assert torch.ops.aadff.psf_points(m(2, 121, 3), m(4), m(4), lc, m(64), m(2, 3, 2, 256), m(2, 3, 2, 2048), 11, True, True, fl).shape == (2, 3, 121, 121)
    assert torch.ops.aadff.psf_points(m(1, 7, 3), m(4), m(4), lc, m(32), m(1, 1, 2, 64), m(1, 1, 2, 0), 5, False, False, fl).shape == (1, 7, 1, 5, 5)
    with pytest.raises(NotImplementedError):
        torch.ops.aadff.render_psf_map(torch.zeros(1, 3, 8, 8), torch.zeros(3, 9, 9), 3)
    schema = str(torch.ops.aadff.psfnet_forward.default._schema)
    assert "Tensor(a" in schema and "flags" in schema                      # flags is declared as mutated


def test_dff_factory_shims(repo_root):
    """dff.factory.get_lens / get_dataset and dff.utils.select_focus_dist resolve like in 2_aber_aware_dff_aif.py:27-60."""
    from dff.factory import get_dataset, get_lens
    from dff.utils import select_focus_dist as sfd
    from deeplens.psfnet import ThinLens
    assert sfd is select_focus_dist
    thin = {"lens": "thinlens", "foc_len": 50.0, "fnum": 1.8, "sensor_size": ["24", "36"], "dataset": "Synthetic", "n": 3}
    args = {"ks": 11, "res": (32, 48), "device": torch.device("cpu"), "train": thin, "test": dict(thin, dataset="Middlebury2014")}
    a, b = get_lens(args)
    assert isinstance(a, ThinLens) and isinstance(b, ThinLens) and a.ps == 24.0 / 32 and a.kernel_size == 11
    args["test"]["dataset"] = "NoSuchSet"
    with pytest.raises(NotImplementedError, match="NoSuchSet"):
        get_dataset(args)
    args["test"]["dataset"] = "Synthetic"
    tr, te = get_dataset(args)
    img, depth = tr[1]
    assert len(tr) == 3 and img.shape == (3, 32, 48) and depth.shape == (1, 32, 48) and 0.4 < float(depth.min()) and float(depth.max()) <= 5.0
    assert not torch.equal(tr[0][0], te[0][0])


def test_middlebury_and_matterport_loaders_on_files_written_here(tmp_path):
    """dff/dataset.py of the reference (:17-52, :170-200) restated with PIL: `im0.png` + 16-bit `depth.png` (mm) per scene ->
    [RGB [3,H,W] in [0,1], depth [1,H,W] in metres], resized like the reference (image: antialiased bilinear; depth: plain
    bilinear at pixel centres, cv.resize's default), selected through dff.factory.get_dataset by the reference's YAML keys."""
    from PIL import Image
    from dff.dataset import Matterport3D, Middlebury
    from dff.factory import get_dataset
    rng = np.random.Generator(np.random.PCG64(5))
    root = tmp_path / "mb"
    truth = {}
    for name in ("Adirondack", "Jadeplant"):
        (root / name).mkdir(parents=True)
        rgb = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)
        dep = rng.integers(500, 6000, (40, 60), dtype=np.uint16)
        Image.fromarray(rgb).save(root / name / "im0.png")
        Image.fromarray(dep).save(root / name / "depth.png")
        truth[name] = (rgb, dep)
    ds = Middlebury(str(root), resize=(40, 60))
    assert len(ds) == 2 and ds.scenes == ["Adirondack", "Jadeplant"]
    img, depth = ds[1]
    rgb, dep = truth["Jadeplant"]
    assert img.shape == (3, 40, 60) and img.dtype == torch.float32 and depth.shape == (1, 40, 60) and depth.dtype == torch.float32
    assert torch.equal(img, torch.from_numpy((rgb / 255.0).astype("float32")).permute(2, 0, 1))
    assert torch.equal(depth[0], torch.from_numpy((dep / 1000).astype("float32")))                 # uint16 mm -> m
    small = Middlebury(str(root), resize=(20, 30))[0]
    assert small[0].shape == (3, 20, 30) and small[1].shape == (1, 20, 30)
    d64 = torch.from_numpy(truth["Adirondack"][1] / 1000)[None, None]
    want = torch.nn.functional.interpolate(d64, size=(20, 30), mode="bilinear", align_corners=False)[0].float()
    assert torch.equal(small[1], want)                                                             # exact 2x: mean of 2x2 blocks
    assert torch.allclose(small[1][0, 3, 4], torch.from_numpy(truth["Adirondack"][1][6:8, 8:10] / 1000).mean().float(), atol=1e-6)
    args = {"res": (20, 30), "train": {"dataset": "Synthetic", "n": 2}, "test": {"dataset": "Middlebury2014"}, "Middlebury2014_val": str(root)}
    tr, te = get_dataset(args)
    assert len(te) == 2 and te[0][0].shape == (3, 20, 30) and 0.5 <= float(te[0][1].min()) and float(te[0][1].max()) <= 6.0
    mp3 = tmp_path / "mp"
    (mp3 / "rgb" / "s0" / "undistorted_color_images").mkdir(parents=True)
    (mp3 / "depth" / "s0" / "render_depth").mkdir(parents=True)
    Image.fromarray(rgb).save(mp3 / "rgb" / "s0" / "undistorted_color_images" / "a.jpg", quality=95)
    Image.fromarray((dep.astype(np.uint32) * 4).clip(0, 65535).astype(np.uint16)).save(mp3 / "depth" / "s0" / "render_depth" / "a.png")
    m = Matterport3D(str(mp3 / "rgb"), str(mp3 / "depth"), resize=(40, 60), train=False)
    i2, d2 = m[0]
    assert len(m) == 1 and i2.shape == (3, 40, 60) and d2.shape == (1, 40, 60)
    assert torch.allclose(d2[0], torch.from_numpy((dep / 1000).astype("float32")), atol=1e-6)      # 4000 units per metre
    np.random.seed(0)
    i3, d3 = Matterport3D(str(mp3 / "rgb"), str(mp3 / "depth"), resize=(40, 60), train=True)[0]   # augmentation path runs
    assert i3.shape == (3, 40, 60) and d3.shape == (1, 40, 60) and float(d3.min()) >= 0


def test_per_surface_newton_step_tolerance_and_flag_bits(repo_root):
    """Host logic of two round-3 changes: (1) the Newton step tolerance packed per surface keeps kappa * tol^2 <= 4e-6 (kappa =
    largest |d^2 sag / d r^2| over the aperture) and never exceeds 10 um; spheres and stops carry 0 (they do not iterate);
    (2) the flags word maps to the reference's errors: bit 2 = 'sensor position is negative.' (optics.py:1176)."""
    from deeplens.optics import raise_psf_flags
    lens = Lensgroup(lens_path(repo_root), post_computation=False, device="cpu")
    tols = [s.newton_step_tol() for s in lens.surfaces]
    assert [t > 0 for t in tols] == [k == 2 for k in (s.kind() for s in lens.surfaces)]
    for s, t in zip(lens.surfaces, tols):
        if t > 0:
            assert t <= 1e-2 and s.pack(0.589).newton_step_tol == np.float32(t)
            rr = np.linspace(0, s.r, 1001)
            c, k = s.c.item(), s.k.item()
            curv = c / (1 - (1 + k) * c * c * rr * rr) ** 1.5 + sum(float(a) * (2 * j + 2) * (2 * j + 1) * rr ** (2 * j) for j, a in enumerate(s.ai.tolist()))
            assert np.abs(curv).max() * t * t <= 4.1e-6
    assert 3e-3 < tols[8] < 6e-3 and 3e-3 < tols[9] < 6e-3            # rf50mm's aspheres: 4.9 and 3.7 um, not the 10 um round 2 assumed
    raise_psf_flags(0)
    with pytest.raises(AssertionError, match="sensor position is negative"):
        raise_psf_flags(4)
    with pytest.raises(AssertionError, match="No sampled rays is valid"):
        raise_psf_flags(2)
    with pytest.raises(FloatingPointError):
        raise_psf_flags(1 | 4)
    with pytest.warns(RuntimeWarning):
        raise_psf_flags(8)


def _write_exr(path, planes, compression, lines):
    """Scan-line OpenEXR writer for the tests (OpenEXR file layout: magic, version 2, attributes, offset table, chunks of
    `lines` rows stored channel by channel; ZIP/ZIPS/RLE chunks byte-split and delta-predicted like ImfZip.cpp)."""
    import struct
    import zlib
    names = sorted(planes)
    H, W = planes[names[0]].shape
    ptype = {np.dtype("uint32"): 0, np.dtype("float16"): 1, np.dtype("float32"): 2}

    def attr(name, typ, payload):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload

    chl = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", ptype[planes[n].dtype], 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<4i", 0, 0, W - 1, H - 1)
    head = struct.pack("<ii", 20000630, 2) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([compression])) \
        + attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") \
        + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) \
        + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"

    def rle(b):
        out, i = bytearray(), 0
        while i < len(b):
            j = i
            while j + 1 < len(b) and b[j + 1] == b[i] and j - i < 126:
                j += 1
            if j > i + 1:
                out += struct.pack("b", j - i) + b[i:i + 1]
                i = j + 1
            else:
                k = i
                while k < len(b) and k - i < 127 and not (k + 2 < len(b) and b[k] == b[k + 1] == b[k + 2]):
                    k += 1
                out += struct.pack("b", -(k - i)) + b[i:k]
                i = k
        return bytes(out)

    chunks = []
    for y0 in range(0, H, lines):
        raw = b"".join(planes[n][y].astype(planes[n].dtype.newbyteorder("<")).tobytes() for y in range(y0, min(H, y0 + lines)) for n in names)
        data = raw
        if compression in (1, 2, 3):
            t = np.frombuffer(raw, np.uint8)
            t = np.concatenate([t[0::2], t[1::2]]).astype(np.int64)
            t[1:] = (t[1:] - t[:-1] + 128) & 0xFF
            enc = t.astype(np.uint8).tobytes()
            data = zlib.compress(enc) if compression in (2, 3) else rle(enc)
            if len(data) >= len(raw):
                data = raw
        chunks.append((y0, data))
    table_at = len(head)
    pos = table_at + 8 * len(chunks)
    offs = []
    for _, data in chunks:
        offs.append(pos)
        pos += 8 + len(data)
    with open(path, "wb") as f:
        f.write(head + struct.pack(f"<{len(offs)}Q", *offs) + b"".join(struct.pack("<ii", y, len(d)) + d for y, d in chunks))


def test_exr_reader_and_flyingthings_realworld_loaders(tmp_path):
    """dff/exr.py (scan-line OpenEXR: NONE / RLE / ZIPS / ZIP, HALF / FLOAT / UINT) on files written here, then the two
    remaining dataset classes of the reference (dff/dataset.py:55-110 FlyingThings3D, :207-246 RealWorld) on top of it and
    through dff.factory.get_dataset with the reference's YAML keys."""
    import random
    from PIL import Image
    from dff.dataset import FlyingThings3D, RealWorld
    from dff.exr import read_exr
    from dff.factory import get_dataset
    rng = np.random.Generator(np.random.PCG64(9))
    H, W = 37, 50                                   # 37 rows: a ragged last ZIP chunk (16 rows per chunk)
    disp = (rng.random((H, W), dtype=np.float32) * 30 + 5).astype(np.float32)
    smooth = np.repeat(np.repeat(rng.random((5, 5), dtype=np.float32), 10, 1), 8, 0)[:H, :W].copy()   # runs: RLE / ZIP really compress
    for comp, lines in ((0, 1), (1, 1), (2, 1), (3, 16)):
        for arr in (disp, smooth):
            p = tmp_path / f"c{comp}.exr"
            _write_exr(p, {"Y": arr}, comp, lines)
            assert np.array_equal(read_exr(p), arr), comp
    rgbh = {"R": disp.astype(np.float16), "G": (disp * 2).astype(np.float16), "B": smooth.astype(np.float16), "A": np.ones((H, W), np.float16)}
    _write_exr(tmp_path / "rgba.exr", rgbh, 3, 16)
    got = read_exr(tmp_path / "rgba.exr")
    assert got.shape == (H, W, 4) and got.dtype == np.float32                                        # OpenCV order: B G R A
    assert np.array_equal(got[..., 0], rgbh["B"].astype(np.float32)) and np.array_equal(got[..., 2], rgbh["R"].astype(np.float32))
    _write_exr(tmp_path / "zu.exr", {"Z": disp, "id": rng.integers(0, 1 << 30, (H, W)).astype(np.uint32)}, 2, 1)
    assert read_exr(tmp_path / "zu.exr").shape == (H, W, 2)
    bad = bytearray((tmp_path / "c3.exr").read_bytes())
    bad[bad.index(b"compression\0compression\0") + 28] = 4                                          # PIZ
    (tmp_path / "piz.exr").write_bytes(bytes(bad))
    with pytest.raises(NotImplementedError, match="PIZ"):
        read_exr(tmp_path / "piz.exr")
    with pytest.raises(ValueError, match="not an OpenEXR"):
        (tmp_path / "x.exr").write_bytes(b"12345678")
        read_exr(tmp_path / "x.exr")

    ft = tmp_path / "ft"
    imgs = {}
    for scene in ("a", "b"):
        (ft / scene).mkdir(parents=True)
        _write_exr(ft / scene / "disp.exr", {"Y": disp}, 3, 16)
        for name in ("10.5", "20", "40", "AiF"):
            imgs[scene, name] = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
            Image.fromarray(imgs[scene, name]).save(ft / scene / f"{name}.png")
    ds = FlyingThings3D(str(ft), resize=(H, W), train=False)
    aif, depth = ds[0]
    assert len(ds) == 2 and aif.shape == (3, H, W) and depth.shape == (1, H, W)
    assert torch.equal(aif, torch.from_numpy((imgs[ds.scenes[0], "AiF"] / 255.0).astype("float32")).permute(2, 0, 1))     # RGB
    assert torch.equal(depth[0], torch.from_numpy(disp / 20))
    random.seed(1)
    stack, depth2, dists = FlyingThings3D(str(ft), resize=(20, 30), train=False, fs_num=2)[1]
    random.seed(1)
    picked = random.sample(["10.5", "20", "40"], 2)
    assert stack.shape == (2, 3, 20, 30) and depth2.shape == (1, 20, 30) and dists.tolist() == [float(n) / 20 for n in picked]
    want0 = torch.nn.functional.interpolate(torch.from_numpy(imgs[ds.scenes[1], picked[0]][..., ::-1].astype(np.float32) / np.float32(255)).permute(2, 0, 1)[None],
                                            size=(20, 30), mode="bilinear", align_corners=False)[0]
    assert torch.equal(stack[0], want0)                                                               # B G R, plain bilinear
    np.random.seed(3)
    random.seed(2)
    s3, d3, f3 = FlyingThings3D(str(ft), resize=(20, 30), train=True, fs_num=3)[0]                    # augmentation on a stack
    assert s3.shape == (3, 3, 20, 30) and d3.shape == (1, 20, 30) and len(f3) == 3

    rw = tmp_path / "rw" / "scene0"
    (rw / "align").mkdir(parents=True)
    (rw / "depth").mkdir()
    shots = {}
    for i, mm in enumerate((600, 1200, 2500)):
        shots[mm] = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        Image.fromarray(shots[mm]).save(rw / "align" / f"{i:02d}_dist{mm}_f2.png")
    d16 = rng.integers(0, 65536, (H, W)).astype(np.uint16)
    Image.fromarray(d16).save(rw / "depth" / "depth.png")
    stack, depth, dists = RealWorld(str(tmp_path / "rw"), resize=(H, W), depth=True)[0]
    assert stack.shape == (3, 3, H, W) and dists.tolist() == [0.6, 1.2, 2.5]
    assert torch.equal(stack[1], torch.from_numpy(shots[1200][..., ::-1].astype(np.float32) / np.float32(255)).permute(2, 0, 1))
    assert torch.equal(depth[0], torch.from_numpy(((d16 / 65535 * 3000 + 500) / 1000).astype("float32")))
    assert float(RealWorld(str(tmp_path / "rw"), resize=(H, W))[0][1].abs().max()) == 0.0            # depth=False: zeros
    args = {"res": (20, 30), "train": {"dataset": "FlyingThings3D"}, "test": {"dataset": "RealWorld"}, "FlyingThings3D_train": str(ft),
            "RealWorld_val": str(tmp_path / "rw")}
    np.random.seed(0)
    tr, te = get_dataset(args)
    assert len(tr) == 2 and tr[0][0].shape == (3, 20, 30) and tr[0][1].shape == (1, 20, 30) and te[0][0].shape == (3, 3, 20, 30)


def test_public_signatures_match_the_reference(golden_dir):
    """Fixture G16 (tests/golden/make_signatures.py: names and defaults of the reference's functions and methods, read from
    its source by ast): every one this package also defines takes the reference's parameters in the reference's order with
    the reference's defaults; extra parameters are allowed only AFTER them and only with defaults (opt-in switches)."""
    import ast
    import importlib
    import inspect
    import json
    ref = json.load(open(os.path.join(golden_dir, "g16_signatures.json")))

    def norm(src):
        try:
            v = ast.literal_eval(src)
            return round(v, 12) if isinstance(v, float) else (list(v) if isinstance(v, tuple) else v)
        except Exception:
            return src                                            # a name (DEPTH, GEO_SPP, DEFAULT_WAVE ...): compared as text

    consts = {}
    import deeplens.basics as basics
    for k in dir(basics):
        if k.isupper():
            consts[k] = getattr(basics, k)
    checked, extras, missing, bad = 0, [], [], []

    def check(cond, msg):
        if not cond:
            bad.append(msg)
        return cond

    for modname, entries in ref.items():
        mod = importlib.import_module(modname)
        for qual, want in entries.items():
            obj = mod
            for part in qual.split("."):
                obj = getattr(obj, part, None)
                if obj is None:
                    break
            if obj is None or not callable(obj):
                missing.append(f"{modname}.{qual}")
                continue
            try:
                got = list(inspect.signature(obj).parameters.values())
            except (TypeError, ValueError):
                continue
            if "." in qual and not isinstance(inspect.getattr_static(getattr(mod, qual.split(".")[0]), qual.split(".")[1]), (staticmethod,)) \
                    and got and got[0].name != "self" and want and want[0][0] == "self":
                want = want[1:]                                   # bound through the class: inspect drops nothing, but be lenient
            names = [p.name if p.kind not in (p.VAR_POSITIONAL, p.VAR_KEYWORD) else ("*" if p.kind == p.VAR_POSITIONAL else "**") + p.name for p in got]
            wn = [w[0] for w in want]
            if not check(names[:len(wn)] == wn, f"{modname}.{qual}: parameters {names} do not start with the reference's {wn}"):
                continue
            for p, (_, dflt) in zip(got, want):
                if dflt is None:
                    check(p.default is inspect.Parameter.empty or p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD), f"{modname}.{qual}: '{p.name}' has a default the reference lacks")
                    continue
                if not check(p.default is not inspect.Parameter.empty, f"{modname}.{qual}: '{p.name}' lost its default {dflt}"):
                    continue
                d = norm(dflt)
                if isinstance(d, str) and d in consts:
                    d = consts[d]
                mine = p.default
                mine = list(mine) if isinstance(mine, tuple) else (round(mine, 12) if isinstance(mine, float) else mine)
                if isinstance(d, str) and not isinstance(mine, str):
                    continue                                      # an expression (e.g. torch.device('cpu')): not comparable as data
                check(mine == d, f"{modname}.{qual}: default of '{p.name}' is {mine!r}, the reference has {dflt}")
            for p in got[len(wn):]:
                check(p.default is not inspect.Parameter.empty or p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD), f"{modname}.{qual}: extra parameter '{p.name}' without a default")
                extras.append(f"{modname}.{qual}:{p.name}")
            checked += 1
    print(f"\n{checked} signatures equal to the reference's; opt-in extras: {extras}; not defined here: {len(missing)}")
    assert not bad, "\n".join(bad)
    assert checked >= 80
    hot = ["deeplens.optics.Lensgroup.psf", "deeplens.optics.Lensgroup.psf_map", "deeplens.optics.Lensgroup.refocus", "deeplens.optics.Lensgroup.trace",
           "deeplens.optics.Lensgroup.render_single_img", "deeplens.optics.Lensgroup.analysis", "deeplens.psfnet.PSFNet.render",
           "deeplens.psfnet.PSFNet.calc_psf_map", "deeplens.psfnet.PSFNet.evaluate_psf_score", "deeplens.psfnet.PSFNet.train_psfnet",
           "deeplens.render_psf.render_psf_map", "deeplens.render_psf.local_psf_render", "dff.utils.select_focus_dist", "dff.factory.get_dataset"]
    assert not [h for h in hot if h in missing], [h for h in hot if h in missing]


def test_pfm_reader(tmp_path):
    """pfmreader.read_pfm / read_and_clean_pfm (reference: pfmreader.py:5-50) on files written here: both byte orders, one and
    three channels, bottom-up rows, inf / NaN cleaned; then 0_warm_up_with_pfm.py's disparity -> depth rule."""
    from pfmreader import disparity_to_depth_mm, read_and_clean_pfm, read_pfm
    rng = np.random.Generator(np.random.PCG64(2))
    a = rng.random((5, 7), dtype=np.float32) * 200
    a[1, 2], a[3, 3], a[4, 0] = np.inf, np.nan, -np.inf
    for endian, scale in (("<", b"-1.0"), (">", b"1.0")):
        p = tmp_path / f"d{scale.decode()}.pfm"
        p.write_bytes(b"Pf\n7 5\n" + scale + b"\n" + np.flipud(a).astype(endian + "f4").tobytes())
        got = read_pfm(str(p))
        assert got.shape == (5, 7) and np.array_equal(got, a, equal_nan=True)
        clean = read_and_clean_pfm(str(p))
        assert clean[1, 2] == 0 and clean[3, 3] == 0 and clean[4, 0] == 0 and np.array_equal(clean[0], a[0])
    c = rng.random((4, 3, 3), dtype=np.float32)
    (tmp_path / "c.pfm").write_bytes(b"PF\n3 4\n-1.0\n" + np.flipud(c).astype("<f4").tobytes())
    assert np.array_equal(read_pfm(str(tmp_path / "c.pfm")), c)
    (tmp_path / "bad.pfm").write_bytes(b"Pf\nseven five\n-1.0\n")
    with pytest.raises(Exception, match="Malformed PFM header"):
        read_pfm(str(tmp_path / "bad.pfm"))
    z = disparity_to_depth_mm(np.float32(100.0), 4161.221, 176.252, 209.059)
    assert z == pytest.approx(4161.221 * 176.252 / 309.059)


def test_strict_count_table_logic():
    """aadff/strict_stack.py: any-bits -> the reference's iteration count (`while (|ft| > 5e-5).any() and it < 10`,
    deeplens/surfaces.py:547) and the acceptance test of a speculated count, against a literal loop."""
    from aadff.strict_stack import counts_of_masks, prediction_holds
    rng = np.random.default_rng(0)
    masks = np.concatenate((rng.integers(0, 1024, 500), [0, 1, 3, 7, 1023, 511, 0b1011, 0b0111111111])).astype(np.uint32)

    def loop_count(m):                      # the reference's loop on the per-iteration any() results
        it = 0
        above = True                        # enters with ft = MAXT
        while above and it < 10:
            it += 1
            above = bool((m >> (it - 1)) & 1)
        return it
    want = np.array([loop_count(int(m)) for m in masks])
    assert np.array_equal(counts_of_masks(masks), want)
    curved = np.ones(1, dtype=bool)
    for m, n_true in zip(masks, want):
        for n in range(1, 11):
            seen = np.uint32(m) & np.uint32((1 << n) - 1)          # a fused launch only sees the iterations it ran
            assert bool(prediction_holds(np.array([[seen]]), np.array([[n]]), curved)[0]) == (n == n_true), (m, n, n_true)
    # flat surfaces are ignored
    assert prediction_holds(np.array([[0, 5]], dtype=np.uint32), np.array([[1, 4]]), np.array([True, False]))[0]
    # check_counts: the first failing surface IN CROSSING ORDER is corrected (too many iterations: the first clear bit + 1; too few:
    # n + 1), nothing behind it is touched; rows that hold are returned unchanged
    from aadff.strict_stack import check_counts
    true = np.array([0x3ff, 0b0111, 0b01111, 0b011], dtype=np.uint32)          # counts 10, 4, 5, 3
    cur = np.ones(4, dtype=bool)
    seen = lambda pred: true & ((np.uint32(1) << np.asarray(pred).astype(np.uint32)) - np.uint32(1))
    for pred, want_ok, want_fix in (([10, 4, 5, 3], True, [10, 4, 5, 3]), ([9, 4, 5, 3], False, [10, 4, 5, 3]), ([10, 6, 5, 3], False, [10, 4, 5, 3]),
                                    ([10, 3, 5, 3], False, [10, 4, 5, 3]), ([10, 4, 4, 4], False, [10, 4, 5, 4])):
        ok, fix = check_counts(seen(pred)[None], np.array([pred]), cur, range(4))
        assert ok.tolist() == [want_ok] and fix.tolist() == [want_fix], (pred, ok, fix)
    ok, fix = check_counts(seen([10, 4, 4, 4])[None], np.array([[10, 4, 4, 4]]), cur, [3, 2, 1, 0])      # backward trace: surface 3 comes first
    assert ok.tolist() == [False] and fix.tolist() == [[10, 4, 4, 3]]
    both = np.array([[[10, 4, 4, 4], [10, 4, 5, 3]]])                                                    # [batch, phase, surface]
    ok, fix = check_counts(np.stack([seen(both[0, 0]), seen(both[0, 1])])[None], both, cur, range(4))
    assert ok.tolist() == [[False, True]] and fix.tolist() == [[[10, 4, 5, 4], [10, 4, 5, 3]]]


def test_host_pupil_points_equal_torch_bit_for_bit():
    """aadff_host_pupil_points (strict mode's pupil / aperture points in three library calls per stack) against the tensor operations
    of the reference (deeplens/optics.py:480-486, deeplens/surfaces.py:188-199) that `_pupil_points` restates: same bits, for the
    reference's call-by-call [2048] shape, the batched shape and ragged lengths; and the routines torch itself uses on this machine
    are found (a silent fall-back to the tensor path would cost the strict mode ~1 ms per stack)."""
    from aadff import strict_stack as ss
    vec = ss._sleef()
    assert vec is not None, "the vector cos / sin / sqrt routines of torch's CPU kernels were not found or do not reproduce torch"
    g = torch.Generator().manual_seed(77)
    per = 3 * 2048 + 2 * 999
    u = torch.rand(5 * per, generator=g)
    for n, off_t, off_r, radius, z in ((2048, 0, 2048, 7.25, -3.5), (999, 3 * 2048, 3 * 2048 + 999, torch.tensor(3.0177), 0.0),
                                       (1, 5, 9, 1e-3, 12.0), (2048, 2048, 4096, np.float64(6.1), np.float32(0.3))):
        t_off = np.arange(5, dtype=np.int64) * per + off_t
        r_off = np.arange(5, dtype=np.int64) * per + off_r
        have = torch.empty(5, n, 3)
        ss._pupil_rows(vec, u, t_off, r_off, n, radius, z, have)
        rows = u.view(5, per)
        batched = ss._pupil_points(rows[:, off_t:off_t + n], rows[:, off_r:off_r + n], radius, z)
        assert torch.equal(batched.contiguous().view(torch.int32), have.view(torch.int32))
        for i in range(5):                                                   # the reference's own shape: one [n] call per batch
            one = ss._pupil_points(u[t_off[i]:t_off[i] + n].clone(), u[r_off[i]:r_off[i] + n].clone(), radius, z)
            assert torch.equal(one.view(torch.int32), have[i].view(torch.int32))
    with pytest.raises(RuntimeError, match="kind"):
        ss._pupil_rows(vec[:3] + (5, None), u, t_off, r_off, 4, 1.0, 0.0, torch.empty(5, 4, 3))


def test_two_pass_row_draws_equal_one_pass_bit_exactly():
    """aadff_host_mt19937_rows (round 5: the focus draws of every slice first, the PSF rows from per-row generator snapshots behind
    the refocus launch): the block equals torch.rand of the whole block bit for bit, the heads alone after pass 1, and torch's
    generator is left exactly where the single draw leaves it - for the bench layout and for ragged sizes that straddle the
    generator's 624-word regenerations."""
    import ctypes as C
    from aadff import _abi
    lib = _abi.load_library()
    for S, per, head, seed in ((10, 20480, 4096, 5), (3, 1000, 7, 1), (4, 624, 624, 2), (5, 625, 0, 3), (1, 5, 2, 4)):
        torch.manual_seed(seed)
        torch.rand(11)                                       # a generator that is not at a regeneration boundary
        st0 = torch.get_rng_state()
        want = torch.rand(S * per)
        after = torch.get_rng_state()
        st = st0.clone()
        out = torch.full((S * per,), -1.0)
        snaps = torch.empty(S * 2504, dtype=torch.uint8)
        assert lib.aadff_host_mt19937_rows(C.c_void_p(st.data_ptr()), st.numel(), S, per, head, C.c_void_p(out.data_ptr()), C.c_void_p(snaps.data_ptr()), 0) == 0
        assert torch.equal(st, after)
        assert torch.equal(out.view(S, per)[:, :head], want.view(S, per)[:, :head]) and bool((out.view(S, per)[:, head:] == -1).all())
        assert lib.aadff_host_mt19937_rows(None, 5056, S, per, head, C.c_void_p(out.data_ptr()), C.c_void_p(snaps.data_ptr()), 1) == 0
        assert torch.equal(out, want)


def test_vectorised_host_reductions_equal_the_call_by_call_forms(monkeypatch):
    """Round 6: the per-slice host arithmetic of a strict / edge stack (np.mean per slice, torch.sum / atan per slice, psf_diff's
    object points per slice: deeplens/optics.py:1175-1178, :1205-1217, :945-950) as one array operation each.  Bit for bit the
    call-by-call forms - also with slices that need the reference's filtering (dead rays, NaN, non-positive distances) - and the
    self-check switches a form off (keeping the call-by-call result) when a build of numpy / torch does not agree."""
    from aadff import strict_stack as ss
    rng = np.random.default_rng(11)

    class FakeLens:
        r_last, sensor_size = 21.64, [30.60358, 30.60358]

        def entrance_pupil(self):
            return 19.809343, 13.351469

    lens = FakeLens()
    S = 10
    for trial in range(60):
        fd = (59.6 + rng.random((S, 2048), dtype=np.float32) * 0.05).astype(np.float32)
        alive = np.ones((S, 2048), dtype=bool)
        if trial % 3:
            alive[rng.integers(S), rng.integers(2048, size=40)] = False
            fd[rng.integers(S), rng.integers(2048)] = np.nan
            fd[rng.integers(S), rng.integers(2048)] = -1.0
        assert ss._d_sensor_of(fd, alive) == ss._d_sensor_loop(fd, alive)
        tan = torch.from_numpy(rng.random((S, 100), dtype=np.float32) * 0.9 - 0.45)
        ra = (torch.from_numpy(rng.random((S, 100), dtype=np.float32)) > 0.15).float()
        h, f, n = ss._fov_of(lens, tan, ra)
        assert h == ss._fov_loop(tan, ra) and f == [lens.r_last / np.tan(v) for v in h]
        pts = torch.from_numpy(rng.random((121, 3), dtype=np.float32) * 2 - 1)
        pts[:, 2] = -float(400 + rng.random() * 6000)
        hf = [float(np.float32(0.3 + rng.random() * 0.2)) for _ in range(S)]
        a, b = ss._object_points(lens, pts, hf), ss._object_points_loop(lens, pts, hf)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    # a slice with no countable ray: NaN mean -> the reference's assertion
    fd = np.full((2, 2048), 59.0, dtype=np.float32)
    alive = np.ones((2, 2048), dtype=bool)
    alive[1] = False
    with pytest.raises(AssertionError, match="sensor position is negative"):
        ss._d_sensor_of(fd, alive)
    # all-NaN tangent sums fall back to 0.5 rad like the reference
    tan, ra = torch.full((2, 100), 0.3), torch.ones(2, 100)
    ra[1] = 0
    assert ss._fov_of(lens, tan, ra)[0][1] == 0.5 == ss._fov_loop(tan, ra)[1]
    # a build on which a vectorised form disagrees: switched off with a warning, the call-by-call result is what comes back
    monkeypatch.setattr(ss, "_POINTS_CHECKED", [0])
    monkeypatch.setitem(ss._HostFast.left, "points", ss._HostFast.CHECKS)
    real = ss._object_points_loop
    monkeypatch.setattr(ss, "_object_points_loop", lambda *a: real(*a) + 1.0)
    with pytest.warns(RuntimeWarning, match="switched off"):
        got = ss._object_points(lens, pts, hf)
    assert torch.equal(got, real(lens, pts, hf) + 1.0) and not ss._HostFast.use("points")
    monkeypatch.setattr(ss, "_object_points_loop", real)
    assert torch.equal(ss._object_points(lens, pts, hf), real(lens, pts, hf))
    monkeypatch.setitem(ss._HostFast.left, "points", 0)


def test_two_variant_words_pick_exactly_the_neighbouring_pairs():
    """`strict_stack._two_variant_words`: a batch becomes a two-variant psf_map job (aadff_strict_psf_points_alt) when exactly ONE curved
    surface of its chief row has two NEIGHBOURING counts on record - and only then."""
    from aadff import strict_stack as ss
    MS = _abi.MAX_SURF
    curved = np.zeros(MS, dtype=bool)
    curved[[0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 11]] = True
    bit = lambda *ns: np.uint16(sum(1 << n for n in ns))
    seen = np.zeros((7, MS), dtype=np.uint16)
    seen[:, :12] = bit(3)
    seen[1, 9] = bit(4, 5)                    # the case of rf50mm: 4 <-> 5 at the second aspheric surface
    seen[2, 9] = bit(4, 6)                    # not neighbours
    seen[3, 9], seen[3, 8] = bit(4, 5), bit(2, 3)   # two undecided surfaces
    seen[4, 9] = bit(4, 5, 6)                 # three counts
    seen[5, 5] = bit(1, 2)                    # the stop: not a curved surface, ignored
    seen[6, 0] = bit(9, 10)                   # the top of the range
    w = ss._two_variant_words(seen, curved)
    assert w.dtype == np.int32 and w.tolist() == [-1, 9 | 4 << 8, -1, -1, -1, -1, 0 | 9 << 8]
