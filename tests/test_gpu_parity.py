"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C
ABI of include/aadff.h, against (a) the golden vectors generated from the reference and
(b) the oracle on the same seeded inputs.

Tolerances (BASELINE.json north_star): rendered images <= 1e-4 relative L2;
PSFs <= 2e-3 relative L2 (fp32 trace noise floor ~5e-4); validity-mask mismatches <= 1e-4
of rays; focus scalars <= 1e-5 relative.  Pure convolutions are held to 2e-6 abs.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from aadff import _abi                                   # noqa: E402
from aadff.focal_stack import StackPlan, render_focal_stack_m1, render_focal_stack_m2   # noqa: E402
from aadff.synth import mlp_state_dict, synth_depth_mm, synth_rgb           # noqa: E402
from deeplens import monte_carlo as dl_mc                # noqa: E402
import importlib                                          # noqa: E402
rp = importlib.import_module("deeplens.render_psf")       # the package re-exports a same-named function
from deeplens.basics import GEO_SPP, WAVE_RGB, Ray        # noqa: E402
from deeplens.optics import Lensgroup                     # noqa: E402
from deeplens.psfnet import PSFNet, ThinLens              # noqa: E402
from oracle import conv as oconv                          # noqa: E402
from oracle import psfnet as opsf                         # noqa: E402
from oracle.lens import OracleLens                        # noqa: E402

DEV = "cuda:0"
IMG_TOL, PSF_TOL, CONV_ATOL = 1e-4, 2e-3, 2e-6


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def tt(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def lens_path(repo_root, name="rf50mm"):
    return os.path.join(repo_root, "lenses", name, "lens.json")


def test_native_library_is_loaded():
    lib = _abi.require_gpu()
    n_cu, lds = C.c_int(), C.c_int()
    arch = C.create_string_buffer(64)
    assert lib.aadff_device_info(C.byref(n_cu), C.byref(lds), arch, 64) == 0
    assert arch.value.decode().startswith("gfx950"), arch.value
    assert n_cu.value == 256
    assert any("libaadff.so" in l for l in open("/proc/self/maps"))


# ================================================================= G5: convolutions
@pytest.fixture(scope="module")
def g5(golden_dir):
    return np.load(os.path.join(golden_dir, "g5_conv_small.npz"))


@pytest.mark.parametrize("tag", list("abcde"))
def test_render_psf_map_golden(g5, tag):
    img, pm, grid = tt(g5[f"map_{tag}_img"]).to(DEV), tt(g5[f"map_{tag}_psf"]).to(DEV), int(g5[f"map_{tag}_grid"])
    out = rp.render_psf_map(img, pm, grid)
    assert out.is_cuda and out.shape == img.shape
    assert np.abs(out.cpu().numpy() - g5[f"map_{tag}_out"]).max() <= CONV_ATOL


@pytest.mark.parametrize("tag", list("ab"))
def test_render_psf_golden(g5, tag):
    out = rp.render_psf(tt(g5[f"uni_{tag}_img"]).to(DEV), tt(g5[f"uni_{tag}_psf"]).to(DEV))
    assert np.abs(out.cpu().numpy() - g5[f"uni_{tag}_out"]).max() <= CONV_ATOL


@pytest.mark.parametrize("tag,ks", [("a", 11), ("b", 5), ("c", 3), ("3d", 5)])
def test_local_psf_render_golden(g5, tag, ks):
    out = rp.local_psf_render(tt(g5[f"loc_{tag}_img"]).to(DEV), tt(g5[f"loc_{tag}_psf"]).to(DEV), kernel_size=ks)
    assert out.shape == g5[f"loc_{tag}_out"].shape
    assert np.abs(out.cpu().numpy() - g5[f"loc_{tag}_out"]).max() <= CONV_ATOL


def test_local_psf_render_high_res_keeps_seams(g5):
    out = rp.local_psf_render_high_res(tt(g5["hr_img"]).to(DEV), tt(g5["hr_psf"]).to(DEV), patch_size=[16, 20], kernel_size=11)
    assert np.abs(out.cpu().numpy() - g5["hr_out"]).max() <= CONV_ATOL


def test_cpu_tensors_round_trip_through_the_gpu(g5):
    out = rp.render_psf_map(tt(g5["map_a_img"]), tt(g5["map_a_psf"]), 5)
    assert not out.is_cuda
    assert np.abs(out.numpy() - g5["map_a_out"]).max() <= CONV_ATOL


@pytest.mark.parametrize("ks", [3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 31])
def test_render_psf_map_all_kernel_sizes_vs_oracle(ks):
    """fast path (templated ks) and generic path (17, 31) against the oracle."""
    rng = np.random.Generator(np.random.PCG64(ks))
    g, H, W = 3, 70, 101
    img = tt(rng.random((2, 3, H, W), dtype=np.float32))
    pm = tt(rng.random((3, g * ks, g * ks), dtype=np.float32)) / (ks * ks)
    want = oconv.render_psf_map(img, pm, g).numpy()
    got = rp.render_psf_map(img.to(DEV), pm.to(DEV), g).cpu().numpy()
    assert np.abs(got - want).max() <= 4e-6


@pytest.mark.parametrize("H,W,g", [(480, 640, 11), (11, 11, 11), (33, 1000, 7), (256, 256, 1), (97, 64, 64)])
def test_render_psf_map_ragged_shapes_vs_oracle(H, W, g):
    rng = np.random.Generator(np.random.PCG64(H * 7 + W))
    ks = 11 if min(H, W) > 5 else 3
    if min(H, W) <= 11:
        ks = 3
    img = tt(rng.random((1, 3, H, W), dtype=np.float32))
    pm = tt(rng.random((3, g * ks, g * ks), dtype=np.float32)) / (ks * ks)
    want = oconv.render_psf_map(img, pm, g).numpy()
    got = rp.render_psf_map(img.to(DEV), pm.to(DEV), g).cpu().numpy()
    assert np.abs(got - want).max() <= 4e-6


@pytest.mark.parametrize("B,H,W,g,S", [(1, 50, 50, 3, 3), (2, 97, 131, 1, 5), (1, 64, 230, 1, 17), (1, 33, 40, 4, 10),
                                        (1, 201, 97, 2, 8), (1, 120, 120, 11, 13), (1, 25, 300, 2, 4)])
def test_render_psf_map_stack_slice_batched_path_vs_oracle(B, H, W, g, S):
    """The ks = 11 stack path (slices batched on the MFMA M dimension; >= 3 slices): chunk tails (S not a multiple
    of 4), more than 16 slices (several passes), patches wider than one 96-column tile, bands shorter than 24 rows,
    odd patch origins (unaligned 8-byte stores), scaled inputs.  Oracle = the reference loop, slice by slice."""
    rng = np.random.Generator(np.random.PCG64(B * 1000 + H * 7 + W + S))
    ks = 11
    img = tt(rng.random((B, 3, H, W), dtype=np.float32)) * 37.5 - 3.0          # not in [0,1]: exercises the tile pre-scale
    maps = tt(rng.random((S, 3, g * ks, g * ks), dtype=np.float32)) / (ks * ks)
    maps[S // 2] *= 1e-3                                                       # per-slice tap pre-scale
    got = rp.render_psf_map_stack(img.to(DEV), maps.to(DEV), g).cpu().numpy()
    assert got.shape == (B, 3, S, H, W)
    for s in range(S):
        want = oconv.render_psf_map(img, maps[s], g).numpy()
        assert np.abs(got[:, :, s] - want).max() <= 4e-6 * 40, f"slice {s}"


@pytest.mark.parametrize("B,Cn,H,W,g,S,ks", [(1, 3, 50, 50, 3, 1, 11), (2, 3, 97, 131, 1, 1, 11), (1, 3, 64, 230, 1, 2, 11), (1, 1, 33, 40, 4, 2, 11),
                                             (1, 3, 201, 97, 2, 1, 9), (1, 4, 120, 120, 11, 1, 11), (1, 3, 25, 300, 2, 2, 9), (1, 3, 203, 203, 2, 1, 11),
                                             (1, 3, 61, 83, 3, 7, 9), (1, 3, 13, 13, 1, 1, 11), (1, 2, 12, 200, 2, 1, 9), (1, 3, 400, 520, 3, 1, 11)])
def test_render_psf_map_block_gemm_path_vs_oracle_and_toeplitz(B, Cn, H, W, g, S, ks, monkeypatch):
    """Lone slices at ks 9 / 11 (round 4: `conv_psf_map_blk_kernel`, a 4 x 4 block of output pixels on the MFMA M dimension;
    the S = 1 entry `render_psf_map` takes it, stacks of the same inputs take the Toeplitz / slice-batched forms): patches wider than one 96-column tile, column blocks cut by the patch border inside a 4-pixel store, bands shorter
    than 24 rows and 8-row groups cut by the patch border, images smaller than one band (every row reflected), interior bands (the
    branch-free staging) next to border bands, odd patch origins (16-byte stores at 4-byte alignment), B > 1, C != 3,
    inputs far from [0, 1] (tile pre-scale) and a 1e-3 PSF (tap pre-scale).  Oracle = the reference's loop; and the Toeplitz
    form (AADFF_CONV_PATH=toeplitz), which carries the same exact operand split, to 1e-6 of the data range."""
    rng = np.random.Generator(np.random.PCG64(B * 1000 + H * 7 + W + S + ks))
    img = tt(rng.random((B, Cn, H, W), dtype=np.float32)) * 37.5 - 3.0
    maps = tt(rng.random((S, Cn, g * ks, g * ks), dtype=np.float32)) / (ks * ks)
    maps[S // 2, 0] *= 1e-3
    got = rp.render_psf_map_stack(img.to(DEV), maps.to(DEV), g)
    assert got.shape == (B, Cn, S, H, W)
    one = rp.render_psf_map(img.to(DEV), maps[0].to(DEV), g)                   # S = 1: the block-GEMM kernel (S >= 2: Toeplitz / slice-batched)
    assert (one - got[:, :, 0]).abs().max().item() <= 4e-6 * 40
    monkeypatch.setenv("AADFF_CONV_PATH", "toeplitz")
    toe = rp.render_psf_map_stack(img.to(DEV), maps.to(DEV), g)
    monkeypatch.delenv("AADFF_CONV_PATH")
    assert (toe - got).abs().max().item() <= 4e-6 * 40
    gotn = got.cpu().numpy()
    for s in range(S):
        want = oconv.render_psf_map(img, maps[s], g).numpy()
        assert np.abs(gotn[:, :, s] - want).max() <= 4e-6 * 40, f"slice {s}"
        # every slice through the block-GEMM kernels: the default pick, the round-4 kernel (fixed 14 x 14 window, 24-row bands) and
        # the round-5 one (window ks + 3, band height by patch height) - ks 9 and tall-patch ks 11 lone slices default to the latter
        for force in (None, "0", "1"):
            if force is None:
                monkeypatch.delenv("AADFF_CONV_BLKW", raising=False)
            else:
                monkeypatch.setenv("AADFF_CONV_BLKW", force)
            lone = rp.render_psf_map(img.to(DEV), maps[s].to(DEV), g).cpu().numpy()
            assert np.abs(lone - want).max() <= 4e-6 * 40, f"lone slice {s}, AADFF_CONV_BLKW={force}"
        monkeypatch.delenv("AADFF_CONV_BLKW", raising=False)


@pytest.mark.parametrize("B,Cn,H,W,g,S,ks", [(1, 3, 50, 50, 3, 1, 13), (2, 3, 97, 131, 1, 1, 21), (1, 3, 64, 230, 1, 2, 15), (1, 1, 33, 40, 4, 2, 17),
                                             (1, 3, 201, 97, 2, 1, 19), (1, 4, 120, 120, 7, 1, 13), (1, 3, 25, 300, 2, 3, 21), (1, 3, 203, 203, 2, 1, 21),
                                             (1, 3, 61, 83, 3, 7, 15), (1, 3, 13, 13, 1, 1, 21), (1, 2, 12, 200, 2, 1, 17), (1, 3, 400, 520, 3, 1, 21)])
def test_render_psf_map_wide_block_gemm_path_vs_oracle_and_the_other_paths(B, Cn, H, W, g, S, ks, monkeypatch):
    """ks 13 ... 21 (round 5: `conv_psf_map_blkw_kernel`, the block-GEMM form with a window of up to 24 x 24 pixels per 4 x 4 output
    block; lone slices of every such ks and stacks at ks 13 / 17 / 21 - 21 is the reference's `render_single_img` size,
    deeplens/optics.py:779-783 - take it by default): the same edge cases as the ks 9 / 11 test above (patches wider than one tile, blocks and row groups cut by
    the patch border, images smaller than a band with every row and column reflected, interior bands next to border bands, odd patch
    origins, B > 1, C != 3, inputs far from [0, 1], a 1e-3 PSF), forced for every slice count (AADFF_CONV_BLKW=1) and compared with
    the oracle, with the wide Toeplitz form (=0) and with the packed-FMA / generic kernels (AADFF_CONV_PATH=valu).  Non-finite pixels:
    the contract of the GEMM forms - whatever the reference poisons is non-finite, nothing beyond 128 px is."""
    rng = np.random.Generator(np.random.PCG64(B * 1000 + H * 7 + W + S + ks))
    img = tt(rng.random((B, Cn, H, W), dtype=np.float32)) * 37.5 - 3.0
    maps = tt(rng.random((S, Cn, g * ks, g * ks), dtype=np.float32)) / (ks * ks)
    maps[S // 2, 0] *= 1e-3
    default = rp.render_psf_map_stack(img.to(DEV), maps.to(DEV), g)
    monkeypatch.setenv("AADFF_CONV_BLKW", "1")
    got = rp.render_psf_map_stack(img.to(DEV), maps.to(DEV), g)
    monkeypatch.setenv("AADFF_CONV_BLKW", "0")
    toe = rp.render_psf_map_stack(img.to(DEV), maps.to(DEV), g)
    monkeypatch.delenv("AADFF_CONV_BLKW")
    monkeypatch.setenv("AADFF_CONV_PATH", "valu")
    valu = rp.render_psf_map_stack(img.to(DEV), maps.to(DEV), g)
    monkeypatch.delenv("AADFF_CONV_PATH")
    assert got.shape == (B, Cn, S, H, W)
    tol = 4e-6 * 40
    assert (toe - got).abs().max().item() <= tol and (valu - got).abs().max().item() <= tol and (default - got).abs().max().item() <= tol
    if S == 1 or ks in (13, 17, 21):
        assert torch.equal(default, got)                                       # the default dispatch IS this kernel there
    gotn = got.cpu().numpy()
    for s in range(S):
        want = oconv.render_psf_map(img, maps[s], g).numpy()
        assert np.abs(gotn[:, :, s] - want).max() <= tol, f"slice {s}"
    # one NaN pixel
    if H > ks and W > ks:
        bad = img.clone()
        bad[0, 0, H // 2, W // 3] = float("nan")
        monkeypatch.setenv("AADFF_CONV_BLKW", "1")
        a = rp.render_psf_map_stack(bad.to(DEV), maps.to(DEV), g)
        monkeypatch.delenv("AADFF_CONV_BLKW")
        monkeypatch.setenv("AADFF_CONV_PATH", "valu")
        b = rp.render_psf_map_stack(bad.to(DEV), maps.to(DEV), g)
        monkeypatch.delenv("AADFF_CONV_PATH")
        na, nb = ~torch.isfinite(a), ~torch.isfinite(b)
        assert int(nb.sum()) > 0 and bool((na | ~nb).all())                    # everything the reference poisons is non-finite here too
        far = torch.ones(H, W, dtype=torch.bool, device=a.device)
        far[max(0, H // 2 - 128):H // 2 + 129, max(0, W // 3 - 128):W // 3 + 129] = False
        assert not bool(na[0, 0, :, far].any()) and not bool(na[0, 1:].any()) and not bool(na[1:].any())   # nothing farther than 128 px, no other plane


@pytest.mark.parametrize("Cn", [1, 2, 4, 6])
def test_channel_counts_other_than_three_vs_oracle(Cn):
    """The reference's functions are channel-generic (depthwise conv2d with groups = C; the gather stacks the kernel C
    times): grey images, pairs, RGBA and 6 planes through the single-slice conv, the slice-batched stack conv and the
    per-pixel gather (compile-time channel paths exist for C = 1 and 3 only; the rest take the runtime-C code)."""
    rng = np.random.Generator(np.random.PCG64(100 + Cn))
    B, H, W, g, ks, S = 2, 61, 83, 3, 11, 5
    img = tt(rng.random((B, Cn, H, W), dtype=np.float32))
    maps = tt(rng.random((S, Cn, g * ks, g * ks), dtype=np.float32)) / (ks * ks)
    got1 = rp.render_psf_map(img.to(DEV), maps[0].to(DEV), g).cpu().numpy()
    assert np.abs(got1 - oconv.render_psf_map(img, maps[0], g).numpy()).max() <= 4e-6
    gots = rp.render_psf_map_stack(img.to(DEV), maps.to(DEV), g).cpu().numpy()
    for s in range(S):
        assert np.abs(gots[:, :, s] - oconv.render_psf_map(img, maps[s], g).numpy()).max() <= 4e-6, s
    psf = tt(rng.random((B, H, W, ks, ks), dtype=np.float32)) / (ks * ks)
    gotl = rp.local_psf_render(img.to(DEV), psf.to(DEV), ks).cpu().numpy()
    assert np.abs(gotl - oconv.local_psf_render(img, psf, ks).numpy()).max() <= 4e-6


def test_render_psf_map_1024_golden_crops(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_conv_1024.npz"))
    img = tt(synth_rgb(1024, 1024))[None].to(DEV)
    out = rp.render_psf_map(img, tt(g["psf_map"]).to(DEV), 11)[0].cpu().numpy()
    for k in g.files:
        if k.startswith("crop_"):
            y, x = (int(v) for v in k[5:].split("_"))
            assert np.abs(out[:, y:y + 64, x:x + 64] - g[k]).max() <= CONV_ATOL, k
    assert out.astype(np.float64).sum((1, 2)) == pytest.approx(g["sums"], rel=1e-6)


def test_conv_hdr_dynamic_range_within_a_band():
    """The MFMA paths carry fp32 pixels as fp16 hi/lo pairs after ONE power-of-two scale per image tile: a tile whose
    pixels span 1e-4 .. 1e3 (seven decades) must still come out like an fp32 FMA chain, i.e. with an absolute error
    proportional to the LARGEST pixel under the PSF, not to the local value."""
    rng = np.random.Generator(np.random.PCG64(404))
    H = W = 192
    img = (10.0 ** rng.uniform(-4.0, 3.0, size=(1, 3, H, W))).astype(np.float32)
    maps = rng.random((10, 3, 33, 33), dtype=np.float32)
    maps /= maps.reshape(10, 3, 3, 11, 3, 11).sum((3, 5), keepdims=True).reshape(10, 3, 3, 1, 3, 1).repeat(11, 3).repeat(11, 5).reshape(10, 3, 33, 33)
    want = np.stack([oconv.render_psf_map(tt(img), tt(m), 3).numpy() for m in maps], 2)        # [1,3,10,H,W]
    got_stack = rp.render_psf_map_stack(tt(img).to(DEV), tt(maps).to(DEV), 3).cpu().numpy()     # slice-batched GEMM
    got_one = rp.render_psf_map(tt(img).to(DEV), tt(maps[0]).to(DEV), 3).cpu().numpy()          # Toeplitz GEMM
    bound = 1e-6 * float(img.max())
    assert np.abs(got_stack - want).max() <= bound
    assert np.abs(got_one - want[:, :, 0]).max() <= bound
    # and relative to the result itself the bulk is at fp32 level: the scale is per tile, not per image
    rel = np.abs(got_stack - want) / np.abs(want)
    assert np.median(rel) <= 2e-7


def test_conv_non_finite_pixel_poisons_a_bounded_neighbourhood():
    """One inf (or NaN) pixel: F.conv2d in the reference turns the 11x11 support non-finite (inf where the tap is
    positive, NaN where it is zero).  The GEMM forms scale every image TILE by a power of two taken from the tile's
    largest pixel, multiply the pixel with the zero padding of their K extent and split inf into (inf, NaN): they return
    NaN for the whole tile that staged the bad pixel (at most a 34 x 108 pixel band with its halo).  Contract: everything
    the reference poisons is non-finite here too, and everything farther than 128 pixels from the bad pixel is unaffected
    (documented in DESIGN.md; a non-finite input image is outside what the renderer is specified for)."""
    rng = np.random.Generator(np.random.PCG64(405))
    H = W = 352
    by, bx = 170, 181
    base = rng.random((1, 3, H, W), dtype=np.float32)
    maps = rng.random((4, 3, 22, 22), dtype=np.float32) / 60.5           # taps sum to ~1 per PSF
    for bad in (np.inf, np.nan):
        img = base.copy()
        img[0, 1, by, bx] = bad
        want = np.stack([oconv.render_psf_map(tt(img), tt(m), 2).numpy() for m in maps], 2)
        clean = np.stack([oconv.render_psf_map(tt(base), tt(m), 2).numpy() for m in maps], 2)
        for got in (rp.render_psf_map_stack(tt(img).to(DEV), tt(maps).to(DEV), 2).cpu().numpy(),
                    np.stack([rp.render_psf_map(tt(img).to(DEV), tt(m).to(DEV), 2).cpu().numpy() for m in maps], 2)):
            assert not np.isfinite(got[~np.isfinite(want)]).any()
            yy, xx = np.mgrid[0:H, 0:W]
            far = np.maximum(np.abs(yy - by), np.abs(xx - bx)) > 128
            assert far.sum() > 10000
            assert np.isfinite(got[..., far]).all() and np.abs(got[..., far] - clean[..., far]).max() <= CONV_ATOL
            assert np.isfinite(got[:, [0, 2]]).all()                  # other channels untouched


def test_custom_ops_opcheck_and_compile_tracing():
    """torch.ops.aadff.* (north_star: "HIP kernels through PyTorch-ROCm custom ops"): torch.library.opcheck validates
    schema, fake-tensor shapes and dispatch registration; a torch.compile'd function that calls the ops traces without a
    graph break (aot_eager backend: no code generation) and returns the eager result."""
    from aadff import ops  # noqa: F401
    rng = np.random.Generator(np.random.PCG64(9))
    img = tt(rng.random((1, 3, 48, 40), dtype=np.float32)).to(DEV)
    pm = tt(rng.random((3, 33, 33), dtype=np.float32)).to(DEV)
    maps = tt(rng.random((4, 3, 33, 33), dtype=np.float32)).to(DEV)
    psf = tt(rng.random((1, 48, 40, 5, 5), dtype=np.float32)).to(DEV)
    tests = ("test_schema", "test_faketensor")
    torch.library.opcheck(torch.ops.aadff.render_psf_map.default, (img, pm, 3), test_utils=tests)
    torch.library.opcheck(torch.ops.aadff.render_psf_map_stack.default, (img, maps, 3), test_utils=tests)
    torch.library.opcheck(torch.ops.aadff.local_psf_render.default, (img, psf, 5), test_utils=tests)
    torch.library.opcheck(torch.ops.aadff.thinlens_render.default, (img, -img[:, :1] * 3000 - 300, torch.tensor([-900.0], device=DEV), 11, 50.0, 1.8,
                                                                     0.02, 200.0, 20000.0), test_utils=tests)

    def f(x, m, p):
        a = torch.ops.aadff.render_psf_map_stack(x, m, 3)            # [1,3,4,H,W]
        return torch.ops.aadff.local_psf_render(a[:, :, 0].contiguous(), p, 5) + a.sum(2)

    want = f(img, maps, psf)
    got = torch.compile(f, backend="aot_eager", fullgraph=True)(img, maps, psf)
    assert torch.equal(got, want)
    assert np.abs(want.cpu().numpy() - (oconv.local_psf_render(oconv.render_psf_map(img.cpu(), maps[0].cpu(), 3), psf.cpu(), 5)
                                        + sum(oconv.render_psf_map(img.cpu(), m.cpu(), 3) for m in maps)).numpy()).max() <= 2e-4
    x = img.clone().requires_grad_(True)
    y = torch.ops.aadff.render_psf_map(x, pm, 3)                      # no autograd formula: forward-only
    with pytest.raises(RuntimeError):
        y.sum().backward()


def test_stack_fused_equals_per_slice():
    """The stack path (slice-batched GEMM, 4 slices per MFMA) and the single-slice path (Toeplitz GEMM) carry the same
    exact fp16 hi/lo operand split; they differ only in fp32 summation order."""
    rng = np.random.Generator(np.random.PCG64(3))
    S, g, ks, H, W = 10, 11, 11, 1024, 1024
    img = tt(synth_rgb(H, W))[None].to(DEV)
    maps = tt(rng.random((S, 3, g * ks, g * ks), dtype=np.float32)).to(DEV) / 121
    stack = rp.render_psf_map_stack(img, maps, g)
    assert stack.shape == (1, 3, S, H, W)
    for s in (0, 4, 9):
        assert (stack[:, :, s] - rp.render_psf_map(img, maps[s], g)).abs().max().item() <= 1e-6


def test_conv_properties_at_full_size():
    """Size-independent properties at 1024^2: delta PSF = identity, a normalised PSF keeps a
    constant image constant (reflect padding), linearity in the image."""
    g, ks, H, W = 11, 11, 1024, 1024
    img = tt(synth_rgb(H, W))[None].to(DEV)
    delta = torch.zeros(3, g, g, ks, ks)
    delta[..., ks // 2, ks // 2] = 1
    delta = delta.permute(0, 1, 3, 2, 4).reshape(3, g * ks, g * ks).to(DEV)
    # the MFMA path carries operands as hi+lo fp16 pairs (22 bits): identity to 2^-22 relative, not bit-exact
    assert (rp.render_psf_map(img, delta, g) - img).abs().max().item() <= 5e-7
    rng = np.random.Generator(np.random.PCG64(1))
    p = tt(rng.random((3, g, g, ks, ks), dtype=np.float32))
    p = (p / p.sum((-1, -2), keepdim=True)).permute(0, 1, 3, 2, 4).reshape(3, g * ks, g * ks).contiguous().to(DEV)
    const = torch.full((1, 3, H, W), 0.625, device=DEV)
    assert (rp.render_psf_map(const, p, g) - 0.625).abs().max().item() <= 2e-6
    a = rp.render_psf_map(img, p, g)
    b = rp.render_psf_map(img.flip(-1), p, g)
    ab = rp.render_psf_map(0.5 * img + 0.25 * img.flip(-1), p, g)
    assert (ab - (0.5 * a + 0.25 * b)).abs().max().item() <= 2e-6


def test_conv_argument_errors():
    img = torch.rand(1, 3, 32, 32, device=DEV)
    with pytest.raises(AssertionError, match="should be odd"):
        rp.render_psf_map(img, torch.rand(3, 8, 8, device=DEV), 2)
    with pytest.raises(RuntimeError, match="grid"):
        rp.render_psf_map(img, torch.rand(3, 3 * 40, 3 * 40, device=DEV), 40)
    with pytest.raises(RuntimeError, match="forward-only"):
        rp.render_psf_map(img.clone().requires_grad_(True), torch.rand(3, 6, 6, device=DEV), 2)


# ================================================================= G1: lens scalars, pupils, refocus
@pytest.mark.parametrize("key", ["rf50mm@1024x1024", "rf50mm@480x640", "50mm_f2.8@1024x1024"])
def test_lens_scalars_and_pupils(golden_dir, repo_root, key):
    g = json.load(open(os.path.join(golden_dir, "g1_scalars.json")))[key]
    name, res = key.split("@")
    lens = Lensgroup(lens_path(repo_root, name), sensor_res=tuple(int(v) for v in res.split("x")), device=DEV)
    for k in ("d_sensor", "hfov", "foclen", "fnum"):
        assert getattr(lens, k) == pytest.approx(g["load"][k], rel=1e-5), k
    assert lens.aper_idx == g["load"]["aper_idx"]
    assert lens.entrance_pupil() == pytest.approx(tuple(g["entrance_pupil"]), rel=1e-5)
    assert lens.exit_pupil() == pytest.approx(tuple(g["exit_pupil"]), rel=1e-5)
    assert lens.entrance_pupil(shrink_pupil=True) == pytest.approx(tuple(g["entrance_pupil_shrunk"]), rel=1e-5)


@pytest.mark.parametrize("name", ["rf50mm", "50mm_f2.8"])
def test_refocus_seeded(golden_dir, repo_root, name):
    g = json.load(open(os.path.join(golden_dir, "g1_scalars.json")))[f"{name}@1024x1024"]["refocus"]
    lens = Lensgroup(lens_path(repo_root, name), sensor_res=(1024, 1024), device=DEV)
    for f, want in g.items():
        torch.manual_seed(0)
        lens.refocus(float(f))
        for k in ("d_sensor", "hfov", "foclen", "fnum"):
            assert getattr(lens, k) == pytest.approx(want[k], rel=1e-5), (f, k)


def test_named_glass_lens_end_to_end(golden_dir, repo_root):
    """G14: a prescription whose elements NAME catalogue glasses of all three dispersion branches (COC: Schott, N-LAK34:
    Sellmeier, SF5: table n/V; deeplens/basics.py:298-336) - load scalars, pupils, refocus scalars and a PSF map against
    the reference.  (Round 2 answered named glasses with the Cauchy formula: n off by 1e-4, which moves every PSF.)"""
    g = json.load(open(os.path.join(golden_dir, "g14_glass.json")))["named_lens"]
    gm = np.load(os.path.join(golden_dir, "g14_named_psf_map.npz"))
    path = lens_path(repo_root, "rf50mm_named")
    lens = Lensgroup(path, sensor_res=(256, 256), device=DEV)
    for k in ("d_sensor", "hfov", "foclen", "fnum", "pixel_size"):
        assert getattr(lens, k) == pytest.approx(g["load"][k], rel=1e-5), k
    assert lens.aper_idx == g["load"]["aper_idx"]
    assert list(lens.entrance_pupil()) == pytest.approx(g["entrance_pupil"], rel=1e-5)
    assert list(lens.exit_pupil()) == pytest.approx(g["exit_pupil"], rel=1e-5)
    for f, want in g["refocus"].items():
        torch.manual_seed(0)
        lens.refocus(float(f))
        for k in ("d_sensor", "hfov", "foclen", "fnum"):
            assert getattr(lens, k) == pytest.approx(want[k], rel=1e-5), (f, k)
    lens = Lensgroup(path, sensor_res=(256, 256), device=DEV)
    torch.manual_seed(0)
    lens.refocus(-1500.0)
    pm = lens.psf_map(depth=-1200.0, grid=5, ks=11, spp=512).cpu().numpy()
    assert lens.d_sensor == pytest.approx(float(gm["d_sensor"]), rel=1e-5)
    assert rel_l2(pm, gm["psf_map"]) <= 2e-3


def test_four_coefficient_asphere_keeps_the_reference_wart(golden_dir, repo_root):
    """G15: ai_degree == 4 evaluates r^8 with the r^6 coefficient in the reference (surfaces.py:313); reproduced through
    the surface table (deeplens/surfaces.py: Aspheric.ai8 / pack): refocus scalars, per-surface states, PSF map."""
    g = json.load(open(os.path.join(golden_dir, "g15_ai4.json")))
    z = np.load(os.path.join(golden_dir, "g15_ai4.npz"))
    path = lens_path(repo_root, "rf50mm_ai4")
    lens = Lensgroup(path, sensor_res=(256, 256), device=DEV)
    assert lens.surfaces[8].ai_degree == 4 and float(lens.surfaces[8].ai8) == float(lens.surfaces[8].ai6) != float(lens.surfaces[8].ai[3])
    for f, want in g["refocus"].items():
        torch.manual_seed(0)
        lens.refocus(float(f))
        for k in ("d_sensor", "hfov", "foclen", "fnum"):
            assert getattr(lens, k) == pytest.approx(want[k], rel=1e-5), (f, k)
    torch.manual_seed(0)
    lens.refocus(-1500.0)
    ray = Ray(tt(z["ray_o0"]).clone(), tt(z["ray_d0"]).clone(), wvln=0.589, device=DEV)
    ray.d = tt(z["ray_d0"]).clone().to(DEV)                   # stored after Ray() normalised them: do not normalise twice
    for i, s in enumerate(lens.surfaces):
        ray = s.ray_reaction(ray)
        ra = ray.ra.cpu().numpy()
        assert np.array_equal(ra, z["states_ra"][i]), f"surface {i} validity"
        alive = ra > 0
        assert np.abs(ray.o.cpu().numpy() - z["states_o"][i])[alive].max() <= 2e-4, f"surface {i}"     # 1.5 m away: 1e-4 mm of fp32 noise
        assert np.abs(ray.d.cpu().numpy() - z["states_d"][i])[alive].max() <= 1e-5, f"surface {i}"       # 1e-4 mm x curvature
    torch.manual_seed(0)
    pm = lens.psf_map(depth=-1200.0, grid=5, ks=11, spp=512).cpu().numpy()
    assert lens.d_sensor == pytest.approx(float(z["d_sensor"]), rel=1e-5)
    assert rel_l2(pm, z["psf_map"]) <= 2e-3


def test_psf_with_no_ray_inside_the_window_is_nan_like_the_reference(repo_root):
    """A point whose rays all fall outside the ks x ks window has psf.sum() == 0 and the reference returns 0/0 = NaN for
    it (deeplens/optics.py:978, monte_carlo.py:37).  3x3 window at 1024^2 under heavy defocus, 64 rays: 21 of the 27
    (point, wavelength) PSFs are NaN in the reference - the same ones here, the others match; a NaN patch of the map turns
    exactly that patch of the rendered image into NaN (render_psf_map convolves patch by patch, render_psf.py:61-71)."""
    from oracle.lens import OracleLens
    ora = OracleLens(lens_path(repo_root), sensor_res=(1024, 1024))
    torch.manual_seed(0)
    ora.refocus(-500.0)
    torch.manual_seed(2)
    want = ora.psf_map(depth=-8000.0, grid=3, ks=3, spp=64).numpy()
    lens = Lensgroup(lens_path(repo_root), sensor_res=(1024, 1024), device=DEV)
    torch.manual_seed(0)
    lens.refocus(-500.0)
    torch.manual_seed(2)
    got_t = lens.psf_map(depth=-8000.0, grid=3, ks=3, spp=64)
    got = got_t.cpu().numpy()
    nan_w, nan_g = np.isnan(want), np.isnan(got)
    assert nan_w.sum() == 21 * 9 and np.array_equal(nan_w, nan_g)
    assert np.abs(got[~nan_g] - want[~nan_w]).max() <= 2e-2                 # a handful of rays per PSF: one ray = 1/5 of it
    img = tt(synth_rgb(96, 96))[None].to(DEV)
    out = rp.render_psf_map(img, got_t, 3).cpu().numpy()[0]                 # [3,96,96], patches of 32 x 32
    for c in range(3):
        for gi in range(3):
            for gj in range(3):
                patch = out[c, 32 * gi:32 * gi + 32, 32 * gj:32 * gj + 32]
                assert np.isnan(patch).all() == bool(nan_g[c, 3 * gi, 3 * gj]) and np.isnan(patch).any() == bool(nan_g[c, 3 * gi, 3 * gj])


def test_refocus_without_a_valid_ray_raises_like_the_reference(repo_root):
    """refocus on an object 1 mm in front of the first surface: no ray gets through, np.mean([]) is NaN and the reference
    stops with 'sensor position is negative.' (deeplens/optics.py:1176).  Same error from Lensgroup.refocus, and from a
    pipelined stack through the flags word (bit 2) although only one of its focus states fails."""
    lens = Lensgroup(lens_path(repo_root), sensor_res=(64, 64), device=DEV)
    good = lens.d_sensor
    torch.manual_seed(0)
    lens.refocus(-1.0)
    with pytest.raises(AssertionError, match="sensor position is negative"):
        lens.d_sensor
    lens = Lensgroup(lens_path(repo_root), sensor_res=(64, 64), device=DEV)
    img = tt(synth_rgb(64, 64))[None].to(DEV)
    plan = StackPlan(lens, 4, 64, 64, 1, 3, 3, 11, 256)
    torch.manual_seed(0)
    render_focal_stack_m1(lens, img, -1500.0, [-800.0, -1.0, -2000.0, -4000.0], 3, 11, 256, plan=plan, update_lens=False)
    with pytest.raises(AssertionError, match="sensor position is negative"):
        plan.check_flags()
    torch.manual_seed(0)
    render_focal_stack_m1(lens, img, -1500.0, [-800.0, -1200.0, -2000.0, -4000.0], 3, 11, 256, plan=plan, update_lens=False)
    plan.check_flags()                                                      # the flag was consumed; a healthy stack raises nothing
    assert good > 0


def test_square_aperture_is_refused_cleanly():
    """Square apertures (Surface(is_square=True) / Aspheric(square=True), deeplens/surfaces.py:17-20,330,416) are outside the
    path; they cannot come from a lens file (read_lens_json never passes the flag) and the constructors say so."""
    from deeplens.surfaces import Aspheric, Surface
    for make in (lambda: Surface(5.0, 0.0, "air", "air", is_square=True, device="cpu"),
                 lambda: Aspheric(5.0, 0.0, c=0.0, mat1="air", mat2="air", square=True, device="cpu")):
        with pytest.raises(NotImplementedError, match="square apertures"):
            make()
    # the reference overwrites `is_square` with `square` (surfaces.py:330): is_square=True alone leaves a ROUND aperture
    s = Aspheric(5.0, 0.0, c=0.0, mat1="air", mat2="air", is_square=True, device="cpu")
    assert s.is_square is False and s.h == pytest.approx(5.0 * np.sqrt(2))


# ================================================================= G2/G3: trace and splat
@pytest.fixture(scope="module")
def g23(golden_dir):
    return np.load(os.path.join(golden_dir, "g2_g3_trace_splat.npz"))


@pytest.fixture(scope="module")
def lens_foc2000(repo_root):
    lens = Lensgroup(lens_path(repo_root), sensor_res=(1024, 1024), device=DEV)
    torch.manual_seed(0)
    lens.refocus(-2000.0)
    return lens


def test_per_surface_states(g23, lens_foc2000):
    """Surface by surface through Aspheric.ray_reaction -> aadff_trace_rays."""
    ray = Ray(tt(g23["ray_o0"]).clone(), tt(g23["ray_d0"]).clone(), wvln=0.589, device=DEV)
    for i, s in enumerate(lens_foc2000.surfaces):
        ray = s.ray_reaction(ray)
        ra = ray.ra.cpu().numpy()
        assert np.array_equal(ra, g23["states_ra"][i]), f"surface {i} validity"
        alive = ra > 0
        # positions carry |t|*ulp error from surface 0, reached from t~1500 mm where one fp32 ulp
        # is 1.2e-4 mm; later surfaces inherit it
        assert np.abs(ray.o.cpu().numpy() - g23["states_o"][i])[alive].max() <= 4e-4, f"surface {i} o"
        assert np.abs(ray.d.cpu().numpy() - g23["states_d"][i])[alive].max() <= 2e-6, f"surface {i} d"
        # dead rays keep their last state (the reference leaves them in place too)
        assert np.abs(ray.o.cpu().numpy() - g23["states_o"][i])[~alive].max(initial=0) <= 4e-4


def _trace_points(lens, pobj, u_theta, u_r, spp, wvln=0.589, shrunk=False):
    pz, pr = lens.entrance_pupil(shrink_pupil=shrunk)
    N = pobj.shape[0]
    # keep the uploads referenced until the launch: an inline `.to(DEV)` temporary is freed (and
    # its block reused by the next upload) before the kernel is even enqueued
    pobj, u_theta, u_r = pobj.to(DEV).contiguous(), u_theta.to(DEV), u_r.to(DEV)
    o = torch.empty((spp, N, 3), device=DEV)
    d = torch.empty((spp, N, 3), device=DEV)
    ra = torch.empty((spp, N), device=DEV)
    _abi.call("aadff_trace_points", _abi.ptr(pobj), N, _abi.ptr(u_theta), _abi.ptr(u_r),
              spp, float(pz), float(pr), _abi.ptr(lens._table([wvln])), len(lens.surfaces), _abi.ptr(lens._state_device()),
              _abi.ptr(o), _abi.ptr(d), _abi.ptr(ra), _abi.stream_ptr(torch.device(DEV)))
    return o, d, ra


def test_sensor_hits_from_stored_uniforms(g23, lens_foc2000):
    lens = lens_foc2000
    assert lens.d_sensor == pytest.approx(float(g23["d_sensor"]), rel=1e-5)
    o, d, ra = _trace_points(lens, tt(g23["points_obj"]), tt(g23["u_theta"]), tt(g23["u_r"]), 256)
    ra_h, want_ra = ra.cpu().numpy() > 0, g23["sensor_ra"] > 0
    assert (ra_h != want_ra).mean() <= 1e-4
    both = ra_h & want_ra
    err = np.abs(o[..., :2].cpu().numpy() - g23["sensor_xy"])[both]
    assert err.mean() <= 2e-5 and err.max() <= 5e-4          # fp32-vs-fp64 floor: mean 5e-6, max 4e-5 (Appendix D)
    assert np.abs(d.cpu().numpy() - g23["final_d"])[both].max() <= 5e-6


def test_splat_matches_reference_histogram(g23, lens_foc2000):
    """aadff_psf_splat on the reference's own sensor hits: isolates the histogram."""
    o = torch.zeros((256, 121, 3))
    o[..., :2] = tt(g23["sensor_xy"])
    ray = Ray.__new__(Ray)
    ray.o, ray.ra, ray.d = o.to(DEV), tt(g23["sensor_ra"]).float().to(DEV), None
    raw = dl_mc.forward_integral(ray, ps=float(g23["pixel_size"]), ks=11, pointc_ref=tt(g23["centre"]).to(DEV))
    assert np.abs(raw.cpu().numpy() - g23["psf_raw"]).max() <= 2e-4      # atomics: sum order
    nrm = raw / raw.sum((-1, -2), keepdim=True)
    assert rel_l2(nrm.cpu().numpy(), g23["psf"]) <= 1e-5


def test_chief_ray_centres_and_psfs_from_stored_uniforms(g23, lens_foc2000):
    """Fused kernel with the stored draws: centres and PSFs vs the reference."""
    lens = lens_foc2000
    N, spp, ks = 121, 256, 11
    u_main = torch.stack((tt(g23["u_theta"]), tt(g23["u_r"])))[None].contiguous().to(DEV)        # [L=1,2,spp]
    u_chief = torch.stack((tt(g23["c_theta"]), tt(g23["c_r"])))[None].contiguous().to(DEV)
    psf = torch.empty((N, 1, ks, ks), device=DEV)
    cen = torch.empty((1, N, 2), device=DEV)
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    pts = tt(g23["points"]).to(DEV)
    _abi.call("aadff_psf_points", _abi.ptr(pts), 1, N, 1, _abi.ptr(lens._table([0.589])),
              _abi.ptr(lens._table([0.589])), lens._lens_const(), _abi.ptr(lens._state_device()), _abi.ptr(u_main), spp, 2 * spp, 2 * spp,
              _abi.ptr(u_chief), GEO_SPP, 2 * GEO_SPP, 2 * GEO_SPP, ks, 1, 0, _abi.ptr(psf), _abi.ptr(cen), _abi.ptr(flags),
              _abi.stream_ptr(torch.device(DEV)))
    assert int(flags.item()) == 0
    assert np.abs(cen[0].cpu().numpy() - g23["centre"]).max() <= 2e-5      # mm; pixel = 0.03 mm
    # 256 rays per PSF: each ray carries 8x the weight it has at spp 2048, where the fp32 floor
    # is 5e-4; 4e-3 here
    assert rel_l2(psf[:, 0].cpu().numpy(), g23["psf"]) <= 2 * PSF_TOL


def test_backward_trace_entrance_pupil_rays(g23, lens_foc2000):
    lens = lens_foc2000
    M = 32
    aper = lens.surfaces[lens.aper_idx]
    phi = torch.arange(-0.5, 0.5, 1.0 / M)
    o = torch.tensor([[aper.r, 0, aper.d.item()]]).repeat(M, 1).to(torch.float32)
    d = torch.stack((torch.sin(phi), torch.zeros_like(phi), -torch.cos(phi)), axis=-1)
    ray, valid, _ = lens.trace(Ray(o, d, device=DEV), lens_range=range(0, lens.aper_idx))
    assert np.array_equal(ray.ra.cpu().numpy(), g23["back_ra"])
    alive = g23["back_ra"] > 0
    assert np.abs(ray.o.cpu().numpy() - g23["back_o"])[alive].max() <= 2e-5
    assert np.abs(ray.d.cpu().numpy() - g23["back_d"])[alive].max() <= 2e-6


# ================================================================= G4: psf_map end to end (seeded host RNG)
@pytest.mark.parametrize("name,res,foc,depth,spp", [("rf50mm", (1024, 1024), -2000.0, -1500.0, 2048),
                                                   ("50mm_f2.8", (256, 256), -1000.0, -1250.0, 512)])
def test_psf_map_seeded(golden_dir, repo_root, name, res, foc, depth, spp):
    g = np.load(os.path.join(golden_dir, "g4_psf_map.npz"))
    key = name.replace(".", "_")
    lens = Lensgroup(lens_path(repo_root, name), sensor_res=res, device=DEV)
    torch.manual_seed(0)
    lens.refocus(foc)
    pm = lens.psf_map(depth=depth, grid=11, ks=11, spp=spp)
    assert pm.shape == (3, 121, 121) and pm.is_cuda
    assert lens.d_sensor == pytest.approx(float(g[f"{key}_d_sensor"]), rel=1e-5)
    assert rel_l2(pm.cpu().numpy(), g[f"{key}_psf_map"]) <= PSF_TOL
    # and the map renders the same image as the reference's map (the 1e-4 budget applies to the IMAGE)
    H, W = (256, 256)
    img = tt(synth_rgb(H, W))[None]
    want = oconv.render_psf_map(img, tt(g[f"{key}_psf_map"]), 11).numpy()
    got = rp.render_psf_map(img.to(DEV), pm, 11).cpu().numpy()
    assert rel_l2(got, want) <= IMG_TOL


def test_psf_single_point_and_nocenter(golden_dir, repo_root):
    g = np.load(os.path.join(golden_dir, "g4_psf_map.npz"))
    lens = Lensgroup(lens_path(repo_root), sensor_res=(480, 640), device=DEV)
    torch.manual_seed(3)
    p = lens.psf([0.3, -0.4, -1200.0], ks=11, spp=1024)
    assert p.shape == (11, 11)
    assert rel_l2(p.cpu().numpy(), g["single_point_psf"]) <= PSF_TOL
    torch.manual_seed(3)
    p = lens.psf(torch.tensor([[0.3, -0.4, -1200.0], [0.0, 0.0, -3000.0]]), ks=11, spp=1024, center=False)
    assert rel_l2(p.cpu().numpy(), g["nocenter_psf"]) <= PSF_TOL


@pytest.mark.parametrize("spp", [3000, 4096])
def test_psf_more_rays_than_one_compaction_chunk_vs_oracle(repo_root, spp):
    """spp above the 2048-ray compaction chunk of the fused kernel's main pass (the training path uses 4096) and a
    ragged last chunk, against the oracle on the same host-RNG stream."""
    pts = torch.tensor([[0.0, 0.0, -1500.0], [0.7, -0.5, -900.0], [-0.95, 0.95, -4000.0]])
    ora = OracleLens(lens_path(repo_root), sensor_res=(512, 512))
    torch.manual_seed(21)
    ora.refocus(-1500.0)
    want = ora.psf(pts, ks=11, spp=spp).numpy()
    lens = Lensgroup(lens_path(repo_root), sensor_res=(512, 512), device=DEV)
    torch.manual_seed(21)
    lens.refocus(-1500.0)
    got = lens.psf(pts, ks=11, spp=spp).cpu().numpy()
    assert got.shape == want.shape == (3, 11, 11)
    assert rel_l2(got, want) <= PSF_TOL
    assert got.sum((1, 2)) == pytest.approx(1.0, abs=1e-5)


def test_psf_map_signature_defaults_ks51(repo_root):
    """Lensgroup.psf_map() with the reference's defaults (grid 7, ks 51, optics.py:1006) and render_psf_map on it
    (generic-ks convolution kernel) against the oracle on the same RNG stream, at a reduced ray count."""
    ora = OracleLens(lens_path(repo_root), sensor_res=(256, 256))
    torch.manual_seed(5)
    want = ora.psf_map(depth=-1500.0, spp=256).numpy()
    lens = Lensgroup(lens_path(repo_root), sensor_res=(256, 256), device=DEV)
    torch.manual_seed(5)
    got = lens.psf_map(depth=-1500.0, spp=256)
    assert got.shape == (3, 7 * 51, 7 * 51)
    assert rel_l2(got.cpu().numpy(), want) <= 2 * PSF_TOL          # 256 rays over 51x51 bins: one ray = 0.4 % of a PSF
    img = tt(synth_rgb(256, 256))[None]
    ref = oconv.render_psf_map(img, tt(want), 7).numpy()
    out = rp.render_psf_map(img.to(DEV), got, 7).cpu().numpy()
    assert rel_l2(out, ref) <= IMG_TOL


@pytest.mark.parametrize("variant", ["conic", "hyperbolic", "degree5", "degree8"])
def test_conic_and_hyperbolic_aspheres_vs_oracle(repo_root, tmp_path, variant):
    """Branches rf50mm does not reach: an asphere with conic constant k != 0 (shape-domain test k > -1, Newton start
    from the conic root) and one with k <= -1 (no shape-domain limit); refocus scalars, PSFs and the in-place /
    compacted fused trace against the oracle on the same RNG stream."""
    d = json.load(open(lens_path(repo_root)))
    if variant == "conic":
        d["surfaces"][2].update({"type": "Aspheric", "k": -0.6, "ai": [0.0, 1.5e-6, -2e-9, 0.0, 0.0, 0.0]})
        d["surfaces"][8]["k"] = 0.35
    elif variant == "hyperbolic":
        d["surfaces"][8]["k"] = -1.5
        d["surfaces"][9]["k"] = -1.0
    elif variant == "degree5":                                   # five even-asphere coefficients (reference: surfaces.py:314)
        d["surfaces"][8]["ai"] = d["surfaces"][8]["ai"][:5]
    else:                                                        # eight: the kernels' AADFF_MAX_AI Horner branch
        d["surfaces"][9]["ai"] = list(d["surfaces"][9]["ai"]) + [3e-16, -1e-18]
    path = str(tmp_path / "lens.json")
    json.dump(d, open(path, "w"))
    pts = torch.tensor([[0.0, 0.0, -2000.0], [0.6, 0.6, -1200.0], [-0.9, 0.3, -5000.0], [0.2, -0.98, -800.0]])
    ora = OracleLens(path, sensor_res=(512, 512))
    torch.manual_seed(13)
    ora.refocus(-1800.0)
    want = ora.psf(pts, ks=11, spp=2048).numpy()
    lens = Lensgroup(path, sensor_res=(512, 512), device=DEV)
    torch.manual_seed(13)
    lens.refocus(-1800.0)
    assert lens.d_sensor == pytest.approx(ora.d_sensor, rel=1e-5) and lens.hfov == pytest.approx(ora.hfov, rel=1e-5)
    got = lens.psf(pts, ks=11, spp=2048).cpu().numpy()
    assert rel_l2(got, want) <= PSF_TOL


def test_no_valid_chief_ray_raises_like_the_reference(repo_root):
    """A point far outside the field loses every chief ray: the reference asserts "No sampled rays is valid."
    (optics.py:901); the kernels flag it and the Python mirror raises the same AssertionError - round 5: DEFERRED, the per-call API
    no longer waits for the GPU inside psf / psf_map (`flags.item()` per call made the reference's own loop host-bound): the error
    is raised (a) by `check_flags()`, (b) at the next host read-back of the lens state, (c) a few calls later through the pinned
    mirror of the flags word, without any synchronisation; `sync_flags = True` keeps the round-4 behaviour (raise inside the call)."""
    bad, good = torch.tensor([[40.0, 40.0, -300.0]]), torch.tensor([[0.1, 0.1, -1500.0]])
    lens = Lensgroup(lens_path(repo_root), sensor_res=(256, 256), device=DEV)
    torch.manual_seed(0)
    lens.psf(bad, ks=11, spp=256)                                           # returns: asynchronous
    with pytest.raises(AssertionError, match="No sampled rays is valid"):
        lens.check_flags()                                                  # (a)
    lens.check_flags()                                                      # consumed
    torch.manual_seed(0)
    assert lens.psf(good, ks=11, spp=256).shape == (1, 11, 11)              # the lens is still usable
    lens.psf(bad, ks=11, spp=256)
    lens.refocus(-1500.0)
    with pytest.raises(AssertionError, match="No sampled rays is valid"):
        lens.d_sensor                                                       # (b) the read-back that follows a refocus
    assert lens.d_sensor > 0                                                # the state itself is fine
    lens.psf(bad, ks=11, spp=256)
    with pytest.raises(AssertionError, match="No sampled rays is valid"):   # (c) never read back, never checked: the mirror catches up
        for _ in range(40):
            lens.psf(good, ks=11, spp=256)
            torch.cuda.synchronize()                                        # (only to make the test deterministic: the publish launch has run)
    lens.check_flags()
    lens.sync_flags = True
    torch.manual_seed(0)
    with pytest.raises(AssertionError, match="No sampled rays is valid"):
        lens.psf(bad, ks=11, spp=256)
    torch.manual_seed(0)
    assert lens.psf(good, ks=11, spp=256).shape == (1, 11, 11)


def test_psf_rgb_layout_matches_psf_map(repo_root):
    lens = Lensgroup(lens_path(repo_root), sensor_res=(256, 256), device=DEV)
    torch.manual_seed(1)
    pm = lens.psf_map(depth=-1500.0, grid=3, ks=11, spp=256)
    torch.manual_seed(1)
    pts = lens.point_source_grid(depth=-1500.0, grid=3).reshape(-1, 3)
    rgb = lens.psf_rgb(pts, ks=11, spp=256)
    assert rgb.shape == (9, 3, 11, 11)
    tiled = rgb.reshape(3, 3, 3, 11, 11).permute(2, 0, 3, 1, 4).reshape(3, 33, 33)
    assert (tiled - pm).abs().max().item() <= 1e-6          # LDS float atomics: sum order varies run to run
    assert pm.sum().item() == pytest.approx(27.0, rel=1e-5)      # every PSF sums to 1


# ================================================================= G8: M1 focal stack (config 0 scale)
def test_focal_stack_m1_golden(golden_dir, repo_root):
    g = np.load(os.path.join(golden_dir, "g8_stack_m1.npz"))
    H = W = 256
    lens = Lensgroup(lens_path(repo_root), sensor_res=(H, W), device=DEV)
    img = tt(synth_rgb(H, W))[None].to(DEV)
    torch.manual_seed(0)
    stack, maps = render_focal_stack_m1(lens, img, float(g["dbar"]), g["fds"], grid=11, ks=11, spp=GEO_SPP, return_maps=True)
    assert stack.shape == (1, 3, 5, H, W)
    s = stack[0].cpu().numpy()
    assert rel_l2(maps.cpu().numpy(), g["psf_maps"]) <= PSF_TOL
    assert rel_l2(s[:, :, 96:160, 96:160], g["crop"]) <= IMG_TOL
    assert rel_l2(s[:, :, 64:192, 64:192], g["centre_f16"].astype(np.float32)) <= 5e-4     # fp16 fixture
    assert s.astype(np.float64).sum((2, 3)) == pytest.approx(g["sums"], rel=1e-4)
    # the lens is left focused at the last distance, as the reference loop leaves it
    assert lens.d_sensor == pytest.approx(float(g["d_sensor"][-1]), rel=1e-5)


def test_focal_stack_m1_equals_sequential_api(repo_root):
    """The 3-launch batched stack == the reference-shaped per-slice loop on the same RNG stream."""
    H = W = 128
    lens = Lensgroup(lens_path(repo_root), sensor_res=(H, W), device=DEV)
    img = tt(synth_rgb(H, W))[None].to(DEV)
    fds = [-600.0, -900.0, -1500.0, -4000.0]
    torch.manual_seed(7)
    batched = render_focal_stack_m1(lens, img, -1200.0, fds, grid=5, ks=11, spp=512).clone()
    torch.manual_seed(7)
    sl = []
    for f in fds:
        lens.refocus(f)
        sl.append(rp.render_psf_map(img, lens.psf_map(depth=-1200.0, grid=5, ks=11, spp=512), 5))
    assert (batched - torch.stack(sl, dim=2)).abs().max().item() <= 5e-6    # histogram atomics + refocus partial sums: sum order


def test_focal_stack_m1_vs_oracle_full_pipeline(repo_root):
    """Whole M1 pipeline (refocus -> psf_map -> conv) vs the oracle at a small size, second lens."""
    H = W = 96
    img = tt(synth_rgb(H, W, seed=5))[None]
    fds = [-700.0, -1500.0, -3000.0, -6000.0]
    ora = OracleLens(lens_path(repo_root, "50mm_f2.8"), sensor_res=(H, W))
    torch.manual_seed(11)
    want, _ = opsf.focal_stack_m1(ora, img, -1800.0, fds, grid=5, ks=11, spp=1024)
    lens = Lensgroup(lens_path(repo_root, "50mm_f2.8"), sensor_res=(H, W), device=DEV)
    torch.manual_seed(11)
    got = render_focal_stack_m1(lens, img.to(DEV), -1800.0, fds, grid=5, ks=11, spp=1024)
    assert rel_l2(got.cpu().numpy(), want.numpy()) <= IMG_TOL


# ================================================================= G6/G7: PSFNet (M2) and thin lens
@pytest.fixture(scope="module")
def g67(golden_dir):
    return np.load(os.path.join(golden_dir, "g6_g7_psfnet.npz"))


@pytest.fixture(scope="module")
def psfnet64(repo_root):
    net = PSFNet(lens_path(repo_root), sensor_res=(64, 64), kernel_size=11, device=DEV)
    net.psfnet.load_state_dict({k: tt(v) for k, v in mlp_state_dict(seed=4321).items()})
    return net


def test_mlp_golden(g67, psfnet64):
    with torch.no_grad():
        y = psfnet64.psfnet(tt(g67["mlp_in"]).to(DEV))
    assert rel_l2(y.cpu().numpy(), g67["mlp_out"]) <= 1e-5


def test_fused_mlp_pred_golden_and_vs_torch(g67, psfnet64):
    """aadff_psfnet_forward mode 0 (the whole MLP in one HIP kernel, fp16 hi/lo MFMA) against the reference's
    MLP outputs (G6) and against torch fp32 on sizes that are not multiples of the 128-pixel tile."""
    assert psfnet64.mlp_precision == "fp32" and psfnet64._fused(DEV) is not None
    with torch.no_grad():
        y = psfnet64.pred(tt(g67["mlp_in"]).to(DEV))
    assert y.shape == (g67["mlp_in"].shape[0], 11, 11)
    assert rel_l2(y.reshape(y.shape[0], -1).cpu().numpy(), g67["mlp_out"]) <= 1e-5
    rng = np.random.Generator(np.random.PCG64(77))
    for n in (1, 127, 129, 1000):
        x = tt(rng.random((n, 4), dtype=np.float32)).to(DEV)
        x[:, :2] = x[:, :2] * 2 - 1
        with torch.no_grad():
            a = psfnet64.pred(x).reshape(n, -1)
            b = psfnet64.psfnet(x)
        assert (a - b).abs().max().item() <= 2e-7 and rel_l2(a.cpu().numpy(), b.cpu().numpy()) <= 2e-6, n
        assert a.sum(-1).cpu().numpy() == pytest.approx(1.0, abs=1e-5)
    with torch.enable_grad():                                   # training keeps the autograd path
        x = tt(rng.random((8, 4), dtype=np.float32)).to(DEV)
        assert psfnet64.pred(x).requires_grad


def test_fused_mlp_fp16_single_pass_mode(g67, psfnet64):
    """mlp_precision="fp16" (opt-in): the fused kernel with one fp16 MFMA per product.  PSFs within 2e-3 rel-L2 of torch
    fp32 (measured ~3e-4), rendered images within 5e-4; the default mode is untouched."""
    rng = np.random.Generator(np.random.PCG64(31))
    x = tt(rng.random((1000, 4), dtype=np.float32)).to(DEV)
    x[:, :2] = x[:, :2] * 2 - 1
    img = tt(synth_rgb(64, 64, seed=11))[None].to(DEV)
    depth = -tt(synth_depth_mm(64, 64, seed=12))[None, None].to(DEV)
    fds = torch.tensor([[-500.0, -1500.0, -5000.0]], device=DEV)
    net = PSFNet.__new__(PSFNet)
    net.__dict__.update(psfnet64.__dict__)
    net._packed = None
    with torch.no_grad():
        ref_psf = net.psfnet(x)
        ref_stack = net.render_stack(img, depth, fds)
        net.mlp_precision = "fp16"
        got_psf = net.pred(x).reshape(1000, -1)
        got_stack = net.render_stack(img, depth, fds)
        got_one = net.render(img, depth, fds[:, 1])
    e_psf, e_img = rel_l2(got_psf.cpu().numpy(), ref_psf.cpu().numpy()), rel_l2(got_stack.cpu().numpy(), ref_stack.cpu().numpy())
    assert 1e-6 < e_psf <= 2e-3 and e_img <= 5e-4, (e_psf, e_img)            # really the reduced-precision path, and close
    assert (got_one - got_stack[:, :, 1]).abs().max().item() <= 1e-6
    assert got_psf.sum(-1).cpu().numpy() == pytest.approx(1.0, abs=1e-5)
    assert psfnet64.mlp_precision == "fp32"


def test_fused_mlp_small_weights(repo_root):
    """Weights of magnitude ~1e-3 (and down to 1e-7): the lo halves of the fp16 split are all subnormal there (spacing 6e-8
    absolute, see aadff/psfnet_pack.py).  The fused kernel must still match torch fp32 to 2e-7: in particular the MFMA f16
    path must not flush subnormal operands (a flush would cost ~1e-5)."""
    net = PSFNet(lens_path(repo_root), sensor_res=(64, 64), kernel_size=11, device=DEV)
    rng = np.random.Generator(np.random.PCG64(5))
    sd = {}
    for k, v in mlp_state_dict(seed=99).items():
        if k.endswith("weight"):
            scale = 10.0 ** rng.uniform(-4.0, -1.0, size=v.shape).astype(np.float32)       # |w| spread over 1e-5 .. 1e-1 of the init
            v = (v * scale).astype(np.float32)
        sd[k] = tt(v)
    net.psfnet.load_state_dict(sd)
    x = tt(rng.random((777, 4), dtype=np.float32)).to(DEV)
    x[:, :2] = x[:, :2] * 2 - 1
    with torch.no_grad():
        a = net.pred(x).reshape(777, -1)
        b = net.psfnet(x)
    assert net._fused(DEV) is not None
    assert (a - b).abs().max().item() <= 2e-7 and rel_l2(a.cpu().numpy(), b.cpu().numpy()) <= 2e-6


def test_fused_mlp_activation_overflow_raises(repo_root):
    """A hidden activation above 65504 does not fit the fp16 hi half of the split operand: the fused kernel raises flag
    bit 4 and the host raises ActivationOverflow instead of returning inf/NaN PSFs; the torch mode still renders."""
    from aadff.psfnet_pack import ActivationOverflow
    net = PSFNet(lens_path(repo_root), sensor_res=(16, 16), kernel_size=11, device=DEV)
    sd = {k: tt(v) for k, v in mlp_state_dict(seed=4321).items()}
    sd["net.2.weight"] = sd["net.2.weight"] * 3e4                 # second layer: activations ~1e5
    net.psfnet.load_state_dict(sd)
    x = torch.rand(300, 4, device=DEV)
    with torch.no_grad():
        assert float(torch.relu(net.psfnet.net[:4](x)).max()) > 65504.0
        with pytest.raises(ActivationOverflow):
            net.pred(x)
        img, depth = torch.rand(1, 3, 16, 16, device=DEV), -torch.rand(1, 1, 16, 16, device=DEV) * 3000 - 300
        with pytest.raises(ActivationOverflow):
            net.render_stack(img, depth, torch.tensor([[-500.0, -900.0]], device=DEV))
        net.mlp_precision = "torch"
        out = net.render(img, depth, torch.tensor([-500.0], device=DEV))
        assert torch.isfinite(out).all()
    # the shipped rf50mm checkpoint stays far inside the range (numbers recorded from the reference's ckpt, no weights)
    rec = json.load(open(os.path.join(repo_root, "tests", "golden", "g11_ckpt_activation_range.json")))
    assert max(l["max_abs_preact"] for l in rec["layers"]) < 100.0 and max(l["max_abs_weight"] for l in rec["layers"]) < 2.0


def test_fused_render_equals_torch_mlp_plus_gather(psfnet64):
    """mode 1 (MLP + gather fused) == torch MLP + aadff_local_psf_render, incl. a ragged last tile and B = 2."""
    rng = np.random.Generator(np.random.PCG64(78))
    img = tt(rng.random((2, 3, 50, 37), dtype=np.float32)).to(DEV)
    depth = -tt(rng.random((2, 1, 50, 37), dtype=np.float32) * 4000 + 300).to(DEV)
    fd = torch.tensor([-800.0, -2500.0], device=DEV)
    net = PSFNet.__new__(PSFNet)
    net.__dict__.update(psfnet64.__dict__)
    net.sensor_res, net._packed = (50, 37), None
    fused = net.render(img, depth, fd)
    net.mlp_precision = "torch"
    ref = net.render(img, depth, fd)
    assert fused.shape == ref.shape == img.shape
    assert rel_l2(fused.cpu().numpy(), ref.cpu().numpy()) <= 2e-6


def test_render_stack_equals_per_slice_render(psfnet64):
    """PSFNet.render_stack (a whole [N,C,S,H,W] stack in one fused launch) == stacking render() slice by slice."""
    rng = np.random.Generator(np.random.PCG64(81))
    img = tt(rng.random((2, 3, 64, 64), dtype=np.float32)).to(DEV)
    depth = -tt(rng.random((2, 1, 64, 64), dtype=np.float32) * 4000 + 300).to(DEV)
    fds = -tt(np.sort(rng.random((2, 5), dtype=np.float32) * 4000 + 400, axis=1)).to(DEV)
    stack = psfnet64.render_stack(img, depth, fds)
    assert stack.shape == (2, 3, 5, 64, 64)
    for i in range(5):
        assert (stack[:, :, i] - psfnet64.render(img, depth, fds[:, i])).abs().max().item() <= 1e-6, i


def test_psfnet_render_golden(g67, psfnet64):
    img = tt(synth_rgb(64, 64, seed=11))[None].to(DEV)
    depth = -tt(synth_depth_mm(64, 64, seed=12))[None, None].to(DEV)
    for i, f in enumerate(g67["render_fds"]):
        out = psfnet64.render(img, depth, torch.tensor([float(f)], device=DEV))
        assert rel_l2(out.cpu().numpy(), g67["render_out"][i:i + 1]) <= IMG_TOL, f
    out3 = psfnet64.render(img[0], depth[0, 0], -1500.0)
    assert rel_l2(out3.cpu().numpy(), g67["render3d_out"]) <= IMG_TOL
    img2 = torch.cat((img, torch.flip(img, [-1])), 0)
    depth2 = torch.cat((depth, torch.flip(depth, [-2])), 0)
    outb = psfnet64.render(img2, depth2, torch.tensor([-700.0, -2500.0], device=DEV))
    assert rel_l2(outb.cpu().numpy(), g67["render_b2_out"]) <= IMG_TOL


def test_focal_stack_m2_vs_oracle(psfnet64):
    sd = {k: tt(v) for k, v in mlp_state_dict(seed=4321).items()}
    img = tt(synth_rgb(64, 64, seed=11))[None]
    depth_m = tt(synth_depth_mm(64, 64, seed=12))[None, None] / 1e3
    want = opsf.focal_stack_m2(sd, img, depth_m, 5).numpy()
    got, fds = render_focal_stack_m2(psfnet64, img.to(DEV), depth_m.to(DEV), 5)
    assert got.shape == (1, 3, 5, 64, 64)
    assert rel_l2(got.cpu().numpy(), want) <= IMG_TOL


def test_config5_stack_at_its_size_golden(golden_dir, repo_root):
    """BASELINE.json config 5's render AT ITS SIZE (the suite ran M2 at 64 x 64 only): configs/aber_aware_dff_dfv.yml:19-21
    = bs 2, n_stack 8, 480 x 640, ks 11, through the loop of 2_aber_aware_dff_dfv.py:101-107 - `select_focus_dist(depth, 8)` then
    `PSFNet.render(aif, -depth * 1e3, -foc_dist * 1e3)` per slice - here `render_focal_stack_m2` (ONE fused launch for the
    [2,3,8,480,640] stack).  Fixture G7b = the reference's own output (focus distances, three 64 x 64 crops, 16 x 16 block means
    and float64 sums of every (sample, slice)); the per-slice `PSFNet.render` loop must give the same pixels as the stack entry."""
    g = np.load(os.path.join(golden_dir, "g7b_config5_stack.npz"))
    H, W, B, S = 480, 640, 2, 8
    net = PSFNet(lens_path(repo_root), sensor_res=(H, W), kernel_size=11, device=DEV)
    net.psfnet.load_state_dict({k: tt(v) for k, v in mlp_state_dict(seed=4321).items()})
    aif = tt(np.stack([synth_rgb(H, W, seed=31 + b) for b in range(B)])).to(DEV)
    depth = (tt(np.stack([synth_depth_mm(H, W, seed=41 + b) for b in range(B)]))[:, None] / 1e3).to(DEV)
    got, fds = render_focal_stack_m2(net, aif, depth, S)
    assert got.shape == (B, 3, S, H, W) and np.array_equal(fds.cpu().numpy(), g["focus_dists"])
    out = got.cpu().numpy()
    for name, (ys, xs) in {"a": (slice(0, 64), slice(0, 64)), "b": (slice(208, 272), slice(288, 352)), "c": (slice(416, 480), slice(576, 640))}.items():
        assert rel_l2(out[:, :, :, ys, xs], g[f"crop_{name}"]) <= IMG_TOL, name
        assert np.abs(out[:, :, :, ys, xs] - g[f"crop_{name}"]).max() <= 2e-5, name
    blocks = out.astype(np.float64).reshape(B, 3, S, H // 16, 16, W // 16, 16).mean((4, 6))
    assert rel_l2(blocks, g["block_means"]) <= 2e-6
    assert out.astype(np.float64).sum((3, 4)) == pytest.approx(g["sums"], rel=2e-6)
    for i in (0, 5):                                                            # the reference's own call, slice by slice
        sl = net.render(aif, -depth * 1e3, -fds[:, i] * 1e3)
        assert (sl - got[:, :, i]).abs().max().item() <= 2e-6, i


def test_config0_warm_up_stack_256_vs_oracle(repo_root, tmp_path):
    """BASELINE.json configs[0] (0_warm_up.py scale: rf50mm, 256x256 RGB-D, 5-slice stack through PSFNet.render) against
    the oracle, and the example script that replays 0_warm_up.py's call sequence runs end to end."""
    import subprocess
    import sys
    sd = {k: tt(v) for k, v in mlp_state_dict().items()}
    net = PSFNet(lens_path(repo_root), sensor_res=(256, 256), kernel_size=11, device=DEV)
    net.psfnet.load_state_dict(sd)
    img = tt(synth_rgb(256, 256))[None]
    depth_m = tt(synth_depth_mm(256, 256))[None, None] / 1e3
    want = opsf.focal_stack_m2(sd, img, depth_m, 5).numpy()
    got, fds = render_focal_stack_m2(net, img.to(DEV), depth_m.to(DEV), 5)
    assert got.shape == (1, 3, 5, 256, 256) and rel_l2(got.cpu().numpy(), want) <= IMG_TOL
    env = dict(os.environ, PYTHONPATH=os.path.join(repo_root, "aberration-aware-depth-from-focus_amd"))
    p = subprocess.run([sys.executable, os.path.join(repo_root, "examples", "0_warm_up_synthetic.py"), str(tmp_path)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert (tmp_path / "aberrated_defocused_img.png").exists() and (tmp_path / "stack_m1_4.png").exists()


def test_thinlens_golden(g67):
    img = tt(synth_rgb(64, 64, seed=11))[None].to(DEV)
    depth = -tt(synth_depth_mm(64, 64, seed=12))[None, None].to(DEV)
    thin = ThinLens(foc_len=50.0, fnum=1.8, kernel_size=11, sensor_size=[24.0, 24.0], sensor_res=(64, 64))
    out = thin.render(img, depth, torch.tensor([-1500.0], device=DEV))
    assert rel_l2(out.cpu().numpy(), g67["thin_out"]) <= IMG_TOL


def test_thinlens_kernel_matches_tensor_form_and_edge_cases(g67):
    """aadff_thinlens_render (PSF evaluated in the gather kernel) vs the reference's literal tensor form on the same GPU
    (ThinLens.render_psf_tensor -> local_psf_render): batches with different focus distances, positive depths (no sign
    flip), ragged widths, one channel, and the coc clamp: a pixel exactly in focus has coc 0 -> clamped to 0.1 px -> only
    the centre tap survives the cut -> the pixel is copied."""
    thin = ThinLens(foc_len=50.0, fnum=1.8, kernel_size=11, sensor_size=[24.0, 24.0], sensor_res=(64, 64))
    assert np.abs(thin.coc(-tt(synth_depth_mm(64, 64, seed=12))[None, None], torch.full((1, 1, 64, 64), -1500.0)).numpy()
                  - g67["thin_coc"]).max() <= 1e-5 * np.abs(g67["thin_coc"]).max()
    rng = np.random.Generator(np.random.PCG64(77))
    for (N, C, H, W, ks, sign) in ((2, 3, 40, 72, 11, -1), (1, 3, 33, 130, 11, 1), (2, 1, 20, 64, 5, -1), (1, 4, 17, 50, 7, 1)):
        t = ThinLens(foc_len=50.0, fnum=2.8, kernel_size=ks, sensor_size=[24.0, 36.0], sensor_res=(H, W))
        img = tt(rng.random((N, C, H, W), dtype=np.float32)).to(DEV)
        depth = sign * tt(np.stack([synth_depth_mm(H, W, seed=50 + i) for i in range(N)]))[:, None].to(DEV)
        fd = sign * torch.tensor([900.0, 2500.0][:N], device=DEV)
        got, want = t.render(img, depth, fd), t.render_psf_tensor(img, depth, fd)
        assert got.shape == want.shape == (N, C, H, W)
        assert (got - want).abs().max().item() <= 3e-6, (N, C, H, W, ks, sign)
    t = ThinLens(foc_len=50.0, fnum=1.8, kernel_size=11, sensor_size=[24.0, 24.0], sensor_res=(32, 64))
    img = tt(rng.random((1, 3, 32, 64), dtype=np.float32)).to(DEV)
    depth = torch.full((1, 1, 32, 64), -1500.0, device=DEV)
    out = t.render(img, depth, torch.tensor([-1500.0], device=DEV))
    assert torch.equal(out, img)                                          # delta PSF


@pytest.mark.parametrize("overlap", [False, True])
def test_pipelined_training_batches_equal_unpipelined(repo_root, overlap):
    """overlap=True: the producer runs one batch ahead on its own stream (same host draws in the same order).
    aadff.training.TrainingDataPlan (one pinned block per batch uploaded inside the refocus launch, two launches, no
    sync) against the call-by-call form (refocus, then psf) on the same numpy/torch RNG streams, for 2*RING + 5 batches:
    every pinned block is reused twice, the guard wait and the flags mirror run."""
    from aadff.training import TrainingDataPlan
    net = PSFNet(lens_path(repo_root), sensor_res=(480, 640), kernel_size=11, device=DEV)
    n = 2 * TrainingDataPlan.RING + 5
    np.random.seed(11)
    torch.manual_seed(11)
    want = [net._get_training_data_unpipelined(bs=32, spp=512) for _ in range(n)]
    np.random.seed(11)
    torch.manual_seed(11)
    plan = TrainingDataPlan(net, 32, 512, overlap=overlap)
    assert (plan.side is not None) == overlap
    got = []
    for j in range(n):
        inp, psf = plan.next(prefetch=j + 1 < n)
        got.append((inp.clone(), psf.clone()))
    plan.check_flags()
    lens_ref = PSFNet(lens_path(repo_root), sensor_res=(480, 640), kernel_size=11, device=DEV)
    torch.manual_seed(5)
    lens_ref.refocus(float(got[-1][0][0, 3].item()) * (lens_ref.d_max - lens_ref.d_min) + lens_ref.d_min)
    assert net.d_sensor == pytest.approx(lens_ref.d_sensor, rel=1e-4)          # the lens is left focused where the last batch was traced
    assert len(plan.guards) <= TrainingDataPlan.RING // TrainingDataPlan.GUARD_EVERY + 2
    for i, ((ia, pa), (ib, pb)) in enumerate(zip(got, want)):
        assert torch.equal(ia.cpu(), ib.cpu()), i
        # the staged refocus sums its rays in four quarters: d_sensor moves by an ulp, PSFs by sum-order noise
        assert rel_l2(pa.cpu().numpy(), pb.cpu().numpy()) <= 5e-4, i
        assert pa.sum(-1).cpu().numpy() == pytest.approx(1.0, abs=1e-5)


def test_get_training_data_shapes_and_normalisation(repo_root):
    net = PSFNet(lens_path(repo_root), sensor_res=(480, 640), kernel_size=11, device=DEV)
    np.random.seed(0)
    torch.manual_seed(0)
    inp, psf = net.get_training_data(bs=128, spp=4096)
    assert inp.shape == (128, 4) and psf.shape == (128, 121)
    s = psf.sum(-1).cpu().numpy()
    ok = np.isfinite(s)
    assert ok.mean() > 0.9 and np.abs(s[ok] - 1).max() <= 1e-5


def test_psfnet_render_bf16_mode_is_opt_in_and_close(g67, psfnet64):
    """Opt-in bf16 MLP: not a parity mode (tolerance 5e-3 on the image), default stays fp32."""
    img = tt(synth_rgb(64, 64, seed=11))[None].to(DEV)
    depth = -tt(synth_depth_mm(64, 64, seed=12))[None, None].to(DEV)
    assert psfnet64.mlp_precision == "fp32"
    psfnet64.mlp_precision = "bf16"
    try:
        out = psfnet64.render(img, depth, torch.tensor([-1500.0], device=DEV))
    finally:
        psfnet64.mlp_precision = "fp32"
    assert rel_l2(out.cpu().numpy(), g67["render_out"][2:3]) <= 5e-3


def test_empty_inputs_return_empty():
    e = torch.empty(0, 3, 32, 32, device=DEV)
    assert rp.render_psf_map(e, torch.rand(3, 6, 6, device=DEV), 2).shape == (0, 3, 32, 32)
    assert rp.render_psf(e, torch.rand(3, 3, 3, device=DEV)).shape == (0, 3, 32, 32)
    assert rp.local_psf_render(e, torch.empty(0, 32, 32, 3, 3, device=DEV), 3).shape == (0, 3, 32, 32)
    img = torch.rand(1, 3, 32, 32, device=DEV)
    assert rp.render_psf_map_stack(img, torch.empty(0, 3, 6, 6, device=DEV), 2).shape == (1, 3, 0, 32, 32)


def test_full_resolution_middlebury_size_properties():
    """1988 x 2880 (the reference's Middlebury depth maps), grid 11, ks 11, 2 images x 4 slices: the delta PSF
    is the identity in every slice, and slices of the stack equal the single-slice render (ragged 180/181-px
    and 261/262-px patches, non-multiple-of-32 everything)."""
    H, W, g, ks, S = 1988, 2880, 11, 11, 4
    rng = np.random.Generator(np.random.PCG64(8))
    img = tt(rng.random((2, 3, H, W), dtype=np.float32)).to(DEV)
    delta = torch.zeros(3, g, g, ks, ks)
    delta[..., ks // 2, ks // 2] = 1
    delta = delta.permute(0, 1, 3, 2, 4).reshape(3, g * ks, g * ks).to(DEV)
    maps = tt(rng.random((S, 3, g * ks, g * ks), dtype=np.float32)).to(DEV) / 121
    maps[1] = delta
    stack = rp.render_psf_map_stack(img, maps, g)
    assert stack.shape == (2, 3, S, H, W)
    assert (stack[:, :, 1] - img).abs().max().item() <= 5e-7
    assert (stack[:, :, 3] - rp.render_psf_map(img, maps[3], g)).abs().max().item() <= 1e-6
    want = oconv.render_psf_map(img[1:, :, 900:1300, 1700:2100].cpu(), maps[2].cpu(), 1) if False else None
    # spot-check one interior patch against the oracle (patch (5,5): rows 903..1083, cols 1309..1570)
    hb = [int(i / g * H) for i in range(g + 1)]
    wb = [int(j / g * W) for j in range(g + 1)]
    y0, y1, x0, x1 = hb[5], hb[6], wb[5], wb[6]
    crop = img[0:1, :, y0 - 5:y1 + 5, x0 - 5:x1 + 5].cpu()
    k = maps[2][:, 5 * ks:6 * ks, 5 * ks:6 * ks].cpu()
    ref = torch.nn.functional.conv2d(crop, torch.flip(k, [1, 2]).unsqueeze(1), groups=3)
    assert (stack[0:1, :, 2, y0:y1, x0:x1].cpu() - ref).abs().max().item() <= 4e-6


# ================================================================= script-facing API surface
def test_train_psfnet_runs_and_checkpoints(repo_root, tmp_path):
    """1_fit_psfnet.py's calls: analysis, write_lens_json, train_psfnet (2 iterations), evaluate_psf, load_net."""
    net = PSFNet(lens_path(repo_root), sensor_res=(480, 640), kernel_size=11, device=DEV)
    pic = net.analysis(save_name=str(tmp_path / "lens"))                # PSF-map picture (7 x 7 fields, ks 51) off the PSF-grid kernel
    assert pic == str(tmp_path / "lens") + "_psf20000mm.png" and os.path.getsize(pic) > 10_000
    net.write_lens_json(str(tmp_path / "lens.json"))
    np.random.seed(0)
    torch.manual_seed(0)
    net.train_psfnet(iters=1, bs=16, lr=1e-4, spp=256, evaluate_every=1, result_dir=str(tmp_path))
    assert (tmp_path / "PSFNet_mlp.pkl").exists() and (tmp_path / "iter1.png").exists()
    net.evaluate_psf(result_dir=str(tmp_path))
    # calc_psf_map / evaluate_psf_score (psfnet.py:215-243, 305-366; both call Lensgroup.psf with a keyword it lacks in the
    # reference snapshot - built as they are meant): the ray-traced grid equals psf() on the same points and seed
    torch.manual_seed(3)
    pm = net.calc_psf_map(-1500.0, -1200.0, psf_grid=(3, 4))
    assert pm.shape == (3, 3 * 11, 4 * 11) and torch.equal(pm[0], pm[2])
    torch.manual_seed(3)
    net.refocus(depth=-1500.0)
    xs, ys = net._grid_points((3, 4))
    want = net.psf(points=torch.stack((xs, ys, torch.full_like(xs, -1200.0)), -1), ks=11, spp=net.spp, center=True)
    assert torch.allclose(pm[0, 11:22, 22:33].cpu(), want[1 * 4 + 2].cpu(), atol=1e-6)       # float atomics: sum-order noise
    assert float(xs[0]) == pytest.approx(-0.875) and float(ys[0]) == pytest.approx(5 / 6)      # tile centres: 4 columns, 3 rows
    net.foc_z_arr, net.psf_grid, net.spp = net.foc_z_arr[:2], [2, 3], 256
    np.random.seed(1)
    torch.manual_seed(1)
    inp, maps = net.get_training_psf_map(bs=3, psf_grid=(2, 2), psf_map_size=(16, 16))
    assert inp.shape == (3, 2) and maps.shape == (3, 3, 16, 16) and bool((inp[:, 1] == inp[0, 1]).all()) and float(maps.min()) >= 0
    net.vis_psf_map(want[:3], filename=str(tmp_path / "vis.png"))
    assert os.path.getsize(tmp_path / "vis.png") > 1000
    l1, l2 = net.evaluate_psf_score(vis=True, result_dir=str(tmp_path))
    assert 0 < l2 <= l1 < 1 and len([f for f in os.listdir(tmp_path) if f.endswith("_gt.png")]) == 2 * 40       # a 1-iteration net: errors are large, finite
    sd = torch.load(tmp_path / "PSFNet_mlp.pkl", map_location="cpu")
    assert sorted(sd)[:2] == ["net.0.bias", "net.0.weight"] and sd["net.20.weight"].shape == (121, 256)
    net2 = PSFNet(str(tmp_path / "lens.json"), sensor_res=(480, 640), kernel_size=11, device=DEV)
    net2.load_net(str(tmp_path / "PSFNet_mlp.pkl"))
    for k in ("d_sensor", "hfov", "foclen", "fnum"):
        assert getattr(net2, k) == pytest.approx(getattr(Lensgroup(lens_path(repo_root), sensor_res=(480, 640), device=DEV), k), rel=1e-6)


def test_graph_captured_training_step_matches_eager(repo_root):
    """train_psfnet's HIP-graph step (static buffers, capturable AdamW, closed-form cosine schedule) follows the
    eager torch loop with CosineAnnealingLR: same weights after 8 iterations on the same data."""
    from deeplens.psfnet import _TrainStep
    rng = np.random.Generator(np.random.PCG64(9))
    data = [(tt(rng.random((32, 4), dtype=np.float32)).to(DEV), tt(rng.random((32, 121), dtype=np.float32)).to(DEV) / 121)
            for _ in range(8)]
    sd0 = {k: tt(v) for k, v in mlp_state_dict(seed=77).items()}
    nets = []
    for mode in ("graph", "eager", "reference"):
        net = PSFNet(lens_path(repo_root), sensor_res=(64, 64), kernel_size=11, device=DEV)
        net.psfnet.load_state_dict(sd0)
        if mode == "reference":
            opt = torch.optim.AdamW(net.psfnet.parameters(), 1e-3)
            sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=20, eta_min=0)
            for inp, psf in data:
                pred = net.psfnet(inp)
                opt.zero_grad()
                torch.nn.functional.mse_loss(pred.float(), psf).backward()
                opt.step()
                sch.step()
        else:
            step = _TrainStep(net.psfnet, 1e-3, 20, 32, 121, torch.device(DEV), False, mode == "graph")
            for inp, psf in data:
                step(inp, psf)
            assert (step.graph is not None) == (mode == "graph")
        torch.cuda.synchronize()
        nets.append({k: v.detach().cpu().numpy() for k, v in net.psfnet.state_dict().items()})
    for k in nets[0]:
        assert rel_l2(nets[0][k], nets[2][k]) <= 1e-4 or np.abs(nets[0][k] - nets[2][k]).max() <= 1e-6, k
        assert rel_l2(nets[1][k], nets[2][k]) <= 1e-4 or np.abs(nets[1][k] - nets[2][k]).max() <= 1e-6, k


@pytest.mark.parametrize("chain", ["1", "0"])
@pytest.mark.parametrize("bs,ks,hidden,layers", [(128, 11, 256, 8), (50, 9, 128, 3), (256, 7, 64, 1), (8, 11, 256, 2)])
def test_fit_kernels_gradients_match_torch(bs, ks, hidden, layers, chain, monkeypatch):
    """The hand-written fit step (csrc/mlp_train.hip via aadff/mlp_fit.py): prediction and every dW / db of one
    forward + backward against torch autograd on the same network (deeplens/psfnet.py:94-106).  bf16 operands: tolerance
    1e-2 against torch's bf16 autocast, 2e-2 against fp32 autograd.  Ragged batches / widths exercise the zero padding."""
    import copy
    from aadff.mlp_fit import FusedFit, supported
    from deeplens.psfnet_arch import MLP
    torch.manual_seed(bs + ks)
    net = MLP(4, ks * ks, hidden, layers).to(DEV)
    ref = copy.deepcopy(net)
    rng = np.random.Generator(np.random.PCG64(bs))
    inp = tt(rng.random((bs, 4), dtype=np.float32) * 2 - 1).to(DEV)
    psf = tt(rng.random((bs, ks * ks), dtype=np.float32)).to(DEV)
    psf /= psf.sum(-1, keepdim=True)
    assert supported(net, bs)
    monkeypatch.setenv("AADFF_FIT_CHAIN", chain)                 # "1": 3-launch chain form, "0": layer-by-layer GEMM launches
    fit = FusedFit(net, 1e-3, 100, bs, torch.device(DEV))
    assert fit.chain == (chain == "1")
    grad, pred = fit.gradients(inp, psf)
    torch.cuda.synchronize()
    lin = [m for m in ref.net if isinstance(m, torch.nn.Linear)]
    want = {}
    for autocast in (True, False):
        ref.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            out = ref(inp)
        torch.nn.functional.mse_loss(out.float(), psf).backward()
        want[autocast] = (out.detach().float().cpu().numpy(), [(m.weight.grad.cpu().numpy().copy(), m.bias.grad.cpu().numpy().copy()) for m in lin])
    assert rel_l2(pred.cpu().numpy(), want[True][0]) <= 1e-6 and rel_l2(pred.cpu().numpy(), want[False][0]) <= 5e-3
    for l, m in enumerate(lin):
        gw = grad[fit.w_off[l]:fit.w_off[l] + m.weight.numel()].view_as(m.weight).cpu().numpy()
        gb = grad[fit.b_off[l]:fit.b_off[l] + m.bias.numel()].cpu().numpy()
        for got, k in ((gw, 0), (gb, 1)):
            assert rel_l2(got, want[True][1][l][k]) <= 1e-2, (l, k)
            # against fp32 autograd: no further from it than torch's own bf16 autocast is (the error grows with depth)
            assert rel_l2(got, want[False][1][l][k]) <= max(1.5 * rel_l2(want[True][1][l][k], want[False][1][l][k]), 5e-3), (l, k)


@pytest.mark.parametrize("chain", ["1", "0"])
def test_fit_kernels_follow_torch_adamw_and_cosine_schedule(chain, monkeypatch):
    """20 fused steps (one HIP graph after two plain runs) against torch.optim.AdamW + CosineAnnealingLR in fp32 on the same
    batches: parameters agree to bf16-gradient noise, the module's own tensors hold the result (state_dict works)."""
    import copy
    from deeplens.psfnet import _TrainStep
    from deeplens.psfnet_arch import MLP
    net = MLP(4, 121, 256, 8).to(DEV)
    net.load_state_dict({k: tt(v) for k, v in mlp_state_dict(seed=4321).items()})
    ref = copy.deepcopy(net)
    rng = np.random.Generator(np.random.PCG64(3))
    data = []
    for _ in range(4):
        inp = tt(rng.random((128, 4), dtype=np.float32) * 2 - 1).to(DEV)
        psf = tt(rng.random((128, 121), dtype=np.float32)).to(DEV) ** 4
        data.append((inp, psf / psf.sum(-1, keepdim=True)))
    monkeypatch.setenv("AADFF_FIT_CHAIN", chain)
    step = _TrainStep(net, 1e-3, 20, 128, 121, torch.device(DEV), True, True)
    assert step.fused is not None and step.fused.chain == (chain == "1")
    opt = torch.optim.AdamW(ref.parameters(), 1e-3)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=20, eta_min=0)
    first = last = None
    for it in range(20):
        inp, psf = data[it % 4]
        pred = step(inp, psf)
        loss = float(((pred - psf) ** 2).mean())
        first, last = (loss if first is None else first), loss
        opt.zero_grad()
        torch.nn.functional.mse_loss(ref(inp), psf).backward()
        opt.step()
        sch.step()
    torch.cuda.synchronize()
    assert (step.fused.graph is not None or len(step.fused.bound) == 4) and int(step.fused.step_dev.item()) == 20 and last < first
    assert (len(step.fused.bound) == 4) == (chain == "1")           # chain form: the four recurring batches are read in place
    for (k, a), b in zip(net.state_dict().items(), ref.state_dict().values()):
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) <= 2e-2, k
    with torch.no_grad():                                       # the bf16 operand copies follow the fp32 master parameters
        h = step.fused.flat.to(torch.bfloat16)
        assert torch.equal(step.fused.p16[step.fused.dst.long()], h)


def test_fit_kernels_can_be_switched_off(monkeypatch):
    from deeplens.psfnet import _TrainStep
    from deeplens.psfnet_arch import MLP
    monkeypatch.setenv("AADFF_FIT_KERNELS", "torch")
    step = _TrainStep(MLP(4, 121, 64, 2).to(DEV), 1e-3, 20, 16, 121, torch.device(DEV), True, True)
    assert step.fused is None
    monkeypatch.delenv("AADFF_FIT_KERNELS")
    wide = torch.nn.Module()
    wide.net = torch.nn.Sequential(torch.nn.Linear(4, 512), torch.nn.ReLU(), torch.nn.Linear(512, 121), torch.nn.Sigmoid()).to(DEV)
    wide.forward = lambda x: torch.nn.functional.normalize(wide.net(x), p=1, dim=-1)
    assert _TrainStep(wide, 1e-3, 20, 16, 121, torch.device(DEV), True, True).fused is None      # width 512: torch-GEMM path


def test_train_psfnet_bf16_runs_on_the_fit_kernels_end_to_end(repo_root, tmp_path):
    """Config 4 as the script runs it (1_fit_psfnet.py with bf16): train_psfnet(autocast_bf16=True) takes the hand-written fit
    step and the one-batch-ahead producer, writes its checkpoints / preview tiles, leaves the module's own parameters updated
    (state_dict round trip) and the fused inference kernel sees the new weights."""
    net = PSFNet(lens_path(repo_root), sensor_res=(64, 64), kernel_size=11, device=DEV)
    net.psfnet.load_state_dict({k: tt(v) for k, v in mlp_state_dict(seed=4321).items()})
    before = {k: v.detach().clone() for k, v in net.psfnet.state_dict().items()}
    torch.manual_seed(0)
    np.random.seed(0)
    net.train_psfnet(iters=7, bs=16, lr=1e-3, spp=256, evaluate_every=4, result_dir=str(tmp_path), autocast_bf16=True)
    torch.cuda.synchronize()
    assert (tmp_path / "iter4.png").exists() and (tmp_path / "iter4_PSFNet_mlp.pkl").exists() and (tmp_path / "PSFNet_mlp.pkl").exists()
    sd = torch.load(str(tmp_path / "PSFNet_mlp.pkl"), map_location="cpu")
    moved = [k for k in before if not torch.equal(sd[k], before[k].cpu())]
    assert len(moved) == len(before)                              # every weight and bias took a step
    assert all(torch.isfinite(v).all() for v in sd.values())
    x = tt(np.random.Generator(np.random.PCG64(5)).random((50, 4), dtype=np.float32)).to(DEV)
    with torch.no_grad():
        assert (net.pred(x) - net.psfnet(x).reshape(50, 11, 11)).abs().max().item() <= 2e-7
    assert net._training_plan(16, 256).pending is None          # the last iteration did not draw a batch it would not use


def test_fused_inference_sees_weights_after_graph_training(repo_root, tmp_path):
    """The packed weights of the fused kernel are rebuilt after train_psfnet (graph replays do not bump tensor versions)."""
    net = PSFNet(lens_path(repo_root), sensor_res=(64, 64), kernel_size=11, device=DEV)
    net.psfnet.load_state_dict({k: tt(v) for k, v in mlp_state_dict(seed=4321).items()})
    x = tt(np.random.Generator(np.random.PCG64(5)).random((50, 4), dtype=np.float32)).to(DEV)
    with torch.no_grad():
        before = net.pred(x).clone()
    torch.manual_seed(0)
    np.random.seed(0)
    net.train_psfnet(iters=6, bs=16, lr=1e-3, spp=256, evaluate_every=1000, result_dir=str(tmp_path))
    with torch.no_grad():
        fused = net.pred(x)
        ref = net.psfnet(x).reshape(50, 11, 11)
    assert (fused - ref).abs().max().item() <= 2e-7
    assert (fused - before).abs().max().item() > 1e-6          # the weights did move


def test_render_single_img_psf_branch(repo_root):
    """Lensgroup.render_single_img(method='psf'): 7x7 grid, ks 21 - the only in-repo caller of render_psf_map
    in the reference (deeplens/optics.py:779-783)."""
    lens = Lensgroup(lens_path(repo_root), sensor_res=(128, 128), device=DEV)
    rng = np.random.Generator(np.random.PCG64(0))
    img = (rng.random((96, 128, 3)) * 255).astype(np.uint8)
    torch.manual_seed(0)
    out = lens.render_single_img(img, depth=-2000.0, method="psf")
    assert out.shape == (96, 128, 3) and out.dtype == np.uint8
    assert list(lens.sensor_res) == [128, 128]                       # restored
    assert abs(float(out.mean()) - float(img.mean())) < 2.0          # a normalised PSF preserves the mean level
    with pytest.raises(NotImplementedError):
        lens.render_single_img(img, method="raytracing")


def test_lens_state_properties_round_trip(repo_root):
    """d_sensor / hfov are device-resident; assigning d_sensor (as scripts may) is honoured by the kernels."""
    lens = Lensgroup(lens_path(repo_root), sensor_res=(256, 256), device=DEV)
    torch.manual_seed(0)
    lens.refocus(-1000.0)
    d1 = lens.d_sensor
    lens.d_sensor = d1 + 0.25
    assert lens.d_sensor == pytest.approx(d1 + 0.25, abs=1e-5)
    lens.post_computation()
    assert lens.hfov < 0.41 and lens.foclen == pytest.approx(lens.r_last / np.tan(lens.hfov), rel=1e-6)
    torch.manual_seed(1)
    a = lens.psf(torch.tensor([[0.0, 0.0, -1000.0]]), ks=11, spp=512)
    lens.d_sensor = d1
    torch.manual_seed(1)
    b = lens.psf(torch.tensor([[0.0, 0.0, -1000.0]]), ks=11, spp=512)
    assert (a - b).abs().max().item() > 1e-3                           # the sensor shift changed the PSF


def test_focal_stack_m1_layered_vs_oracle(repo_root):
    """RGB-D aware grid rendering (depth quantised into layers) against the same composition of oracle primitives."""
    from aadff.focal_stack import depth_layers, render_focal_stack_m1_layered
    H = W = 64
    img = tt(synth_rgb(H, W, seed=21))[None]
    depth = -tt(synth_depth_mm(H, W, seed=22, dmin=600.0, dmax=4000.0))[None, None]
    depth[0, 0, :3, :3] = 0.0                                      # invalid pixels
    fds = [-700.0, -1500.0, -3500.0]
    ora = OracleLens(lens_path(repo_root), sensor_res=(H, W))
    torch.manual_seed(4)
    want = opsf.focal_stack_m1_layered(ora, img, depth, fds, layers=3, grid=3, ks=11, spp=512)
    lens = Lensgroup(lens_path(repo_root), sensor_res=(H, W), device=DEV)
    torch.manual_seed(4)
    got = render_focal_stack_m1_layered(lens, img.to(DEV), depth.to(DEV), fds, layers=3, grid=3, ks=11, spp=512)
    assert got.shape == (1, 3, 3, H, W)
    assert rel_l2(got.cpu().numpy(), want.numpy()) <= IMG_TOL
    idx_g, cen_g = depth_layers(depth.to(DEV), 3)
    idx_o, cen_o = opsf.depth_layers(depth, 3)
    assert torch.equal(idx_g.cpu(), idx_o) and torch.allclose(cen_g.cpu(), cen_o)


@pytest.mark.parametrize("hw,S,L,grid", [((1024, 1024), 10, 4, 11), ((200, 328), 5, 3, 4), ((96, 64), 2, 7, 2)])
def test_focal_stack_m1_layered_fused_equals_composition(repo_root, hw, S, L, grid):
    """Round 5: the fused M1-layered stack (one PSF launch for all (slice, layer) pairs, `aadff_render_psf_map_stack_layered`: the L
    candidates of a pixel from one staged band, only the pixel's own layer written) against the composition it replaces (S x L
    candidate slices through the stack convolution + torch.gather) - at the bench size (1024^2 x 10 x 4 layers), a ragged image
    whose patches are narrower than a tile, and L = 7 (pairs that straddle the 4-map chunks).  The convolution on the SAME PSF maps
    is BIT-equal pixel for pixel; the whole function (its PSF histograms are float atomics, run-to-run 1e-7) to 2e-6."""
    from aadff.focal_stack import depth_layers, render_focal_stack_m1_layered
    H, W = hw
    img = tt(synth_rgb(H, W, seed=31))[None].to(DEV)
    depth = -tt(synth_depth_mm(H, W, seed=32, dmin=600.0, dmax=4000.0))[None, None].to(DEV)
    depth[0, 0, :5, :7] = 0.0                                      # invalid pixels -> farthest layer
    fds = -np.linspace(650.0, 3800.0, S)
    lens = Lensgroup(lens_path(repo_root), sensor_res=(H, W), device=DEV)
    torch.manual_seed(9)
    a = render_focal_stack_m1_layered(lens, img, depth, fds, layers=L, grid=grid, ks=11, spp=256)
    state_a = (lens.d_sensor, lens.hfov)
    torch.manual_seed(9)
    b = render_focal_stack_m1_layered(lens, img, depth, fds, layers=L, grid=grid, ks=11, spp=256, fused=False)
    assert a.shape == b.shape == (1, 3, S, H, W)
    assert float((a - b).abs().max()) <= 2e-6 and not torch.isnan(a).any()
    assert state_a == (lens.d_sensor, lens.hfov)
    # the convolution alone, same maps: bit-equal; two images, so that the (b, c) plane indexing of the layer map is exercised
    g = torch.Generator().manual_seed(5)
    maps = torch.rand(S * L, 3, grid * 11, grid * 11, generator=g).to(DEV)
    maps = maps / maps.reshape(S * L, 3, grid, 11, grid, 11).sum((3, 5), keepdim=True).reshape(S * L, 3, grid, 1, grid, 1).expand(-1, -1, -1, 11, -1, 11).reshape_as(maps)
    x = torch.stack((img[0], img[0].flip(-1))).contiguous()
    idx, _ = depth_layers(torch.stack((depth[0], depth[0].flip(-2))), L)
    lidx = idx.reshape(2, H, W).to(torch.uint8).contiguous()
    fused = torch.full((2, 3, S, H, W), float("nan"), device=DEV)
    _abi.call("aadff_render_psf_map_stack_layered", _abi.ptr(x), _abi.ptr(maps), _abi.ptr(lidx), _abi.ptr(fused), 2, 3, S, L, H, W, grid, 11, _abi.stream_ptr(torch.device(DEV)))
    tmp = torch.empty((2, 3, S * L, H, W), device=DEV)
    _abi.call("aadff_render_psf_map_stack", _abi.ptr(x), _abi.ptr(maps), _abi.ptr(tmp), 2, 3, S * L, H, W, grid, 11, _abi.stream_ptr(torch.device(DEV)))
    want = torch.gather(tmp.view(2, 3, S, L, H, W), 3, idx.reshape(2, 1, 1, 1, H, W).expand(2, 3, S, 1, H, W)).squeeze(3)
    assert torch.equal(fused, want), (int((fused != want).sum()), int(torch.isnan(fused).sum()))


def _staged_vs_copy(repo_root, monkeypatch, steps, lib=None):
    from aadff import focal_stack as fs
    H = W = 64
    lens = Lensgroup(lens_path(repo_root), sensor_res=(H, W), device=DEV)
    img = tt(synth_rgb(H, W))[None].to(DEV)
    fds = [-600.0, -800.0, -1100.0, -1500.0, -2500.0, -5000.0]
    outs, bits = {}, {}
    for staged in (True, False):
        monkeypatch.setattr(fs, "STAGED_UPLOAD", staged)
        if lib is not None:
            monkeypatch.setattr(_abi, "_lib", lib if staged else _abi.load_library())
        plan = fs.StackPlan(lens, len(fds), H, W, 1, 3, 5, 11, 512)
        res = []
        for step in range(steps):
            torch.manual_seed(100 + step)
            o, m = render_focal_stack_m1(lens, img, -1200.0, fds, grid=5, ks=11, spp=512, plan=plan, return_maps=True)
            res.append((o.clone(), m.clone()))
            assert len(plan.guards) <= fs.StackPlan.RING // fs.StackPlan.GUARD_EVERY + 2      # guard events are pruned
        torch.cuda.synchronize()
        bits[staged] = int(plan.flags.item())
        assert (plan.stage_generation > 0) == staged
        outs[staged] = res
    for (a, ma), (b, mb) in zip(outs[True], outs[False]):
        # histogram atomics: sum order; the staged refocus sums its rays in 4 quarters, d_sensor moves by an fp32
        # ulp and now and then one of the 512 rays of a point flips across a validity edge (one ray = 1/400 of a PSF)
        assert rel_l2(ma.cpu().numpy(), mb.cpu().numpy()) <= 5e-4
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) <= 4e-5
    assert (outs[True][0][0] - outs[True][1][0]).abs().max().item() > 1e-4   # different seeds do differ
    return bits


def test_staged_upload_equals_plain_copy_over_repeated_steps(repo_root, monkeypatch):
    """aadff_refocus_staged + aadff_psf_points_staged (upload folded into the launches, per-state counters
    reused across steps) == the hipMemcpyAsync path on the same RNG stream, for 2*RING + GUARD_EVERY steps of one plan:
    every pinned block is reused twice, so the guard-event wait, its pruning and the flags mirror all run."""
    from aadff import focal_stack as fs
    bits = _staged_vs_copy(repo_root, monkeypatch, 2 * fs.StackPlan.RING + fs.StackPlan.GUARD_EVERY)
    assert bits[True] & 8 == 0                 # the copy workgroups were on time


def test_staged_upload_late_block_is_read_from_pinned_memory(repo_root, monkeypatch):
    """HIP guarantees no dispatch order, so a PSF workgroup may find its state's uniform block not yet copied.  The
    build with -DAADFF_STAGE_SPIN_MAX=0 (csrc/libaadff_latestage.so) gives up after ONE poll: those workgroups must
    then read their draws from the pinned block directly and produce the same stacks, with flag bit 3 as a warning."""
    path = os.path.join(os.path.dirname(_abi.LIB_PATH), "libaadff_latestage.so")
    if not os.path.exists(path):
        pytest.skip("csrc/libaadff_latestage.so not built (make -C csrc libaadff_latestage.so)")
    late = _abi.load_library(path)
    bits = _staged_vs_copy(repo_root, monkeypatch, 6, lib=late)
    assert bits[True] & ~8 == 0 and bits[False] == 0
    if bits[True] & 8:
        from deeplens.optics import raise_psf_flags
        with pytest.warns(RuntimeWarning, match="arrived late"):
            raise_psf_flags(bits[True])


def test_per_call_api_without_copies_equals_the_copy_path(repo_root, monkeypatch):
    """Round 5: `refocus` reads its draws and depth from the pinned call block over PCIe and `psf_map` / `psf` / `psf_rgb` launches
    stage theirs with their first workgroups (`aadff_psf_points_staged`, S = 1, first_slice 0; the points are read where they lie) -
    no hipMemcpyAsync in front of a launch.  Same draws, same kernels: the lens state is bit-equal to the copy path's
    (`AADFF_CALL_ZERO_COPY=0`), the PSFs equal to the histogram's float atomics, over a sequence that changes the launch shape (the
    per-block completion counters restart), reuses every ring block more than once and ends where the copy path leaves the generator."""
    from deeplens.optics import _CallRing
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("AADFF_CALL_ZERO_COPY", mode)
        lens = Lensgroup(lens_path(repo_root, "rf50mm"), sensor_res=(128, 128), device=DEV)
        torch.manual_seed(5)
        outs = []
        for i in range(2 * _CallRing.SLOTS + 3):
            lens.refocus(-1500.0 - 40.0 * i)
            outs.append(torch.tensor([lens.d_sensor, lens.hfov, lens.foclen, lens.fnum], dtype=torch.float64))
            if i % 3 == 0:
                outs.append(lens.psf_map(depth=-1200.0, grid=5, ks=11, spp=512).cpu().double().flatten())
            elif i % 3 == 1:
                outs.append(lens.psf(torch.tensor([[0.3, -0.2, -900.0], [0.0, 0.0, -2000.0]]), ks=11, spp=1024).cpu().double().flatten())
            else:
                outs.append(lens.psf_rgb(torch.tensor([0.1, 0.4, -1700.0]), ks=9, spp=256).cpu().double().flatten())
        lens.check_flags()
        assert (lens._ring.mapped is not None) == (mode == "1")
        res[mode] = (outs, torch.rand(1).item())
    monkeypatch.delenv("AADFF_CALL_ZERO_COPY")
    assert res["1"][1] == res["0"][1]
    for a, b in zip(res["1"][0], res["0"][0]):
        if a.numel() == 4:
            assert torch.equal(a, b)                                             # refocus: one workgroup, fixed summation order
        else:
            assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item()    # PSF histograms: float atomics


def test_paired_band_convolution_is_bit_equal_to_one_band_per_workgroup(monkeypatch):
    """AADFF_CONV_PAIR=N (round-3 experiment kept as an option: a workgroup renders two consecutive bands of every N-th
    (patch, plane), the second band prefetched by LDS-DMA into the memory the tap rows no longer need) against the default
    one-band-per-workgroup launch: bit-equal stacks on ragged shapes, every pixel written, repeatable.  (The race this
    comparison exposed - an inline-asm operand read whose destination shared a register with its address - is fixed in
    both forms: early-clobber outputs, prefetched registers owned until they have landed.)"""
    st = _abi.stream_ptr(torch.device(DEV))
    for (H, W, S, G, Cn) in ((1024, 1024, 10, 11, 3), (480, 640, 5, 7, 3), (333, 517, 3, 4, 2), (97, 131, 4, 1, 3)):
        rng = np.random.Generator(np.random.PCG64(H + W))
        img = tt(rng.random((1, Cn, H, W), dtype=np.float32)).to(DEV)
        maps = tt(rng.random((S, Cn, G * 11, G * 11), dtype=np.float32)).to(DEV) / 121
        outs = {}
        for mode in ("0", "1", "3", "1"):
            monkeypatch.setenv("AADFF_CONV_PAIR", mode)
            out = torch.full((1, Cn, S, H, W), -7.0, device=DEV)
            _abi.call("aadff_render_psf_map_stack", _abi.ptr(img), _abi.ptr(maps), _abi.ptr(out), 1, Cn, S, H, W, G, 11, st)
            torch.cuda.synchronize()
            assert int((out == -7.0).sum()) == 0
            if mode in outs:
                assert torch.equal(outs[mode], out), "paired launch is not repeatable"
            outs[mode] = out
        assert torch.equal(outs["0"], outs["1"]) and torch.equal(outs["0"], outs["3"]), (H, W, S, G, Cn)
    monkeypatch.delenv("AADFF_CONV_PAIR")


def test_strided_stack_convolution_writes_unit_major_layouts():
    """aadff_render_psf_map_stack_strided: plane (b, c, s) at out + (b*C + c)*stride_bc + s*stride_s.  With stride_bc = H*W,
    stride_s = k*C*H*W the slices land as every k-th [C,H,W] unit of a caller's buffer (a rank's place in the unit-order
    all-gather buffer) - bit-equal to the contiguous stack, nothing else touched; overlapping planes and
    strides below one plane are refused."""
    st = _abi.stream_ptr(torch.device(DEV))
    rng = np.random.Generator(np.random.PCG64(77))
    for (H, W, S, G, Cn, k) in ((256, 320, 10, 5, 3, 1), (97, 131, 4, 3, 3, 3), (128, 128, 2, 2, 1, 2)):
        img = tt(rng.random((1, Cn, H, W), dtype=np.float32)).to(DEV)
        maps = tt(rng.random((S, Cn, G * 11, G * 11), dtype=np.float32)).to(DEV) / 121
        ref = torch.empty((1, Cn, S, H, W), device=DEV)
        _abi.call("aadff_render_psf_map_stack", _abi.ptr(img), _abi.ptr(maps), _abi.ptr(ref), 1, Cn, S, H, W, G, 11, st)
        units = torch.full((S * k + 2, Cn, H, W), -3.0, device=DEV)
        _abi.call("aadff_render_psf_map_stack_strided", _abi.ptr(img), _abi.ptr(maps), C.c_void_p(units.data_ptr() + 4 * Cn * H * W),
                  H * W, k * Cn * H * W, 1, Cn, S, H, W, G, 11, st)
        torch.cuda.synchronize()
        for sl in range(S):
            assert torch.equal(units[1 + sl * k], ref[0, :, sl]), (H, W, sl)
        mask = torch.ones(units.shape[0], dtype=torch.bool)
        mask[1:1 + S * k:k] = False
        assert bool((units[mask.to(DEV)] == -3.0).all())
    with pytest.raises(RuntimeError, match="overlap"):
        _abi.call("aadff_render_psf_map_stack_strided", _abi.ptr(img), _abi.ptr(maps), _abi.ptr(units), H * W, H * W, 1, 3, S, H, W, G, 11, st)
    with pytest.raises(RuntimeError, match="below one plane"):
        _abi.call("aadff_render_psf_map_stack_strided", _abi.ptr(img), _abi.ptr(maps), _abi.ptr(units), H * W - 1, S * H * W, 1, 1, S, H, W, G, 11, st)


def test_time_next_launch_attaches_events_to_the_kernel():
    """aadff_time_next_launch (bench.py's roofline / trace blocks): the armed call produces identical results, the two HIP
    events attached to the dispatch give a positive kernel time no longer than the bracket of two stream events, the arming
    is consumed by exactly one launch; the single-slice convolution (no slice-batched kernel) records the events around the call."""
    hip = C.CDLL("libamdhip64.so")

    def event():
        e = C.c_void_p()
        assert hip.hipEventCreate(C.byref(e)) == 0
        return e

    def elapsed(a, b):
        ms = C.c_float()
        assert hip.hipEventSynchronize(b) == 0 and hip.hipEventElapsedTime(C.byref(ms), a, b) == 0
        return ms.value

    H = W = 256
    rng = np.random.Generator(np.random.PCG64(21))
    img = tt(synth_rgb(H, W))[None].to(DEV)
    st = _abi.stream_ptr(torch.device(DEV))
    for S in (10, 1):
        maps = tt(rng.random((S, 3, 55, 55), dtype=np.float32)).to(DEV) / 121
        a, b = torch.empty((1, 3, S, H, W), device=DEV), torch.empty((1, 3, S, H, W), device=DEV)
        _abi.call("aadff_render_psf_map_stack", _abi.ptr(img), _abi.ptr(maps), _abi.ptr(a), 1, 3, S, H, W, 5, 11, st)
        e0, e1 = event(), event()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        _abi.call("aadff_time_next_launch", e0, e1)
        _abi.call("aadff_render_psf_map_stack", _abi.ptr(img), _abi.ptr(maps), _abi.ptr(b), 1, 3, S, H, W, 5, 11, st)
        t1.record()
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        kernel_ms, bracket_ms = elapsed(e0, e1), t0.elapsed_time(t1)
        assert 0 < kernel_ms <= bracket_ms + 2e-3, (S, kernel_ms, bracket_ms)
        # consumed: an unarmed call leaves the events alone
        _abi.call("aadff_render_psf_map_stack", _abi.ptr(img), _abi.ptr(maps), _abi.ptr(b), 1, 3, S, H, W, 5, 11, st)
        torch.cuda.synchronize()
        assert elapsed(e0, e1) == kernel_ms
        assert hip.hipEventDestroy(e0) == 0 and hip.hipEventDestroy(e1) == 0
    with pytest.raises(RuntimeError, match="both events or neither"):
        _abi.call("aadff_time_next_launch", event(), None)
    # the PSF-grid kernel: same PSFs with and without the events
    lens = Lensgroup(lens_path(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), sensor_res=(128, 128), device=DEV)
    pts = torch.tensor([[0.0, 0.0, -1500.0], [0.5, -0.3, -1500.0]])
    torch.manual_seed(3)
    want = lens.psf(pts, ks=11, spp=512)
    e0, e1 = event(), event()
    torch.manual_seed(3)
    _abi.call("aadff_time_next_launch", e0, e1)
    got = lens.psf(pts, ks=11, spp=512)
    torch.cuda.synchronize()
    assert rel_l2(got.cpu().numpy(), want.cpu().numpy()) <= 1e-5 and 0 < elapsed(e0, e1) < 50.0


def _render_steps(repo_root, make, steps, consume):
    from aadff import focal_stack as fs
    H = W = 64
    lens = Lensgroup(lens_path(repo_root), sensor_res=(H, W), device=DEV)
    img = tt(synth_rgb(H, W))[None].to(DEV)
    fds = [-600.0, -800.0, -1100.0, -1500.0, -2500.0, -5000.0]
    obj = make(fs, lens, len(fds), H, W)
    res = []
    for step in range(steps):
        torch.manual_seed(300 + step)
        res.append(consume(obj, lens, img, fds))
    torch.cuda.synchronize()
    return obj, res


def test_focus_traces_on_side_stream_give_the_same_stacks(repo_root):
    """StackPlan(overlap_refocus=True) launches the focus traces of a stack on the plan's side stream (they run beside the
    previous stack's kernels) with one focus-state block per ring slot: over 2*RING + 3 steps the stacks equal those of
    the in-line order, and the caller's stream sees finished outputs (clone on the current stream right after the call)."""
    from aadff.focal_stack import StackPlan
    run = lambda ov: _render_steps(
        repo_root, lambda fs, lens, S, H, W: fs.StackPlan(lens, S, H, W, 1, 3, 5, 11, 512, overlap_refocus=ov), 2 * StackPlan.RING + 3,
        lambda plan, lens, img, fds: render_focal_stack_m1(lens, img, -1200.0, fds, grid=5, ks=11, spp=512, plan=plan).clone())
    (pa, a), (pb, b) = run(True), run(False)
    assert pa.side is not None and pb.side is None and int(pa.flags.item()) & ~8 == 0
    for x, y in zip(a, b):
        assert rel_l2(x.cpu().numpy(), y.cpu().numpy()) <= 1e-5           # histogram atomics: sum order only
    assert (a[0] - a[1]).abs().max().item() > 1e-4


def test_stack_pipeline_two_streams_equals_one_plan(repo_root):
    """StackPipeline(depth=2): stacks alternate between two plans on two streams (two in flight).  The host still draws the
    samples in call order, so stack i equals stack i of a single plan; `done` orders the consumer behind the slot's stream."""
    def consume(pipe, lens, img, fds):
        out, done = pipe.render(lens, img, -1200.0, fds)
        if done is not None:
            torch.cuda.current_stream().wait_event(done)
        return out.clone()
    mk = lambda depth: (lambda fs, lens, S, H, W: fs.StackPipeline(lens, S, H, W, 1, 3, 5, 11, 512, depth=depth))
    (p2, a), (p1, b) = _render_steps(repo_root, mk(2), 19, consume), _render_steps(repo_root, mk(1), 19, consume)
    assert len(p2.plans) == 2 and p2.streams[0] is not None and p1.streams == [None]
    p2.check_flags()
    for x, y in zip(a, b):
        assert rel_l2(x.cpu().numpy(), y.cpu().numpy()) <= 1e-5
    assert (a[0] - a[1]).abs().max().item() > 1e-4
    p2.wait()                                                               # no-op ordering call must not raise


def test_staged_upload_rejects_unpinned_host_block(repo_root):
    lens = Lensgroup(lens_path(repo_root), sensor_res=(64, 64), device=DEV)
    u_host = torch.rand(4 * GEO_SPP)                           # pageable, not device-mapped
    u_dev = torch.empty(4 * GEO_SPP, device=DEV)
    dep = torch.tensor([-1000.0], device=DEV)
    states = torch.zeros(C.sizeof(_abi.LensState), dtype=torch.uint8, device=DEV)
    with pytest.raises(RuntimeError, match="not pinned"):
        _abi.call("aadff_refocus_staged", _abi.ptr(dep), 1, C.c_void_p(u_host.data_ptr()), _abi.ptr(u_dev), 2 * GEO_SPP,
                  GEO_SPP, 2 * GEO_SPP, _abi.ptr(lens._table([0.589])), lens._lens_const(), _abi.ptr(states),
                  _abi.ptr(torch.zeros(16, dtype=torch.int32, device=DEV)), _abi.stream_ptr(torch.device(DEV)))


def test_packed_division_and_sqrt_of_the_strict_kernels_are_ieee():
    """csrc/strict_math2.h `div2` (v_rcp_f32 + packed-FMA refinement + v_div_fixup_f32, no v_div_scale_f32) against the compiler's
    IEEE float32 division, bit for bit: operands over the ranges a trace produces (|x| in 2^-80 .. 2^80, any signs), exact zeros,
    infinities, NaNs, numerators down to the edge of what the missing pre-scaling covers (2^-100), quotients near 1 - the mismatch
    count must be ZERO there.  Outside (numerators below 2^-104, exponent gaps above 96) the pre-scaled form is needed: reported."""
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(7)
    n = 1 << 22

    def rnd(lo, hi, n):
        mant = 1 + torch.rand(n, generator=g)
        e = torch.randint(lo, hi + 1, (n,), generator=g).float()
        sign = torch.where(torch.rand(n, generator=g) < 0.5, -1.0, 1.0)
        return (sign * mant * torch.exp2(e)).float()

    cases = {
        "trace magnitudes": (rnd(-40, 14, n), rnd(-30, 14, n)),
        "wide": (rnd(-80, 80, n), rnd(-40, 40, n)),
        "near one": (1 + (torch.rand(n, generator=g) - 0.5) * 1e-3, 1 + (torch.rand(n, generator=g) - 0.5) * 1e-3),
        "small numerators": (rnd(-100, -60, n), rnd(-4, 4, n)),
    }
    special = torch.tensor([0.0, -0.0, float("inf"), -float("inf"), float("nan"), 1.0, -1.0, 3.0, 1e-30, 1e30, 5e-5, 2.0 ** -100])
    cases["specials"] = (special.repeat_interleave(len(special)), special.repeat(len(special)))
    for name, (a, b) in cases.items():
        ad, bd = a.float().to(dev).contiguous(), b.float().to(dev).contiguous()
        mm = torch.zeros(17, dtype=torch.int32, device=dev)
        _abi.call("aadff_selftest_strict_ops", _abi.ptr(ad), _abi.ptr(bd), ad.numel(), 0, _abi.ptr(mm), _abi.stream_ptr(dev))
        m = mm.cpu().numpy().view(np.uint32)
        first = [(int(m[1 + 2 * k]), float(a[int(m[1 + 2 * k])]), float(b[int(m[1 + 2 * k])])) for k in range(min(int(m[0]), 8))]
        assert m[0] == 0, (name, int(m[0]), first)
    # the shared reciprocal of sag / d sag (op 2): EVERY float 1 + sf in [1, 2] as the denominator whose square is divided by, against
    # the compiler's n / (d * d), with numerators of the magnitudes (opsf + a / (2 sf)) * c takes, of any magnitude, and the hard ones
    # next to multiples of d * d
    d_all = torch.arange(0x3f800000, 0x40000001, dtype=torch.int32).view(torch.float32).to(dev)
    for name, num in (("curvature-sized", rnd(-12, 3, d_all.numel())), ("wide", rnd(-60, 60, d_all.numel())),
                      ("near k d^2", (d_all.cpu() * d_all.cpu()) * torch.randint(1, 9, (d_all.numel(),), generator=g).float() * (1 + (torch.randint(-2, 3, (d_all.numel(),), generator=g).float() * 2.0 ** -23)))):
        mm = torch.zeros(17, dtype=torch.int32, device=dev)
        _abi.call("aadff_selftest_strict_ops", _abi.ptr(num.float().to(dev).contiguous()), _abi.ptr(d_all), d_all.numel(), 2, _abi.ptr(mm), _abi.stream_ptr(dev))
        m = mm.cpu().numpy().view(np.uint32)
        assert m[0] == 0, (name, int(m[0]), [(float(num[int(m[1 + 2 * k])]), float(d_all[int(m[1 + 2 * k])])) for k in range(min(int(m[0]), 8))])
    a, b = rnd(-126, -100, n), rnd(-4, 100, n)                                  # outside the domain: informative
    mm = torch.zeros(17, dtype=torch.int32, device=dev)
    _abi.call("aadff_selftest_strict_ops", _abi.ptr(a.to(dev)), _abi.ptr(b.to(dev)), n, 0, _abi.ptr(mm), _abi.stream_ptr(dev))
    print(f"packed division outside its domain (numerators 2^-126..2^-100): {int(mm[0])} of {n} quotients differ from IEEE")
    # square root: the arguments of the trace (1 - a in [2^-24, 1], squared norms, 0, negatives -> NaN, inf, NaN), every float in [1, 4)
    roots = {"unit interval": torch.rand(n, generator=g), "norms": rnd(-20, 40, n).abs(), "near zero of 1 - a": (torch.arange(1, 4097).float() * 2.0 ** -24),
             "specials": torch.tensor([0.0, -0.0, float("inf"), float("nan"), -1.0, -1e-30, 1.0, 2.0, 4.0, 2.0 ** -96, 3.0]),
             "all of [1, 4)": torch.arange(0x3f800000, 0x40800000, dtype=torch.int32).view(torch.float32)}
    for name, a in roots.items():
        ad = a.float().to(dev).contiguous()
        _abi.call("aadff_selftest_strict_ops", _abi.ptr(ad), None, ad.numel(), 1, _abi.ptr(mm), _abi.stream_ptr(dev))
        m = mm.cpu().numpy().view(np.uint32)
        assert m[0] == 0, (name, int(m[0]), [float(a[int(m[1 + 2 * k])]) for k in range(min(int(m[0]), 8))])
