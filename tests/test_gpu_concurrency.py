"""Strict kernels beside a busy second stream.  Round 6 found the strict (packed float32) trace kernels giving other results for a few
rays - always the second ray of a lane - whenever an MFMA kernel ran on another stream, our own convolution included; the cause is one
instruction form (v_pk_mul_f32 / v_pk_add_f32 with op_sel on the second source, produced by the SLP vectoriser), removed by building
the strict translation units with -fno-slp-vectorize (csrc/Makefile, tools/check_isa.py).  Before the fix this test failed in 198 of
200 launches (profiles/r06_concurrency_probe_grid.txt)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from aadff import _abi, strict_stack as ss                 # noqa: E402
from deeplens.optics import Lensgroup                     # noqa: E402
from test_gpu_margins import _psf_level_inputs, lp        # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_strict_trace_is_bit_stable_beside_the_convolution(repo_root):
    dev = torch.device(DEV)
    lens = Lensgroup(lp(repo_root), sensor_res=(512, 512), device=DEV, parity="strict")
    S, grid, spp = 4, 5, 1024
    a = _psf_level_inputs(lens, S, grid, spp, seed=11)
    N, B, n_surf = a["N"], a["B"], len(lens.surfaces)
    n = spp * N
    o0, d0, r0 = (torch.empty(B, n, 3, device=dev), torch.empty(B, n, 3, device=dev), torch.empty(B, n, device=dev))
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = ss._trace(o0, d0, r0, n, B, a["tabs"], len(a["wv"]), n_surf, a["bt_main"], True, flag, dev, a["points"], a["pset"], a["pm"], N, a["zs"])
    pred = torch.from_numpy(np.ascontiguousarray(ss._masks_to_counts(scratch, B), dtype=np.int32)).to(dev)

    def fused():
        o1, d1, r1 = torch.empty_like(o0), torch.empty_like(d0), torch.empty_like(r0)
        bits = torch.empty((B, 2, _abi.MAX_SURF), dtype=torch.int32, device=dev)
        _abi.call("aadff_trace_rays_strict_fused", _abi.ptr(o1), _abi.ptr(d1), _abi.ptr(r1), n, B, _abi.ptr(a["tab_dev"]), len(a["wv"]), n_surf,
                  _abi.ptr(a["bt_main"]), _abi.ptr(a["points"]), _abi.ptr(a["pset"]), _abi.ptr(a["pm"]), N, 0, n_surf, 1, _abi.ptr(a["zs"]),
                  _abi.ptr(pred), _abi.ptr(bits), 0, 0, None, None, None, _abi.stream_ptr(dev))
        return torch.cat([o1.view(torch.int32).reshape(-1), d1.view(torch.int32).reshape(-1), r1.view(torch.int32).reshape(-1), bits.reshape(-1)])

    def batched():
        o1, d1, r1 = torch.empty_like(o0), torch.empty_like(d0), torch.empty_like(r0)
        ss._trace(o1, d1, r1, n, B, a["tabs"], len(a["wv"]), n_surf, a["bt_main"], True, flag, dev, a["points"], a["pset"], a["pm"], N, a["zs"])
        return torch.cat([o1.view(torch.int32).reshape(-1), d1.view(torch.int32).reshape(-1), r1.view(torch.int32).reshape(-1)])

    img = torch.rand(1, 3, 1024, 1024, device=dev)
    psf = torch.rand(3, 121, 121, device=dev)
    psf /= psf.sum()
    out = torch.empty(1, 3, 1024, 1024, device=dev)
    side = torch.cuda.Stream()
    for name, fn, launches in (("fused", fused, 60), ("per-surface", batched, 20)):
        ref = fn()
        torch.cuda.synchronize()
        differing = 0
        for _ in range(launches):
            with torch.cuda.stream(side):                 # the lone-slice MFMA convolution, back to back, while the trace runs
                for _ in range(30):
                    _abi.call("aadff_render_psf_map", _abi.ptr(img), _abi.ptr(psf), _abi.ptr(out), 1, 3, 1024, 1024, 11, 11, C.c_void_p(side.cuda_stream))
            got = fn()
            torch.cuda.synchronize()
            differing += int(not torch.equal(got, ref))
        assert differing == 0, f"{name} strict trace: {differing} of {launches} launches beside the convolution differ from a quiet launch"


@pytest.mark.parametrize("s_alt", [9, 3])       # the second aspheric surface (the real case) and a spherical one
def test_two_variant_jobs_equal_the_single_variant_launches(repo_root, margin, s_alt):
    """`aadff_strict_psf_points_alt` (round 6): a batch whose chief count at one aspheric surface flips between n and n + 1 is rendered
    under both counts in ONE launch.  Checked against two ordinary launches of `aadff_strict_psf_points` (row with n + 1, row with n):
    centres bit-identical, PSF maps equal to the histogram's float atomics, the chief any-bits of both variants identical to what the
    ordinary launches report - for a real level (counts of a seed run) with EVERY batch made a two-variant job at one surface."""
    dev = torch.device(DEV)
    lens = Lensgroup(lp(repo_root), sensor_res=(512, 512), device=DEV, parity="strict")
    S, grid, spp, ks = 2, 5, 512, 11
    from deeplens.basics import GEO_SPP
    a = _psf_level_inputs(lens, S, grid, spp, seed=5)
    N, B, n_surf, MS = a["N"], a["B"], len(lens.surfaces), _abi.MAX_SURF
    curved = ss._curved(lens)
    _, cnt, _ = ss._level3_batched(lens, None, a["points"], a["pset"], a["pc"], a["pm"], a["zs"], a["bt_chief"], a["bt_main"], a["tabs"], len(a["wv"]), n_surf, N, spp, ks, dev)
    cnt = np.ascontiguousarray(cnt, dtype=np.int32)                              # [B, 2, MS] true counts
    assert curved[s_alt] and (cnt[:, 0, s_alt] >= 2).all() and (cnt[:, 0, s_alt] < 10).all()

    def run(rows, alt=None):
        maps = torch.empty((B, grid * ks, grid * ks), device=dev)
        centre = torch.empty((B, N, 2), device=dev)
        bits = torch.empty((B, 2, 2, MS), dtype=torch.int32, device=dev)
        av = torch.empty(B, dtype=torch.int32, device=dev)
        pred = torch.from_numpy(rows).to(dev)
        args = [_abi.ptr(a["points"]), N, B, None, _abi.ptr(a["pset"]), _abi.ptr(a["tab_dev"]), len(a["wv"]), n_surf, _abi.ptr(a["bt_main"]), _abi.ptr(a["bt_chief"]),
                _abi.ptr(a["zs"]), _abi.ptr(a["pm"]), spp, _abi.ptr(a["pc"]), GEO_SPP, _abi.ptr(pred), float(lens.pixel_size), ks, grid, _abi.ptr(maps),
                _abi.ptr(centre), _abi.ptr(bits), _abi.ptr(av)]
        if alt is None:
            _abi.call("aadff_strict_psf_points", *args, _abi.stream_ptr(dev))
            return maps, centre, bits.cpu().numpy(), av.cpu().numpy()
        m2, c2 = torch.empty_like(maps), torch.empty_like(centre)
        b2 = torch.empty((B, 2, MS), dtype=torch.int32, device=dev)
        av2 = torch.empty(B, dtype=torch.int32, device=dev)
        _abi.call("aadff_strict_psf_points_alt", *args, _abi.ptr(torch.from_numpy(alt).to(dev)), _abi.ptr(m2), _abi.ptr(c2), _abi.ptr(b2), _abi.ptr(av2), _abi.stream_ptr(dev))
        return maps, centre, bits.cpu().numpy(), av.cpu().numpy(), m2, c2, b2.cpu().numpy(), av2.cpu().numpy()

    # true count n* at the surface: the pair (n* - 1, n*) - the usual case, truth is the higher - and (n*, n* + 1) - truth is the lower
    for shift in (0, 1):
        n_lo = cnt[:, 0, s_alt] - 1 + shift
        hi_rows, lo_rows = cnt.copy(), cnt.copy()
        hi_rows[:, 0, s_alt] = n_lo + 1
        lo_rows[:, 0, s_alt] = n_lo
        alt = (s_alt | (n_lo << 8)).astype(np.int32)
        m_hi, c_hi, b_hi, v_hi = run(hi_rows)
        m_lo, c_lo, b_lo, v_lo = run(lo_rows)
        m, c, b, v, m2, c2, b2, v2 = run(hi_rows, alt)
        print(f"two-variant jobs at surface {s_alt}, shift {shift}: counts at the surface {cnt[:, 0, s_alt].tolist()}, longest list of noted rays per job {b2[:, 0, MS - 1].tolist()} of {GEO_SPP}")
        assert (v2 >= 0).all(), "a two-variant job ran out of list space"
        assert torch.equal(c.view(torch.int32), c_hi.view(torch.int32)) and torch.equal(c2.view(torch.int32), c_lo.view(torch.int32))
        assert np.array_equal(b, b_hi) and np.array_equal(v, v_hi) and np.array_equal(v2 > 0, v_lo > 0)
        assert np.array_equal(b2[:, 0, :n_surf], b_lo[:, 0, 0, :n_surf]) and np.array_equal(b2[:, 1], b_lo[:, 0, 1])      # chief any / nan bits of the lower-count variant
        margin(f"two-variant psf_map jobs at surface {s_alt} (pair n*{'' if shift else ' - 1'}..): PSF maps of the n + 1 variant vs an ordinary launch, max |d| / max",
               float((m - m_hi).abs().max() / m_hi.max()), 2e-6)
        margin(f"two-variant psf_map jobs at surface {s_alt} (pair n*{'' if shift else ' - 1'}..): PSF maps of the n variant vs an ordinary launch, max |d| / max",
               float((m2 - m_lo).abs().max() / m_lo.max()), 2e-6)
        differ = int((c_hi.view(torch.int32) != c_lo.view(torch.int32)).any(-1).sum())
        print(f"two-variant jobs, shift {shift}: centres that differ between the two counts: {differ} of {B * N}")


def test_strict_stacks_without_relaunches(repo_root, margin):
    """With the two-variant jobs a strict stack needs no level-3 re-launch once the count table knows both values (VERDICT r5 #2:
    at most 2 per 20 stacks), and its maps equal those of the re-launching form (AADFF_STRICT_ALT=0) on the same draws."""
    H = W = 512
    S, grid, spp = 10, 11, 2048
    from aadff.synth import synth_depth_mm
    depth = synth_depth_mm(1024, 1024, seed=5678)
    dbar, fds = -float(depth.mean()), [float(f) for f in -np.linspace(depth.min(), depth.max(), S)]
    out = {}
    for alt_on in (True, False):
        ss.ALT_JOBS = alt_on
        try:
            lens = Lensgroup(lp(repo_root), sensor_res=(H, W), device=DEV, parity="strict")
            torch.manual_seed(21)
            for _ in range(40):                                                      # seeds the table and lets it see both counts of the batches that flip
                ss.strict_psf_maps(lens, dbar, fds, grid, 11, spp)
            stats0 = dict(ss.StrictCounts.of(lens).stats)
            torch.manual_seed(22)
            maps = [ss.strict_psf_maps(lens, dbar, fds, grid, 11, spp).clone() for _ in range(20)]
            torch.cuda.synchronize()
            st = ss.StrictCounts.of(lens).stats
            out[alt_on] = (maps, {k: st[k] - stats0.get(k, 0) for k in st})
        finally:
            ss.ALT_JOBS = True
    print("strict stacks, two-variant jobs on :", out[True][1])
    print("strict stacks, two-variant jobs off:", out[False][1])
    worst = max(float((x - y).abs().max() / y.max()) for x, y in zip(out[True][0], out[False][0]))
    margin("strict stacks: two-variant jobs vs re-launches, PSF maps max |d| / max over 20 stacks", worst, 2e-6)
    # (replayed_batches counts psf_map batches only; a batch that shows its second count for the first time still costs one re-launch)
    assert out[True][1]["replayed_batches"] <= 2, out[True][1]
    assert out[True][1].get("alt_taken", 0) > 0 and out[False][1]["replayed_batches"] >= 20, (out[True][1], out[False][1])
