"""parity="strict" for a whole focal stack in THREE batched traces instead of 72 S single ones.

The reference's loop (2_aber_aware_dff_aif.py:104-114 over deeplens/optics.py:1155-1217 refocus / calc_fov and :888-1026 psf_map)
makes, per slice, one 2048-ray focus trace, one 100-ray field-of-view trace and 3 x 2 traces of spp x N rays; each is its own
Newton batch (`while (|ft| > 5e-5).any()`, deeplens/surfaces.py:547).  Slices do not depend on each other, and within a slice
only  focus -> d_sensor -> field of view -> hfov -> object points  is a chain.  So the stack is three LEVELS, each ONE call of
`aadff_trace_rays_strict_batched` (one launch per surface for all batches of the level, every batch with its own iteration
counts, wavelength table and sensor plane; csrc/strict.hip):

  level 1   S focus batches (2048 rays)                    host: the reference's focus-distance arithmetic, np.mean per slice
  level 2   S field-of-view batches (100 rays, backward)   host: tan / sum / atan per slice
  level 3   3 S main batches + 3 S chief batches           rays BUILT on the device (o2 - o, F.normalize), chief-ray centres by
            (spp x N rays each)                            `aadff_strict_centroid` in ATen's summation order, histogram kernel

What stays on the host is what the reference computes there with torch / numpy and what cannot be reproduced off its
libraries: the pupil sampling (`rand * 2 * pi`, MKL's vector sqrt / cos / sin - evaluated for all slices in one call each:
element-wise, position independent, checked in tests), the focus and field-of-view reductions.  The host generator is consumed
in the reference's order (per slice focus theta, focus r, then per wavelength main theta, main r, chief
theta, chief r).  Result = the slice-by-slice strict loop's (`strict_psf_maps_loop`) to the float atomics of the histogram
(tests/test_gpu_margins.py), 17-20x faster (9.5 ms per 10-slice 1024^2 stack)."""
import ctypes as C
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import _abi
from deeplens.basics import DEFAULT_WAVE, GEO_SPP, WAVE_RGB


def strict_psf_maps_loop(lens, depth_plane_mm, focus, grid, ks, spp):
    """The reference's loop, call by call (refocus(f_k) then psf_map, every trace a single `aadff_trace_rays_strict` call)."""
    return torch.stack([(lens.refocus(f), lens.psf_map(depth=depth_plane_mm, grid=grid, ks=ks, spp=spp))[1] for f in focus])


def _pupil_points(theta_u, r_u, radius, z):
    """[..., n] uniforms -> [..., n, 3] pupil / aperture points: theta = u * 2 * pi, r = sqrt(u * R^2), (r cos, r sin, z)
    (deeplens/optics.py:480-486, deeplens/surfaces.py:188-199), the reference's own torch calls on the host."""
    theta = theta_u * 2 * np.pi
    r = torch.sqrt(r_u * radius ** 2)
    return torch.stack((r * torch.cos(theta), r * torch.sin(theta), torch.full_like(r, z)), -1)


def _tables(lens, wvlns):
    """HOST copy of the packed surface tables of `wvlns`, cached with the lens's device tables (`Lensgroup.invalidate()` drops both:
    packing 36 surfaces costs 1.8 ms, an eighth of a strict stack)."""
    key = ("strict-host", tuple(float(w) for w in wvlns))
    arr = lens._table_cache.get(key)
    if arr is None:
        n = len(lens.surfaces)
        arr = (_abi.Surface * (n * len(wvlns)))()
        for li, w in enumerate(wvlns):
            for i, s in enumerate(lens.surfaces):
                arr[li * n + i] = s.pack(w)
        lens._table_cache[key] = arr
    return arr


def _trace(o, d, ra, n, B, tabs, n_tables, n_surf, batch_table, forward, flags, dev, points=None, point_set=None, pupil=None, N=1, z_sensor=None,
           tbuf=None):
    scratch = torch.empty(2 * B * _abi.MAX_SURF + 1, dtype=torch.int32, device=dev)
    if tbuf is None:
        tbuf = torch.empty(2 * B * n, dtype=torch.float32, device=dev)
    _abi.call("aadff_trace_rays_strict_batched", _abi.ptr(o), _abi.ptr(d), _abi.ptr(ra), n, B, C.byref(tabs), n_tables, n_surf,
              _abi.ptr(batch_table), _abi.ptr(points), _abi.ptr(point_set), _abi.ptr(pupil), N, 0, n_surf, int(forward), _abi.ptr(z_sensor),
              _abi.ptr(scratch), _abi.ptr(tbuf), _abi.ptr(flags), _abi.stream_ptr(dev))


@torch.no_grad()
def strict_psf_maps(lens, depth_plane_mm, focus, grid, ks, spp):
    """PSF maps [S,3,g*ks,g*ks] (device) of a strict-parity lens for the focus distances `focus`, all field points on the plane
    `depth_plane_mm`; leaves the lens focused at the last distance, like the reference's loop."""
    from .focal_stack import stack_uniform_layout
    if ks > _abi.MAX_KS:
        raise ValueError(f"ks={ks} exceeds the kernels' limit {_abi.MAX_KS}")
    S, L, N = len(focus), len(WAVE_RGB), grid * grid
    dev = lens._gpu()
    n_surf = len(lens.surfaces)
    wv = list(WAVE_RGB) + ([] if DEFAULT_WAVE in WAVE_RGB else [DEFAULT_WAVE])
    t_green = wv.index(DEFAULT_WAVE)
    tabs = _tables(lens, wv)
    f32 = torch.float32

    # ---- the stack's draws, in the reference's order (one flat draw = the same generator stream as call by call)
    per, o_main, o_chief, per_l = stack_uniform_layout(spp, L)
    u = lens.sampler.rand_block([S * per]).cpu().reshape(S, per)
    uf = u[:, :2 * GEO_SPP].reshape(S, 2, GEO_SPP)
    rest = u[:, 2 * GEO_SPP:].reshape(S, L, per_l)
    um = rest[:, :, :2 * spp].reshape(S, L, 2, spp)
    uc = rest[:, :, 2 * spp:].reshape(S, L, 2, GEO_SPP)

    flags = torch.zeros(3, dtype=torch.int32, device=dev)
    marks = [("start", time.perf_counter())] if os.environ.get("AADFF_STRICT_TIMING") == "1" else None
    mark = (lambda name: marks.append((name, time.perf_counter()))) if marks is not None else (lambda name: None)
    with torch.cuda.device(dev):
        # ---- level 1: refocus (deeplens/optics.py:1155-1180) - S batches of 2048 rays from the first surface's aperture
        s0 = lens.surfaces[0]
        o = _pupil_points(uf[:, 0], uf[:, 1], s0.r, s0.d.item())                                   # [S,2048,3]
        tgt = torch.zeros(S, 1, 3, dtype=f32)
        tgt[:, 0, 2] = torch.tensor([float(f) for f in focus], dtype=f32)
        d = F.normalize((o - tgt).float(), p=2, dim=-1)                                            # Ray.__init__
        od, dd = o.to(dev).contiguous(), d.to(dev).contiguous()
        rad = torch.ones(S, GEO_SPP, dtype=f32, device=dev)
        bt = torch.full((S,), t_green, dtype=torch.int32, device=dev)
        _trace(od, dd, rad, GEO_SPP, S, tabs, len(wv), n_surf, bt, True, flags[0:1], dev)
        mark("level 1 queued")
        ro, rd, rra = od.cpu(), dd.cpu(), rad.cpu()
        mark("level 1 back on the host")
        # (element-wise IEEE arithmetic: the same bits for all slices at once as slice by slice; the mean stays per slice)
        tt = (rd[..., 0] * ro[..., 0] + rd[..., 1] * ro[..., 1]) / (rd[..., 0] ** 2 + rd[..., 1] ** 2)
        tt = tt * rra
        fd_all = (ro[..., 2] - rd[..., 2] * tt).numpy()
        alive = (rra > 0).numpy()
        d_sensor = []
        for k in range(S):
            focus_d = fd_all[k][alive[k]]
            focus_d = focus_d[~np.isnan(focus_d) & (focus_d > 0)]
            with np.errstate(all="ignore"):
                z = float(np.mean(focus_d)) if len(focus_d) else float("nan")
            assert z > 0, "sensor position is negative."
            d_sensor.append(z)
        mark("d_sensor")
        # ---- level 2: calc_fov (deeplens/optics.py:1187-1217) - S batches of 100 rays from the sensor corner, backward
        M = 100
        pupilz, pupilx = lens.exit_pupil(shrink_pupil=True)
        x2 = torch.linspace(-pupilx, pupilx, M)
        o2 = torch.stack((x2, torch.full_like(x2, 0), torch.full_like(x2, pupilz)), axis=-1)
        o1 = torch.stack([torch.tensor([lens.r_last, 0, z]).repeat(M, 1).to(f32) for z in d_sensor])        # [S,100,3]
        dfov = F.normalize((o2.unsqueeze(0) - o1).float(), p=2, dim=-1)
        od, dd = o1.to(dev).contiguous(), dfov.to(dev).contiguous()
        rad = torch.ones(S, M, dtype=f32, device=dev)
        _trace(od, dd, rad, M, S, tabs, len(wv), n_surf, bt, bool(dfov[0, 0, 2] > 0), flags[1:2], dev)
        mark("level 2 queued")
        rd, rra = dd.cpu(), rad.cpu()
        mark("level 2 back on the host")
        _, enp_r = lens.entrance_pupil()
        hfov, foclen, fnum = [], [], []
        for k in range(S):
            tan_fov = rd[k, :, 0] / rd[k, :, 2]
            fov = torch.atan(torch.sum(tan_fov * rra[k]) / torch.sum(rra[k]))
            h = 0.5 if torch.isnan(fov) else fov.item()
            hfov.append(h)
            foclen.append(lens.r_last / np.tan(h))
            fnum.append(foclen[-1] / enp_r / 2)

        mark("hfov")
        # ---- level 3: psf_map (deeplens/optics.py:888-1026) - per slice and wavelength spp x N main rays and 2048 x N chief rays
        pts = lens.point_source_grid(depth=depth_plane_mm, grid=grid, quater=False).reshape(-1, 3).float()
        pobj = []
        for k in range(S):                                                                       # psf_diff's object points (optics.py:945-950)
            scale = -pts[:, 2] * np.tan(hfov[k]) / lens.r_last
            p = pts.clone()
            p[..., 0] = pts[..., 0] * scale * lens.sensor_size[1] / 2
            p[..., 1] = pts[..., 1] * scale * lens.sensor_size[0] / 2
            pobj.append(p)
        points = torch.stack(pobj).to(dev).contiguous()                                          # [S,N,3]
        enp_z, enp_rr = lens.entrance_pupil()
        pm = _pupil_points(um[:, :, 0], um[:, :, 1], enp_rr, enp_z).reshape(S * L, spp, 3).to(dev).contiguous()
        pc = _pupil_points(uc[:, :, 0], uc[:, :, 1], enp_rr * 0.5, enp_z).reshape(S * L, GEO_SPP, 3).to(dev).contiguous()
        B = S * L
        pset = torch.arange(S, dtype=torch.int32).repeat_interleave(L).to(dev)
        zs = torch.tensor(d_sensor, dtype=f32).repeat_interleave(L).to(dev)
        bt_main = torch.arange(L, dtype=torch.int32).repeat(S).to(dev)
        bt_chief = torch.full((B,), t_green, dtype=torch.int32, device=dev)
        centre = torch.empty((B, N, 2), dtype=f32, device=dev)
        any_valid = torch.zeros(B, dtype=torch.int32, device=dev)
        # chief rays (always the default wavelength, shrunk pupil) -> centres; the main rays then reuse their buffers.  The ray
        # state of a level (B x n x 36 bytes incl. the iterate buffer: 268 MB for the bench stack) is kept on the lens between calls: handing it back to
        # torch's caching allocator made every other call re-allocate it from the driver (65-100 ms instead of 15)
        nmax = max(spp, GEO_SPP) * N
        buf = getattr(lens, "_strict_rays", None)
        if buf is None or buf[0].shape[0] < B * nmax * 3 or buf[0].device != dev:
            buf = lens._strict_rays = (torch.empty(B * nmax * 3, dtype=f32, device=dev), torch.empty(B * nmax * 3, dtype=f32, device=dev),
                                       torch.empty(B * nmax, dtype=f32, device=dev), torch.empty(2 * B * nmax, dtype=f32, device=dev))
        oc, dc, rac = buf[0][:B * GEO_SPP * N * 3].view(B, GEO_SPP * N, 3), buf[1][:B * GEO_SPP * N * 3].view(B, GEO_SPP * N, 3), buf[2][:B * GEO_SPP * N].view(B, GEO_SPP * N)
        _trace(oc, dc, rac, GEO_SPP * N, B, tabs, len(wv), n_surf, bt_chief, True, flags[2:3], dev, points, pset, pc, N, zs, buf[3][:2 * B * GEO_SPP * N])
        _abi.call("aadff_strict_centroid", _abi.ptr(oc), _abi.ptr(rac), GEO_SPP, N, B, _abi.ptr(centre), _abi.ptr(any_valid), _abi.stream_ptr(dev))
        om, dm, ram = buf[0][:B * spp * N * 3].view(B, spp * N, 3), buf[1][:B * spp * N * 3].view(B, spp * N, 3), buf[2][:B * spp * N].view(B, spp * N)
        flag_m = torch.zeros(1, dtype=torch.int32, device=dev)
        _trace(om, dm, ram, spp * N, B, tabs, len(wv), n_surf, bt_main, True, flag_m, dev, points, pset, pm, N, zs, buf[3][:2 * B * spp * N])
        raw = torch.empty((B, N, ks, ks), dtype=f32, device=dev)
        nrm = torch.empty((N, ks, ks), dtype=f32, device=dev)
        st = _abi.stream_ptr(dev)
        for b in range(B):                                                                       # forward_integral (monte_carlo.py:9-57)
            _abi.call("aadff_psf_splat", _abi.ptr(om[b]), _abi.ptr(ram[b]), _abi.ptr(centre[b]), spp, N, float(lens.pixel_size), ks,
                      _abi.ptr(raw[b]), _abi.ptr(nrm), st)
        psf = raw / raw.sum(-1).sum(-1).unsqueeze(-1).unsqueeze(-1)                               # optics.py:978 (0/0 -> NaN like the reference)
        maps = psf.reshape(S, L, grid, grid, ks, ks).permute(0, 1, 2, 4, 3, 5).reshape(S, L, grid * ks, grid * ks).contiguous()   # make_grid, padding 0
        mark("level 3 queued")
        bits = (flags.cpu().tolist(), int(flag_m.item()), any_valid.cpu())
        mark("level 3 done")
        if marks is not None:
            lens._strict_timing = [(b[0], round((b[1] - a[1]) * 1e3, 3)) for a, b in zip(marks, marks[1:])]
    if any(bits[0]) or bits[1]:
        raise FloatingPointError("found nan in ft in non-diff newton method.")
    assert bool(bits[2].all()), "No sampled rays is valid."
    # the lens is left focused at the last distance
    lens._state_sync()
    hs = lens._state_host
    hs.d_sensor, hs.hfov, hs.tan_hfov = float(d_sensor[-1]), float(hfov[-1]), float(np.tan(hfov[-1]))
    hs.foclen, hs.fnum = float(foclen[-1]), float(fnum[-1])
    if lens._state_dev is not None:
        lens._state_upload()
    lens._strict_stack_scalars = {"d_sensor": d_sensor, "hfov": hfov}
    return maps
