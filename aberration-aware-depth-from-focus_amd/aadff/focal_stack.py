"""Focal-stack rendering: the whole stack in three kernel launches, sharded over GPUs.

Mode M1 (north-star path): per slice refocus(f_k) -> psf_map(depth plane) ->
render_psf_map, i.e. the loop of 2_aber_aware_dff_aif.py:104-114 with the grid renderer
(deeplens/optics.py:779-783) — here as ONE refocus launch over the S focus distances,
ONE fused PSF launch over S x 3 wavelengths x g^2 field points and ONE stack-fused
convolution; the S lens states stay on the device in between.
Mode M2: the per-pixel-PSF path the training scripts call (PSFNet.render per slice).
"""
import contextlib
import ctypes as C
import os

import numpy as np
import torch

from . import _abi
from deeplens.basics import DEFAULT_WAVE, GEO_SPP, WAVE_RGB


def select_focus_dist(depth, num, mode="linear", center=True):
    """`num` focus distances between the min and max VALID (>0) depth of each sample
    (reference: dff/utils.py:4-50).  depth [B,1,H,W] -> [B,n], sorted.
    'linear': n = num evenly spaced distances.  'importance': the two extremes plus rejection samples from a
    triangular density peaking at the mean valid depth, drawn with np.random in the reference's call order; like
    the reference it works for B = 1 only and returns num - 2 distances (its loop runs `while len < num - 2`,
    dff/utils.py:33-45).  `center` is accepted and unused, as in the reference."""
    assert num > 3, "Focal stack size is too small"
    dmax = torch.amax(depth, dim=(1, 2, 3))
    big = torch.where(depth > 0, depth, torch.full_like(depth, float("inf")))
    dmin = torch.amin(big, dim=(1, 2, 3))
    if mode == "linear":
        # the reference divides a CPU tensor by the Python scalar num - 1: an IEEE division; on the GPU ATen turns tensor / scalar
        # into tensor * (1 / scalar) (one ulp off for some values) - dividing by a TENSOR keeps the IEEE division on both devices
        den = torch.full_like(dmax, float(num - 1))
        f = [dmin + (i * (dmax - dmin)) / den for i in range(num)]
    elif mode == "importance":
        avg = torch.sum(depth, dim=(1, 2, 3)) / torch.sum(depth > 0, dim=(1, 2, 3))
        f = [dmax, dmin]
        while len(f) < num - 2:
            cand = np.random.rand() * (dmax - dmin) + dmin
            if cand > avg:                                   # B = 1 (bool of a one-element tensor), as in the reference
                rate = (dmax - cand) / (dmax - avg)
            else:
                rate = (cand - dmin) / (avg - dmin)
            if np.random.rand() < rate:
                f.append(cand)
    else:
        raise NotImplementedError
    return torch.sort(torch.stack(f, dim=1), dim=-1)[0]


def stack_uniform_layout(spp, L=3, spp_chief=GEO_SPP, spp_focus=GEO_SPP):
    """Flat per-slice layout of the uniforms in the reference's host-RNG draw order (SURVEY.md
    Appendix B): [focus theta, focus r, (main theta, main r, chief theta, chief r) x L].
    Returns (floats per slice, offset of main, offset of chief, stride between wavelengths)."""
    per_l = 2 * spp + 2 * spp_chief
    return 2 * spp_focus + L * per_l, 2 * spp_focus, 2 * spp_focus + 2 * spp, per_l


def draw_stack_uniforms(sampler, S, spp, L=3, spp_chief=GEO_SPP, spp_focus=GEO_SPP):
    """Views (u_focus [S,2,spp_focus], u_main [S,L,2,spp], u_chief [S,L,2,spp_chief]) of ONE
    flat draw of S slices (used by tests; the renderer consumes the flat block through strides)."""
    per, o_main, o_chief, per_l = stack_uniform_layout(spp, L, spp_chief, spp_focus)
    flat = sampler.rand_block([per * S]).reshape(S, per)
    u_focus = flat[:, :2 * spp_focus].reshape(S, 2, spp_focus)
    rest = flat[:, 2 * spp_focus:].reshape(S, L, per_l)
    return u_focus, rest[:, :, :2 * spp].reshape(S, L, 2, spp), rest[:, :, 2 * spp:].reshape(S, L, 2, spp_chief)


STAGED_UPLOAD = os.environ.get("AADFF_STAGED_UPLOAD", "1") != "0"
# staged path: focus traces of a stack on the plan's side stream, beside the previous stack's kernels.  Opt-in: +1-3 % of
# throughput, but in ~1 of 6 stacks the traces land beside the convolution, whose own duration then reads ~5 us longer -
# the default keeps every kernel alone on the device so that per-kernel measurements mean what they say.
REFOCUS_OVERLAP = os.environ.get("AADFF_REFOCUS_OVERLAP", "0") == "1"
try:                                                          # focus states uploaded by the refocus launch (tuning override, clamped)
    STAGE_FIRST = max(0, min(64, int(os.environ.get("AADFF_STAGE_FIRST", "3"))))
except ValueError:
    STAGE_FIRST = 3


class StackPlan:
    """Pre-allocated device buffers, cached lens tables and a pinned-host ring for the pupil
    samples of repeated M1 stacks of one shape.  The ring lets the host draw step i+1's
    samples while the GPU renders step i (copies are asynchronous from pinned memory)."""
    RING = 8           # pinned/device uniform blocks in flight
    GUARD_EVERY = 4    # staged path: one guard event per 4 steps protects the reuse of a pinned block RING steps later

    def __init__(self, lens, S, H, W, B=1, C_=3, grid=11, ks=11, spp=GEO_SPP, overlap_refocus=None):
        dev = lens._gpu()
        self.lens, self.S, self.grid, self.ks, self.spp, self.dev = lens, S, grid, ks, spp, dev
        # focus states, one block per ring slot: the focus traces of stack i+1 run on a side stream WHILE the PSF-grid and
        # convolution kernels of stack i (which read block i) are still running; `states` is the block of the last stack
        self.states_ring = [torch.zeros(S * C.sizeof(_abi.LensState), dtype=torch.uint8, device=dev) for _ in range(self.RING)]
        self.states = self.states_ring[0]
        # high priority: its few workgroups take the first slots the running PSF-grid kernel frees, instead of queueing
        # behind that kernel's remaining workgroups and then running beside (and slowing) the convolution
        prio = int(os.environ.get("AADFF_REFOCUS_PRIORITY", "-1"))
        self.side = torch.cuda.Stream(dev, priority=prio) if (REFOCUS_OVERLAP if overlap_refocus is None else overlap_refocus) else None
        self.side_events = [torch.cuda.Event() for _ in range(self.RING)]
        self.side_primed = False
        self.psf_maps = torch.empty((S, 3, grid * ks, grid * ks), dtype=torch.float32, device=dev)
        self.out = torch.empty((B, C_, S, H, W), dtype=torch.float32, device=dev)
        self.flags = torch.zeros(1, dtype=torch.int32, device=dev)
        self.flags_mirror = torch.zeros(1, dtype=torch.int32).pin_memory()    # aadff_publish_flags target (staged path)
        self.tab_rgb = lens._table(WAVE_RGB)
        self.tab_green = lens._table([DEFAULT_WAVE])
        self.lc = lens._lens_const()
        self.pts_xy = lens.point_source_grid(depth=0.0, grid=grid).reshape(-1, 3)
        self.conv_events = None      # optional (start, end) torch.cuda.Event pair around the conv launch
        # optional (start, stop) torch.cuda.Event pairs (enable_timing=True, recorded once so that their HIP events exist)
        # to be ATTACHED to the dispatch of the conv / PSF-grid kernel (aadff_time_next_launch): the kernel's own time
        self.conv_kernel_events = None
        self.psf_kernel_events = None
        self.psf_events = None       # optional (start, end) pair around the fused trace/PSF launch
        self.per, self.o_main, self.o_chief, self.per_l = stack_uniform_layout(spp)
        self.u_dev = [torch.empty(S * self.per, dtype=torch.float32, device=dev) for _ in range(self.RING)]
        self.u_pin = [torch.empty(S * self.per, dtype=torch.float32).pin_memory() for _ in range(self.RING)]
        self.u_evt = [None] * self.RING
        self.guards = {}
        self.turn = 0
        # staged upload (aadff_refocus_staged + aadff_psf_points_staged): the first STAGE_FIRST states' draws are
        # copied behind the focus traces, the rest by leading workgroups of the PSF launch
        self.stage_counters = torch.zeros(S, dtype=torch.int32, device=dev)
        self.refocus_scratch = torch.zeros(S * 16, dtype=torch.int32, device=dev)     # 64 B per focus state
        self.stage_generation = 0
        self._geo = {}

    def uniforms(self, sampler):
        """Device block of this step's uniforms, drawn in the reference's order."""
        k = self.turn % self.RING
        self.turn += 1
        n = self.S * self.per
        if sampler.on_device:
            self.u_dev[k] = sampler.rand_block([n])
            return self.u_dev[k]
        if self.u_evt[k] is not None:
            self.u_evt[k].synchronize()          # the copy that last used this pinned slot has finished
        if hasattr(sampler, "rand_into"):
            sampler.rand_into(self.u_pin[k])     # == the reference's call-by-call draws (same generator stream)
        else:
            self.u_pin[k].copy_(sampler.rand_block([n]))
        self.u_dev[k].copy_(self.u_pin[k], non_blocking=True)
        self.u_evt[k] = torch.cuda.Event()
        self.u_evt[k].record()
        return self.u_dev[k]

    def uniforms_host(self, sampler):
        """(pinned host block, device block) of this step's uniforms WITHOUT the upload: the caller hands
        both to aadff_refocus_staged, which copies inside the refocus launch, then calls `staged()`."""
        k = self.turn % self.RING
        last = self.turn - self.RING                # step that last used this block
        self.turn += 1
        if self.u_evt[k] is not None:
            self.u_evt[k].synchronize()          # (copy path) the upload that last used this pinned slot has finished
            self.u_evt[k] = None
        if last >= 0:
            g = self.guards.get(last // self.GUARD_EVERY)     # recorded at step 4 m + 3 >= last, at least 4 steps ago
            if g is not None:
                g.synchronize()
                self._poll_mirror()                          # flags of every step up to that guard, no extra sync
        if hasattr(sampler, "rand_into"):
            sampler.rand_into(self.u_pin[k])
        else:
            self.u_pin[k].copy_(sampler.rand_block([self.S * self.per]))
        return self.u_pin[k], self.u_dev[k]

    def staged(self):
        """End of a staged step: every GUARD_EVERY-th step records the event later reuses of the pinned blocks wait on
        (an event record per step costs ~3 us of queue gap)."""
        step = self.turn - 1
        if step % self.GUARD_EVERY == self.GUARD_EVERY - 1:
            _abi.call("aadff_publish_flags", _abi.ptr(self.flags), C.c_void_p(self.flags_mirror.data_ptr()), _abi.stream_ptr(self.dev))
            e = torch.cuda.Event()
            e.record()
            m = step // self.GUARD_EVERY
            self.guards[m] = e
            for old in [q for q in self.guards if q < m - self.RING // self.GUARD_EVERY - 1]:
                del self.guards[old]

    def _poll_mirror(self):
        """Pipelined stacks: the flags word as of the last guard step, read from its pinned mirror (the kernels' error
        conditions reach the host a few steps late instead of never; `check_flags` is the synchronous form)."""
        bits = int(self.flags_mirror[0])
        if bits:
            from deeplens.optics import raise_psf_flags
            self.flags_mirror[0] = 0
            self.flags.zero_()
            raise_psf_flags(bits)

    def check_flags(self):
        """Raise the reference's errors for anything the kernels of earlier steps flagged (synchronises)."""
        from deeplens.optics import raise_psf_flags
        bits = int(self.flags.item())
        if bits:
            self.flags.zero_()
            self.flags_mirror[0] = 0
        raise_psf_flags(bits)

    def geometry(self, focus, depth_plane_mm):
        """Device copies of the focus distances [S] and of the field points [S,N,3] at this depth plane, cached per
        (focus list, plane): a sharded run alternates between scenes and must not re-upload them every time."""
        key = (tuple(focus), float(depth_plane_mm))
        hit = self._geo.get(key)
        if hit is None:
            if len(self._geo) >= 64:
                self._geo.clear()
            pts = self.pts_xy.clone()
            pts[:, 2] = float(depth_plane_mm)
            hit = (torch.tensor(focus, dtype=torch.float32).to(self.dev), pts.unsqueeze(0).repeat(self.S, 1, 1).contiguous().to(self.dev))
            self._geo[key] = hit
        return hit


def _conv_into_units(x, maps, dest, B, C_, S, H, W, grid, ks, st):
    """Stack convolution with slice k written as the [C,H,W] unit `units[first + k*step]` of the caller's contiguous
    [M,C,H,W] buffer (aadff_render_psf_map_stack_strided); returns `units`."""
    units, first, ustep = dest
    assert B == 1 and units.is_cuda and units.is_contiguous() and units.dtype == torch.float32 and tuple(units.shape[1:]) == (C_, H, W)
    assert ustep >= 1 and 0 <= first and first + (S - 1) * ustep < units.shape[0], "dest units out of range"
    _abi.call("aadff_render_psf_map_stack_strided", _abi.ptr(x), _abi.ptr(maps),
              C.c_void_p(units.data_ptr() + 4 * first * C_ * H * W), H * W, ustep * C_ * H * W, 1, C_, S, H, W, grid, ks, st)
    return units


def strict_psf_maps(lens, depth_plane_mm, focus, grid, ks, spp):
    """PSF maps [S,3,g*ks,g*ks] of a strict-parity lens in the reference's host-RNG order (2_aber_aware_dff_aif.py:104-114 with
    deeplens/optics.py:779-783): three batched traces for the whole stack (aadff/strict_stack.py); AADFF_STRICT_BATCHED=0 runs
    the reference's loop call by call (refocus(f_k) then psf_map per slice, 72 single traces per slice) - same result."""
    from . import strict_stack
    if os.environ.get("AADFF_STRICT_BATCHED", "1") == "0":
        return strict_stack.strict_psf_maps_loop(lens, depth_plane_mm, focus, grid, ks, spp)
    return strict_stack.strict_psf_maps(lens, depth_plane_mm, focus, grid, ks, spp)


@torch.no_grad()
def render_focal_stack_m1(lens, img, depth_plane_mm, focus_mm, grid=11, ks=11, spp=GEO_SPP, plan=None,
                          return_maps=False, update_lens=True, dest=None):
    """[B,C,S,H,W] aberrated focal stack of `img` [B,C,H,W] for S focus distances (mm < 0),
    all scene points on one depth plane (mm < 0).  Three kernel launches, no host
    synchronisation; with a reused `plan` the output buffer is reused too.

    `dest=(units, first, step)` (B = 1): slice k is written as the [C,H,W] unit `units[first + k*step]` of a caller's
    contiguous [M,C,H,W] buffer instead of into the plan's stack (aadff_render_psf_map_stack_strided) - a sharded run
    renders straight into its all-gather buffer; the return value is then `units`."""
    focus = [float(f) for f in np.asarray(focus_mm, dtype=np.float64).reshape(-1)]
    S = len(focus)
    B, C_, H, W = img.shape
    assert tuple(lens.sensor_res) == (H, W), "lens.sensor_res must match the image"
    if getattr(lens, "parity", "fast") != "fast":
        # verification mode: the reference's own loop (refocus -> psf_map per slice, 2_aber_aware_dff_aif.py:104-114) with the
        # strict trace behind every call; only the convolution is shared with the fast path (deterministic to 2e-6 abs)
        dev = lens._gpu()
        keep = None if update_lens else _abi.LensState.from_buffer_copy(bytes(lens._state_sync()))
        try:
            maps = strict_psf_maps(lens, depth_plane_mm, focus, grid, ks, spp).to(dev).contiguous()
        finally:
            if keep is not None:                                 # update_lens=False: the lens stays focused where it was
                lens._state_host = keep
                if lens._state_dev is not None:
                    lens._state_upload()
        x = _abi.f32c(img, dev)
        with torch.cuda.device(dev):
            if dest is None:
                out = torch.empty((B, C_, S, H, W), dtype=torch.float32, device=dev)
                _abi.call("aadff_render_psf_map_stack", _abi.ptr(x), _abi.ptr(maps), _abi.ptr(out), B, C_, S, H, W, grid, ks, _abi.stream_ptr(dev))
            else:
                out = _conv_into_units(x, maps, dest, B, C_, S, H, W, grid, ks, _abi.stream_ptr(dev))
        return (out, maps) if return_maps else out
    own_plan = plan is None
    if own_plan:
        plan = StackPlan(lens, S, H, W, B, C_, grid, ks, spp)
    dev = plan.dev
    with torch.cuda.device(dev):
        dep, pts = plan.geometry(focus, depth_plane_mm)
        x = _abi.f32c(img, dev)
        N = grid * grid
        st = _abi.stream_ptr(dev)
        stage = None
        if lens.sampler.on_device or not STAGED_UPLOAD or plan.per % 4:
            u = plan.uniforms(lens.sampler)
            ub = u.data_ptr()
            _abi.call("aadff_refocus", _abi.ptr(dep), S, C.c_void_p(ub), GEO_SPP, plan.per, _abi.ptr(plan.tab_green),
                      plan.lc, _abi.ptr(plan.states), st)
        else:       # host draws: the upload rides on the refocus launch
            slot = plan.turn % plan.RING
            u_pin, u = plan.uniforms_host(lens.sampler)          # waits (host) until the stack that last used this slot is done
            ub = u.data_ptr()
            first = min(S, STAGE_FIRST)
            plan.states = plan.states_ring[slot]
            rst = st
            if plan.side is not None:
                # The focus traces depend on nothing the previous stack produces: launched on the plan's side stream they
                # run beside the previous stack's PSF-grid / convolution kernels (a handful of workgroups, latency-bound).
                if not plan.side_primed:
                    plan.side.wait_stream(torch.cuda.current_stream(dev))     # plan buffers were filled on the caller's stream
                    plan.side_primed = True
                rst = C.c_void_p(plan.side.cuda_stream)
            _abi.call("aadff_refocus_staged", _abi.ptr(dep), S, C.c_void_p(u_pin.data_ptr()), C.c_void_p(ub), first * plan.per,
                      GEO_SPP, plan.per, _abi.ptr(plan.tab_green), plan.lc, _abi.ptr(plan.states), _abi.ptr(plan.refocus_scratch), rst)
            if plan.side is not None:
                ev = plan.side_events[slot]
                ev.record(plan.side)
                torch.cuda.current_stream(dev).wait_event(ev)
            plan.stage_generation += 1
            stage = _abi.Stage(u_pin.data_ptr(), ub, plan.per, first, plan.stage_generation & 0xFFFFFFFF,
                               plan.stage_counters.data_ptr())
        if plan.psf_events is not None:
            plan.psf_events[0].record()
        if plan.psf_kernel_events is not None:
            _abi.call("aadff_time_next_launch", C.c_void_p(plan.psf_kernel_events[0].cuda_event), C.c_void_p(plan.psf_kernel_events[1].cuda_event))
        if stage is None:
            _abi.call("aadff_psf_points", _abi.ptr(pts), S, N, 3, _abi.ptr(plan.tab_rgb), _abi.ptr(plan.tab_green),
                      plan.lc, _abi.ptr(plan.states), C.c_void_p(ub + 4 * plan.o_main), spp, plan.per, plan.per_l,
                      C.c_void_p(ub + 4 * plan.o_chief), GEO_SPP, plan.per, plan.per_l, ks, 1, 1,
                      _abi.ptr(plan.psf_maps), None, _abi.ptr(plan.flags), st)
        else:
            _abi.call("aadff_psf_points_staged", _abi.ptr(pts), S, N, 3, _abi.ptr(plan.tab_rgb), _abi.ptr(plan.tab_green),
                      plan.lc, _abi.ptr(plan.states), C.c_void_p(ub + 4 * plan.o_main), spp, plan.per, plan.per_l,
                      C.c_void_p(ub + 4 * plan.o_chief), GEO_SPP, plan.per, plan.per_l, ks, 1, 1,
                      _abi.ptr(plan.psf_maps), None, _abi.ptr(plan.flags), C.byref(stage), st)
        if plan.psf_events is not None:
            plan.psf_events[1].record()
        if plan.conv_events is not None:
            plan.conv_events[0].record()
        kev = plan.conv_kernel_events
        if kev is not None:
            _abi.call("aadff_time_next_launch", C.c_void_p(kev[0].cuda_event), C.c_void_p(kev[1].cuda_event))
        if dest is None:
            _abi.call("aadff_render_psf_map_stack", _abi.ptr(x), _abi.ptr(plan.psf_maps), _abi.ptr(plan.out), B, C_, S,
                      H, W, grid, ks, st)
        else:
            _conv_into_units(x, plan.psf_maps, dest, B, C_, S, H, W, grid, ks, st)
        if plan.conv_events is not None:
            plan.conv_events[1].record()
        if stage is not None:
            plan.staged()        # pinned-slot guard: recorded behind the conv launch (an event record between the PSF and
                                 # conv launches widens that queue gap by ~4 us)
    if update_lens:      # the reference leaves the lens focused at the last distance
        nb = C.sizeof(_abi.LensState)
        lens._state_device().copy_(plan.states[(S - 1) * nb:S * nb])
        lens._state_stale = True
    if own_plan:
        plan.check_flags()                               # one-shot calls: same sync point as the reference's asserts
    res = plan.out if dest is None else dest[0]
    return (res, plan.psf_maps) if return_maps else res


class StackPipeline:
    """`depth` M1 stacks in flight on `depth` HIP streams (one StackPlan each, used in turn).  Within one stream a stack is
    three dependent launches (focus traces -> PSF grid -> convolution) with ~6-12 us of queue gap between them and a
    half-empty machine at the tail of every kernel; a second stack on another stream fills both (the PSF-grid kernel of
    stack i+1 is VALU-bound, the convolution of stack i is LDS/MFMA/HBM-bound).  The samples are still drawn on the host
    in call order (the reference's RNG stream), so every stack is the same as with a single plan.

    render() returns (out, done): `out` [B,C,S,H,W] belongs to the slot's plan and is complete when `done` (an event on
    the slot's stream) has fired: `torch.cuda.current_stream().wait_event(done)` or `wait()` before consuming it.  The
    slot is reused `depth` calls later.  By default the slot's stream first waits for the caller's current stream (inputs
    written there are complete; the previous consumer of the slot's output has finished); `inputs_ready=True` skips that
    barrier when the caller guarantees both (same resident image, outputs consumed)."""

    def __init__(self, lens, S, H, W, B=1, C_=3, grid=11, ks=11, spp=GEO_SPP, depth=2, overlap_refocus=None):
        self.dev, self.depth = lens._gpu(), int(depth)
        if overlap_refocus is None:
            overlap_refocus = REFOCUS_OVERLAP and self.depth == 1      # with >= 2 streams the other stack already fills the gap
        self.plans = [StackPlan(lens, S, H, W, B, C_, grid, ks, spp, overlap_refocus=overlap_refocus) for _ in range(self.depth)]
        self.streams = [torch.cuda.Stream(self.dev) for _ in range(self.depth)] if self.depth > 1 else [None]
        self.done = [None] * self.depth
        self.turn = 0
        self.args = (grid, ks, spp)

    def render(self, lens, img, depth_plane_mm, focus_mm, inputs_ready=False, update_lens=False):
        k = self.turn % self.depth
        self.turn += 1
        plan, st = self.plans[k], self.streams[k]
        if st is None:
            out = render_focal_stack_m1(lens, img, depth_plane_mm, focus_mm, *self.args, plan=plan, update_lens=update_lens)
            return out, None
        if not inputs_ready:
            st.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(st):
            out = render_focal_stack_m1(lens, img, depth_plane_mm, focus_mm, *self.args, plan=plan, update_lens=update_lens)
            if self.done[k] is None:
                self.done[k] = torch.cuda.Event()
            self.done[k].record(st)
        return out, self.done[k]

    def wait(self):
        """The caller's current stream waits for every stack rendered so far."""
        cur = torch.cuda.current_stream(self.dev)
        for e in self.done:
            if e is not None:
                cur.wait_event(e)

    def check_flags(self):
        for p in self.plans:
            p.check_flags()


def depth_layers(depth_mm, layers):
    """Quantise a depth map (mm, < 0 = in front of the lens; 0 = invalid) into `layers` bins that are uniform
    in depth between the nearest and the farthest valid pixel.  Returns (index map int64 like depth, centre
    depth of each layer [layers], mm < 0).  Invalid pixels go to the farthest layer."""
    d = -depth_mm                                    # positive distances
    valid = d > 0
    dmin = torch.where(valid, d, torch.full_like(d, float("inf"))).amin()
    dmax = d.amax()
    width = torch.clamp((dmax - dmin) / layers, min=1e-6)
    idx = torch.clamp(((d - dmin) / width).floor().long(), 0, layers - 1)
    idx = torch.where(valid, idx, torch.full_like(idx, layers - 1))
    centres = -(dmin + (torch.arange(layers, device=d.device, dtype=torch.float32) + 0.5) * width)
    return idx, centres


@torch.no_grad()
def render_focal_stack_m1_layered(lens, img, depth_mm, focus_mm, layers=4, grid=11, ks=11, spp=GEO_SPP, fused=True, return_parts=False):
    """RGB-D aware grid rendering ("M1-layered"): the depth map is quantised into `layers`
    planes; for every focus distance and layer one PSF map is ray traced and the slice is the per-pixel
    selection  out[s] = sum_l [layer == l] * render_psf_map(img, psf_map[s, l]).
    Composes reference primitives only (refocus, psf_map, render_psf_map).  Host-RNG order: per focus
    distance the refocus draws, then the psf_map draws of layer 0, 1, ...   img [B,C,H,W], depth_mm
    [B,1,H,W] (mm, < 0), focus_mm [S] (mm, < 0) -> [B,C,S,H,W].
    Round 5: three launches - refocus, ONE PSF-grid launch for all S x L (slice, layer) pairs, and (ks 11) a stack convolution that
    takes the layer-index map and writes [B,C,S,H,W] directly (`aadff_render_psf_map_stack_layered`); `fused=False` (and other ks)
    renders the S x L candidate slices and gathers, L x the output bytes."""
    focus = [float(f) for f in np.asarray(focus_mm, dtype=np.float64).reshape(-1)]
    S, L = len(focus), int(layers)
    B, C_, H, W = img.shape
    assert tuple(lens.sensor_res) == (H, W), "lens.sensor_res must match the image"
    assert 1 <= L <= 254, "1..254 depth layers"
    dev = lens._gpu()
    x = _abi.f32c(img, dev)
    idx, centres = depth_layers(_abi.f32c(depth_mm, dev), L)
    N, nb = grid * grid, C.sizeof(_abi.LensState)
    per, o_main, o_chief, per_l = stack_uniform_layout(spp)              # per-layer block = per - 2*GEO_SPP
    layer_block = per - o_main
    slice_stride = o_main + L * layer_block
    u = lens.sampler.rand_block([S * slice_stride]).to(dev)
    states = torch.zeros(S * nb, dtype=torch.uint8, device=dev)
    dep = torch.tensor(focus, dtype=torch.float32, device=dev)
    maps = torch.empty((S * L, 3, grid * ks, grid * ks), dtype=torch.float32, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    # field points of the (slice, layer) pairs, built on the device (no read-back of the layer centres): the grid at z = centre[l]
    pts = lens.point_source_grid(depth=0.0, grid=grid).reshape(1, -1, 3).to(dev).repeat(L, 1, 1)
    pts[:, :, 2] = centres.to(torch.float32).reshape(L, 1)
    pts = pts.repeat(S, 1, 1).contiguous()                               # [S*L, N, 3], pair p = s * L + l
    tab_rgb, tab_green, lc = lens._table(WAVE_RGB), lens._table([DEFAULT_WAVE]), lens._lens_const()
    with torch.cuda.device(dev):
        st = _abi.stream_ptr(dev)
        _abi.call("aadff_refocus", _abi.ptr(dep), S, C.c_void_p(u.data_ptr()), GEO_SPP, slice_stride, _abi.ptr(tab_green), lc,
                  _abi.ptr(states), st)
        # ONE PSF launch for all S x L pairs: pair p takes focus state p // L (the states repeated L times) and its own block of
        # draws - the per-slice blocks minus the refocus draws are [S, L * layer_block] = [S * L, layer_block] once made contiguous
        st_rep = states.view(S, nb).repeat_interleave(L, 0).contiguous()
        u_pairs = u.view(S, slice_stride)[:, o_main:].contiguous()
        base = u_pairs.data_ptr()
        _abi.call("aadff_psf_points", _abi.ptr(pts), S * L, N, 3, _abi.ptr(tab_rgb), _abi.ptr(tab_green), lc, _abi.ptr(st_rep),
                  C.c_void_p(base), spp, layer_block, per_l, C.c_void_p(base + 4 * 2 * spp), GEO_SPP, layer_block, per_l, ks, 1, 1,
                  _abi.ptr(maps), None, _abi.ptr(flags), st)
        if ks == 11 and fused:
            # the L candidates of a pixel are computed from one staged band and only the pixel's own layer is written: 1 x the output
            lidx = idx.reshape(B, H, W).to(torch.uint8).contiguous()
            out = torch.empty((B, C_, S, H, W), dtype=torch.float32, device=dev)
            _abi.call("aadff_render_psf_map_stack_layered", _abi.ptr(x), _abi.ptr(maps), _abi.ptr(lidx), _abi.ptr(out), B, C_, S, L, H, W, grid, ks, st)
        else:
            tmp = torch.empty((B, C_, S * L, H, W), dtype=torch.float32, device=dev)
            _abi.call("aadff_render_psf_map_stack", _abi.ptr(x), _abi.ptr(maps), _abi.ptr(tmp), B, C_, S * L, H, W, grid, ks, st)
            sel = idx.reshape(B, 1, 1, 1, H, W).expand(B, C_, S, 1, H, W)
            out = torch.gather(tmp.view(B, C_, S, L, H, W), 3, sel).squeeze(3)
    lens._state_device().copy_(states[(S - 1) * nb:S * nb])
    lens._state_stale = True
    lens._m1l_flags = flags                                               # raise_psf_flags(int(flags.item())) is the caller's (synchronising) check
    if return_parts:                                                      # (stack, PSF maps [S*L,3,g*ks,g*ks], layer index [B,H,W] uint8)
        return out, maps, idx.reshape(B, H, W).to(torch.uint8).contiguous()
    return out


@torch.no_grad()
def render_focal_stack_m2(lens, img, depth_m, n_stack):
    """[B,C,S,H,W] stack with per-pixel PSFs: depth in metres (> 0 valid), focus distances by
    the 'linear' rule, slices via lens.render(img, -depth*1e3, -f*1e3)
    (reference: 2_aber_aware_dff_aif.py:104-114)."""
    fds = select_focus_dist(depth_m, n_stack)
    if hasattr(lens, "render_stack") and len(img.shape) == 4:        # PSFNet: the whole stack in one fused launch
        return lens.render_stack(img, -depth_m * 1e3, -fds * 1e3), fds
    return torch.stack([lens.render(img, -depth_m * 1e3, -fds[:, i] * 1e3) for i in range(n_stack)], dim=2), fds


def shard_units(n_units, rank, world, block=1):
    """Unit ownership (u // block) = rank (mod world); block = 1 is SURVEY.md §8e's round robin (aadff.dist.shard_units)."""
    from .dist import shard_units as f
    return f(n_units, rank, world, block)


class RowSampler:
    """The rows `rows` (ascending) of the `[n_rows, per]` block of draws that starts at the generator's current position,
    taken straight from the host generator: the rows in between are SKIPPED (aadff_host_mt19937_discard: the generator's
    regenerations only), not produced and dropped.  A rank of a sharded job owns some slices of a scene's stack and must
    consume exactly the draws the whole-stack render would give those slices.  The generator is left
    behind the last owned row, not behind the block (every scene is re-seeded)."""
    on_device = False

    def __init__(self, base, rows, per):
        self.base, self.rows, self.per = base, list(rows), int(per)
        assert all(b > a for a, b in zip(self.rows, self.rows[1:])), "rows must ascend"
        self.preset = None

    def rand_into(self, out):
        per, rows = self.per, self.rows
        if self.preset is not None or out.numel() != len(rows) * per or not out.is_contiguous() or out.dim() != 1:
            return self._fallback().rand_into(out)
        pos = i = 0
        while i < len(rows):
            j = i
            while j + 1 < len(rows) and rows[j + 1] == rows[j] + 1:
                j += 1                                           # a run of consecutive rows is one draw
            self.base.skip((rows[i] - pos) * per)
            self.base.rand_into(out[i * per:(j + 1) * per])
            pos, i = rows[j] + 1, j + 1
        self.preset = False
        return out

    def _fallback(self):
        if not self.preset:
            assert self.preset is None, "RowSampler: the rows were already consumed"
            block = torch.empty(len(self.rows) * self.per)
            self.rand_into(block)
            self.preset = PresetSampler(block)
        return self.preset

    def rand(self, n):
        return self._fallback().rand(n)

    def rand_block(self, sizes):
        return self._fallback().rand_block(sizes)


class PresetSampler:
    """Replays uniforms drawn earlier (host or device) in place of the lens's sampler.  `rows` selects rows of a 2-D host
    block without materialising the selection (plain memcpy per row: torch's parallel CPU copies and gathers stall for
    milliseconds on a box that exposes more logical CPUs than its cgroup quota lets run)."""

    def __init__(self, block, rows=None):
        self.on_device = block.is_cuda
        if rows is not None and not block.is_cuda and block.dim() == 2 and block.is_contiguous():
            self.block2d, self.rows, self.block = block, list(rows), None
        else:
            self.block2d, self.rows = None, None
            self.block = (block if rows is None else block[list(rows)]).reshape(-1)
        self.pos = 0

    def _take(self, n):
        if self.block is None:                      # lazily flatten (device consumers / odd sizes)
            self.block = self.block2d[self.rows].reshape(-1)
        out = self.block[self.pos:self.pos + n]
        assert out.numel() == n, "preset uniforms exhausted"
        self.pos += n
        return out

    def rand(self, n):
        return self._take(int(n))

    def rand_block(self, sizes):
        return self._take(int(sum(sizes)))

    def rand_into(self, out):
        n, es = out.numel(), out.element_size()
        if self.block is None and self.pos == 0 and out.is_contiguous() and n == len(self.rows) * self.block2d.shape[1]:
            per = self.block2d.shape[1]
            for i, r in enumerate(self.rows):
                C.memmove(out.data_ptr() + i * per * es, self.block2d.data_ptr() + r * per * es, per * es)
            self.pos = n
            return out
        src = self._take(n)
        if src.is_cuda or not src.is_contiguous() or not out.is_contiguous():
            out.copy_(src)
        else:
            C.memmove(out.data_ptr(), src.data_ptr(), n * es)
        return out


class SceneUnitRenderer:
    """M1 rendering of (scene, slice) units for the sharded configuration (BASELINE.json config 3: 16 scenes x 10
    slices over 8 GPUs, units u = scene*S + slice dealt to the ranks by aadff.dist.shard_units).

    A unit's pixels must not depend on which rank renders it or on what else that rank renders, so every rank
    draws the uniforms of the WHOLE stack of a scene in the reference's order (`seed_scene(scene)` then one flat
    draw: 0.1 ms on the host) and consumes only the rows of the slices it owns.  The slices a rank owns within one
    scene go through ONE refocus / PSF-grid / convolution launch triple."""

    def __init__(self, lens, scenes, S, grid=11, ks=11, spp=GEO_SPP, seed_scene=None, streams=1):
        """`seed_scene(scene)` seeds torch's CPU generator for a scene (default: the CPU half of `torch.manual_seed(scene)`,
        2 us instead of 26-125).  `streams` > 1: consecutive scene groups go to that many HIP streams in turn, each with
        plans of its own, so the small launches of one group (a rank of an 8-rank job owns 1-2 slices of a scene in the
        round-robin partition) run beside those of the next and the PSF-grid kernel of one scene beside the convolution
        of another (StackPipeline's effect: +10 % on whole stacks)."""
        self.lens, self.scenes, self.S, self.grid, self.ks, self.spp = lens, scenes, S, grid, ks, spp
        self.seed_scene = seed_scene or (lambda scene: torch.default_generator.manual_seed(int(scene)))
        self.per = stack_uniform_layout(spp)[0]
        self.plans = {}
        self.n_streams, self.streams, self.turn = max(1, int(streams)), None, 0

    def n_units(self):
        return len(self.scenes) * self.S

    def render(self, units, out=None, out_index=None, after_group=None):
        """`units`: unit ids in any order -> `out[out_index[k]]` = unit `units[k]` (default: `out` [len(units), C, H, W],
        out_index[k] = k).  The slices of one scene whose destinations form an arithmetic progression - always the case for
        a rank's share in ascending order, in the local layout and in the all-gather buffer alike - are written by the
        convolution itself (aadff_render_psf_map_stack_strided); anything else is rendered into the plan's stack and copied.
        `after_group(positions)` is called after the launches of each scene group, with the stream they were queued on as
        the current stream: the sharded renderer starts the group's all-gather chunks from it.  On return the caller's
        current stream waits for every group."""
        lens, S = self.lens, self.S
        by_scene = {}
        for pos, u in enumerate(units):
            by_scene.setdefault(u // S, []).append((u % S, pos if out_index is None else out_index[pos]))
        img0 = self.scenes[0][0]
        B, C_, H, W = img0.shape
        dev = lens._gpu()
        if out is None:
            out = torch.empty((len(units), C_, H, W), dtype=torch.float32, device=dev)
        cur = torch.cuda.current_stream(dev)
        multi = self.n_streams > 1 and len(by_scene) > 1
        if multi:
            if self.streams is None:
                self.streams = [torch.cuda.Stream(dev) for _ in range(self.n_streams)]
            for st in self.streams:
                st.wait_stream(cur)                       # inputs and `out` are ready on the caller's stream
        saved = lens.sampler
        host_rows = not saved.on_device and hasattr(saved, "skip")
        try:
            for scene, items in by_scene.items():
                img, depth_plane_mm, focus = self.scenes[scene]
                focus = [float(f) for f in np.asarray(focus, dtype=np.float64).reshape(-1)]
                assert len(focus) == S
                items = sorted(items)
                sl = [k for k, _ in items]
                dst = [q for _, q in items]
                self.seed_scene(scene)
                if host_rows:
                    lens.sampler = RowSampler(saved, sl, self.per)               # only the owned slices' draws, rest skipped
                else:
                    block = saved.rand_block([S * self.per]).reshape(S, self.per)        # the whole stack's draws
                    lens.sampler = PresetSampler(block, rows=sl)
                n = len(sl)
                slot = self.turn % self.n_streams if multi else 0
                self.turn += 1
                plan = self.plans.get((n, slot))
                if plan is None:
                    plan = self.plans[(n, slot)] = StackPlan(lens, n, H, W, 1, C_, self.grid, self.ks, self.spp)
                step = dst[1] - dst[0] if n > 1 else 1
                direct = (B == 1 and step >= 1 and all(dst[i] == dst[0] + i * step for i in range(n)) and out.is_cuda
                          and out.is_contiguous() and out.dtype == torch.float32
                          and 16 * step * C_ * H * W + 4 * H * W < (1 << 32))          # the kernel's 32-bit store offsets (conv.hip)
                with (torch.cuda.stream(self.streams[slot]) if multi else contextlib.nullcontext()):
                    st = render_focal_stack_m1(lens, img, depth_plane_mm, [focus[k] for k in sl], self.grid, self.ks, self.spp,
                                               plan=plan, update_lens=False, dest=(out, dst[0], step) if direct else None)
                    if not direct:
                        for i, q in enumerate(dst):
                            out[q].copy_(st[0, :, i])
                    if after_group is not None:
                        after_group(dst)
        finally:
            lens.sampler = saved
            if multi:
                for st in self.streams:
                    cur.wait_stream(st)
        return out

    def check_flags(self):
        for plan in self.plans.values():
            plan.check_flags()


def render_scenes_sharded(renderer, gather=True, stream=None, block=None):
    """This rank's share of all (scene, slice) units through the HIP renderer and, with `gather`, the full
    `[n_scenes*S, C, H, W]` set in unit order on every rank.  Returns `(out, mine)`, or with `stream`
    always `(out, mine, done_or_None)` (aadff.dist.render_sharded semantics).

    `block`: units are dealt to the ranks in blocks of that many consecutive units (aadff.dist.shard_units); default
    `scene_block` = whole scenes when there are enough of them (config 3 on 8 ranks: 2 scenes per rank, 2 launch triples
    and 2 scenes' draws per step instead of 16), 1 = SURVEY's u = r (mod world).

    No copy and no reorder anywhere: the buffer is laid out `[rows, world, block, C, H, W]`, whose row i holds the units
    (i*world)*block .. (i*world + world)*block - 1 - rank r's i-th block is block i*world + r, so this IS unit order.
    The convolution writes each of this rank's slices straight to its place (`aadff_render_psf_map_stack_strided`), and
    row i is completed by ONE in-place all-gather of that row (block x 12.6 MB per rank at 1024^2) as soon as the scene
    groups that produce this rank's block of it have been launched: with `stream` the gathers run on that side stream,
    overlapped with the rendering of the following scenes, per-row chunks instead of one monolithic collective at the end."""
    import torch.distributed as dist
    from . import dist as adist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    n = renderer.n_units()
    if block is None:
        block = adist.scene_block(n, renderer.S, world)
    mine = adist.shard_units(n, rank, world, block)
    share = adist.padded_share(n, world, block)
    img0 = renderer.scenes[0][0]
    dev = renderer.lens._gpu()
    unit_shape = tuple(img0.shape[1:])
    if not gather or (world == 1 and not adist.grouped()):
        local = torch.empty((share,) + unit_shape, dtype=torch.float32, device=dev)
        if len(mine) < share:
            local[len(mine):].zero_()                   # padding units of the equal share: defined, like dist.render_sharded's
        renderer.render(mine, out=local[:len(mine)])
        return (local[:len(mine)] if world == 1 else local, mine) if stream is None else (local[:len(mine)] if world == 1 else local, mine, None)
    full = torch.empty((share * world,) + unit_shape, dtype=torch.float32, device=dev)
    n_rows = share // block
    rows = full.view((n_rows, world, block) + unit_shape)
    owned = [0] * n_rows                                # units of this rank per row still to be queued
    for u in mine:
        owned[u // (world * block)] += 1
    for i in range(n_rows):                             # padding units of the equal-share gather (own slots past n)
        if owned[i] < block:
            rows[i, rank, owned[i]:].zero_()
    cur = torch.cuda.current_stream(dev)
    side = stream if stream is not None else cur
    if stream is not None:
        full.record_stream(stream)
        side.wait_stream(cur)                           # the padding zero_() above is queued on `cur`: a rank that owns nothing
                                                        # would otherwise gather its rows before they are written
    state = {"next": 0}

    def gather_ready():
        """Rows are gathered in ascending order on every rank (collectives must be issued in the same order everywhere)."""
        while state["next"] < n_rows and owned[state["next"]] == 0:
            with torch.cuda.stream(side):
                adist.gather_row(rows[state["next"]], rank)
            state["next"] += 1

    def after_group(dst):
        here = torch.cuda.current_stream(dev)           # the stream the group's launches were queued on
        if side != here:
            ev = torch.cuda.Event()
            ev.record(here)
            side.wait_event(ev)
        for q in dst:                                   # destinations are unit ids: the buffer is in unit order
            owned[q // (world * block)] -= 1
        gather_ready()

    renderer.render(mine, out=full, out_index=mine, after_group=after_group)
    gather_ready()                                      # rows in which this rank owns nothing
    assert state["next"] == n_rows
    if stream is None:
        return full[:n], mine
    done = torch.cuda.Event()
    done.record(stream)
    return full[:n], mine, done
