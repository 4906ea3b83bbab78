"""parity="strict" for a whole focal stack: three LEVELS instead of 72 S single traces, each level ONE launch.

The reference's loop (2_aber_aware_dff_aif.py:104-114 over deeplens/optics.py:1155-1217 refocus / calc_fov and :888-1026 psf_map)
makes, per slice, one 2048-ray focus trace, one 100-ray field-of-view trace and 3 x 2 traces of spp x N rays; each is its own
Newton batch (`while (|ft| > 5e-5).any()`, deeplens/surfaces.py:547).  Slices do not depend on each other, and within a slice
only  focus -> d_sensor -> field of view -> hfov -> object points  is a chain.  So the stack is three levels:

  level 1   S focus batches (2048 rays)                    host: the reference's focus-distance arithmetic, np.mean per slice
  level 2   S field-of-view batches (100 rays, backward)   host: tan / sum / atan per slice
  level 3   3 S main batches + 3 S chief batches           rays BUILT on the device (o2 - o, F.normalize), chief-ray centres in
            (spp x N rays each)                            ATen's summation order, histogram, normalisation, psf_map tiling

Round 5 (`AADFF_STRICT_FUSED=1`, default): every level is one launch of csrc/strict_fused.hip - a ray crosses all surfaces in
registers - with the batch-wide Newton counts SPECULATED from a per-lens table (`StrictCounts`) and verified afterwards from the
any-bits the launch saw: n is the reference's count for (batch, surface) <=> bits 0..n-2 are set and (bit n-1 is clear or n == 10).
The table is seeded by one run of the round-4 form below (`aadff_trace_rays_strict_batched`: one launch pair per surface, counting
passes), which also REPLAYS any batch whose prediction failed and corrects its row.  Same rays, same per-ray arithmetic: identical
d_sensor / hfov, PSF maps equal to the float atomics of the histogram (tests/test_gpu_margins.py).

What stays on the host is what the reference computes there with torch / numpy and what cannot be reproduced off its
libraries: the pupil sampling (`rand * 2 * pi`, MKL's vector sqrt / cos / sin - evaluated for all slices in one call each:
element-wise, position independent, checked in tests), the focus and field-of-view reductions.  The host generator is consumed
in the reference's order (per slice focus theta, focus r, then per wavelength main theta, main r, chief
theta, chief r)."""
import concurrent.futures
import ctypes as C
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import _abi
from deeplens.basics import DEFAULT_WAVE, GEO_SPP, WAVE_RGB

MAX_ITER = 10


def strict_psf_maps_loop(lens, depth_plane_mm, focus, grid, ks, spp):
    """The reference's loop, call by call (refocus(f_k) then psf_map, every trace a single `aadff_trace_rays_strict` call: the
    round-3 form the batched and fused forms are tested against - the per-call API's own fused halves are switched off here)."""
    keep = getattr(lens, "_strict_calls_fused", True)
    lens._strict_calls_fused = False
    try:
        return torch.stack([(lens.refocus(f), lens.psf_map(depth=depth_plane_mm, grid=grid, ks=ks, spp=spp))[1] for f in focus])
    finally:
        lens._strict_calls_fused = keep


def _run_steps(steps):
    try:
        while True:
            next(steps).synchronize()
    except StopIteration as done:
        return done.value


def calls_fused(lens):
    return getattr(lens, "_strict_calls_fused", True) and os.environ.get("AADFF_STRICT_CALLS_FUSED", "1") != "0"


@torch.no_grad()
def strict_refocus_call(lens, depth):
    """`Lensgroup.refocus` of a strict lens (deeplens/optics.py:1155-1180 + post_computation): the refocus and calc_fov levels of
    `_strict_psf_maps_steps` for one state - two fused launches on speculated counts instead of ~50 per-surface ones."""
    _run_steps(_strict_psf_maps_steps(lens, float(depth), [float(depth)], 1, 1, GEO_SPP, phase="focus"))


@torch.no_grad()
def strict_psf_map_call(lens, depth, grid, ks, spp):
    """`Lensgroup.psf_map` of a strict lens at its current state (deeplens/optics.py:888-1026): the psf_map level of
    `_strict_psf_maps_steps` for one state; [3, grid*ks, grid*ks] on the device."""
    return _run_steps(_strict_psf_maps_steps(lens, float(depth), [None], grid, ks, spp, phase="psf"))[0]


def _pupil_points(theta_u, r_u, radius, z):
    """[..., n] uniforms -> [..., n, 3] pupil / aperture points: theta = u * 2 * pi, r = sqrt(u * R^2), (r cos, r sin, z)
    (deeplens/optics.py:480-486, deeplens/surfaces.py:188-199), the reference's own torch calls on the host."""
    theta = theta_u * 2 * np.pi
    r = torch.sqrt(r_u * radius ** 2)
    return torch.stack((r * torch.cos(theta), r * torch.sin(theta), torch.full_like(r, z)), -1)


_SLEEF = []


def _sleef():
    """(cos, sin, sqrt addresses, kind) of the float32 vector cosine / sine torch's CPU kernels call on this machine (MKL's
    vmsCos / vmsSin / vmsSqrt, kind 1, or Sleef's u10 routines of width 8 / 16 - all exported by libtorch_cpu.so), or None when they are not
    there or their output differs from torch's own on a probe block (then the pupil points stay with torch: `_pupil_points`).  `aadff_host_pupil_points`
    evaluates a stack's 70 batches of pupil points with them in three calls that hold no interpreter lock."""
    if _SLEEF:
        return _SLEEF[0]
    got = None
    try:
        width = 1 if torch.backends.mkl.is_available() else {"AVX512": 16, "AVX2": 8}.get(torch.backends.cpu.get_cpu_capability())
        if width is not None and os.environ.get("AADFF_STRICT_HOST_PUPIL", "1") != "0":
            lib = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libtorch_cpu.so"))
            names = ("vmsCos", "vmsSin", "vmsSqrt") if width == 1 else (f"Sleef_cosf{width}_u10", f"Sleef_sinf{width}_u10")
            fn = [C.cast(getattr(lib, name), C.c_void_p) for name in names] + [None]
            local = None
            if width == 1:
                # MKL's vector math may fan a 2048-element call out over its OpenMP team; from a second host thread that is a second
                # team beside torch's (measured: 5 ms per call inside a CPU quota).  Thread-local setting, restored after each use.
                local = lib.MKL_Set_Num_Threads_Local
                local.argtypes, local.restype = [C.c_int], C.c_int
            got = (fn[0], fn[1], fn[2], width, local)
            g = torch.Generator().manual_seed(20261003)
            n = 4099                                                      # not a multiple of the vector width: tails too
            u = torch.rand(2 * n, generator=g)
            u[:3] = torch.tensor([0.0, 1.0 - 2.0 ** -24, 2.0 ** -24])
            for radius, z in ((6.789, -1.25), (torch.tensor(11.0317), 0.0)):
                want = _pupil_points(u[:n], u[n:], radius, z)
                have = torch.empty(n, 3)
                _pupil_rows(got, u, np.array([0], dtype=np.int64), np.array([n], dtype=np.int64), n, radius, z, have)
                if not torch.equal(want.view(torch.int32), have.view(torch.int32)):
                    got = None
                    break
    except (OSError, AttributeError, RuntimeError):
        got = None
    _SLEEF.append(got)
    return got


def _pupil_rows(fns, u, theta_off, r_off, n, radius, z, out):
    """rows of pupil points through `aadff_host_pupil_points`: row i from u[theta_off[i]:+n], u[r_off[i]:+n] into out[i] ([rows,n,3]
    float32, contiguous, host); the scalars rounded exactly as `_pupil_points`' tensor-times-scalar operations round them"""
    R2 = float(torch.ones((), dtype=torch.float32) * radius ** 2)
    assert u.dtype == torch.float32 and u.is_contiguous() and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == len(theta_off) * n * 3
    lib = _abi.load_library()
    prev = fns[4](1) if fns[4] is not None else None
    try:
        rc = _pupil_call(lib, fns, u, theta_off, r_off, n, R2, z, out)
    finally:
        if prev is not None:
            fns[4](prev)
    if rc != 0:
        raise RuntimeError("aadff_host_pupil_points failed: " + lib.aadff_last_error().decode(errors="replace"))


def _pupil_call(lib, fns, u, theta_off, r_off, n, R2, z, out):
    return lib.aadff_host_pupil_points(C.c_void_p(u.data_ptr()), len(theta_off), C.c_void_p(theta_off.ctypes.data), C.c_void_p(r_off.ctypes.data),
                   n, float(np.float32(np.pi)), R2, float(np.float32(z)), C.c_void_p(out.data_ptr()), fns[0], fns[1], fns[2], fns[3])


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


# ---------------------------------------------------------------------------------------------------------------- count table
def counts_of_masks(any_bits):
    """any-bits words -> the iterations the reference's loop runs: the first iteration whose bit is clear, at most ten
    (`iterations_of`, csrc/strict_math.h)."""
    m = np.asarray(any_bits).astype(np.uint32)
    n = np.full(m.shape, MAX_ITER, dtype=np.int32)
    for it in range(MAX_ITER - 1, -1, -1):
        n = np.where((m >> np.uint32(it)) & 1 == 0, it + 1, n)
    return n


def prediction_holds(any_bits, pred, curved):
    """[..., n_surf] any-bits seen by a fused launch that ran pred iterations -> [...] bool: every curved surface of the batch
    shows "bits 0..n-2 set and (bit n-1 clear or n == 10)"."""
    m = np.asarray(any_bits).astype(np.uint32)
    n = np.asarray(pred).astype(np.uint32)
    low = (np.uint32(1) << (n - 1)) - np.uint32(1)
    ok = ((m & low) == low) & ((n == MAX_ITER) | (((m >> (n - 1)) & 1) == 0))
    return np.all(ok | ~np.asarray(curved, dtype=bool), axis=-1)


def check_counts(any_bits, pred, curved, order):
    """The any-bits a fused launch saw under the predicted counts -> (ok [...], corrected [..., n_surf]).  Surfaces are examined in the
    order the rays cross them (`order`); at the FIRST curved surface whose prediction n fails the batch is marked bad and its row
    corrected there - an earlier iteration without any ray above the tolerance gives the true count (first clear bit + 1); all n
    bits set with n < 10 means the loop runs on: n + 1 is tried - and nothing behind that surface is looked at (its bits are
    meaningless)."""
    idx = np.asarray(order, dtype=np.int64)
    pred = np.asarray(pred, dtype=np.int32)
    m = np.asarray(any_bits).astype(np.uint32)[..., idx]
    n = pred[..., idx].astype(np.uint32)
    low = (np.uint32(1) << (n - 1)) - np.uint32(1)
    short = (m & low) != low
    more = ~short & (n < MAX_ITER) & (((m >> (n - 1)) & 1) == 1)
    bad = (short | more) & np.asarray(curved, dtype=bool)[idx]
    if not bad.any():                                     # the usual case: every prediction of the launch holds
        return np.ones(bad.shape[:-1], dtype=bool), pred
    ok = ~bad.any(-1)
    fix = pred.copy()
    if not ok.all():
        first = bad.argmax(-1)[..., None]                 # the first failing surface in crossing order
        take = lambda a: np.take_along_axis(a, first, -1)
        corrected = np.where(take(short), counts_of_masks(take(m | ~low)), take(n).astype(np.int32) + 1)
        where = idx[first]
        np.put_along_axis(fix, where, np.where(ok[..., None], np.take_along_axis(pred, where, -1), corrected), -1)
    return ok, fix


class StrictCounts:
    """Predicted batch-wide Newton counts of one lens, per level: key -> int32 [B, (2,) MAX_SURF].  Lives in the lens's table cache
    (dropped by `Lensgroup.invalidate()`); `stats` counts seed runs, fused runs and replayed batches."""

    def __init__(self):
        self.rows = {}
        self.seen = {}                                    # key -> uint16 [..., MAX_SURF]: bit n set = count n has been the truth
        self.stats = {"seeded": 0, "fused": 0, "replayed_batches": 0, "fused_replays": 0, "per_surface_replays": 0}

    def learn(self, key, rows):
        """`rows` are TRUE counts of this call: they become the prediction and join the alternatives seen."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        self.rows[key] = rows
        bit = (np.uint16(1) << rows.astype(np.uint16)).astype(np.uint16)
        old = self.seen.get(key)
        self.seen[key] = bit if old is None or old.shape != bit.shape else (old | bit)

    def candidates(self, key, b, limit=4):
        """Count rows to try for batch b of a cheap level: the last truth, then single-surface deviations to every other count seen."""
        row = self.rows[key][b]
        out = [row]
        seen = self.seen[key][b]
        for i in np.nonzero(seen & (seen - 1))[0]:        # surfaces with more than one count on record
            for n in range(1, MAX_ITER + 1):
                if (int(seen[i]) >> n) & 1 and n != row[i] and len(out) < limit:
                    alt = row.copy()
                    alt[i] = n
                    out.append(alt)
        return out

    @staticmethod
    def of(lens):
        t = lens._table_cache.get("strict-counts")
        if t is None:
            t = lens._table_cache["strict-counts"] = StrictCounts()
        return t


ALT_JOBS = os.environ.get("AADFF_STRICT_ALT", "1") != "0"       # two-variant psf_map jobs (aadff_strict_psf_points_alt) instead of re-launches


def _two_variant_words(seen_chief, curved):
    """[B, MS] uint16 masks of the chief counts on record (bit n = count n has been the truth) -> [B] int32: surface | n << 8 where
    exactly ONE curved surface of the batch's chief row has more than one count on record and these are two neighbours n, n + 1;
    -1 otherwise."""
    s = np.asarray(seen_chief).astype(np.uint32) * np.asarray(curved, dtype=np.uint32)[None, :]
    multi = (s & (s - 1)) != 0
    one = multi.sum(-1) == 1
    idx = multi.argmax(-1)
    v = s[np.arange(len(s)), idx]
    low = v & (~v + np.uint32(1))
    pair = one & (v == (low | (low << np.uint32(1))))
    n_lo = np.zeros(len(s), dtype=np.int64)
    n_lo[pair] = np.log2(low[pair].astype(np.float64)).astype(np.int64)
    out = np.where(pair & (n_lo >= 1) & (n_lo + 1 <= MAX_ITER), idx | (n_lo << 8), -1).astype(np.int32)
    return out


def _curved(lens):
    c = lens._table_cache.get("strict-curved")
    if c is None:
        c = np.zeros(_abi.MAX_SURF, dtype=bool)
        for i, s in enumerate(lens.surfaces):
            c[i] = s.pack(DEFAULT_WAVE).kind != _abi.SURF_STOP
        lens._table_cache["strict-curved"] = c
    return c


# ---------------------------------------------------------------------------------------------------------------- launches
def _trace(o, d, ra, n, B, tabs, n_tables, n_surf, batch_table, forward, flags, dev, points=None, point_set=None, pupil=None, N=1, z_sensor=None,
           tbuf=None):
    """Round-4 form: one launch pair per surface with counting passes; returns the scratch words ([B][MAX_SURF] any-bits first)."""
    scratch = torch.empty(2 * B * _abi.MAX_SURF + 1, dtype=torch.int32, device=dev)
    if tbuf is None:
        tbuf = torch.empty(2 * B * n, dtype=torch.float32, device=dev)
    _abi.call("aadff_trace_rays_strict_batched", _abi.ptr(o), _abi.ptr(d), _abi.ptr(ra), n, B, C.byref(tabs), n_tables, n_surf,
              _abi.ptr(batch_table), _abi.ptr(points), _abi.ptr(point_set), _abi.ptr(pupil), N, 0, n_surf, int(forward), _abi.ptr(z_sensor),
              _abi.ptr(scratch), _abi.ptr(tbuf), _abi.ptr(flags), _abi.stream_ptr(dev))
    return scratch


def _masks_to_counts(scratch, B):
    return counts_of_masks(scratch[:B * _abi.MAX_SURF].cpu().numpy().view(np.uint32).reshape(B, _abi.MAX_SURF))


def _level1_batched(lens, uf, focus, S, tabs, n_tables, n_surf, bt, dev):
    """Round-4 form of level 1 (refocus, deeplens/optics.py:1155-1180): rays built on the host, one launch pair per surface, the
    focus-distance arithmetic on the host.  Returns (fd [S,2048] numpy, alive [S,2048] numpy bool, counts [S,MAX_SURF])."""
    f32 = torch.float32
    s0 = lens.surfaces[0]
    o = _pupil_points(uf[:, 0], uf[:, 1], s0.r, s0.d.item())                                   # [S,2048,3]
    tgt = torch.zeros(S, 1, 3, dtype=f32)
    tgt[:, 0, 2] = torch.tensor([float(f) for f in focus], dtype=f32)
    d = F.normalize((o - tgt).float(), p=2, dim=-1)                                            # Ray.__init__
    od, dd = o.to(dev).contiguous(), d.to(dev).contiguous()
    rad = torch.ones(S, GEO_SPP, dtype=f32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = _trace(od, dd, rad, GEO_SPP, S, tabs, n_tables, n_surf, bt, True, flag, dev)
    ro, rd, rra = od.cpu(), dd.cpu(), rad.cpu()
    if int(flag.item()):
        raise FloatingPointError("found nan in ft in non-diff newton method.")
    # (element-wise IEEE arithmetic: the same bits for all slices at once as slice by slice; the mean stays per slice)
    tt = (rd[..., 0] * ro[..., 0] + rd[..., 1] * ro[..., 1]) / (rd[..., 0] ** 2 + rd[..., 1] ** 2)
    tt = tt * rra
    return (ro[..., 2] - rd[..., 2] * tt).numpy(), (rra > 0).numpy(), _masks_to_counts(scratch, S)


class _HostFast:
    """The per-slice host reductions of a stack as ONE array operation each where that gives the reference's bits: numpy's mean along
    the contiguous axis and ATen's sum(1) run the same pairwise / vectorised inner loop per row as the 1-D calls the reference makes,
    and the element-wise float32 operations of psf_diff's object points do not depend on the batch shape.  That is a property of
    the installed numpy / torch builds, so it is CHECKED, not assumed: the first CHECKS uses of each form also run the reference's
    call-by-call form and compare bit for bit; a mismatch switches the form off for the process (with a warning) and the
    call-by-call result is used.  AADFF_HOST_FAST=0 switches all of them off (0.5 ms of Python per stack)."""
    CHECKS = 3
    on = os.environ.get("AADFF_HOST_FAST", "1") != "0"
    left = {"mean": CHECKS, "fov": CHECKS, "points": CHECKS}

    @classmethod
    def use(cls, what):
        return cls.on and cls.left[what] >= 0

    @classmethod
    def verify(cls, what, same):
        """called with the comparison of the two forms while checks are left; returns True when the fast result may be used"""
        if not same:
            import warnings
            cls.left[what] = -1
            warnings.warn(f"aadff: vectorised host form '{what}' differs from the call-by-call form on this numpy / torch build; switched off",
                          RuntimeWarning, stacklevel=3)
            return False
        cls.left[what] -= 1
        if cls.left[what] < 0:
            cls.left[what] = 0
        return True


class _HostNative:
    """The native host driver of the short levels (csrc/stack_host.cpp).  AADFF_HOST_NATIVE=0: the Python form throughout."""
    on = os.environ.get("AADFF_HOST_NATIVE", "1") != "0"

    @classmethod
    def switch_off(cls, what):
        import warnings
        cls.on = False
        warnings.warn(f"aadff: native host driver differs from the Python form ({what}); switched off", RuntimeWarning, stacklevel=3)


def _d_sensor_loop(fd_all, alive):
    """np.mean of the valid positive crossing distances per slice (optics.py:1175-1178), the reference's calls slice by slice."""
    out = []
    for k in range(fd_all.shape[0]):
        focus_d = fd_all[k][alive[k]]
        focus_d = focus_d[~np.isnan(focus_d) & (focus_d > 0)]
        with np.errstate(all="ignore"):
            z = float(np.mean(focus_d)) if len(focus_d) else float("nan")
        out.append(z)
    return out


_MEAN_CHECKED = [0]


def _d_sensor_of(fd_all, alive, weights=None):
    """np.mean of the valid positive crossing distances per slice (optics.py:1175-1178): `aadff_host_masked_mean_f32` - numpy's pairwise
    float32 summation and float64 division restated in C, all slices in one call (the reference's filtering and np.mean slice by
    slice are ~10 us of numpy overhead each) - checked against numpy itself on its first uses (`_HostFast`).  `weights`: the float32
    ray weights the mask `alive` came from (rows countable where weight > 0)."""
    S, n = fd_all.shape
    fast = _HostFast.use("mean") and fd_all.dtype == np.float32 and fd_all.flags.c_contiguous
    if fast:
        w = weights if weights is not None and weights.dtype == np.float32 and weights.flags.c_contiguous else alive.astype(np.float32)
        out_a, scratch = np.empty(S, dtype=np.float32), np.empty(max(n, 1), dtype=np.float32)
        rc = _abi.load_library().aadff_host_masked_mean_f32(C.c_void_p(fd_all.ctypes.data), C.c_void_p(w.ctypes.data), S, n, C.c_void_p(scratch.ctypes.data),
                                                            C.c_void_p(out_a.ctypes.data))
        fast = rc == 0
    if fast:
        out = [float(v) for v in out_a]
        if _MEAN_CHECKED[0] < _HostFast.CHECKS:
            _MEAN_CHECKED[0] += 1
            want = _d_sensor_loop(fd_all, alive)
            same = all((a == b) or (a != a and b != b) for a, b in zip(out, want))
            if not _HostFast.verify("mean", same):
                out = want
    else:
        out = _d_sensor_loop(fd_all, alive)
    for z in out:
        assert z > 0, "sensor position is negative."
    return out


def _fov_geometry(lens, d_sensor):
    """calc_fov's rays (optics.py:1187-1204): o1 [S,3] sensor corners, o2 [100,3] points across the shrunk exit pupil."""
    M = 100
    o2 = lens._table_cache.get("strict-fov-o2")                 # depends on the surfaces only (`Lensgroup.invalidate()` drops the cache)
    if o2 is None:
        pupilz, pupilx = lens.exit_pupil(shrink_pupil=True)
        x2 = torch.linspace(-pupilx, pupilx, M)
        o2 = lens._table_cache["strict-fov-o2"] = torch.stack((x2, torch.full_like(x2, 0), torch.full_like(x2, pupilz)), axis=-1)
    # one tensor of Python floats -> float32, as torch.tensor([r_last, 0, z]) rounds each of them
    o1 = torch.tensor([[lens.r_last, 0, z] for z in d_sensor], dtype=torch.float32)                # [S,3]
    return o1, o2


def _level2_batched(lens, d_sensor, S, tabs, n_tables, n_surf, bt, dev):
    """Round-4 form of level 2 (calc_fov, optics.py:1187-1217).  Returns (tan_fov [S,100], ra [S,100] host tensors, counts)."""
    f32 = torch.float32
    o1, o2 = _fov_geometry(lens, d_sensor)
    M = o2.shape[0]
    o1 = o1.unsqueeze(1).repeat(1, M, 1)
    dfov = F.normalize((o2.unsqueeze(0) - o1).float(), p=2, dim=-1)
    od, dd = o1.to(dev).contiguous(), dfov.to(dev).contiguous()
    rad = torch.ones(S, M, dtype=f32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = _trace(od, dd, rad, M, S, tabs, n_tables, n_surf, bt, bool(dfov[0, 0, 2] > 0), flag, dev)
    rd, rra = dd.cpu(), rad.cpu()
    if int(flag.item()):
        raise FloatingPointError("found nan in ft in non-diff newton method.")
    return rd[..., 0] / rd[..., 2], rra, _masks_to_counts(scratch, S)


def _fov_loop(tan_fov, rra):
    out = []
    for k in range(tan_fov.shape[0]):
        fov = torch.atan(torch.sum(tan_fov[k] * rra[k]) / torch.sum(rra[k]))
        out.append(0.5 if torch.isnan(fov) else fov.item())
    return out


_FOV_CHECKED = [0]


def _fov_of(lens, tan_fov, rra):
    """hfov / foclen / fnum per slice from the traced tangents (optics.py:1205-1217, 1097-1102, 186-187).  The S sums and arctangents
    are one sum(1) and one atan over [S] (the same vectorised inner loop per row / the same element-wise routine; `_HostFast`)."""
    _, enp_r = lens.entrance_pupil()
    if _HostFast.use("fov") and tan_fov.is_contiguous() and rra.is_contiguous():
        fov = torch.atan((tan_fov * rra).sum(1) / rra.sum(1))
        hfov = [0.5 if v != v else v for v in fov.tolist()]
        if _FOV_CHECKED[0] < _HostFast.CHECKS:
            _FOV_CHECKED[0] += 1
            want = _fov_loop(tan_fov, rra)
            if not _HostFast.verify("fov", hfov == want):
                hfov = want
    else:
        hfov = _fov_loop(tan_fov, rra)
    foclen = [lens.r_last / np.tan(h) for h in hfov]
    fnum = [f / enp_r / 2 for f in foclen]
    return hfov, foclen, fnum


def _object_points_loop(lens, pts, hfov):
    """psf_diff's object points per slice (optics.py:945-950), the reference's tensor operations slice by slice."""
    pobj = []
    for h in hfov:
        scale = -pts[:, 2] * np.tan(h) / lens.r_last
        p = pts.clone()
        p[..., 0] = pts[..., 0] * scale * lens.sensor_size[1] / 2
        p[..., 1] = pts[..., 1] * scale * lens.sensor_size[0] / 2
        pobj.append(p)
    return torch.stack(pobj)


_POINTS_CHECKED = [0]


def _object_points(lens, pts, hfov):
    """psf_diff's object points per slice (optics.py:945-950): [S,N,3] float32 on the host.  Every operation of the reference is an
    element-wise float32 one between a tensor and a Python scalar (which ATen rounds to float32 once): the S slices are one
    broadcast of the same operations in the same order, the scalars rounded the same way (`_HostFast` checks it)."""
    if not (_HostFast.use("points") and pts.dtype == torch.float32):
        return _object_points_loop(lens, pts, hfov)
    # numpy float32 arithmetic: the same IEEE single-precision multiply / divide per element as ATen's CPU kernels, at a fifth of
    # the per-call overhead (eight small operations per stack)
    p = pts.numpy()
    tanh = np.array([np.tan(h) for h in hfov], dtype=np.float64).astype(np.float32)[:, None]             # [S,1]: the scalar as ATen rounds it
    r_last, sw, sh, two = (np.float32(v) for v in (lens.r_last, lens.sensor_size[1], lens.sensor_size[0], 2.0))
    scale = ((-p[:, 2])[None, :] * tanh) / r_last                                                         # [S,N]
    o = np.empty((len(hfov),) + p.shape, dtype=np.float32)
    o[..., 0] = ((p[:, 0][None, :] * scale) * sw) / two
    o[..., 1] = ((p[:, 1][None, :] * scale) * sh) / two
    o[..., 2] = p[:, 2][None, :]
    out = torch.from_numpy(o)
    if _POINTS_CHECKED[0] < _HostFast.CHECKS:
        _POINTS_CHECKED[0] += 1
        want = _object_points_loop(lens, pts, hfov)
        if not _HostFast.verify("points", torch.equal(out.view(torch.int32), want.view(torch.int32))):
            out = want
    return out


def _level3_batched(lens, sel, points, pset, pc, pm, zs, bt_chief, bt_main, tabs, n_tables, n_surf, N, spp, ks, dev):
    """Round-4 form of level 3 for the batches `sel` (LongTensor on the device, or None for all): chief trace -> centres -> main trace
    -> histogram per batch.  Returns (psf [B',N,ks,ks] normalised, counts [B',2,MAX_SURF], any_valid [B'] (host))."""
    f32 = torch.float32
    if sel is not None:
        pset, pc, pm, zs, bt_chief, bt_main = (t[sel].contiguous() for t in (pset, pc, pm, zs, bt_chief, bt_main))
    B = pset.shape[0]
    centre = torch.empty((B, N, 2), dtype=f32, device=dev)
    any_valid = torch.zeros(B, dtype=torch.int32, device=dev)
    # The ray state of a level (B x n x 36 bytes incl. the iterate buffer: 268 MB for the bench stack) is kept on the lens between
    # calls: handing it back to torch's caching allocator made every other call re-allocate it from the driver (65-100 ms instead
    # of 15).  `release_buffers(lens)` / `Lensgroup.invalidate()` drop it.
    nmax = max(spp, GEO_SPP) * N
    buf = lens._table_cache.get("strict-rays")
    if buf is None or buf[0].shape[0] < B * nmax * 3 or buf[0].device != dev:
        buf = lens._table_cache["strict-rays"] = (torch.empty(B * nmax * 3, dtype=f32, device=dev), torch.empty(B * nmax * 3, dtype=f32, device=dev),
                                                  torch.empty(B * nmax, dtype=f32, device=dev), torch.empty(2 * B * nmax, dtype=f32, device=dev))
    flag = torch.zeros(2, dtype=torch.int32, device=dev)
    oc, dc, rac = buf[0][:B * GEO_SPP * N * 3].view(B, GEO_SPP * N, 3), buf[1][:B * GEO_SPP * N * 3].view(B, GEO_SPP * N, 3), buf[2][:B * GEO_SPP * N].view(B, GEO_SPP * N)
    sc = _trace(oc, dc, rac, GEO_SPP * N, B, tabs, n_tables, n_surf, bt_chief, True, flag[0:1], dev, points, pset, pc, N, zs, buf[3][:2 * B * GEO_SPP * N])
    _abi.call("aadff_strict_centroid", _abi.ptr(oc), _abi.ptr(rac), GEO_SPP, N, B, _abi.ptr(centre), _abi.ptr(any_valid), _abi.stream_ptr(dev))
    om, dm, ram = buf[0][:B * spp * N * 3].view(B, spp * N, 3), buf[1][:B * spp * N * 3].view(B, spp * N, 3), buf[2][:B * spp * N].view(B, spp * N)
    sm = _trace(om, dm, ram, spp * N, B, tabs, n_tables, n_surf, bt_main, True, flag[1:2], dev, points, pset, pm, N, zs, buf[3][:2 * B * spp * N])
    raw = torch.empty((B, N, ks, ks), dtype=f32, device=dev)
    nrm = torch.empty((N, ks, ks), dtype=f32, device=dev)
    st = _abi.stream_ptr(dev)
    for b in range(B):                                                                       # forward_integral (monte_carlo.py:9-57)
        _abi.call("aadff_psf_splat", _abi.ptr(om[b]), _abi.ptr(ram[b]), _abi.ptr(centre[b]), spp, N, float(lens.pixel_size), ks,
                  _abi.ptr(raw[b]), _abi.ptr(nrm), st)
    psf = raw / raw.sum(-1).sum(-1).unsqueeze(-1).unsqueeze(-1)                               # optics.py:978 (0/0 -> NaN like the reference)
    cnt = np.stack((_masks_to_counts(sc, B), _masks_to_counts(sm, B)), 1)
    if flag.cpu().any():
        raise FloatingPointError("found nan in ft in non-diff newton method.")
    return psf, cnt, any_valid.cpu()


def _tile(psf, grid, ks):
    """[B,N,ks,ks] -> [B,g*ks,g*ks]: make_grid with padding 0 (deeplens/optics.py:1025)."""
    B = psf.shape[0]
    return psf.reshape(B, grid, grid, ks, ks).permute(0, 1, 3, 2, 4).reshape(B, grid * ks, grid * ks).contiguous()


def release_buffers(lens):
    """Drop the ray-state buffers the round-4 form keeps on the lens (268 MB for the bench stack)."""
    lens._table_cache.pop("strict-rays", None)


_WORKER = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="aadff-strict-host")   # see strict_psf_maps
JOBS_PER_BATCH = 4          # candidate count rows per batch of a cheap level (levels 1, 2)
FUSED_ROUNDS = 4            # corrected re-launches before a batch goes back to the per-surface form

# parity="edge" (round 6): levels 1 and 2 as above - d_sensor and hfov must be the reference's to the last bit, an ulp in either
# re-draws the rounding noise of every ray behind it (tools/edge_sim.py: a re-trace from an hfov one ulp off is worth nothing) -
# and level 3 on the FAST kernel, which leaves the one discontinuous decision of the histogram (the window test of
# deeplens/monte_carlo.py:37) open for the rays within EDGE_DELTA_MM of the window edge; those few rays (0.1-0.5 % of the rays
# inside the window, a few hundred per slice) are re-traced in the strict arithmetic from the reference's host-exact pupil and
# object points and decided there (aadff_psf_points_edge -> aadff_strict_edge_retrace -> aadff_psf_normalise).
EDGE_DELTA_MM = float(os.environ.get("AADFF_EDGE_DELTA_MM", "2e-4"))    # 5 x the largest fast-vs-reference hit distance measured (4.1e-5 mm, G2)
EDGE_CAP = int(os.environ.get("AADFF_EDGE_CAP", "8192"))               # deferred rays per (slice, wavelength) batch; more: strict psf_map for that stack


def is_edge(lens):
    return getattr(lens, "parity", "fast") == "edge"


class _Stage:
    """Pinned host blocks and their device twins for one stack shape (S, L, N, spp): ONE upload of all pupil points in front of
    level 1, one small parameter block up and one result block down per launch (every `.to(device)` / `.cpu()` of a small tensor is
    30-100 us of its own).  Integer and float words share a block (int32 storage, float32 views)."""

    def __init__(self, dev, S, L, N, spp, t_green, phase="all"):
        B, MS, M = S * L, _abi.MAX_SURF, 100
        i32, f32 = torch.int32, torch.float32
        self.key = (S, L, N, spp, t_green, phase)
        self.J = J = JOBS_PER_BATCH * S
        self.n_pf, self.n_pm, self.n_pc = S * GEO_SPP * 3, B * spp * 3, B * GEO_SPP * 3
        self.h_pupil = torch.empty(self.n_pf + self.n_pm + self.n_pc, dtype=f32, pin_memory=True)
        self.d_pupil = torch.empty(self.n_pf + self.n_pm + self.n_pc, dtype=f32, device=dev)
        # parameter blocks.  levels 1 / 2: [geometry G | job -> batch J | pred J*MS]  (G = S*3 axis points | S*3 sensor corners + M*3 pupil points)
        #                   level 3: [z_sensor B | object points S*N*3 | pred B*2*MS | two-variant word B];  its replays: [job -> batch B | pred B*2*MS]
        self.G = [S * 3, S * 3 + M * 3]
        sizes = [self.G[0] + J + J * MS, self.G[1] + J + J * MS, B + S * N * 3 + B * 2 * MS + B, B + B * 2 * MS]
        self.h_par = [torch.empty(n, dtype=i32, pin_memory=True) for n in sizes]
        self.d_par = [torch.empty(n, dtype=i32, device=dev) for n in sizes]
        # result blocks: levels 1 / 2 [value J*n | ra J*n | bits J*2*MS], level 3 [bits B*4*MS | any_valid B | second variant: chief bits B*2*MS |
        # any_valid B], its replays [bits B*4*MS | any_valid B]
        rs = [2 * J * GEO_SPP + J * 2 * MS, 2 * J * M + J * 2 * MS, B * 4 * MS + B + B * 2 * MS + B, B * 4 * MS + B]
        self.h_res = [torch.empty(n, dtype=i32, pin_memory=True) for n in rs]
        self.d_res = [torch.empty(n, dtype=i32, device=dev) for n in rs]
        self.pset = torch.arange(S, dtype=i32).repeat_interleave(L).to(dev)
        self.bt_main = torch.arange(L, dtype=i32).repeat(S).to(dev)
        self.bt_green = torch.full((max(B, J),), t_green, dtype=i32, device=dev)
        self.zeros = torch.zeros(J, dtype=i32, device=dev)
        self.events = [torch.cuda.Event() for _ in sizes]
        # where the uniforms of every batch of pupil points sit in the stack's flat block of draws (stack_uniform_layout)
        from .focal_stack import stack_uniform_layout
        per, o_main, o_chief, per_l = stack_uniform_layout(spp, L)
        if phase == "focus":                                 # the draws of a refocus call alone / of a psf_map call alone
            per = 2 * GEO_SPP
        elif phase == "psf":
            per, o_main, o_chief = L * per_l, 0, 2 * spp
        sl = (np.arange(S, dtype=np.int64)[:, None] * per + np.arange(L, dtype=np.int64)[None, :] * per_l).reshape(-1)
        self.off_focus = np.arange(S, dtype=np.int64) * per
        self.off_main, self.off_chief = sl + o_main, sl + o_chief

    def native_levels(self, lens, counts, keys, curved, tab_dev, n_tables, n_surf, S):
        """aadff_levels_t for csrc/stack_host.cpp (the short levels' host work as one call per wait), with the candidate count rows
        of both levels written into the parameter blocks - rewritten only when the table's rows / alternatives have changed.
        Returns (struct, cand1, cand2): cand = [(batch, row)] in job order."""
        nl = getattr(self, "_native", None)
        if nl is None:
            lv = _abi.Levels()
            lv.S, lv.n_surf, lv.n_tables, lv.jobs_max, lv.fov_rays = S, n_surf, n_tables, self.J, 100
            lv.tables_dev, lv.bt_green, lv.zeros = tab_dev.data_ptr(), self.bt_green.data_ptr(), self.zeros.data_ptr()
            for i in (0, 1):
                setattr(lv, f"h_par{i + 1}", self.h_par[i].data_ptr()); setattr(lv, f"d_par{i + 1}", self.d_par[i].data_ptr())
                setattr(lv, f"h_res{i + 1}", self.h_res[i].data_ptr()); setattr(lv, f"d_res{i + 1}", self.d_res[i].data_ptr())
            lv.h_pupil, lv.d_pupil = self.h_pupil.data_ptr(), self.d_pupil.data_ptr()
            for i in range(_abi.MAX_SURF):
                lv.curved[i] = 1 if curved[i] else 0
            o1, o2 = _fov_geometry(lens, [1.0] * S)
            self.h_par[1][S * 3:S * 3 + 100 * 3].view(torch.float32).view(100, 3).copy_(o2)         # constant per lens
            ev = [torch.cuda.Event(), torch.cuda.Event()]
            for e in ev:
                e.record()                                   # creates the HIP event behind it: the driver records it by handle
            nl = self._native = {"lv": lv, "ver": [None, None], "cand": [None, None], "first": [None, None], "events": ev,
                                 "chosen": np.empty(S, dtype=np.int32), "dsens": np.empty(S, dtype=np.float32), "scratch": np.empty(GEO_SPP, dtype=np.float32),
                                 "tan": np.empty((S, 100), dtype=np.float32), "ra": np.empty((S, 100), dtype=np.float32),
                                 "focus": np.empty(S, dtype=np.float32), "o2z": o2[0, 2].clone(), "checked": 0}
        lv = nl["lv"]
        for lvl in (0, 1):
            # the candidate rows in the parameter block stay while the table says the same (compared by content: 2 x S x 32 words)
            rows_now, seen_now = counts.rows[keys[lvl]], counts.seen[keys[lvl]]
            ver = nl["ver"][lvl]
            if ver is None or not (np.array_equal(ver[0], rows_now) and np.array_equal(ver[1], seen_now)):
                ver = (rows_now.copy(), seen_now.copy())
                cand = [(b, row) for b in range(S) for row in counts.candidates(keys[lvl], b, JOBS_PER_BATCH)]
                G, J = self.G[lvl], len(cand)
                h = self.h_par[lvl].numpy()
                h[G:G + J] = [b for b, _ in cand]
                h[G + self.J:G + self.J + J * _abi.MAX_SURF] = np.stack([row for _, row in cand]).astype(np.int32).reshape(-1)
                nl["cand"][lvl], nl["ver"][lvl] = cand, ver
                first = np.full(S, -1, dtype=np.int32)
                for j in range(J - 1, -1, -1):
                    first[cand[j][0]] = j
                nl["first"][lvl] = first                  # the job that carries each batch's predicted row
                setattr(lv, "J1" if lvl == 0 else "J2", J)
        return nl

    def edge_buffers(self, dev, S, L, N, per, ks):
        """buffers of the edge-exact level 3 (allocated at the first use, re-allocated when ks changes): the stack's uniforms (the fast
        kernel samples the pupil itself), the S lens states and normalised field points, the deferred-ray lists and raw histograms"""
        e = getattr(self, "edge", None)
        if e is None or e["ks"] != ks:
            B, i32, f32 = S * L, torch.int32, torch.float32
            nst = S * (C.sizeof(_abi.LensState) // 4)
            e = self.edge = {
                "ks": ks, "nst": nst,
                "h_u": torch.empty(S * per, dtype=f32, pin_memory=True), "d_u": torch.empty(S * per, dtype=f32, device=dev),
                "h_in": torch.empty(nst + S * N * 3, dtype=i32, pin_memory=True), "d_in": torch.empty(nst + S * N * 3, dtype=i32, device=dev),
                "count": torch.zeros(B + 1, dtype=i32, device=dev),                 # [B] counts | flags word
                "list": torch.empty(B * EDGE_CAP, dtype=i32, device=dev),
                "raw": torch.empty(B * N * ks * ks, dtype=f32, device=dev),
                "slope": torch.empty(B * N * 2, dtype=f32, device=dev),
                "states_prov": torch.zeros(nst, dtype=i32, device=dev),          # the fast refocus kernel's states (provisional pass)
                "h_focus": torch.empty(2 * S, dtype=f32, pin_memory=True), "d_focus": torch.empty(2 * S, dtype=f32, device=dev),   # [focus S | tan_exact S]
                "h_back": torch.zeros(B + 1, dtype=i32, pin_memory=True),
                "uploaded": torch.cuda.Event(), "done": torch.cuda.Event(), "busy": False,
            }
            for name in ("uploaded", "done"):
                e[name].record()                             # creates the HIP events: the native driver records them by handle
        return e

    @staticmethod
    def of(lens, dev, S, L, N, spp, t_green, phase="all"):
        name = "strict-stage" if phase == "all" else "strict-stage-" + phase
        st = lens._table_cache.get(name)
        if st is None or st.key != (S, L, N, spp, t_green, phase) or st.d_pupil.device != dev:
            st = lens._table_cache[name] = _Stage(dev, S, L, N, spp, t_green, phase)
        return st

    def submit(self, i, n_up, n_down, launch, stream):
        """enqueue: upload the first n_up words of parameter block i, launch, download the first n_down result words; returns the
        event recorded behind the download (the generators of this module `yield` it: the sequential driver waits for it, the
        pipeline goes on with another stack until it has happened)"""
        with torch.cuda.stream(stream):
            self.d_par[i][:n_up].copy_(self.h_par[i][:n_up], non_blocking=True)
            launch(self.d_par[i], self.d_res[i])
            self.h_res[i][:n_down].copy_(self.d_res[i][:n_down], non_blocking=True)
            self.events[i].record(stream)
        return self.events[i]

    def result(self, i):
        """the result block of the last `submit(i, ...)` once its event has happened"""
        return self.h_res[i].numpy()


def _ptr_at(t, word):
    return C.c_void_p(t.data_ptr() + 4 * word)


def _nan_in_run(nan_bits, pred, curved):
    """a NaN residual in an iteration the reference runs (it exits there, deeplens/surfaces.py:555-558)"""
    ran = (np.uint32(1) << np.asarray(pred).astype(np.uint32)) - np.uint32(1)
    return bool(((np.asarray(nan_bits).astype(np.uint32) & ran) != 0)[..., curved].any())


def _speculate_small(counts, key, st, lvl, S, n, curved, order, launch, stream, after_first_submit=None):
    """A cheap level (S batches of n rays) on speculated counts: every batch is traced under each candidate row of the table at
    once (`StrictCounts.candidates`: 2048-ray batches cost nothing, a second round trip costs 60-100 us), the first candidate whose
    any-bits confirm it is the batch's result; batches without one are re-launched with the row `check_counts` corrected.
    `launch(J, par, res)` enqueues the kernel for J jobs.  A generator (yields the event of every launch; `yield from` it): returns
    (out0 [S,n], out1 [S,n]) float32, or None when some batch is still unconfirmed after FUSED_ROUNDS launches (the caller falls back
    to the per-surface form)."""
    MS, G = _abi.MAX_SURF, st.G[lvl]
    cand = [(b, row) for b in range(S) for row in counts.candidates(key, b, JOBS_PER_BATCH)]
    out0, out1 = np.empty((S, n), dtype=np.float32), np.empty((S, n), dtype=np.float32)
    truth = np.empty((S, MS), dtype=np.int32)
    done = np.zeros(S, dtype=bool)
    h = st.h_par[lvl].numpy()
    for rnd in range(FUSED_ROUNDS):
        J = len(cand)
        pred = np.stack([row for _, row in cand]).astype(np.int32)
        h[G:G + J] = [b for b, _ in cand]
        h[G + st.J:G + st.J + J * MS] = pred.reshape(-1)
        ev = st.submit(lvl, G + st.J + J * MS, 2 * J * n + J * 2 * MS, lambda par, res: launch(J, par, res), stream)
        if after_first_submit is not None:                  # host work of the caller that can run while this launch is in flight
            after_first_submit()
            after_first_submit = None
        yield ev
        r = st.result(lvl)
        counts.stats["fused"] += 1
        bits = r[2 * J * n:2 * J * n + J * 2 * MS].view(np.uint32).reshape(J, 2, MS)
        ok, fix = check_counts(bits[:, 0], pred, curved, order)
        if _nan_in_run(bits[ok, 1], pred[ok], curved):
            raise FloatingPointError("found nan in ft in non-diff newton method.")
        nxt = {}
        for j, (b, row) in enumerate(cand):
            if done[b]:
                continue
            if ok[j]:
                out0[b] = r[j * n:(j + 1) * n].view(np.float32)
                out1[b] = r[(J + j) * n:(J + j + 1) * n].view(np.float32)
                truth[b], done[b] = row, True
                nxt.pop(b, None)
            elif b not in nxt:
                nxt[b] = fix[j]
        if done.all():
            counts.learn(key, truth)
            return out0, out1
        cand = [(b, row) for b, row in nxt.items() if not done[b]]
        counts.stats["fused_replays"] += 1
    return None


@torch.no_grad()
def strict_psf_maps(lens, depth_plane_mm, focus, grid, ks, spp, fused=None):
    """PSF maps [S,3,g*ks,g*ks] (device) of a strict-parity lens for the focus distances `focus`, all field points on the plane
    `depth_plane_mm`; leaves the lens focused at the last distance, like the reference's loop."""
    steps = _strict_psf_maps_steps(lens, depth_plane_mm, focus, grid, ks, spp, fused)
    try:
        while True:
            next(steps).synchronize()                        # every yield is the event of a launch whose result the next step reads
    except StopIteration as done:
        return done.value


def _strict_psf_maps_steps(lens, depth_plane_mm, focus, grid, ks, spp, fused=None, phase="all"):
    """`strict_psf_maps` as a generator: every host wait of the fused form - the round trips of the two short levels, the psf_map
    launch, its re-launches - is a `yield` of the event to wait for.  `strict_psf_maps` waits right there; `StrictPipeline` goes on
    with whichever other stack's event has happened.  Returns the maps (StopIteration.value).
    `phase`: "all" = a stack (refocus -> calc_fov -> psf_map per slice); "focus" = the refocus / calc_fov half only (the lens state is
    set, None returned) and "psf" = the psf_map half only at the lens's current state - what `Lensgroup.refocus` and `.psf_map` of a
    strict lens call (one state each): the same kernels, tables and checks, the draws of each call in the reference's order."""
    from .focal_stack import stack_uniform_layout
    if ks > _abi.MAX_KS:
        raise ValueError(f"ks={ks} exceeds the kernels' limit {_abi.MAX_KS}")
    if fused is None:
        fused = os.environ.get("AADFF_STRICT_FUSED", "1") != "0"
    t_enter = time.perf_counter()
    S, L, N = len(focus), len(WAVE_RGB), grid * grid
    B, MS = S * L, _abi.MAX_SURF
    dev = lens._gpu()
    n_surf = len(lens.surfaces)
    wv = list(WAVE_RGB) + ([] if DEFAULT_WAVE in WAVE_RGB else [DEFAULT_WAVE])
    t_green = wv.index(DEFAULT_WAVE)
    tabs = _tables(lens, wv)
    tab_dev = lens._table(wv)
    counts = StrictCounts.of(lens)
    curved = _curved(lens)
    f32 = torch.float32
    assert phase in ("all", "focus", "psf")
    keys = (("focus", S), ("fov", S), ("psf", B, N, spp))
    need = keys if phase == "all" else (keys[:2] if phase == "focus" else keys[2:])
    fused = fused and all(k in counts.rows for k in need)      # no table yet: this call is the seed run (round-4 form throughout)

    # ---- the draws, in the reference's order (one flat draw = the same generator stream as call by call)
    per, o_main, o_chief, per_l = stack_uniform_layout(spp, L)
    if phase == "focus":
        per = 2 * GEO_SPP
    elif phase == "psf":
        per = L * per_l
    edge = is_edge(lens) and fused and phase != "focus"
    eb = None
    if edge:
        # the fast kernel of level 3 samples the pupil itself from the raw uniforms: they are drawn straight into the pinned block
        # its upload starts from (the previous stack's upload out of that block has long completed - its event is checked all the same)
        eb = _Stage.of(lens, dev, S, L, N, spp, t_green, phase).edge_buffers(dev, S, L, N, per, ks)
        if eb["busy"]:
            eb["uploaded"].synchronize()
        u = lens.sampler.rand_into(eb["h_u"]).view(S, per)
    else:
        u = lens.sampler.rand_block([S * per]).cpu().reshape(S, per)
    uf = um = uc = None
    if phase != "psf":
        uf = u[:, :2 * GEO_SPP].reshape(S, 2, GEO_SPP)
    if phase != "focus":
        rest = u[:, (2 * GEO_SPP if phase == "all" else 0):].reshape(S, L, per_l)
        um = rest[:, :, :2 * spp].reshape(S, L, 2, spp)
        uc = rest[:, :, 2 * spp:].reshape(S, L, 2, GEO_SPP)

    marks = [("start", time.perf_counter())] if os.environ.get("AADFF_STRICT_TIMING") == "1" else None
    mark = (lambda name: marks.append((name, time.perf_counter()))) if marks is not None else (lambda name: None)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev)
        sp = _abi.stream_ptr(dev)
        enp_z, enp_rr = lens.entrance_pupil()
        s0 = lens.surfaces[0]
        pts = lens.point_source_grid(depth=depth_plane_mm, grid=grid, quater=False).reshape(-1, 3).float()
        fwd_order, bwd_order = list(range(n_surf)), list(range(n_surf - 1, -1, -1))
        st = None
        bt_green = None
        if fused:
            st = _Stage.of(lens, dev, S, L, N, spp, t_green, phase)
            hp = st.h_pupil
            vec = _sleef()
            # the two short levels may run on a stream of their own (StrictPipeline: a high-priority one, so that they do not queue
            # behind the psf_map launch of the stack in front); every level ends with a host wait, which orders them with level 3
            s12 = getattr(lens, "_strict_fast_stream", None)
            if s12 is None and edge and phase == "all":
                # an edge stack starts its PSF kernel BEFORE the short levels (provisional pass below): they need a stream of their
                # own also outside a pipeline, a high-priority one (their few workgroups take the first slots the PSF kernel frees)
                s12 = lens._table_cache.get("edge-side-stream")
                if s12 is None:
                    s12 = lens._table_cache["edge-side-stream"] = torch.cuda.Stream(dev, priority=-1)
            s12 = stream if s12 is None else s12
            sp12 = C.c_void_p(s12.cuda_stream)
        prov = edge and phase == "all" and os.environ.get("AADFF_EDGE_PROVISIONAL", "1") != "0"
        if edge:
            maps = torch.empty((S, L, grid * ks, grid * ks), dtype=f32, device=dev)
            centre = torch.empty((B, N, 2), dtype=f32, device=dev)
            hin, nst, du, cnt = eb["h_in"], eb["nst"], eb["d_u"], eb["count"]
            if eb.get("pts_key") != (float(depth_plane_mm), grid):
                hin[nst:].view(f32).view(S, N, 3).copy_(pts.unsqueeze(0).expand(S, N, 3))
                eb["pts_key"] = (float(depth_plane_mm), grid)
                eb["d_in"][nst:].copy_(hin[nst:], non_blocking=True)
            o_main_w = 0 if phase == "psf" else 2 * GEO_SPP
            surf_bytes = C.sizeof(_abi.Surface)

            def launch_edge(states_ptr, slope_ptr):
                _abi.call("aadff_psf_points_edge", _ptr_at(eb["d_in"], nst), S, N, L, _abi.ptr(tab_dev), C.c_void_p(tab_dev.data_ptr() + t_green * n_surf * surf_bytes),
                          lens._lens_const(), states_ptr, _ptr_at(du, o_main_w), spp, per, per_l, _ptr_at(du, o_main_w + 2 * spp), GEO_SPP, per, per_l, ks,
                          EDGE_DELTA_MM, _abi.ptr(eb["raw"]), _abi.ptr(centre), slope_ptr, _ptr_at(cnt, 0), _abi.ptr(eb["list"]), EDGE_CAP, _ptr_at(cnt, B), sp)
        def edge_struct():
            """aadff_edge_stack_t of this stage's buffers (csrc/stack_host.cpp), built once per buffer set"""
            es = eb.get("es")
            if es is None:
                es = eb["es"] = _abi.EdgeStack()
                es.S, es.L, es.N, es.spp, es.ks, es.n_surf, es.n_tables, es.t_green, es.cap = S, L, N, spp, ks, n_surf, len(wv), t_green, EDGE_CAP
                es.per, es.per_l, es.o_main, es.n_pm = per, per_l, o_main_w, st.n_pm
                es.delta, es.pixel_size, es.lc = EDGE_DELTA_MM, float(lens.pixel_size), lens._lens_const()
                es.tables_dev = tab_dev.data_ptr()
                es.h_u, es.d_u, es.h_focus, es.d_focus = eb["h_u"].data_ptr(), du.data_ptr(), eb["h_focus"].data_ptr(), eb["d_focus"].data_ptr()
                es.d_pts, es.states_prov = eb["d_in"].data_ptr() + 4 * nst, eb["states_prov"].data_ptr()
                es.raw, es.slope, es.count, es.list, es.h_back = eb["raw"].data_ptr(), eb["slope"].data_ptr(), cnt.data_ptr(), eb["list"].data_ptr(), eb["h_back"].data_ptr()
                es.h_par3, es.d_par3, es.pset, es.bt_main = st.h_par[2].data_ptr(), st.d_par[2].data_ptr(), st.pset.data_ptr(), st.bt_main.data_ptr()
                es.h_pupil_main, es.d_pupil_main = hp.data_ptr() + 4 * st.n_pf, st.d_pupil.data_ptr() + 4 * st.n_pf
                eb["es_keep"] = (tab_dev, hp)                    # what the struct points into beyond eb / st
            return es

        def provisional_pass():
            if native:
                rc = lib.aadff_edge_provisional(C.byref(edge_struct()), C.c_void_p(nl["focus"].ctypes.data), _abi.ptr(centre), sp)
                if rc != 0:
                    raise RuntimeError("aadff_edge_provisional failed: " + lib.aadff_last_error().decode(errors="replace"))
                mark("provisional pass queued")
                return
            # ---- provisional pass (edge stacks): the PSF kernel needs lens states, and the exact ones come out of two host round trips
            # (levels 1 and 2 below, ~1 ms of latency with the GPU idle).  The interior rays do not care about a few ulps of d_sensor /
            # hfov - only the border decisions do, and those are taken by the re-trace in the exact world - so the fast refocus kernel
            # provides PROVISIONAL states from the same draws and the fast PSF kernel starts right away, beside the short levels; the
            # re-trace moves each centre into the exact world (csrc/strict_fused.hip: EdgeRetraceArgs).
            du.copy_(eb["h_u"], non_blocking=True)
            eb["h_focus"][:S] = torch.tensor([float(f) for f in focus], dtype=f32)
            eb["d_focus"][:S].copy_(eb["h_focus"][:S], non_blocking=True)
            cnt[B:].zero_()
            _abi.call("aadff_refocus", _abi.ptr(eb["d_focus"]), S, _abi.ptr(du), GEO_SPP, per, C.c_void_p(tab_dev.data_ptr() + t_green * n_surf * surf_bytes),
                      lens._lens_const(), _abi.ptr(eb["states_prov"]), sp)
            launch_edge(_abi.ptr(eb["states_prov"]), _abi.ptr(eb["slope"]))
            mark("provisional pass queued")
        # (queued right behind level 1's launch, below: the host work of queueing it then overlaps the level-1 kernel)
        if fused and phase != "focus":
            def psf_pupils():
                # the psf_map pupil points are not needed before level 3: a worker thread evaluates them (torch releases the GIL in
                # sqrt / cos / sin) while this thread goes through levels 1 and 2, which are launch and round-trip latency.  Slice by
                # slice: below ATen's grain size (32768 elements) an element-wise op stays on the calling thread - a second OpenMP
                # team next to the main thread's oversubscribes a CPU quota (spinning workers: 60-80 ms stalls were measured)
                # (edge: the chief rays stay with the fast kernel, only the main points are needed in the reference's arithmetic)
                if vec is not None:                      # two calls into the library, no interpreter lock held
                    _pupil_rows(vec, u, st.off_main, st.off_main + spp, spp, enp_rr, enp_z, hp[st.n_pf:st.n_pf + st.n_pm])
                    if not edge:
                        _pupil_rows(vec, u, st.off_chief, st.off_chief + GEO_SPP, GEO_SPP, enp_rr * 0.5, enp_z, hp[st.n_pf + st.n_pm:])
                    return
                pm_h = hp[st.n_pf:st.n_pf + st.n_pm].view(S, L, spp, 3)
                pc_h = hp[st.n_pf + st.n_pm:].view(S, L, GEO_SPP, 3)
                for k in range(S):
                    pm_h[k].copy_(_pupil_points(um[k, :, 0], um[k, :, 1], enp_rr, enp_z))
                    if not edge:
                        pc_h[k].copy_(_pupil_points(uc[k, :, 0], uc[k, :, 1], enp_rr * 0.5, enp_z))

            pupils_ready = _WORKER.submit(psf_pupils)
        if phase == "psf":                                   # psf_map at the lens's current state (the reference reads self.d_sensor / self.hfov)
            hs0 = lens._state_sync()
            d_sensor, hfov, foclen, fnum = [float(hs0.d_sensor)] * S, [float(hs0.hfov)] * S, [float(hs0.foclen)] * S, [float(hs0.fnum)] * S
        got = None
        # ---- levels 1 and 2 through the native host driver (csrc/stack_host.cpp): what the Python below does between two waits, as one
        # call each; any status != 0 (a batch no candidate row confirms, a NaN residual) falls through to the Python form of that level
        native = fused and phase == "all" and vec is not None and _HostNative.on
        l1_done = l2_done = prov_done = False
        if native:
            nl = st.native_levels(lens, counts, keys, curved, tab_dev, len(wv), n_surf, S)
            lv, lib = nl["lv"], _abi.load_library()
            nl["focus"][:] = focus
            if prov and os.environ.get("AADFF_EDGE_PROV_FIRST", "1") != "0":
                # the provisional pass needs nothing but the draws: queued FIRST, the 0.36 ms of fast refocus + PSF kernel start while this
                # thread is still in MKL for the focus rays' aperture points (level 1 then runs beside them on its high-priority stream)
                provisional_pass()
                prov_done = True
            R2_first = float(torch.ones((), dtype=f32) * s0.r ** 2)
            prev = vec[4](1) if vec[4] is not None else None                      # MKL: one thread for this thread's calls (see _pupil_rows)
            try:
                rc = lib.aadff_levels_focus_submit(C.byref(lv), C.c_void_p(u.data_ptr()), C.c_void_p(st.off_focus.ctypes.data), float(np.float32(np.pi)), R2_first,
                                                   float(np.float32(s0.d.item())), C.c_void_p(nl["focus"].ctypes.data), vec[0], vec[1], vec[2], vec[3], sp12,
                                                   C.c_void_p(nl["events"][0].cuda_event))
            finally:
                if prev is not None:
                    vec[4](prev)
            if rc != 0:
                raise RuntimeError("aadff_levels_focus_submit failed: " + lib.aadff_last_error().decode(errors="replace"))
            mark("level 1 rays")
            if prov and not prov_done:
                provisional_pass()
                prov_done = True
            yield nl["events"][0]
            counts.stats["fused"] += 1
            status = lib.aadff_levels_focus_finish(C.byref(lv), C.c_void_p(nl["chosen"].ctypes.data), C.c_void_p(nl["dsens"].ctypes.data), C.c_void_p(nl["scratch"].ctypes.data))
            if status == 0:
                d_sensor = [float(v) for v in nl["dsens"]]
                if nl["checked"] < _HostFast.CHECKS:                               # the driver's mean against numpy's own, on its first uses
                    nl["checked"] += 1
                    J1, r1 = lv.J1, st.result(0)
                    rows_v = np.stack([r1[j * GEO_SPP:(j + 1) * GEO_SPP].view(np.float32) for j in nl["chosen"]])
                    rows_w = np.stack([r1[(J1 + j) * GEO_SPP:(J1 + j + 1) * GEO_SPP].view(np.float32) for j in nl["chosen"]])
                    want = _d_sensor_loop(rows_v, rows_w > 0)
                    if not all((a == b) or (a != a and b != b) for a, b in zip(d_sensor, want)):
                        _HostNative.switch_off("np.mean of the focus distances")
                        d_sensor = want
                for z in d_sensor:
                    assert z > 0, "sensor position is negative."
                if not np.array_equal(nl["chosen"], nl["first"][0]):                # a batch took an alternative row: it becomes the prediction
                    counts.learn(keys[0], np.stack([nl["cand"][0][j][1] for j in nl["chosen"]]))
                l1_done = True
                counts.stats["native"] = counts.stats.get("native", 0) + 1
            else:
                counts.stats["native_fallbacks"] = counts.stats.get("native_fallbacks", 0) + 1
        if fused and phase != "psf" and not l1_done:
            # every pupil point of the stack comes from the reference's host calls; the focus ones ride in front of level 1
            if vec is not None:
                _pupil_rows(vec, u, st.off_focus, st.off_focus + GEO_SPP, GEO_SPP, s0.r, s0.d.item(), hp[:st.n_pf])
            else:
                hp[:st.n_pf].view(S, GEO_SPP, 3).copy_(_pupil_points(uf[:, 0], uf[:, 1], s0.r, s0.d.item()))
            with torch.cuda.stream(s12):
                st.d_pupil[:st.n_pf].copy_(hp[:st.n_pf], non_blocking=True)
            mark("level 1 rays")
            # ---- level 1: refocus (deeplens/optics.py:1155-1180) - rays from the first surface's aperture points away from (0, 0, focus)
            t = st.h_par[0][:S * 3].view(f32).view(S, 3)
            t.zero_()
            t[:, 2] = torch.tensor([float(f) for f in focus], dtype=f32)

            def launch1(J, par, res):
                G = st.G[0]
                _abi.call("aadff_trace_rays_strict_fused", None, None, None, GEO_SPP, J, _abi.ptr(tab_dev), len(wv), n_surf, _abi.ptr(st.bt_green),
                          _ptr_at(par, 0), _ptr_at(par, G), _abi.ptr(st.d_pupil), 1, 0, n_surf, 1, None, _ptr_at(par, G + st.J),
                          _ptr_at(res, 2 * J * GEO_SPP), 1, 1, _ptr_at(res, 0), _ptr_at(res, J * GEO_SPP), _ptr_at(par, G), sp12)

            got = yield from _speculate_small(counts, keys[0], st, 0, S, GEO_SPP, curved, fwd_order, launch1, s12,
                                              after_first_submit=provisional_pass if prov and not prov_done else None)
            if got is not None:
                fd_all, alive, w_focus = got[0], got[1] > 0, got[1]
        bt_green = st.bt_green if st is not None else torch.full((B,), t_green, dtype=torch.int32, device=dev)
        if phase != "psf" and (not fused or got is None) and not l1_done:
            if fused:
                counts.stats["per_surface_replays"] += 1
            fd_all, alive, cnt = _level1_batched(lens, uf, focus, S, tabs, len(wv), n_surf, bt_green[:S], dev)
            w_focus = None
            counts.learn(keys[0], cnt)
        mark("level 1 back on the host")
        if phase != "psf" and not l1_done:
            d_sensor = _d_sensor_of(fd_all, alive, w_focus)
        mark("d_sensor")
        if native and l1_done:
            # ---- level 2, native: sensor corners into the block, launch, wait, the confirmed jobs' tangents back
            dsa = np.asarray(d_sensor, dtype=np.float32)
            backward = not bool(nl["o2z"] - torch.tensor(d_sensor[0], dtype=f32) > 0)
            rc = lib.aadff_levels_fov_submit(C.byref(lv), C.c_void_p(dsa.ctypes.data), float(np.float32(lens.r_last)), int(not backward), sp12,
                                             C.c_void_p(nl["events"][1].cuda_event))
            if rc != 0:
                raise RuntimeError("aadff_levels_fov_submit failed: " + lib.aadff_last_error().decode(errors="replace"))
            mark("level 2 rays")
            yield nl["events"][1]
            counts.stats["fused"] += 1
            status = lib.aadff_levels_fov_finish(C.byref(lv), C.c_void_p(nl["chosen"].ctypes.data), C.c_void_p(nl["tan"].ctypes.data), C.c_void_p(nl["ra"].ctypes.data))
            if status == 0:
                tan_fov, rra = torch.from_numpy(nl["tan"]), torch.from_numpy(nl["ra"])
                if not np.array_equal(nl["chosen"], nl["first"][1]):
                    counts.learn(keys[1], np.stack([nl["cand"][1][j][1] for j in nl["chosen"]]))
                l2_done = True
            else:
                counts.stats["native_fallbacks"] = counts.stats.get("native_fallbacks", 0) + 1
        # ---- level 2: calc_fov (deeplens/optics.py:1187-1217) - S batches of 100 rays from the sensor corner, backward
        if fused and phase != "psf" and not l2_done:
            o1, o2 = _fov_geometry(lens, d_sensor)
            M = o2.shape[0]
            backward = not bool(o2[0, 2] - o1[0, 2] > 0)
            st.h_par[1][:S * 3].view(f32).view(S, 3).copy_(o1)
            st.h_par[1][S * 3:S * 3 + M * 3].view(f32).view(M, 3).copy_(o2)

            def launch2(J, par, res):
                G = st.G[1]
                _abi.call("aadff_trace_rays_strict_fused", None, None, None, M, J, _abi.ptr(tab_dev), len(wv), n_surf, _abi.ptr(st.bt_green),
                          _ptr_at(par, 0), _ptr_at(par, G), _ptr_at(par, S * 3), 1, 0, n_surf, int(not backward), None, _ptr_at(par, G + st.J),
                          _ptr_at(res, 2 * J * M), 0, 2, _ptr_at(res, 0), _ptr_at(res, J * M), _abi.ptr(st.zeros), sp12)

            mark("level 2 rays")
            got = yield from _speculate_small(counts, keys[1], st, 1, S, M, curved, bwd_order if backward else fwd_order, launch2, s12)
            if got is not None:
                tan_fov, rra = torch.from_numpy(got[0]), torch.from_numpy(got[1])
        elif not l2_done:
            mark("level 2 rays")
        if phase != "psf" and (not fused or got is None) and not l2_done:
            if fused:
                counts.stats["per_surface_replays"] += 1
            tan_fov, rra, cnt = _level2_batched(lens, d_sensor, S, tabs, len(wv), n_surf, bt_green[:S], dev)
            counts.learn(keys[1], cnt)
        mark("level 2 back on the host")
        if phase != "psf":
            hfov, foclen, fnum = _fov_of(lens, tan_fov, rra)
        mark("hfov")
        if phase == "focus":                                 # refocus + calc_fov of the per-call API: the state is set, no PSFs
            lens._state_sync()
            hs = lens._state_host
            hs.d_sensor, hs.hfov, hs.tan_hfov = float(d_sensor[-1]), float(hfov[-1]), float(np.tan(hfov[-1]))
            hs.foclen, hs.fnum = float(foclen[-1]), float(fnum[-1])
            if lens._state_dev is not None:
                lens._state_upload()
            return None
        # ---- level 3: psf_map (deeplens/optics.py:888-1026) - per slice and wavelength spp x N main rays and 2048 x N chief rays
        edge_native = bool(edge and native and prov and prov_done and l1_done and l2_done)
        # [S,N,3] (the native edge driver computes them itself; the Python form is then only its check on the first stacks)
        pobj = _object_points(lens, pts, hfov) if not edge_native or eb.get("pobj_checked", 0) < _HostFast.CHECKS else None
        edge_done = False
        if edge:
            # ---- level 3, edge-exact: fast kernel (already running on provisional states, or launched here on the exact ones) + strict
            # re-trace of the rays at the window edge + normalisation, no host wait in between
            pred3 = counts.rows[keys[2]]
            h = st.h_par[2]
            hn = h.numpy()
            if eb.get("pred_ref") is not pred3:                 # the table's rows are replaced, never edited: same object = same counts
                hn[B + S * N * 3:B + S * N * 3 + B * 2 * MS].reshape(B, 2, MS)[:] = pred3
                eb["pred_ref"] = pred3
            par = st.d_par[2]
            if edge_native:
                # object points, uploads, re-trace, normalisation, counts back: one call (csrc/stack_host.cpp: aadff_edge_finish)
                pupils_ready.result()
                pn = eb.get("pts_np")
                if pn is None or eb.get("pts_np_key") != (float(depth_plane_mm), grid):
                    pn = eb["pts_np"] = np.ascontiguousarray(pts.numpy(), dtype=np.float32)
                    eb["pts_np_key"] = (float(depth_plane_mm), grid)
                hf64, ds32 = np.asarray(hfov, dtype=np.float64), np.asarray(d_sensor, dtype=np.float32)
                rc = lib.aadff_edge_finish(C.byref(edge_struct()), C.c_void_p(pn.ctypes.data), C.c_void_p(hf64.ctypes.data), C.c_void_p(ds32.ctypes.data),
                                           float(np.float32(lens.r_last)), float(np.float32(lens.sensor_size[1])), float(np.float32(lens.sensor_size[0])),
                                           _abi.ptr(centre), _abi.ptr(maps), sp, C.c_void_p(eb["uploaded"].cuda_event), C.c_void_p(eb["done"].cuda_event))
                if rc != 0:
                    raise RuntimeError("aadff_edge_finish failed: " + lib.aadff_last_error().decode(errors="replace"))
                eb["busy"] = True
                if pobj is not None:                                               # the driver's object points against the reference's tensor operations
                    eb["pobj_checked"] = eb.get("pobj_checked", 0) + 1
                    if not torch.equal(h[B:B + S * N * 3].view(f32).view(S, N, 3).view(torch.int32), pobj.view(torch.int32)):
                        _HostNative.switch_off("psf_diff's object points")
                mark("level 3 inputs")
            else:
                hn[:B].view(np.float32)[:] = np.repeat(np.asarray(d_sensor, dtype=np.float32), L)
                h[B:B + S * N * 3].view(f32).view(S, N, 3).copy_(pobj)
            if edge_native:
                pass
            elif prov:
                eb["h_focus"][S:] = torch.tensor([float(np.tan(v)) for v in hfov], dtype=torch.float64).to(f32)
                eb["d_focus"][S:].copy_(eb["h_focus"][S:], non_blocking=True)
            else:
                # the S lens states as the fast kernel reads them (aadff_lens_state_t: 5 floats, 3 ints per state)
                sw = hin.numpy()[:nst].reshape(S, nst // S)
                sf = sw.view(np.float32)
                sf[:, 0], sf[:, 1], sf[:, 2] = d_sensor, hfov, [float(np.tan(v)) for v in hfov]
                sf[:, 3], sf[:, 4] = foclen, fnum
                sw[:, 5], sw[:, 6], sw[:, 7] = GEO_SPP, 0, 0
                eb["d_in"][:nst].copy_(hin[:nst], non_blocking=True)
                du.copy_(eb["h_u"], non_blocking=True)
                cnt[B:].zero_()
            if not edge_native:
                pupils_ready.result()
                par.copy_(h, non_blocking=True)
                st.d_pupil[st.n_pf:st.n_pf + st.n_pm].copy_(hp[st.n_pf:st.n_pf + st.n_pm], non_blocking=True)
                eb["uploaded"].record(stream)
                eb["busy"] = True
                mark("level 3 inputs")
                if not prov:
                    launch_edge(_ptr_at(eb["d_in"], 0), None)
                _abi.call("aadff_strict_edge_retrace", _ptr_at(par, B), N, B, _abi.ptr(st.pset), _abi.ptr(tab_dev), len(wv), n_surf, _abi.ptr(st.bt_main),
                          _ptr_at(par, 0), _ptr_at(st.d_pupil, st.n_pf), spp, _ptr_at(par, B + S * N * 3), float(lens.pixel_size), ks, _abi.ptr(centre),
                          _ptr_at(cnt, 0), _abi.ptr(eb["list"]), EDGE_CAP, _abi.ptr(eb["raw"]), _ptr_at(cnt, B),
                          _abi.ptr(eb["states_prov"]) if prov else None, _ptr_at(eb["d_focus"], S) if prov else None, _abi.ptr(eb["slope"]) if prov else None, sp)
                _abi.call("aadff_psf_normalise", _abi.ptr(eb["raw"]), S, N, L, float(lens.pixel_size), ks, 1, _abi.ptr(maps), sp)
                eb["h_back"].copy_(cnt, non_blocking=True)
                eb["done"].record(stream)
            yield eb["done"]
            back = eb["h_back"].numpy()
            bits = int(back[B])
            counts.stats["edge"] = counts.stats.get("edge", 0) + 1
            counts.stats["edge_rays"] = counts.stats.get("edge_rays", 0) + int(back[:B].sum())
            lens._edge_last = {"rays": back[:B].copy(), "flags": bits}
            mark("level 3 (edge) waited for")
            if bits & 1:
                raise FloatingPointError("found nan in ft in non-diff newton method.")
            assert not bits & 2, "No sampled rays is valid."
            if bits & 16:                                       # a list overflowed (a caustic along the window edge): the strict psf_map decides every ray
                counts.stats["edge_overflows"] = counts.stats.get("edge_overflows", 0) + 1
                if pobj is None:
                    pobj = _object_points(lens, pts, hfov)
                if vec is not None:
                    _pupil_rows(vec, u, st.off_chief, st.off_chief + GEO_SPP, GEO_SPP, enp_rr * 0.5, enp_z, hp[st.n_pf + st.n_pm:])
                else:
                    hp[st.n_pf + st.n_pm:].view(S, L, GEO_SPP, 3).copy_(_pupil_points(uc[:, :, 0], uc[:, :, 1], enp_rr * 0.5, enp_z))
                pupils_ready = concurrent.futures.Future()
                pupils_ready.set_result(None)
            else:
                edge_done = True
                any_valid = torch.ones(B, dtype=torch.int32)
        if edge_done:
            pass
        elif fused:
            pred3 = counts.rows[keys[2]]
            maps = torch.empty((S, L, grid * ks, grid * ks), dtype=f32, device=dev)
            centre = torch.empty((B, N, 2), dtype=f32, device=dev)
            # two-variant jobs (aadff_strict_psf_points_alt): a batch whose CHIEF count at exactly one surface has been seen at two
            # neighbouring values n, n + 1 (and nothing else undecided in its chief row) is rendered under both in this launch
            # (not inside a StrictPipeline: there the re-launch runs beside the next stack's psf_map launch for free, while the second
            # variant makes every launch 4.6 % longer - measured 3.63 against 3.97 ms per stack at depth 3)
            alt = _two_variant_words(counts.seen[keys[2]][:, 0], curved) if ALT_JOBS and getattr(lens, "_strict_l3_chain", None) is None else None
            if alt is not None and (alt >= 0).any():
                pred3 = pred3.copy()
                ab = np.nonzero(alt >= 0)[0]
                pred3[ab, 0, alt[ab] & 0xff] = (alt[ab] >> 8) + 1
                ak = ("alt", grid, ks)
                if not hasattr(st, "_alt_out"):
                    st._alt_out = {}
                if ak not in st._alt_out:
                    st._alt_out[ak] = (torch.empty((B, grid * ks, grid * ks), dtype=f32, device=dev), torch.empty((B, N, 2), dtype=f32, device=dev))
                maps_alt, centre_alt = st._alt_out[ak]
                counts.stats["alt_jobs"] = counts.stats.get("alt_jobs", 0) + len(ab)
            else:
                alt = None
            h = st.h_par[2]
            h[:B].view(f32).copy_(torch.tensor(d_sensor, dtype=f32).repeat_interleave(L))
            h[B:B + S * N * 3].view(f32).view(S, N, 3).copy_(pobj)
            h[B + S * N * 3:B + S * N * 3 + B * 2 * MS].view(B, 2, MS).copy_(torch.from_numpy(pred3))
            if alt is not None:
                h[B + S * N * 3 + B * 2 * MS:].copy_(torch.from_numpy(alt))
            n_up3 = h.numel() if alt is not None else h.numel() - B

            def launch3(J, jobs_ptr, pred_ptr, res, on=None, alt_ptr=None):
                # re-launches go to the short levels' stream: behind a pipeline's NEXT psf_map launch they would wait 3 ms (the host
                # has seen the first launch finish, and waits for the re-launch before the convolution is queued: ordered either way)
                par = st.d_par[2]
                if alt_ptr is None:
                    _abi.call("aadff_strict_psf_points", _ptr_at(par, B), N, J, jobs_ptr, _abi.ptr(st.pset), _abi.ptr(tab_dev), len(wv), n_surf,
                              _abi.ptr(st.bt_main), _abi.ptr(st.bt_green), _ptr_at(par, 0), _ptr_at(st.d_pupil, st.n_pf), spp,
                              _ptr_at(st.d_pupil, st.n_pf + st.n_pm), GEO_SPP, pred_ptr, float(lens.pixel_size), ks, grid, _abi.ptr(maps), _abi.ptr(centre),
                              _ptr_at(res, 0), _ptr_at(res, J * 4 * MS), sp if on is None else on)
                else:
                    _abi.call("aadff_strict_psf_points_alt", _ptr_at(par, B), N, J, jobs_ptr, _abi.ptr(st.pset), _abi.ptr(tab_dev), len(wv), n_surf,
                              _abi.ptr(st.bt_main), _abi.ptr(st.bt_green), _ptr_at(par, 0), _ptr_at(st.d_pupil, st.n_pf), spp,
                              _ptr_at(st.d_pupil, st.n_pf + st.n_pm), GEO_SPP, pred_ptr, float(lens.pixel_size), ks, grid, _abi.ptr(maps), _abi.ptr(centre),
                              _ptr_at(res, 0), _ptr_at(res, J * 4 * MS), alt_ptr, _abi.ptr(maps_alt), _abi.ptr(centre_alt),
                              _ptr_at(res, J * 4 * MS + J), _ptr_at(res, J * 4 * MS + J + J * 2 * MS), sp if on is None else on)

            pupils_ready.result()
            st.d_pupil[st.n_pf:].copy_(hp[st.n_pf:], non_blocking=True)
            mark("level 3 inputs")
            chain = getattr(lens, "_strict_l3_chain", None)
            if chain is not None and chain[0] is not None:
                # StrictPipeline: behind the psf_map launch of the stack in front, not beside it - two such launches sharing the chip
                # finish together, and the host would learn the first one's counts 3 ms later
                stream.wait_event(chain[0])
            n_dn3 = B * 4 * MS + B + (B * 2 * MS + B if alt is not None else 0)
            ev3 = st.submit(2, n_up3, n_dn3, lambda par, res: launch3(B, None, _ptr_at(par, B + S * N * 3), res,
                                                                          alt_ptr=None if alt is None else _ptr_at(par, B + S * N * 3 + B * 2 * MS)), stream)
            if chain is not None:
                chain[0] = ev3
            yield ev3
            r = st.result(2)
            mark("level 3 launch waited for")
            counts.stats["fused"] += 1
            hb = r[:B * 4 * MS].view(np.uint32).reshape(B, 2, 2, MS)
            any_valid = torch.from_numpy(r[B * 4 * MS:B * 4 * MS + B].copy())
            ok2, fix = check_counts(hb[:, :, 0], pred3, curved, fwd_order)                        # [B,2]: chief and main of the batch
            ok = ok2.all(-1)
            if alt is not None:
                # a two-variant batch whose run under n + 1 shows that the loop stops at n: the variant under n is the truth if ITS bits
                # (bits_alt: the unchanged ones in front of the undecided surface, its own behind) confirm the rest of its row
                o2 = B * 4 * MS + B
                hb_alt = r[o2:o2 + B * 2 * MS].view(np.uint32).reshape(B, 2, MS)
                av_alt = r[o2 + B * 2 * MS:o2 + B * 2 * MS + B]
                counts.stats["alt_retraced_max"] = max(counts.stats.get("alt_retraced_max", 0), int(hb_alt[ab, 0, MS - 1].max()))
                sa, nl_ = alt[ab] & 0xff, alt[ab] >> 8
                cand = ~ok2[ab, 0] & ok2[ab, 1] & (fix[ab, 0, sa] == nl_) & (av_alt[ab] >= 0)
                if cand.any():
                    cb = ab[cand]
                    rows_lo = pred3[cb, 0].copy()
                    rows_lo[np.arange(len(cb)), alt[cb] & 0xff] = alt[cb] >> 8
                    ok_lo, _ = check_counts(hb_alt[cb, 0], rows_lo, curved, fwd_order)
                    take = cb[ok_lo]
                    if len(take):
                        if _nan_in_run(hb_alt[take, 1][:, None], rows_lo[ok_lo][:, None], curved):
                            raise FloatingPointError("found nan in ft in non-diff newton method.")
                        with torch.cuda.stream(stream):
                            for b_ in take:                  # (a few 70 KB device copies in front of the convolution; the centres stay behind:
                                maps.view(B, grid * ks, grid * ks)[int(b_)].copy_(maps_alt[int(b_)], non_blocking=True)   # nothing reads them)
                        pred3[take, 0] = rows_lo[ok_lo]
                        hb = hb.copy()
                        hb[take, 0, 0] = hb_alt[take, 0]
                        hb[take, 0, 1] = hb_alt[take, 1]
                        any_valid[torch.from_numpy(take)] = torch.from_numpy((av_alt[take] > 0).astype(np.int32))
                        ok = ok.copy()
                        ok[take] = True
                        counts.stats["alt_taken"] = counts.stats.get("alt_taken", 0) + len(take)
            if _nan_in_run(hb[ok][:, :, 1], pred3[ok], curved):
                raise FloatingPointError("found nan in ft in non-diff newton method.")
            truth = pred3.copy()
            bad = np.nonzero(~ok)[0]
            counts.stats["replayed_batches"] += len(bad)
            rnd = 0
            while len(bad) and rnd < FUSED_ROUNDS:          # re-launch the mispredicted batches with their corrected rows
                rnd += 1
                J = len(bad)
                rows = fix[bad] if rnd == 1 else rows_next
                hr = st.h_par[3].numpy()
                hr[:J] = bad
                hr[B:B + J * 2 * MS] = rows.reshape(-1)
                yield st.submit(3, B + J * 2 * MS, J * 4 * MS + J, lambda par, res: launch3(J, _ptr_at(par, 0), _ptr_at(par, B), res, sp12), s12)
                r = st.result(3)
                counts.stats["fused_replays"] += 1
                jb = r[:J * 4 * MS].view(np.uint32).reshape(J, 2, 2, MS)
                okj2, fixj = check_counts(jb[:, :, 0], rows, curved, fwd_order)
                okj = okj2.all(-1)
                if _nan_in_run(jb[okj][:, :, 1], rows[okj], curved):
                    raise FloatingPointError("found nan in ft in non-diff newton method.")
                truth[bad[okj]] = rows[okj]
                any_valid[bad[okj]] = torch.from_numpy(r[J * 4 * MS:J * 4 * MS + J][okj].copy())
                bad, rows_next = bad[~okj], fixj[~okj]
            mark("counts checked, re-launches")
            if len(bad):                                    # still unconfirmed: the per-surface form finds the counts itself
                counts.stats["per_surface_replays"] += 1
                sel = torch.from_numpy(bad).to(dev)
                par = st.d_par[2]
                psf, cnt, av = _level3_batched(lens, sel, par[B:B + S * N * 3].view(f32).view(S, N, 3), st.pset, st.d_pupil[st.n_pf + st.n_pm:].view(B, GEO_SPP, 3),
                                               st.d_pupil[st.n_pf:st.n_pf + st.n_pm].view(B, spp, 3), par[:B].view(f32), st.bt_green[:B], st.bt_main, tabs,
                                               len(wv), n_surf, N, spp, ks, dev)
                maps.view(B, grid * ks, grid * ks)[sel] = _tile(psf, grid, ks)
                truth[bad] = cnt
                any_valid[bad] = av
            counts.learn(keys[2], truth)
        else:
            points = pobj.to(dev).contiguous()
            pm = _pupil_points(um[:, :, 0], um[:, :, 1], enp_rr, enp_z).reshape(B, spp, 3).to(dev).contiguous()
            pc = _pupil_points(uc[:, :, 0], uc[:, :, 1], enp_rr * 0.5, enp_z).reshape(B, GEO_SPP, 3).to(dev).contiguous()
            pset = torch.arange(S, dtype=torch.int32).repeat_interleave(L).to(dev)
            zs = torch.tensor(d_sensor, dtype=f32).repeat_interleave(L).to(dev)
            bt_main = torch.arange(L, dtype=torch.int32).repeat(S).to(dev)
            mark("level 3 inputs")
            psf, cnt, any_valid = _level3_batched(lens, None, points, pset, pc, pm, zs, bt_green, bt_main, tabs, len(wv), n_surf, N, spp, ks, dev)
            counts.learn(keys[2], cnt)
            counts.stats["seeded"] += 1
            maps = _tile(psf, grid, ks).reshape(S, L, grid * ks, grid * ks)
        mark("level 3 done")
        if marks is not None:
            lens._strict_timing = [("draws and setup", round((marks[0][1] - t_enter) * 1e3, 3))] + [(b[0], round((b[1] - a[1]) * 1e3, 3)) for a, b in zip(marks, marks[1:])]
    assert bool(any_valid.bool().all()), "No sampled rays is valid."
    if phase == "psf":
        return maps
    # the lens is left focused at the last distance
    lens._state_sync()
    hs = lens._state_host
    hs.d_sensor, hs.hfov, hs.tan_hfov = float(d_sensor[-1]), float(hfov[-1]), float(np.tan(hfov[-1]))
    hs.foclen, hs.fnum = float(foclen[-1]), float(fnum[-1])
    if lens._state_dev is not None:
        lens._state_upload()
    lens._strict_stack_scalars = {"d_sensor": d_sensor, "hfov": hfov}
    return maps


class StrictPipeline:
    """Strict-parity stacks interleaved on ONE host thread: `depth` strict lenses (count table, staging blocks, HIP stream each; 4 by
    default: 3.3 - 3.5 ms per stack, the psf_map launches back to back on the GPU; 3: 3.6 - 3.8; 2: 4.3 - 4.5; sequential 5.1 - 5.5) take turns, and every host wait of a stack - the round trips of its two short levels, its psf_map launch, a re-launch
    - is a point where the host goes on with whichever other stack is ready (`_strict_psf_maps_steps` yields the event it would wait
    for; the pipeline polls them, oldest stack first).  The short levels and the re-launches run on a high-priority stream so that
    they do not queue behind the 3 ms psf_map launch of another stack, psf_map launches are chained by an event (side by side they
    would finish together), re-launches use the 256-thread form.  Draws are taken at `submit`, in submission order: every stack
    is bit for bit what `render_focal_stack_m1(strict_lens, ...)` returns for it in a sequential loop (to the histograms' float
    atomics).  `submit(...)` returns a handle whose `result()` is (stack [B,C,S,H,W], event recorded behind its convolution).
    (History, DESIGN.md section 2: two host threads - slower than the sequential loop under the interpreter lock; one thread with one
    switch point per stack - 4.4 ms against 5.2 sequential; this form.)"""

    class _Pending:
        def __init__(self, pipe, k, steps):
            self.pipe, self.k, self.steps, self.value, self.event, self.error = pipe, k, steps, None, None, None

        def result(self):
            self.pipe._drive(lambda: self.steps is None)
            if self.error is not None:                       # what this stack raised (the reference's assertions, a NaN in Newton's method, ...)
                raise self.error
            return self.value

    def __init__(self, make_lens, depth=4):
        self.depth = int(depth)
        self.lenses = [make_lens() for _ in range(self.depth)]
        assert all(getattr(l, "parity", "") in ("strict", "edge") for l in self.lenses), "StrictPipeline renders strict- / edge-parity lenses"
        dev = self.lenses[0]._gpu()
        self.streams = [torch.cuda.Stream(dev) for _ in range(self.depth)]
        self.chain = [None]                                  # the event behind the newest psf_map launch, shared by the lenses
        for l in self.lenses:
            l._strict_l3_chain = self.chain
            if os.environ.get("AADFF_STRICT_PRIO", "1") != "0":
                l._strict_fast_stream = torch.cuda.Stream(dev, priority=-1)
        _abi.call("aadff_strict_replay_threads", 256)        # re-launches run beside another stack's psf_map launch
        self.pending = []                                    # oldest first
        self.turn = 0
        self.closed = False
        self.trace = [] if os.environ.get("AADFF_STRICT_PIPE_TRACE") == "1" else None       # (stack, "in" / "out" of a step, seconds)

    def _steps(self, lens, img, depth_plane_mm, focus_mm, grid, ks, spp):
        focus = [float(f) for f in np.asarray(focus_mm, dtype=np.float64).reshape(-1)]
        B, C_, H, W = img.shape
        assert tuple(lens.sensor_res) == (H, W), "lens.sensor_res must match the image"
        dev = lens._gpu()
        maps = yield from _strict_psf_maps_steps(lens, depth_plane_mm, focus, grid, ks, spp)
        maps = maps.to(dev).contiguous()
        x = _abi.f32c(img, dev)
        out = torch.empty((B, C_, len(focus), H, W), dtype=torch.float32, device=dev)
        _abi.call("aadff_render_psf_map_stack", _abi.ptr(x), _abi.ptr(maps), _abi.ptr(out), B, C_, len(focus), H, W, grid, ks, _abi.stream_ptr(dev))
        return out

    def _advance(self, p):
        """run stack p to its next wait (p.event) or to its end (p.value = (stack, event), p.steps = None)"""
        i = p.k % self.depth
        with torch.no_grad(), torch.cuda.stream(self.streams[i]):
            if self.trace is not None:
                self.trace.append((p.k, "in", time.perf_counter()))
            try:
                p.event = next(p.steps)
            except StopIteration as done:
                ev = torch.cuda.Event()
                ev.record(self.streams[i])
                p.value, p.steps, p.event = (done.value, ev), None, None
                self.pending.remove(p)
            except Exception as e:                           # belongs to THIS stack: raised by its handle, the others go on
                p.steps, p.event, p.error = None, None, e
                self.pending.remove(p)
            except BaseException:
                p.steps, p.event = None, None
                self.pending.remove(p)
                raise
            finally:
                if self.trace is not None:
                    self.trace.append((p.k, "out", time.perf_counter()))

    def _drive(self, done):
        """advance ready stacks (oldest first) until done(); when none is ready, wait for the oldest one's event"""
        while not done():
            ready = next((p for p in self.pending if p.event.query()), None)
            if ready is None:
                self.pending[0].event.synchronize()
                ready = self.pending[0]
            self._advance(ready)

    def submit(self, img, depth_plane_mm, focus_mm, grid=11, ks=11, spp=GEO_SPP):
        assert not self.closed, "StrictPipeline is closed"
        k = self.turn
        self.turn += 1
        self._drive(lambda: all(p.k % self.depth != k % self.depth for p in self.pending))      # the stack that has this lens is done
        p = StrictPipeline._Pending(self, k, self._steps(self.lenses[k % self.depth], img, depth_plane_mm, focus_mm, grid, ks, spp))
        self.pending.append(p)
        self._advance(p)                                     # draws, host arithmetic and the first launch of the stack
        while True:                                          # and whatever else is ready meanwhile
            ready = next((q for q in self.pending if q.event.query()), None)
            if ready is None:
                break
            self._advance(ready)
        return p

    def close(self):
        """finish what is pending and hand the re-launch form back to the sequential path (idempotent)"""
        if self.closed:
            return
        self.closed = True
        try:
            self._drive(lambda: not self.pending)
        finally:
            _abi.call("aadff_strict_replay_threads", 1024)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            if not self.closed:
                self.closed = True
                _abi.call("aadff_strict_replay_threads", 1024)
        except Exception:
            pass
