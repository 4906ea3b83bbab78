"""Lensgroup for the MI355X build: the reference's lens-as-PSF-generator API
(deeplens/optics.py: psf* :887-1026, sample_from_points :457, trace* :598-714, refocus
:1155, calc_fov :1187, pupils :1312-1403, read_lens_json :2045) on HIP kernels.

Design: the focus-dependent scalars (d_sensor, hfov, foclen, fnum) live in
a 32-byte `aadff_lens_state_t` ON THE DEVICE; refocus / post_computation write it with a
kernel and every PSF kernel reads it, so a whole focal stack runs without a host round
trip.  The Python attributes of the same names are properties that read it back lazily.
Entrance/exit pupils depend only on the surfaces in front of / behind the stop and are
cached per lens geometry.  Pupil samples come from the HOST torch generator in the
reference's call order so results are comparable sample for sample.
"""
import ctypes as C
import json
import os
import logging
import random                    # noqa: F401  (the next five names ride the star-import chain the reference scripts rely on:
from datetime import datetime    # noqa: F401   `from deeplens.psfnet import *` must provide torch, nn, np, plt, tqdm,
                                 #              save_image, make_grid ... as deeplens/optics.py:5-20 does)
import numpy as np
import torch
import torch.nn.functional as F
from scipy import stats
from tqdm import tqdm            # noqa: F401

class _LazyPyplot:
    """`plt` of the star surface (deeplens/optics.py:13 imports matplotlib.pyplot at module level): the name is there, the
    import - 0.6 s, and a font-cache build of many seconds on a fresh machine - happens at the first attribute access."""
    _mod = None

    def _load(self):
        if _LazyPyplot._mod is None:
            import matplotlib
            matplotlib.use("Agg", force=False)
            import matplotlib.pyplot as mod
            _LazyPyplot._mod = mod
        return _LazyPyplot._mod

    def __getattr__(self, name):
        return getattr(self._load(), name)


try:                             # plotting is out of scope, the NAME is part of the star surface
    import importlib.util as _ilu
    plt = _LazyPyplot() if _ilu.find_spec("matplotlib") is not None else None
except Exception:                # pragma: no cover - matplotlib is present in the image
    plt = None

from aadff import _abi
from aadff.sampling import HostSampler
from .basics import *            # noqa: F401,F403  (star re-exports are part of the API surface)
from .basics import DEFAULT_WAVE, DEPTH, DEVICE, EPSILON, GEO_SPP, WAVE_RGB, DeepObj, Material, Ray
from .monte_carlo import *       # noqa: F401,F403
from .render_psf import *        # noqa: F401,F403
from .render_psf import render_psf_map
from .surfaces import *          # noqa: F401,F403
from .surfaces import Aspheric, pack_table, trace_ray_object
from .utils import *             # noqa: F401,F403
from .utils import make_grid, save_image   # noqa: F401


class _CallRing:
    """Pinned host blocks and their device twins for the per-call API (refocus / psf / psf_rgb / psf_map): the host draws of a call go
    into the next pinned block (one fast MT19937 fill instead of a torch.rand per row), one asynchronous copy takes them to the GPU
    and the kernel reads the rows through strides - no per-row tensors, no `.to(device)` staging copies, no synchronisation.  A block
    is handed out again only after the launch that last used it has completed (an event per block, recorded behind the launch)."""
    SLOTS = 16

    def __init__(self, dev, words):
        self.words = (int(words) + 3) // 4 * 4               # 16-byte aligned blocks (the staged upload copies float4s)
        self.host = torch.empty((self.SLOTS, self.words), dtype=torch.float32, pin_memory=True)
        self.dev = torch.empty((self.SLOTS, self.words), dtype=torch.float32, device=dev)
        self.events = [torch.cuda.Event() for _ in range(self.SLOTS)]
        self.used = [False] * self.SLOTS
        self.turn = 0
        # No copy in front of a launch where the kernel can fetch the block itself (round 5: a copy and the dispatch gap behind it
        # were 15 us of every refocus / psf_map call, a fifth of the reference's slice loop): `mapped` = the address at which the
        # GPU sees the pinned blocks (refocus reads its 16 KB over PCIe; psf launches stage theirs into `dev` with their first
        # workgroups: aadff_psf_points_staged), a completion counter per block for those.
        self.mapped = None
        if os.environ.get("AADFF_CALL_ZERO_COPY", "1") != "0":
            out = C.c_void_p()
            try:
                _abi.call("aadff_host_device_pointer", C.c_void_p(self.host.data_ptr()), C.byref(out))
                self.mapped = out.value
            except RuntimeError:
                self.mapped = None
        self.counters = torch.zeros(self.SLOTS, dtype=torch.int32, device=dev)
        self.gen = [0] * self.SLOTS
        self.shape = [None] * self.SLOTS

    def mapped_ptr(self, k):
        """device-visible address of pinned block k"""
        return self.mapped + 4 * self.words * k

    def stage(self, k, n_u, shape):
        """aadff_stage_t for block k: the launch's first workgroups copy its first n_u words (a multiple of 4) from the pinned block to
        the device block and count themselves into the block's counter; a launch of another shape has another number of copy
        workgroups, so the counter restarts then."""
        if self.shape[k] != shape:
            self.counters[k:k + 1].zero_()
            self.gen[k], self.shape[k] = 0, shape
        self.gen[k] += 1
        return _abi.Stage(self.host.data_ptr() + 4 * self.words * k, self.dev.data_ptr() + 4 * self.words * k, n_u, 0, self.gen[k] & 0xFFFFFFFF,
                          self.counters.data_ptr() + 4 * k)

    def next(self):
        k = self.turn % self.SLOTS
        self.turn += 1
        if self.used[k]:
            self.events[k].synchronize()
        return k, self.host[k], self.dev[k]

    def sent(self, k, n, stream):
        """queue the upload of the first n words of block k"""
        self.dev[k, :n].copy_(self.host[k, :n], non_blocking=True)

    def launched(self, k, stream):
        self.events[k].record(stream)
        self.used[k] = True

    def drain(self):
        """wait for every launch that still reads a block of this ring (before the ring is dropped: with the zero-copy path the
        kernels read the PINNED block itself, which no copy_ - hence no allocator event - protects from being handed out again)"""
        for k in range(self.SLOTS):
            if self.used[k]:
                self.events[k].synchronize()
                self.used[k] = False


def raise_psf_flags(bits):
    """Flags word of aadff_psf_points -> the reference's errors (bit 0: NaN in a Newton residual, surfaces.py:555-558;
    bit 2: a focus state without a positive sensor position, i.e. a refocus that found no valid ray, optics.py:1176;
    bit 1: no valid chief ray for some point, optics.py:901; bit 3: a staged upload was late and the kernel read the
    samples over PCIe instead — results are correct, only the overlap was lost, so this one is a warning)."""
    if bits & 1:
        raise FloatingPointError("found nan in ft in non-diff newton method.")
    assert not bits & 4, "sensor position is negative."
    assert not bits & 2, "No sampled rays is valid."
    if bits & 8:
        import warnings
        warnings.warn("aadff: a staged upload of pupil samples arrived late; those PSF workgroups read their samples from "
                      "pinned host memory (correct, slower)", RuntimeWarning, stacklevel=2)


_LIVE_LENSES = None


def _track_lens(lens):
    """Deferred errors must not vanish with the process: the per-call API is asynchronous (`Lensgroup.check_flags`), so a script
    whose LAST call is a psf / psf_map and which never reads the lens state would otherwise end with NaN PSFs and no message.
    At interpreter exit every live lens with kernels on record is checked once; what the reference would have raised inside the
    call is written to stderr (an exception cannot stop a script that has already ended)."""
    global _LIVE_LENSES
    if _LIVE_LENSES is None:
        import atexit
        import weakref
        _LIVE_LENSES = weakref.WeakSet()

        def _report():
            import sys
            for l in list(_LIVE_LENSES):
                try:
                    if l._state_dev is not None and l._psf_calls:
                        l.check_flags()
                except (AssertionError, FloatingPointError) as e:
                    print(f"aadff: an earlier asynchronous call on lens {getattr(l, 'lens_name', '?')!r} failed and was never checked: {e}",
                          file=sys.stderr, flush=True)
                except Exception:
                    pass                                     # the GPU runtime may already be shutting down

        atexit.register(_report)
    _LIVE_LENSES.add(lens)


class Lensgroup(DeepObj):
    """`parity="strict"` (keyword beyond the reference's signature; default "fast"): every ray trace runs in the reference's own
    float32 operation order on the GPU (`aadff_trace_rays_strict`, one launch pair per surface, batch-wide Newton counts),
    and everything the reference computes on the host with torch / numpy - pupil sampling (sin, cos, sqrt), ray
    normalisation, the focus-distance mean, the field-of-view sum and arctangent, the chief-ray centroid - is computed on the
    host with the same torch / numpy calls.  What that buys, what it cannot, and why it is not the default: DESIGN.md section 2.
    0.24 s per 10-slice 1024^2 stack against 0.34 ms for the fused kernels: a verification mode (round 5: 5 ms, aadff/strict_stack.py).
    `parity="edge"` (round 6): the strict lens with ONE difference - `psf_map` / a focal stack trace their 7.4 M PSF rays with the
    fast kernel and re-trace in the strict arithmetic only the few hundred rays per slice that land within 2e-4 mm of the
    histogram's window edge (deeplens/monte_carlo.py:37), the one place where the last bit of a hit decides anything; refocus and
    calc_fov (d_sensor, hfov) are the strict ones, every other call (psf, trace, pupils) behaves as with "strict"."""

    def __init__(self, filename=None, sensor_res=(1024, 1024), use_roc=False, post_computation=True, device=DEVICE, parity="fast"):
        assert parity in ("fast", "strict", "edge"), "parity is 'fast', 'strict' or 'edge'"
        self.parity = parity
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        self.sampler = HostSampler()
        self._table_cache = {}
        self._pupil_cache = {}
        self._state_dev = None
        self._state_host = _abi.LensState()
        self._state_stale = False
        self._ring = None
        self._flags_mirror = None
        self._psf_calls = 0
        self.sync_flags = os.environ.get("AADFF_SYNC_FLAGS", "0") == "1"      # True: every psf call waits for its flags (the round-4 behaviour)
        self.surfaces, self.materials = [], []
        self.sensor_res = sensor_res
        _track_lens(self)
        if filename is not None:
            self.lens_name = filename
            self.load_file(filename, use_roc, sensor_res, post_computation)

    # ------------------------------------------------------------------ loading
    def load_file(self, filename, use_roc, sensor_res, post_computation=True):
        if filename[-5:] == ".json":
            self.read_lens_json(filename)
        else:
            raise Exception("File format not supported.")      # reference: optics.py:135
        self.find_aperture()
        self.prepare_sensor(sensor_res)
        if post_computation:
            self.post_computation()

    def read_lens_json(self, filename="./test.json"):
        """reference: optics.py:2045-2070 — every surface type becomes an `Aspheric`."""
        self.surfaces, self.materials = [], []
        with open(filename, "r") as f:
            data = json.load(f)
        for sd in data["surfaces"]:
            kw = dict(r=sd["r"], d=sd["d"], c=sd["c"], mat1=sd["mat1"], mat2=sd["mat2"], device="cpu")
            if sd["type"] == "Aspheric":
                s = Aspheric(k=sd["k"], ai=sd["ai"], **kw)
            elif sd["type"] in ("Stop", "Spheric"):
                s = Aspheric(**kw)
            else:
                raise Exception("Surface type not implemented.")
            self.surfaces.append(s)
            self.materials.append(Material(sd["mat1"]))
        self.materials.append(Material(sd["mat2"]))
        self.r_last = data["r_last"]
        self.d_sensor = data["d_sensor"]
        self.invalidate()

    def write_lens_json(self, filename="./test.json"):
        data = {"foclen": self.foclen, "fnum": self.fnum, "r_last": self.r_last, "d_sensor": self.d_sensor,
                "sensor_size": [float(v) for v in self.sensor_size], "surfaces": []}
        for i, s in enumerate(self.surfaces):
            sd = s.surf_dict()
            nxt = self.surfaces[i + 1].d.item() if i < len(self.surfaces) - 1 else self.d_sensor
            sd["d_next"] = nxt - s.d.item()
            data["surfaces"].append(sd)
        with open(filename, "w") as f:
            json.dump(data, f, indent=4)

    def to(self, device=DEVICE):
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        self._state_sync()
        self._state_dev = None
        self._table_cache.clear()
        return self

    def invalidate(self):
        """Call after editing surface parameters in place: drops packed tables and pupils."""
        self._table_cache.clear()
        self._pupil_cache.clear()

    def find_aperture(self):
        """First surface with air on both sides (reference: optics.py:190-198)."""
        self.aper_idx = None
        for i in range(len(self.surfaces) - 1):
            if self.surfaces[i].mat1.A < 1.0003 and self.surfaces[i].mat2.A < 1.0003:
                self.aper_idx = i
                return

    def prepare_sensor(self, sensor_res=[512, 512], sensor_size=None):
        """reference: optics.py:153-175."""
        sensor_res = [sensor_res, sensor_res] if isinstance(sensor_res, int) else sensor_res
        self.sensor_res = sensor_res
        H, W = sensor_res
        if sensor_size is None:
            diag = np.sqrt(H ** 2 + W ** 2)
            self.sensor_size = [2 * self.r_last * H / diag, 2 * self.r_last * W / diag]
        else:
            self.sensor_size = sensor_size
            self.r_last = np.sqrt(sensor_size[0] ** 2 + sensor_size[1] ** 2) / 2
        assert self.sensor_size[0] / self.sensor_size[1] == H / W, "Pixel is not square."
        self.pixel_size = self.sensor_size[0] / sensor_res[0]

    # ------------------------------------------------------------------ device state
    def _gpu(self):
        _abi.require_gpu()
        if self.device.type != "cuda":
            return torch.device("cuda", torch.cuda.current_device())
        return self.device

    def _state_device(self):
        """Device copy of the lens state, uploading pending host edits.  The kernels' error-flags word lives in the 8 bytes behind it
        (`_flags_device`), so one read-back returns both."""
        dev = self._gpu()
        if self._state_dev is None or self._state_dev.device != dev:
            n = C.sizeof(_abi.LensState)
            self._state_buf = torch.zeros(n + 8, dtype=torch.uint8, device=dev)
            self._state_dev = self._state_buf[:n]
            self._flags_dev = self._state_buf[n:n + 4].view(torch.int32)
            self._state_upload()
        return self._state_dev

    def _flags_device(self):
        """int32 view of the flags word the PSF kernels OR into (bits: raise_psf_flags)."""
        self._state_device()
        return self._flags_dev

    def _state_upload(self):
        host = torch.frombuffer(bytearray(bytes(self._state_host)), dtype=torch.uint8)
        self._state_dev.copy_(host)
        self._state_stale = False

    def _state_sync(self):
        """Host view of the state; reads the device copy back if a kernel rewrote it - and with it the flags word of the PSF calls
        since the last read-back: their errors (the reference asserts inside the call, optics.py:901,1176; here the calls are
        asynchronous) are raised at this, the next host read-back, with the reference's exception types."""
        if self._state_stale and self._state_dev is not None:
            raw = bytes(self._state_buf.cpu().numpy().tobytes())
            n = C.sizeof(_abi.LensState)
            self._state_host = _abi.LensState.from_buffer_copy(raw[:n])
            self._state_stale = False
            bits = int.from_bytes(raw[n:n + 4], "little", signed=True)
            if bits:
                self._flags_clear()
            if self._state_host.flags & 1 or bits & 1:
                raise FloatingPointError("found nan in ft in non-diff newton method.")   # reference exits: surfaces.py:555-558
            assert self._state_host.d_sensor > 0, "sensor position is negative."          # optics.py:1176
            raise_psf_flags(bits)
        return self._state_host

    def _flags_clear(self):
        self._flags_device().zero_()
        if self._flags_mirror is not None:
            self._flags_mirror[0] = 0

    def check_flags(self):
        """Raise the reference's errors for anything the kernels of earlier calls flagged (synchronises).  The per-call API is
        asynchronous: `psf_map` / `psf` / `refocus` return without waiting, their error conditions surface at the next host
        read-back (`d_sensor`, `hfov`, ...), at a later call through the pinned mirror of the flags word (published every 8 PSF
        calls), or here.  `lens.sync_flags = True` (or AADFF_SYNC_FLAGS=1) restores the check inside every call."""
        if self._state_dev is None:
            return
        self._state_stale = True
        self._state_sync()

    def _poll_flags(self):
        """Flags as of the last published mirror: no synchronisation, the error reaches the host a few calls late instead of never."""
        if self._flags_mirror is not None and int(self._flags_mirror[0]):
            bits = int(self._flags_mirror[0])
            self._flags_clear()
            raise_psf_flags(bits)

    def _state_set(self, name, value):
        self._state_sync()
        setattr(self._state_host, name, float(value))
        if name == "hfov":
            self._state_host.tan_hfov = float(np.tan(float(value)))
        if self._state_dev is not None:
            self._state_upload()

    d_sensor = property(lambda self: float(self._state_sync().d_sensor), lambda self, v: self._state_set("d_sensor", v))
    hfov = property(lambda self: float(self._state_sync().hfov), lambda self, v: self._state_set("hfov", v))
    foclen = property(lambda self: float(self._state_sync().foclen), lambda self, v: self._state_set("foclen", v))
    fnum = property(lambda self: float(self._state_sync().fnum), lambda self, v: self._state_set("fnum", v))

    def _table(self, wvlns):
        key = (tuple(float(w) for w in wvlns), str(self._gpu()))
        if key not in self._table_cache:
            self._table_cache[key] = pack_table(self.surfaces, list(key[0]), self._gpu())
        return self._table_cache[key]

    def _lens_const(self):
        key = (tuple(self.sensor_res), float(self.pixel_size), float(self.r_last))
        hit = self._table_cache.get("lens-const")
        if hit is not None and hit[0] == key:
            return hit[1]
        enp_z, enp_r = self.entrance_pupil()
        exp_z, exp_r = self.exit_pupil()
        lc = _abi.LensConst()
        lc.n_surf = len(self.surfaces)
        lc.r_last = self.r_last
        lc.sensor_h, lc.sensor_w = self.sensor_size[0], self.sensor_size[1]
        lc.pixel_size = self.pixel_size
        lc.enp_z, lc.enp_r = enp_z, enp_r
        lc.enp_r2, lc.enp_r2_shrunk = enp_r ** 2, (enp_r * 0.5) ** 2
        lc.exp_z, lc.exp_r_shrunk = exp_z, exp_r * 0.5
        lc.first_d = self.surfaces[0].d.item()
        lc.first_r2 = self.surfaces[0].r ** 2
        self._table_cache["lens-const"] = (key, lc)
        return lc

    # ------------------------------------------------------------------ derived quantities
    def post_computation(self):
        """hfov, foclen, fnum for the current d_sensor (reference: optics.py:178-187)."""
        if self.parity != "fast":
            return self._post_computation_strict()
        self.find_aperture()
        st, lc = self._state_device(), self._lens_const()
        with torch.cuda.device(st.device):
            _abi.call("aadff_post_computation", 1, _abi.ptr(self._table([DEFAULT_WAVE])), lc, _abi.ptr(st),
                      _abi.stream_ptr(st.device))
        self._state_stale = True

    def calc_fov(self):
        self.post_computation()
        return self.hfov

    def calc_efl(self):
        return self.r_last / np.tan(self.hfov)

    def calc_scale_pinhole(self, depth):
        return -depth * np.tan(self.hfov) / self.r_last

    @torch.no_grad()
    def refocus(self, depth=DEPTH):
        """Move the sensor to the green-light focus of an on-axis point at `depth` (mm < 0)
        and refresh hfov/foclen/fnum (reference: optics.py:1155-1180).  One kernel, no
        host sync; host RNG order = surface_sample: theta then r (surfaces.py:192-193)."""
        if self.parity != "fast":
            from aadff import strict_stack
            if strict_stack.calls_fused(self):
                return strict_stack.strict_refocus_call(self, depth)
            return self._refocus_strict(depth)
        st, lc = self._state_device(), self._lens_const()
        with _abi.on_device(st.device):
            stream = torch.cuda.current_stream(st.device)
            if self.sampler.on_device:
                u = torch.stack((self.sampler.rand(GEO_SPP), self.sampler.rand(GEO_SPP))).to(st.device)
                dep = torch.tensor([float(depth)], dtype=torch.float32).to(st.device)
                _abi.call("aadff_refocus", _abi.ptr(dep), 1, _abi.ptr(u), GEO_SPP, 2 * GEO_SPP, _abi.ptr(self._table([DEFAULT_WAVE])),
                          lc, _abi.ptr(st), C.c_void_p(stream.cuda_stream))
            else:
                # host draws (theta row, r row: the reference's two torch.rand calls) + the depth in one pinned block, one async copy
                ring = self._call_ring(st.device, 2 * GEO_SPP + 4)
                k, h, d = ring.next()
                self.sampler.rand_into(h[:2 * GEO_SPP])
                h[2 * GEO_SPP] = float(depth)
                if ring.mapped is not None:                  # the one workgroup reads draws and depth from the pinned block itself
                    base = ring.mapped_ptr(k)
                else:
                    ring.sent(k, 2 * GEO_SPP + 4, stream)
                    base = d.data_ptr()
                _abi.call("aadff_refocus", C.c_void_p(base + 8 * GEO_SPP), 1, C.c_void_p(base), GEO_SPP, 2 * GEO_SPP,
                          _abi.ptr(self._table([DEFAULT_WAVE])), lc, _abi.ptr(st), C.c_void_p(stream.cuda_stream))
                ring.launched(k, stream)
        self._state_stale = True

    def _call_ring(self, dev, words):
        if self._ring is None or self._ring.words < words or self._ring.dev.device != dev:
            if self._ring is not None:
                self._ring.drain()                           # queued launches may still read the old ring's pinned blocks
            self._ring = _CallRing(dev, max(words, 3 * (2 * GEO_SPP + 2 * GEO_SPP) + 3 * 128 * 3))
        return self._ring

    def _refocus_strict(self, depth):
        """The reference's refocus line by line (optics.py:1155-1180): host sampling and host reductions are the
        reference's own torch / numpy calls, the trace is the strict one."""
        s0 = self.surfaces[0]
        theta = self.sampler.rand(GEO_SPP) * 2 * np.pi                     # surface_sample, surfaces.py:188-199
        r = torch.sqrt(self.sampler.rand(GEO_SPP) * s0.r ** 2)
        x2, y2 = r * torch.cos(theta), r * torch.sin(theta)
        o = torch.stack((x2, y2, torch.full_like(x2, s0.d.item())), 1)
        d = o - torch.tensor([0, 0, depth], dtype=torch.float32)
        ray, _, _ = self.trace(Ray(o, d, wvln=DEFAULT_WAVE, device="cpu"))
        t = (ray.d[..., 0] * ray.o[..., 0] + ray.d[..., 1] * ray.o[..., 1]) / (ray.d[..., 0] ** 2 + ray.d[..., 1] ** 2)
        t = t * ray.ra
        focus_d = (ray.o[..., 2] - ray.d[..., 2] * t).numpy()
        focus_d = focus_d[ray.ra > 0]
        focus_d = focus_d[~np.isnan(focus_d) & (focus_d > 0)]
        with np.errstate(all="ignore"):
            d_sensor_new = float(np.mean(focus_d)) if len(focus_d) else float("nan")
        assert d_sensor_new > 0, "sensor position is negative."
        self.d_sensor = d_sensor_new
        self.post_computation()

    def _post_computation_strict(self):
        """optics.py:178-187 + calc_fov :1187-1217 with the strict trace and the reference's host arithmetic."""
        self.find_aperture()
        M = 100
        pupilz, pupilx = self.exit_pupil(shrink_pupil=True)
        o1 = torch.tensor([self.r_last, 0, self.d_sensor]).repeat(M, 1).to(torch.float32)
        x2 = torch.linspace(-pupilx, pupilx, M)
        o2 = torch.stack((x2, torch.full_like(x2, 0), torch.full_like(x2, pupilz)), axis=-1)
        ray, _, _ = self.trace(Ray(o1, o2 - o1, wvln=DEFAULT_WAVE, device="cpu"))
        tan_fov = ray.d[..., 0] / ray.d[..., 2]
        fov = torch.atan(torch.sum(tan_fov * ray.ra) / torch.sum(ray.ra))
        hfov = 0.5 if torch.isnan(fov) else fov.item()
        foclen = self.r_last / np.tan(hfov)
        _, enp_r = self.entrance_pupil()
        self._state_sync()
        self._state_host.hfov, self._state_host.tan_hfov = float(hfov), float(np.tan(hfov))
        self._state_host.foclen, self._state_host.fnum = float(foclen), float(foclen / enp_r / 2)
        if self._state_dev is not None:
            self._state_upload()

    # ------------------------------------------------------------------ pupils
    @torch.no_grad()
    def exit_pupil(self, shrink_pupil=False):
        return self.entrance_pupil(entrance=False, shrink_pupil=shrink_pupil)

    @torch.no_grad()
    def entrance_pupil(self, M=32, entrance=True, shrink_pupil=False):
        """(z, radius) of the stop's image through the front (entrance) or rear (exit)
        group: M edge rays traced on the GPU, pairwise x-z intersections and the 10%
        trimmed mean on the host (reference: optics.py:1320-1403).  Cached per geometry."""
        if self.aper_idx is None:
            s = self.surfaces[0] if entrance else self.surfaces[-1]
            return s.d.item(), s.r
        key = (M, entrance)
        if key not in self._pupil_cache:
            aper = self.surfaces[self.aper_idx]
            aper_z, aper_r = aper.d.item(), aper.r
            o = torch.tensor([[aper_r, 0, aper_z]]).repeat(M, 1).to(torch.float32)
            phi = torch.arange(-0.5, 0.5, 1.0 / M)
            dz = -torch.cos(phi) if entrance else torch.cos(phi)
            d = torch.stack((torch.sin(phi), torch.zeros_like(phi), dz), axis=-1)
            rng = range(0, self.aper_idx) if entrance else range(self.aper_idx + 1, len(self.surfaces))
            ray, _, _ = self.trace(Ray(o, d, device="cpu" if self.parity != "fast" else self._gpu()), lens_range=rng)
            ro, rd, ra = ray.o.cpu().numpy(), ray.d.cpu().numpy(), ray.ra.cpu().numpy()
            ii, jj = np.triu_indices(M, 1)
            keep = (ra[ii] != 0) & (ra[jj] != 0)
            ii, jj = ii[keep], jj[keep]
            d1x, d1z, d2x, d2z = rd[ii, 0], rd[ii, 2], rd[jj, 0], rd[jj, 2]
            o1x, o1z, o2x, o2z = ro[ii, 0], ro[ii, 2], ro[jj, 0], ro[jj, 2]
            det = -d1x * d2z + d2x * d1z
            b1 = -d1z * o1x + d1x * o1z
            b2 = -d2z * o2x + d2x * o2z
            oz = (-b1 * d2z + b2 * d1z) / det
            ox = (b2 * d1x - b1 * d2x) / det
            if len(ox) == 0:
                px, pz = aper_r, 0
            else:
                px = float(stats.trim_mean(ox.astype(np.float64), 0.1))
                pz = float(stats.trim_mean(oz.astype(np.float64), 0.1))
                if np.abs(pz) < EPSILON:
                    pz = 0
            self._pupil_cache[key] = (pz, px)
        pz, px = self._pupil_cache[key]
        return (pz, px * 0.5) if shrink_pupil else (pz, px)

    # ------------------------------------------------------------------ ray sampling / tracing
    @torch.no_grad()
    def sample_from_points(self, o=[[0, 0, -10000]], spp=256, wvln=DEFAULT_WAVE, shrink_pupil=False, normalized=False):
        """Rays [spp,N,3] from object points through the entrance pupil; ONE pupil sample
        set shared by all points (reference: optics.py:457-491)."""
        if not torch.is_tensor(o):
            o = torch.tensor(o)
        o = o.float().unsqueeze(0).repeat(spp, 1, 1)
        pupilz, pupilr = self.entrance_pupil(shrink_pupil=shrink_pupil)
        theta = self.sampler.rand(spp) * 2 * np.pi
        r = torch.sqrt(self.sampler.rand(spp) * pupilr ** 2)
        o2 = torch.stack((r * torch.cos(theta), r * torch.sin(theta), torch.full_like(r, pupilz)), 1)
        return Ray(o, o2.unsqueeze(1) - o.cpu(), wvln=wvln, device="cpu" if self.parity != "fast" else self.device)

    def _trace_strict(self, ray, first, last, forward, z_sensor=None):
        """`ray` (any device, any leading shape) through surfaces [first, last) in the reference's operation order; the
        whole bundle is ONE Newton batch, as in one reference call.  Returns a new Ray on the CPU (strict mode keeps rays
        there: the reference's host-side arithmetic on them is then the reference's own)."""
        dev = self._gpu()
        shape = ray.o.shape
        o = _abi.f32c(ray.o, dev).reshape(-1, 3).clone()
        d = _abi.f32c(ray.d, dev).reshape(-1, 3).clone()
        ra = _abi.f32c(ray.ra, dev).reshape(-1).clone()
        n = len(self.surfaces)
        tab = (_abi.Surface * n)(*[s.pack(ray.wvln) for s in self.surfaces])
        scratch = torch.zeros(2 * _abi.MAX_SURF + 1, dtype=torch.int32, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _abi.call("aadff_trace_rays_strict", _abi.ptr(o), _abi.ptr(d), _abi.ptr(ra), o.shape[0], C.byref(tab), first, last, int(forward),
                      int(z_sensor is not None), float(z_sensor if z_sensor is not None else 0.0), _abi.ptr(scratch), _abi.ptr(flag),
                      _abi.stream_ptr(dev))
        if int(flag.item()):
            raise FloatingPointError("found nan in ft in non-diff newton method.")
        out = Ray.__new__(Ray)
        out.wvln, out.coherent, out.device = ray.wvln, False, torch.device("cpu")
        out.o, out.d, out.ra = o.cpu().reshape(shape), d.cpu().reshape(shape), ra.cpu().reshape(shape[:-1])
        return out

    def trace(self, ray, lens_range=None, record=False):
        """Ray in, (ray_out, valid, oss) out; direction from the first ray's d_z
        (reference: optics.py:598-624)."""
        if record:
            raise NotImplementedError("record=True is a plotting aid outside the hot path")
        is_forward = bool(ray.d.reshape(-1, 3)[0, 2] > 0)
        rng = range(0, len(self.surfaces)) if lens_range is None else lens_range
        first, last = (rng.start, rng.stop) if len(rng) else (0, 0)
        if self.parity != "fast":
            out = self._trace_strict(ray, first, last, is_forward)
            return out, (out.ra == 1), None
        out = trace_ray_object(ray, self.surfaces, first, last, is_forward, None, table=self._table([ray.wvln]))
        return out, (out.ra == 1), None

    def trace2sensor(self, ray, record=False, ignore_invalid=False):
        """trace + propagate every ray to z = d_sensor (reference: optics.py:635-661)."""
        if record:
            raise NotImplementedError("record=True is a plotting aid outside the hot path")
        is_forward = bool(ray.d.reshape(-1, 3)[0, 2] > 0)
        if self.parity != "fast":
            return self._trace_strict(ray, 0, len(self.surfaces), is_forward, z_sensor=self.d_sensor)
        return trace_ray_object(ray, self.surfaces, 0, len(self.surfaces), is_forward, self._state_device(),
                                table=self._table([ray.wvln]))

    def trace2obj(self, ray, depth=DEPTH):
        ray, _, _ = self.trace(ray)
        return ray.propagate_to(depth)

    # ------------------------------------------------------------------ PSFs
    def point_source_grid(self, depth, grid=9, normalized=True, quater=False, center=False):
        """[grid,grid,3] field points, x in linspace(-.98,.98), y in linspace(.98,-.98)
        (reference: optics.py:813-860)."""
        if grid == 1:
            x, y = torch.tensor([[0.]]), torch.tensor([[0.]])
            assert not quater, "Quater should be False when grid is 1."
        elif center:
            hb = 1 / 2 / (grid - 1)
            x, y = torch.meshgrid(torch.linspace(-1 + hb, 1 - hb, grid), torch.linspace(1 - hb, -1 + hb, grid), indexing="xy")
        else:
            x, y = torch.meshgrid(torch.linspace(-0.98, 0.98, grid), torch.linspace(0.98, -0.98, grid), indexing="xy")
        pts = torch.stack([x, y, torch.full((grid, grid), depth)], dim=-1)
        if quater:
            bi = grid // 2 if grid % 2 == 0 else grid // 2 + 1
            pts = pts[0:bi, grid // 2:, :]
        if not normalized:
            scale = self.calc_scale_pinhole(depth)
            pts[..., 0] *= scale * self.sensor_size[0] / 2
            pts[..., 1] *= scale * self.sensor_size[1] / 2
        return pts

    def _object_points(self, points):
        scale = self.calc_scale_pinhole(points[:, 2])
        pobj = points.clone()
        pobj[..., 0] = points[..., 0] * scale * self.sensor_size[1] / 2
        pobj[..., 1] = points[..., 1] * scale * self.sensor_size[0] / 2
        return pobj

    @torch.no_grad()
    def psf_center(self, point, method="chief_ray"):
        """Reference PSF centre [N,2] for UN-normalised object points (reference: optics.py:888-913)."""
        if method == "chief_ray":
            ray = self.trace2sensor(self.sample_from_points(point, spp=GEO_SPP, shrink_pupil=True))
            assert (ray.ra == 1).any(), "No sampled rays is valid."
            c = (ray.o * ray.ra.unsqueeze(-1)).sum(0) / ray.ra.unsqueeze(-1).sum(0).add(EPSILON)
            return -c[..., :2]
        if method == "pinhole":
            return -point[..., :2] / self.calc_scale_pinhole(point[..., 2])
        raise Exception("Unsupported method.")

    def _psf_launch(self, points, wvlns, ks, spp, center, map_layout):
        """One fused launch (trace chief + main rays, splat, normalise) for len(wvlns)
        wavelengths.  Host RNG order per wavelength: main theta, main r, chief theta,
        chief r.  Asynchronous: the kernels' error conditions (`raise_psf_flags`) are raised at the next
        host read-back of the lens state, at `check_flags()`, or a few calls later through the pinned mirror of the flags word
        (`sync_flags` restores the wait inside the call)."""
        dev = self._gpu()
        L, N = len(wvlns), points.shape[0]
        from aadff import ops
        self._poll_flags()
        flags = self._flags_device()
        lcl = self._table_cache.get("lens-const-list")
        if lcl is None or lcl[0] != (tuple(self.sensor_res), float(self.pixel_size)):
            lcl = self._table_cache["lens-const-list"] = ((tuple(self.sensor_res), float(self.pixel_size)), ops.lens_const_to_list(self._lens_const()))
        with _abi.on_device(dev):
            stream = torch.cuda.current_stream(dev)
            if self.sampler.on_device:
                mains, chiefs = [], []
                for _ in wvlns:
                    mains += [self.sampler.rand(spp), self.sampler.rand(spp)]
                    if center:
                        chiefs += [self.sampler.rand(GEO_SPP), self.sampler.rand(GEO_SPP)]
                u_main = torch.stack(mains).to(dev)
                u_chief = torch.stack(chiefs).to(dev) if center else None
                pts = _abi.f32c(points, dev)
                um = u_main.reshape(1, L, 2, spp)
                uc = u_chief.reshape(1, L, 2, GEO_SPP) if center else torch.empty((1, L, 2, 0), device=dev)
                out = torch.ops.aadff.psf_points(pts.unsqueeze(0), self._table(wvlns), self._table([DEFAULT_WAVE]), lcl[1],
                                                 self._state_device(), um, uc, ks, bool(center), bool(map_layout), flags)[0]
            else:
                spc = GEO_SPP if center else 0
                n_u = L * (2 * spp + 2 * spc)
                ring = self._call_ring(dev, n_u + 3 * N)
                k, h, d = ring.next()
                self.sampler.rand_into(h[:n_u])              # one fill = the reference's 4 L torch.rand calls, same generator stream
                h[n_u:n_u + 3 * N].view(N, 3).copy_(points)
                staged = ring.mapped is not None and n_u % 4 == 0 and not torch.compiler.is_compiling()
                if not staged:
                    ring.sent(k, n_u + 3 * N, stream)
                if torch.compiler.is_compiling():
                    out = torch.ops.aadff.psf_points_block(d[n_u:n_u + 3 * N].view(N, 3), self._table(wvlns), self._table([DEFAULT_WAVE]), lcl[1],
                                                           self._state_device(), d[:n_u], L, spp, spc, ks, bool(map_layout), flags)
                else:       # the same launch without the custom-op dispatcher (~35 us of Python per call: as much as the rest of the call)
                    g = int(round(N ** 0.5))
                    out = torch.empty((L, g * ks, g * ks) if map_layout else (N, L, ks, ks), dtype=torch.float32, device=dev)
                    per_l, base = 2 * spp + 2 * spc, d.data_ptr()
                    if staged:      # the draws are copied by the launch's first workgroups, the points read where they lie (12 bytes per workgroup)
                        _abi.call("aadff_psf_points_staged", C.c_void_p(ring.mapped_ptr(k) + 4 * n_u), 1, N, L, _abi.ptr(self._table(wvlns)),
                                  _abi.ptr(self._table([DEFAULT_WAVE])), self._lens_const(), _abi.ptr(self._state_device()), C.c_void_p(base), spp, L * per_l, per_l,
                                  C.c_void_p(base + 8 * spp) if spc else None, spc, L * per_l, per_l, ks, int(spc > 0), int(map_layout), _abi.ptr(out), None,
                                  _abi.ptr(flags), C.byref(ring.stage(k, n_u, (N, L, n_u))), C.c_void_p(stream.cuda_stream))
                    else:
                        _abi.call("aadff_psf_points", C.c_void_p(base + 4 * n_u), 1, N, L, _abi.ptr(self._table(wvlns)), _abi.ptr(self._table([DEFAULT_WAVE])),
                                  self._lens_const(), _abi.ptr(self._state_device()), C.c_void_p(base), spp, L * per_l, per_l,
                                  C.c_void_p(base + 8 * spp) if spc else None, spc, L * per_l, per_l, ks, int(spc > 0), int(map_layout), _abi.ptr(out), None,
                                  _abi.ptr(flags), C.c_void_p(stream.cuda_stream))
                ring.launched(k, stream)
            self._psf_calls += 1
            if self.sync_flags:
                self._state_stale = True
                self._state_sync()
            elif self._psf_calls % 8 == 0:
                if self._flags_mirror is None:
                    self._flags_mirror = torch.zeros(1, dtype=torch.int32, pin_memory=True)
                _abi.call("aadff_publish_flags", _abi.ptr(flags), C.c_void_p(self._flags_mirror.data_ptr()), C.c_void_p(stream.cuda_stream))
        return out.to(self.device) if self.device.type != "cuda" else out

    def psf(self, points, ks=31, wvln=DEFAULT_WAVE, spp=GEO_SPP, center=True):
        """[N,ks,ks] (or [ks,ks]) single-wavelength PSFs of NORMALISED points (reference: optics.py:915-983)."""
        return self.psf_diff(points=points, wvln=wvln, ks=ks, spp=spp, center=center)

    def _psf_strict(self, points, wvln, ks, spp, center):
        """psf_diff of the reference, call by call (optics.py:933-983): sample on the host, strict trace, chief-ray centre by
        the reference's own sum on the host, then the histogram kernel (whose only discontinuity, the window test, compares
        the same float32 quantities as monte_carlo.py:37)."""
        from .monte_carlo import forward_integral
        pobj = self._object_points(points)
        ray = self.trace2sensor(self.sample_from_points(pobj, spp=spp, wvln=wvln))
        if center:
            ref = self.psf_center(pobj)
        else:
            ref = points.clone()[:, :2]
            ref[:, 0] *= self.sensor_size[1] / 2
            ref[:, 1] *= self.sensor_size[0] / 2
        dev = self._gpu()
        ray.o, ray.ra = ray.o.to(dev), ray.ra.to(dev)
        psf = forward_integral(ray, ps=self.pixel_size, ks=ks, pointc_ref=ref.to(dev))
        psf = psf / psf.sum(-1).sum(-1).unsqueeze(-1).unsqueeze(-1)             # optics.py:978 (0/0 -> NaN like the reference)
        return psf if self.device.type == "cuda" else psf.to(self.device)

    def psf_diff(self, points, wvln=DEFAULT_WAVE, ks=31, spp=GEO_SPP, center=True):
        if not torch.is_tensor(points):
            points = torch.tensor(points)
        single = len(points.shape) == 1
        if single:
            points = points.unsqueeze(0)
        if self.parity != "fast":
            out = self._psf_strict(points.float(), wvln, ks, spp, center)
            return out.squeeze(0) if single else out
        out = self._psf_launch(points.float(), [wvln], ks, spp, center, False)[:, 0]
        return out.squeeze(0) if single else out

    def psf_rgb(self, points, ks=31, spp=GEO_SPP, center=True):
        """[N,3,ks,ks] (or [3,ks,ks]) PSFs at WAVE_RGB (reference: optics.py:986-1003)."""
        if not torch.is_tensor(points):
            points = torch.tensor(points)
        single = len(points.shape) == 1
        if single:
            points = points.unsqueeze(0)
        if self.parity != "fast":
            out = torch.stack([self._psf_strict(points.float(), w, ks, spp, center) for w in WAVE_RGB], dim=-3)
            return out.squeeze(0) if single else out
        out = self._psf_launch(points.float(), WAVE_RGB, ks, spp, center, False)
        return out.squeeze(0) if single else out

    def psf_map(self, depth=DEPTH, grid=7, ks=51, spp=GEO_SPP, center=True):
        """[3, grid*ks, grid*ks] RGB PSF map at one depth plane, row-major tiling, top row
        = +y (reference: optics.py:1006-1026)."""
        if ks > _abi.MAX_KS:
            raise ValueError(f"ks={ks} exceeds the kernels' limit {_abi.MAX_KS}")
        cache = self._table_cache.setdefault("psf-map-points", {})          # the field grid of (depth, grid): a handful of torch ops per call
        pts = cache.get((float(depth), int(grid)))
        if pts is None:
            if len(cache) >= 64:
                cache.clear()
            pts = cache[(float(depth), int(grid))] = self.point_source_grid(depth=depth, grid=grid, quater=False).reshape(-1, 3).float().contiguous()
        if self.parity != "fast":
            from aadff import strict_stack
            if center and strict_stack.calls_fused(self):
                out = strict_stack.strict_psf_map_call(self, depth, grid, ks, spp)
                return out.to(self.device) if self.device.type != "cuda" else out
            return make_grid(self.psf_rgb(pts, ks=ks, spp=spp, center=center), nrow=grid, padding=0)
        return self._psf_launch(pts, WAVE_RGB, ks, spp, center, True)

    # ------------------------------------------------------------------ image rendering / misc
    @torch.no_grad()
    def render_single_img(self, img_org, depth=DEPTH, spp=64, unwarp=False, save_name=None, return_tensor=False,
                          noise=0, method="raytracing"):
        """Reference signature (optics.py:722).  `method='psf'` is the branch on the focal-stack hot path (:779-783: 7x7 PSF
        grid, ks 21); the default 'raytracing' branch calls `calc_scale_ray`, `render_sample_ray` and `render_compute_image`,
        which the reference snapshot does not define (AttributeError there) - refused here with the way out named."""
        if method != "psf":
            raise NotImplementedError(f"render_single_img(method='{method}'): only method='psf' exists (the reference's 'raytracing' "
                                      "branch calls methods its snapshot lacks); pass method='psf'")
        if not isinstance(img_org, np.ndarray):
            raise Exception("This function only supports ndarray input. If you want to render an image batch, use `render` function.")
        H, W, Cn = img_org.shape
        assert Cn == 3, "Only support RGB image, dtype should be ndarray."
        old = self.sensor_res
        self.prepare_sensor(sensor_res=[H, W])
        img = torch.tensor((img_org / 255.).astype(np.float32)).permute(2, 0, 1).unsqueeze(0).to(self._gpu())
        out = render_psf_map(img, self.psf_map(grid=7, ks=21, depth=depth), grid=7)
        if noise > 0:
            out = torch.clamp(out + torch.randn_like(out) * noise, 0, 1)
        self.prepare_sensor(sensor_res=old)
        if return_tensor:
            return out
        return out[0].mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()

    @torch.no_grad()
    def draw_psf_map(self, grid=7, depth=DEPTH, ks=51, log_scale=False, quater=False, save_name=None):
        """RGB PSF map at `depth`, every field normalised to its own peak, saved as an image (reference:
        optics.py:1773-1803).  The map comes from the fused PSF-grid kernel; the picture is written with matplotlib like the
        reference (scale ruler included) or, without matplotlib, as a plain PNG through `save_image`.  Returns the file name."""
        psf_map = self.psf_map(depth=depth, grid=grid, ks=ks, spp=GEO_SPP, center=True).clone()
        G = grid * ks
        tiles = psf_map.reshape(3, grid, ks, grid, ks)
        psf_map = (tiles / tiles.amax(dim=(0, 2, 4), keepdim=True)).reshape(3, G, G)
        if log_scale:
            psf_map = torch.log(psf_map + 1e-3)
        save_name = f"./psf{-depth}mm.png" if save_name is None else f"{save_name}_psf{-depth}mm.png"
        if plt is None:
            save_image(psf_map.clamp(0, 1), save_name)
            return save_name
        plt.figure(figsize=(10, 10))
        plt.imshow(psf_map.permute(1, 2, 0).cpu().numpy())
        ruler_len = 100                                              # um
        arrow_end = ruler_len / (self.pixel_size * 1e3)
        plt.annotate("", xy=(0, G - 10), xytext=(arrow_end, G - 10), arrowprops=dict(arrowstyle="<->", color="white"))
        plt.text(arrow_end + 10, G - 10, f"{ruler_len} um", color="white", fontsize=12, ha="left")
        plt.axis("off")
        plt.tight_layout(pad=0)
        plt.savefig(save_name, dpi=300)
        plt.close()
        return save_name

    @torch.no_grad()
    def analysis(self, save_name="./test", render=False, multi_plot=False, plot_invalid=True, zmx_format=False, depth=DEPTH,
                 render_unwarp=False, lens_title=None):
        """Reference signature (optics.py:1552).  Of its four products this build makes the one that comes off the hot path -
        the PSF map picture (`draw_psf_map`, ks 51) - and logs the first-order numbers; the 2-D layout drawing, the RMS
        spot statistics (they need `sample_point_source`) and the rendered resolution chart are outside the path (DESIGN.md
        section 8) and are skipped with a log line instead of failing, so scripts that call `lens.analysis(...)` keep running."""
        log = logging.getLogger()
        log.info("lens %s: foclen %.4f mm, F/%.4f, hfov %.5f rad, d_sensor %.5f mm, pixel %.6f mm",
                 getattr(self, "lens_name", "?"), self.foclen, self.fnum, self.hfov, self.d_sensor, self.pixel_size)
        out = self.draw_psf_map(save_name=save_name, ks=51, depth=depth)
        log.info("PSF map written to %s; layout drawing, RMS spot analysis%s are not part of this build", out,
                 " and chart rendering" if render else "")
        return out


Lens = Lensgroup      # `north_star` calls the class optics.Lens
