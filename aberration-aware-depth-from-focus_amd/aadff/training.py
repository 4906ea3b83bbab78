"""Pipelined producer of PSFNet training batches (reference: deeplens/psfnet.py:135-170 get_training_data).

Per batch the reference draws, in this order on the host: np.random.choice over the 20 focus distances, the refocus
samples (2 x 2048), rand(bs) x, rand(bs) y, randn(bs) z, then the PSF samples (main 2 x spp, chief 2 x 2048).  All of it
goes into ONE pinned block per batch,

    [focus theta | focus r | main theta | main r | chief theta | chief r | points bs x 3 | inp bs x 4]

which `aadff_refocus_staged` uploads from inside the refocus launch (the focus workgroups read their own draws over
PCIe meanwhile); `aadff_psf_points` then traces the bs target PSFs from the device copy.  Two launches per batch, no
memcpy, no host synchronisation: the kernels' error flags (NaN residual, no valid chief ray: the reference's asserts)
are published to a pinned mirror every few batches and polled when the ring wraps.
"""
import ctypes as C
import time

import numpy as np
import torch

from . import _abi
from deeplens.basics import DEFAULT_WAVE, GEO_SPP


class TrainingDataPlan:
    RING = 8
    GUARD_EVERY = 4

    def __init__(self, net, bs, spp):
        dev = net._gpu()
        self.net, self.bs, self.spp, self.dev, self.ks = net, int(bs), int(spp), dev, net.kernel_size
        self.o_main = 2 * GEO_SPP
        self.o_chief = self.o_main + 2 * self.spp
        self.o_pts = self.o_chief + 2 * GEO_SPP
        self.o_inp = self.o_pts + 3 * self.bs
        self.per = (self.o_inp + 4 * self.bs + 3) // 4 * 4
        self.u_pin = [torch.zeros(self.per, dtype=torch.float32).pin_memory() for _ in range(self.RING)]
        self.u_dev = [torch.zeros(self.per, dtype=torch.float32, device=dev) for _ in range(self.RING)]
        self.psf = [torch.empty((self.bs, self.ks * self.ks), dtype=torch.float32, device=dev) for _ in range(self.RING)]
        self.flags = torch.zeros(1, dtype=torch.int32, device=dev)
        self.flags_mirror = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.scratch = torch.zeros(16, dtype=torch.int32, device=dev)               # 64 B: one focus state
        self.foc_z32 = net.foc_z_arr.astype(np.float64)
        foc_d = self.foc_z32 * (net.d_max - net.d_min) + net.d_min                  # psfnet.py:147, float64 on the host
        self.dep_all = torch.tensor(foc_d, dtype=torch.float32).to(dev)
        self.tab_green = net._table([DEFAULT_WAVE])
        self.lc = net._lens_const()
        self.guards, self.turn = {}, 0
        self.wait_s = 0.0            # host time spent waiting for the GPU to release a pinned block (GPU-bound indicator)

    def _poll(self):
        bits = int(self.flags_mirror[0])
        if bits:
            from deeplens.optics import raise_psf_flags
            self.flags_mirror[0] = 0
            self.flags.zero_()
            raise_psf_flags(bits)

    def check_flags(self):
        """Synchronous form: raise the reference's errors for anything flagged so far."""
        from deeplens.optics import raise_psf_flags
        bits = int(self.flags.item())
        if bits:
            self.flags.zero_()
            self.flags_mirror[0] = 0
        raise_psf_flags(bits)

    def next(self):
        """(inp [bs,4], psf [bs,ks*ks]) of the next batch as DEVICE views that stay valid for RING - 1 further calls."""
        net, bs, spp = self.net, self.bs, self.spp
        k = self.turn % self.RING
        last = self.turn - self.RING
        if last >= 0:
            g = self.guards.get(last // self.GUARD_EVERY)
            if g is not None:
                t0 = time.perf_counter()
                g.synchronize()                    # the launches that read this pinned block have completed
                self.wait_s += time.perf_counter() - t0
                self._poll()
        pin = self.u_pin[k]
        # ---- host draws, reference order
        idx = int(np.random.choice(len(self.foc_z32)))          # == np.random.choice(foc_z_arr): one randint either way
        foc_z = self.foc_z32[idx]
        net.sampler.rand_into(pin[:self.o_main])                # refocus: theta, r (surfaces.py:192-193)
        x = (torch.rand(bs) - 0.5) * 2
        y = (torch.rand(bs) - 0.5) * 2
        zg = torch.clamp(torch.randn(bs), min=-3, max=3)
        # the reference's masked assignments (z = 0 where z_gauss == 0) as selects: the same fp32 operations per element,
        # a third of the host time of boolean indexing
        z = torch.where(zg > 0, (1 - foc_z) * zg / 3 + foc_z, torch.where(zg < 0, foc_z * zg / 3 + foc_z, torch.zeros_like(zg)))
        net.sampler.rand_into(pin[self.o_main:self.o_pts])      # psf: main theta, main r, chief theta, chief r
        pin[self.o_pts:self.o_inp].view(bs, 3).copy_(torch.stack((x, y, net.z2depth(z)), dim=-1))
        pin[self.o_inp:self.o_inp + 4 * bs].view(bs, 4).copy_(torch.stack((x, y, z, torch.full_like(x, foc_z)), dim=-1))
        # ---- two launches
        dev_blk = self.u_dev[k]
        ub = dev_blk.data_ptr()
        st_dev = net._state_device()
        with torch.cuda.device(self.dev):
            st = _abi.stream_ptr(self.dev)
            _abi.call("aadff_refocus_staged", C.c_void_p(self.dep_all.data_ptr() + 4 * idx), 1, C.c_void_p(pin.data_ptr()),
                      C.c_void_p(ub), self.per, GEO_SPP, self.per, _abi.ptr(self.tab_green), self.lc, _abi.ptr(st_dev),
                      _abi.ptr(self.scratch), st)
            _abi.call("aadff_psf_points", C.c_void_p(ub + 4 * self.o_pts), 1, bs, 1, _abi.ptr(self.tab_green), _abi.ptr(self.tab_green),
                      self.lc, _abi.ptr(st_dev), C.c_void_p(ub + 4 * self.o_main), spp, self.per, 2 * spp,
                      C.c_void_p(ub + 4 * self.o_chief), GEO_SPP, self.per, 2 * GEO_SPP, self.ks, 1, 0,
                      _abi.ptr(self.psf[k]), None, _abi.ptr(self.flags), st)
            if self.turn % self.GUARD_EVERY == self.GUARD_EVERY - 1:
                _abi.call("aadff_publish_flags", _abi.ptr(self.flags), C.c_void_p(self.flags_mirror.data_ptr()), st)
                e = torch.cuda.Event()
                e.record()
                m = self.turn // self.GUARD_EVERY
                self.guards[m] = e
                for old in [q for q in self.guards if q < m - self.RING // self.GUARD_EVERY - 1]:
                    del self.guards[old]
        net._state_stale = True
        self.turn += 1
        return dev_blk[self.o_inp:self.o_inp + 4 * bs].view(bs, 4), self.psf[k]
