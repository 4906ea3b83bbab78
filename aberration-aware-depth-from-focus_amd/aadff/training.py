"""Pipelined producer of PSFNet training batches (reference: deeplens/psfnet.py:135-170 get_training_data).

Per batch the reference draws, in this order on the host: np.random.choice over the 20 focus distances, the refocus
samples (2 x 2048), rand(bs) x, rand(bs) y, randn(bs) z, then the PSF samples (main 2 x spp, chief 2 x 2048).  All of it
goes into ONE pinned block per batch,

    [focus theta | focus r | main theta | main r | chief theta | chief r | points bs x 3 | inp bs x 4]

which `aadff_refocus_staged` uploads from inside the refocus launch (the focus workgroups read their own draws over
PCIe meanwhile); `aadff_psf_points` then traces the bs target PSFs from the device copy.  Two launches per batch, no
memcpy, no host synchronisation: the kernels' error flags (NaN residual, no valid chief ray: the reference's asserts)
are published to a pinned mirror every few batches and polled when the ring wraps.

By default (`overlap=False` / AADFF_FIT_OVERLAP=0 turns it off) the launches go to the plan's own side stream and batch j+1 is produced while the optimisation step of batch j runs (the
step is a chain of small latency-bound kernels on a few CUs; the traces use the rest of the chip): `next()` hands out
the batch launched by the previous call and - unless told `prefetch=False` - draws and launches the following one.  The
host draws are the reference's, in the reference's order (nothing else consumes the RNGs between two batches of the fit
loop); only the moment they are drawn moves.  The focus state is traced into the plan's own two-slot buffer and copied
to the lens when the flags are checked / with the last batch, so nothing on the caller's stream races with the producer.
"""
import ctypes as C
import os
import time

import numpy as np
import torch

from . import _abi
from deeplens.basics import DEFAULT_WAVE, GEO_SPP


class TrainingDataPlan:
    RING = 8
    GUARD_EVERY = 4

    def __init__(self, net, bs, spp, overlap=None):
        dev = net._gpu()
        if overlap is None:
            # On by default: with the batch arithmetic in numpy the host needs 0.10 ms per batch and the GPU (traces 60 us +
            # fit step 137 us back to back) is the limit; side by side they take 0.15 ms (5 070 -> 6 570 it/s).
            overlap = os.environ.get("AADFF_FIT_OVERLAP", "1") != "0"
        self.side = torch.cuda.Stream(dev) if overlap else None
        self.state = [torch.zeros(C.sizeof(_abi.LensState), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.pending, self.last_slot = None, None
        self.net, self.bs, self.spp, self.dev, self.ks = net, int(bs), int(spp), dev, net.kernel_size
        self.o_main = 2 * GEO_SPP
        self.o_chief = self.o_main + 2 * self.spp
        self.o_pts = self.o_chief + 2 * GEO_SPP
        self.o_inp = self.o_pts + 3 * self.bs
        self.per = (self.o_inp + 4 * self.bs + 3) // 4 * 4
        self.u_pin = [torch.zeros(self.per, dtype=torch.float32).pin_memory() for _ in range(self.RING)]
        self.u_np = [t.numpy() for t in self.u_pin]                                  # the same pinned memory as numpy views
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
        self._sync_lens_state()
        if self.side is not None:
            self.side.synchronize()
        bits = int(self.flags.item())
        if bits:
            self.flags.zero_()
            self.flags_mirror[0] = 0
        raise_psf_flags(bits)

    def _sync_lens_state(self):
        """The lens is left focused where the last batch handed out was traced (reference: refocus inside get_training_data)."""
        if self.side is not None and self.last_slot is not None:
            self.net._state_device().copy_(self.state[self.last_slot])
            self.net._state_stale = True

    def next(self, prefetch=True):
        """(inp [bs,4], psf [bs,ks*ks]) of the next batch as DEVICE views that stay valid for RING - 2 further calls; the
        caller's current stream is ordered behind the launches that produce them.  prefetch=False: do not draw / launch
        the following batch (single batches, the last iteration: the RNGs stay where the reference leaves them)."""
        if self.pending is None:
            self.pending = self._launch()
        k, ev, slot = self.pending
        if ev is not None:
            torch.cuda.current_stream(self.dev).wait_event(ev)
        self.last_slot = slot
        self.pending = self._launch() if (prefetch and self.side is not None) else None
        if self.pending is None:
            self._sync_lens_state()
        bs = self.bs
        return self.u_dev[k][self.o_inp:self.o_inp + 4 * bs].view(bs, 4), self.psf[k]

    def _launch(self):
        """Draw batch `turn` on the host and enqueue its two launches (side stream when overlapping)."""
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
        idx = int(np.random.randint(0, len(self.foc_z32)))      # what np.random.choice(foc_z_arr) draws (same stream, 1/3 of the call cost)
        foc_z = self.foc_z32[idx]
        net.sampler.rand_into(pin[:self.o_main])                # refocus: theta, r (surfaces.py:192-193)
        # x, y, z, depth: torch's generator for the draws (the reference's stream), numpy for the 128-element arithmetic on
        # them - the same IEEE fp32 operations per element as the reference's tensor expressions (bit-equal, checked over
        # seeds against both the masked-assignment form and G10), at ~1 us per op instead of ~4
        f32 = np.float32
        x = (torch.rand(bs).numpy() - f32(0.5)) * f32(2)
        y = (torch.rand(bs).numpy() - f32(0.5)) * f32(2)
        zg = torch.randn(bs).numpy()
        zg = np.minimum(np.maximum(zg, f32(-3)), f32(3))
        fz = f32(foc_z)
        z = np.where(zg > 0, f32(1 - foc_z) * zg / f32(3) + fz, np.where(zg < 0, fz * zg / f32(3) + fz, f32(0))).astype(np.float32)
        net.sampler.rand_into(pin[self.o_main:self.o_pts])      # psf: main theta, main r, chief theta, chief r
        blk = self.u_np[k]
        pts = blk[self.o_pts:self.o_inp].reshape(bs, 3)
        pts[:, 0], pts[:, 1], pts[:, 2] = x, y, z * f32(net.d_max - net.d_min) + f32(net.d_min)       # z2depth
        inp = blk[self.o_inp:self.o_inp + 4 * bs].reshape(bs, 4)
        inp[:, 0], inp[:, 1], inp[:, 2], inp[:, 3] = x, y, z, fz
        # ---- two launches
        dev_blk = self.u_dev[k]
        ub = dev_blk.data_ptr()
        slot = self.turn % 2
        st_dev = net._state_device() if self.side is None else self.state[slot]
        with torch.cuda.device(self.dev):
            if self.side is None:
                st, rec = _abi.stream_ptr(self.dev), None
            else:
                # behind everything the consumer has enqueued so far: the step that read this ring slot RING batches ago,
                # and any copy of the focus state to the lens
                self.side.wait_stream(torch.cuda.current_stream(self.dev))
                st, rec = C.c_void_p(self.side.cuda_stream), self.side
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
                e.record(rec) if rec is not None else e.record()
                m = self.turn // self.GUARD_EVERY
                self.guards[m] = e
                for old in [q for q in self.guards if q < m - self.RING // self.GUARD_EVERY - 1]:
                    del self.guards[old]
        ev = None
        if rec is not None:
            self.events = getattr(self, "events", None) or [torch.cuda.Event() for _ in range(self.RING)]
            ev = self.events[k]
            ev.record(rec)
        else:
            net._state_stale = True
        self.turn += 1
        return k, ev, slot
