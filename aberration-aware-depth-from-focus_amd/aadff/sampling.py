"""Uniform-sample sources for pupil / aperture sampling.

`HostSampler` draws from torch's CPU generator in the reference's call order (one
`torch.rand(n)` per reference call): with the same seed the HIP
path consumes exactly the samples the reference would, which is what the 1e-4 parity
budget requires (seed-to-seed Monte-Carlo noise is 1.4e-2).  `DeviceSampler` draws on
the GPU: statistically equivalent, not sample-for-sample comparable.
"""
import ctypes as C

import torch

_FAST = None      # None = untested, True/False after the self-check


def _fast_host_rand_into(out):
    """Fill the 1-D float32 CPU tensor `out` with the next out.numel() draws of torch's global CPU
    generator via aadff_host_mt19937_uniform_f32 (bit-identical to torch.rand, ~7x faster) and
    advance the generator accordingly.  Returns False if the fast path is unavailable."""
    global _FAST
    if _FAST is False:
        return False
    from . import _abi
    try:
        lib = _abi.load_library()
    except RuntimeError:
        _FAST = False
        return False
    if _FAST is None:      # one-time self-check against torch.rand on a cloned state
        st = torch.get_rng_state()
        a = torch.empty(1500)
        s2 = st.clone()
        rc = lib.aadff_host_mt19937_uniform_f32(C.c_void_p(s2.data_ptr()), s2.numel(), 1500, C.c_void_p(a.data_ptr()))
        b = torch.rand(1500)
        ok = rc == 0 and torch.equal(a, b) and torch.equal(s2, torch.get_rng_state())
        torch.set_rng_state(st)
        _FAST = bool(ok)
        if not ok:
            return False
    st = torch.get_rng_state()
    rc = lib.aadff_host_mt19937_uniform_f32(C.c_void_p(st.data_ptr()), st.numel(), out.numel(), C.c_void_p(out.data_ptr()))
    if rc != 0:
        _FAST = False
        return False
    torch.set_rng_state(st)
    return True


def _fast_host_discard(n):
    """Advance torch's global CPU generator past n float32 draws without producing them (aadff_host_mt19937_discard)."""
    if _FAST is not True:
        return False
    from . import _abi
    st = torch.get_rng_state()
    if _abi.load_library().aadff_host_mt19937_discard(C.c_void_p(st.data_ptr()), st.numel(), int(n)) != 0:
        return False
    torch.set_rng_state(st)
    return True


class HostSampler:
    on_device = False

    def skip(self, n):
        """Leave the generator where `torch.rand(n)` would, without the draws (a sharded rank skipping slices it does not own)."""
        if n > 0 and not _fast_host_discard(n):
            torch.rand(int(n))

    def rand(self, n):
        return torch.rand(n)

    def rand_block(self, sizes):
        """Concatenation of torch.rand(s) for s in sizes (one flat draw: the CPU generator yields
        the same stream either way, checked in tests/test_host_logic.py)."""
        out = torch.empty(int(sum(sizes)))
        self.rand_into(out)
        return out

    def rand_into(self, out):
        """Fill a (possibly pinned) 1-D float32 CPU tensor with the next draws of the global generator."""
        if not _fast_host_rand_into(out):
            torch.rand(out.numel(), out=out)
        return out


class DeviceSampler:
    on_device = True

    def __init__(self, device, seed=None):
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        if seed is not None:
            self.gen.manual_seed(seed)

    def rand(self, n):
        return torch.rand(n, device=self.device, generator=self.gen)

    def rand_block(self, sizes):
        return torch.rand(int(sum(sizes)), device=self.device, generator=self.gen)
