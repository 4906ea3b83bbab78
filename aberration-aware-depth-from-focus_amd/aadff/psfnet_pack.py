"""Pack an MLP (list of nn.Linear) for the fused PSF-network kernel (csrc/psfnet.hip, aadff_psfnet_forward):
weights as fp16 (hi, lo) pairs in MFMA A-fragment order, biases padded to 16.

Precision of the split: w = hi + lo + r with hi = fp16(w), lo = fp16(w - hi).  While lo is a NORMAL fp16 number
(|w| >= ~0.06) r <= 2^-22 |w|; below that lo falls into the fp16 subnormal range, whose spacing is 2^-24 = 6e-8
ABSOLUTE, so r <= 3e-8 whatever |w| (for |w| < 6e-5 hi is subnormal too and carries the value to the same 3e-8).  A
256-term dot product with unit activations therefore carries at most ~5e-7 of split error (random signs), the same size
as the rounding of an fp32 accumulation of that length; the MFMA f16 path does not flush subnormal inputs (a flush would
show as 1e-5 errors in tests/test_gpu_parity.py::test_fused_mlp_small_weights).  Activations are split the same way."""
import ctypes as C

import numpy as np
import torch

from . import _abi
from . import ops  # noqa: F401  (registers torch.ops.aadff.*)

MAX_WIDTH, MAX_OUT = 256, 128
EVENT_HOOK = None      # optional callable(start: bool) recording a torch.cuda.Event around the kernel launch (bench.py)


def linears_of(mlp):
    """The nn.Linear layers of deeplens.psfnet_arch.MLP in order (Linear+ReLU ..., Linear+Sigmoid)."""
    return [m for m in mlp.net if isinstance(m, torch.nn.Linear)]


def supported(mlp):
    lin = linears_of(mlp)
    mods = list(mlp.net)
    ok_acts = all(isinstance(mods[2 * i + 1], torch.nn.ReLU) for i in range(len(lin) - 1)) and isinstance(mods[-1], torch.nn.Sigmoid)
    return (ok_acts and len(mods) == 2 * len(lin) and 1 <= len(lin) <= 16 and lin[0].in_features == 4
            and all(l.in_features <= MAX_WIDTH and l.out_features <= MAX_WIDTH for l in lin) and lin[-1].out_features <= MAX_OUT)


class PackedMLP:
    """Device buffers for aadff_psfnet_forward; rebuilt when a parameter changes (version counters)."""

    def __init__(self, mlp, device):
        lin = linears_of(mlp)
        self.key = tuple((p.data_ptr(), p._version) for l in lin for p in (l.weight, l.bias))
        self.n = len(lin)
        self.ins = (C.c_int * self.n)(*[l.in_features for l in lin])
        self.outs = (C.c_int * self.n)(*[l.out_features for l in lin])
        self.n_out = lin[-1].out_features
        planes, biases = [], []
        for l in lin:
            w = l.weight.detach().to(device=device, dtype=torch.float32)          # [out, in]
            n, k = w.shape
            npad, kpad = (n + 15) // 16 * 16, (k + 31) // 32 * 32
            wp = torch.zeros((npad, kpad), dtype=torch.float32, device=device)
            wp[:n, :k] = w
            hi = wp.half()
            lo = (wp - hi.float()).half()
            # [tile, m, step, kg, e] -> [tile, step, plane, lane = kg*16 + m, e]
            f = torch.stack((hi, lo), 0).reshape(2, npad // 16, 16, kpad // 32, 4, 8).permute(1, 3, 0, 4, 2, 5)
            planes.append(f.contiguous().reshape(-1))
            b = torch.zeros(npad, dtype=torch.float32, device=device)
            if l.bias is not None:
                b[:n] = l.bias.detach().to(device=device, dtype=torch.float32)
            biases.append(b)
        self.wpack = torch.cat(planes).contiguous()
        self.bias = torch.cat(biases).contiguous()
        self.flags = torch.zeros(1, dtype=torch.int32, device=device)
        wmax = max(float(l.weight.detach().abs().max()) for l in lin)
        if not wmax <= 65504.0:
            raise ValueError(f"aadff: |weight| up to {wmax:g} does not fit the fp16 hi/lo operand split (max 65504)")

    @staticmethod
    def key_of(mlp):
        return tuple((p.data_ptr(), p._version) for l in linears_of(mlp) for p in (l.weight, l.bias))


class ActivationOverflow(FloatingPointError):
    """A hidden activation of the MLP exceeded the fp16 range of the fused kernel's split operand (flag bit 4)."""


def _check(packed, check):
    """Read the kernel's flags word (one 4-byte copy, synchronises) and raise if an activation overflowed."""
    if check and int(packed.flags.item()) & 16:
        packed.flags.zero_()
        raise ActivationOverflow("aadff: a hidden activation of the PSF network exceeded 65504, the range of the fp16 hi/lo "
                                 "operand split of the fused kernel; use mlp_precision='torch' for this network")


def forward(packed, inp, mode, img=None, ks=0, slices=0, check=True, precision=0):
    """mode 0: [P,4] -> [P,n_out] normalised PSFs; mode 1: img [N,C,H,W] + inp [N*H*W,4] -> [N,C,H,W], or with
    slices = S: inp [N*S*H*W,4] (rows ordered [n][slice][y][x]) -> [N,C,S,H,W] in one launch.
    check: read back the saturation flag after the launch (a hidden activation above 65504 raises ActivationOverflow)."""
    dev = inp.device
    inp = _abi.f32c(inp, dev).reshape(-1, 4)
    P = inp.shape[0]
    st = _abi.stream_ptr(dev)
    with torch.cuda.device(dev):
        if mode == 0:
            out = torch.ops.aadff.psfnet_forward(inp, packed.wpack, packed.bias, list(packed.ins), list(packed.outs), packed.flags, int(precision))
            _check(packed, check)
            return out
        x = _abi.f32c(img, dev)
        N, Cc, H, W = x.shape
        out = torch.empty((N, Cc, slices, H, W), dtype=torch.float32, device=dev) if slices else torch.empty_like(x)
        if EVENT_HOOK is not None:
            EVENT_HOOK(True)
        _abi.call("aadff_psfnet_forward", _abi.ptr(inp), P, _abi.ptr(packed.wpack), _abi.ptr(packed.bias), packed.n,
                  packed.ins, packed.outs, 1, None, _abi.ptr(x), _abi.ptr(out), Cc, H, W, ks, int(slices), int(precision), _abi.ptr(packed.flags), st)
        if EVENT_HOOK is not None:
            EVENT_HOOK(False)
        _check(packed, check)
        return out


def render_rgbd(packed, img, depth, xs, ys, foc_z, d_min, d_range, ks, check=True, precision=0):
    """PSFNet.render / render_stack with the network input generated in the kernel (aadff_psfnet_render_rgbd):
    img [N,C,H,W], depth [N,H,W] mm, xs [W], ys [H], foc_z [N,S] -> [N,C,S,H,W]."""
    dev = img.device
    x, d = _abi.f32c(img, dev), _abi.f32c(depth, dev)
    N, Cc, H, W = x.shape
    fz = _abi.f32c(foc_z, dev).reshape(N, -1)
    S = fz.shape[1]
    xs, ys = _abi.f32c(xs, dev), _abi.f32c(ys, dev)
    assert d.shape == (N, H, W) and xs.shape == (W,) and ys.shape == (H,)
    inv_range = float(np.float32(1.0) / np.float32(d_range))          # tensor / python scalar = tensor * (1 / scalar) in ATen
    if EVENT_HOOK is not None:
        EVENT_HOOK(True)
    out = torch.ops.aadff.psfnet_render_rgbd(x, d, xs, ys, fz, float(d_min), inv_range, packed.wpack, packed.bias, list(packed.ins),
                                             list(packed.outs), ks, packed.flags, int(precision))
    if EVENT_HOOK is not None:
        EVENT_HOOK(False)
    _check(packed, check)
    return out
