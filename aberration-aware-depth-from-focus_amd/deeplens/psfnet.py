"""PSFNet (lens + MLP PSF surrogate) and the thin-lens baseline on MI355X.

Reference: deeplens/psfnet.py — PSFNet(Lensgroup) :14, render :393-441, pred :375-390,
depth2z :447-450, get_training_data :135-170, train_psfnet :79-132, ThinLens :489-570.
The per-pixel PSF application runs in csrc/conv.hip (local_psf_render); ray-traced
training targets come from the fused PSF kernel of csrc/trace.hip.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
from tqdm import tqdm

from .optics import *            # noqa: F401,F403  (star chain is part of the API surface)
from .optics import Lensgroup
from .basics import DeepObj
from .psfnet_arch import *       # noqa: F401,F403
from .psfnet_arch import MLP, initialize_weights
from .render_psf import *        # noqa: F401,F403
from .render_psf import local_psf_render
from .utils import make_grid, save_image   # noqa: F401

DMIN = 200     # [mm]
DMAX = 20000   # [mm]

# pixels per MLP chunk: bounds the [chunk,256] fp32 activations to ~0.5 GB
_MLP_CHUNK = 1 << 19


class _LinearInto(torch.autograd.Function):
    """y = x W^T + b whose backward writes dW and db straight into caller-provided buffers (views of a flat gradient
    buffer) and returns only dx: the parameters are not autograd leaves, nothing is accumulated or zeroed."""

    @staticmethod
    def forward(ctx, x, w, b, gw, gb):
        ctx.save_for_backward(x, w)
        ctx.gw, ctx.gb = gw, gb
        return nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        torch.mm(dy.t(), x, out=ctx.gw)
        torch.sum(dy, 0, out=ctx.gb)
        return (dy @ w if ctx.needs_input_grad[0] else None), None, None, None, None


class _LinearReluInto(torch.autograd.Function):
    """relu(x W^T + b) as ONE GEMM with a ReLU epilogue (torch._addmm_activation -> hipBLASLt), backward as _LinearInto
    after masking dy with the saved output."""

    @staticmethod
    def forward(ctx, x, w, b, gw, gb):
        y = torch._addmm_activation(b, x, w.t(), use_gelu=False)
        ctx.save_for_backward(x, w, y)
        ctx.gw, ctx.gb = gw, gb
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        from aadff import _abi
        dy = dy.contiguous()
        dz = torch.empty_like(dy)
        with torch.cuda.device(dy.device):               # ReLU backward + bias gradient in one launch (csrc/optim.hip)
            _abi.call("aadff_relu_bwd_bias", _abi.ptr(dy), _abi.ptr(y), _abi.ptr(dz), _abi.ptr(ctx.gb), dy.shape[0], dy.shape[1],
                      int(dy.dtype == torch.bfloat16), _abi.stream_ptr(dy.device))
        torch.mm(dz.t(), x, out=ctx.gw)
        return (dz @ w if ctx.needs_input_grad[0] else None), None, None, None, None


class _HeadLoss(torch.autograd.Function):
    """(pred, loss-gradient) of the network head in one launch: pred = F.normalize(sigmoid(z).float(), p=1), and the
    gradient of nn.MSELoss()(pred, target) w.r.t. z, computed in the forward and handed out in the backward (the scalar
    loss itself is not needed by the fit loop; `backward` scales by the incoming gradient of the dummy loss)."""

    @staticmethod
    def forward(ctx, z, target):
        from aadff import _abi
        z = z.contiguous()
        pred = torch.empty(z.shape, dtype=torch.float32, device=z.device)
        dz = torch.empty_like(z)
        with torch.cuda.device(z.device):
            _abi.call("aadff_psfnet_head_loss_grad", _abi.ptr(z), _abi.ptr(target), _abi.ptr(pred), _abi.ptr(dz), z.shape[0], z.shape[1],
                      int(z.dtype == torch.bfloat16), _abi.stream_ptr(z.device))
        ctx.save_for_backward(dz)
        ctx.mark_non_differentiable(pred)
        return pred, z.new_zeros(())                     # the second output stands for the loss in the autograd graph

    @staticmethod
    def backward(ctx, _gpred, gloss):
        (dz,) = ctx.saved_tensors
        return dz, None                                   # gloss == 1 (loss.backward()); kept out of the kernel chain


class _TrainStep:
    """One optimisation step of train_psfnet: MSE(MLP(inp), psf) -> backward -> AdamW, learning rate on the cosine
    schedule of torch.optim.lr_scheduler.CosineAnnealingLR(T_max=iters, eta_min=0).

    graph=False: the plain torch loop (torch.optim.AdamW, closed-form schedule set on the host).
    graph=True (GPU): the parameters are re-homed as views of ONE flat fp32 buffer and the optimiser is the fused HIP
    kernel `aadff_adamw_step` (csrc/optim.hip: AdamW + the cosine schedule from a device step counter, one launch instead
    of the ~60 of torch's capturable AdamW); zero-grad, forward, backward and that launch are captured once in a HIP graph
    after three eager warm-up steps and replayed.  With bf16 the forward/backward run on a bf16 COPY of the parameters that
    the optimiser kernel refreshes (same arithmetic as autocast, which casts the weights to bf16 every step and the
    gradients back, without its ~47 cast kernels per step); sigmoid output, L1 normalisation and the loss stay fp32.
    graph=True with bf16 on a Linear+ReLU chain of widths <= 256 (the reference's MLP): the whole step runs as the
    hand-written MFMA kernels of aadff/mlp_fit.py (25 launches in one HIP graph, no torch autograd); set
    AADFF_FIT_KERNELS=torch to keep the torch-GEMM path."""

    BETAS, EPS, WD = (0.9, 0.999), 1e-8, 0.01            # torch.optim.AdamW defaults (the reference passes only lr)

    def __init__(self, psfnet, lr, iters, bs, nout, dev, bf16, graph):
        self.net, self.cri, self.dev, self.bf16 = psfnet, nn.MSELoss(), dev, bf16 and dev.type == "cuda"
        self.use_graph, self.graph, self.pred = graph, None, None
        self.lr0, self.T, self.t = float(lr), max(1, int(iters)), 0
        if not graph:
            self.optim = torch.optim.AdamW(psfnet.parameters(), lr)
            return
        self.fused = None
        if self.bf16 and os.environ.get("AADFF_FIT_KERNELS", "hip") != "torch":
            from aadff import mlp_fit
            if hasattr(psfnet, "net") and mlp_fit.supported(psfnet, bs) and len(list(psfnet.parameters())) == 2 * sum(
                    isinstance(m, nn.Linear) for m in psfnet.net):
                self.fused = mlp_fit.FusedFit(psfnet, lr, iters, bs, dev)
                return
        self.inp = torch.zeros(bs, 4, device=dev)
        self.psf = torch.zeros(bs, nout, device=dev)
        params = list(psfnet.parameters())
        n = sum(p.numel() for p in params)
        self.n = n
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.m, self.v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.scal = torch.zeros(4, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in params:                                  # the module's tensors become views of the flat buffer
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1).float())
                p.data = self.flat[off:off + k].view_as(p)
                off += k
        self.flat16 = self.flat.to(torch.bfloat16) if self.bf16 else None
        src = self.flat16 if self.bf16 else self.flat
        self.gbuf = torch.zeros(n, dtype=src.dtype, device=dev)
        self.wb, off = {}, 0                                   # id(module parameter) -> (tensor the layers use, its gradient view)
        for p in params:
            k = p.numel()
            # requires_grad only so that the layer outputs join the autograd graph; _LinearInto returns no gradient for them
            self.wb[id(p)] = (src[off:off + k].view_as(p).requires_grad_(True), self.gbuf[off:off + k].view_as(p))
            off += k

    def _forward(self, inp, target=None):
        if not self.use_graph:
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.bf16):
                return self.net(inp)
        # graph mode: the layers run through _LinearInto, whose backward WRITES dW / db into the views of the flat gradient
        # buffer (autograd's own accumulation would add 22 `grad += new` kernels and need the buffer zeroed every step)
        h = inp.to(torch.bfloat16) if self.bf16 else inp
        mods = list(self.net.net)
        for i, mod in enumerate(mods):
            if isinstance(mod, nn.Linear):
                w, b = self.wb[id(mod.weight)], self.wb[id(mod.bias)]
                fused = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU) and mod.bias is not None
                h = (_LinearReluInto if fused else _LinearInto).apply(h, w[0], b[0], w[1], b[1])
            elif isinstance(mod, nn.ReLU):
                if not (i > 0 and isinstance(mods[i - 1], nn.Linear) and mods[i - 1].bias is not None):
                    h = torch.relu(h)
            elif isinstance(mod, nn.Sigmoid):
                if i == len(mods) - 1 and target is not None and h.shape[-1] <= 128:
                    return _HeadLoss.apply(h, target)     # (pred, loss stand-in): sigmoid + L1 normalise + MSE gradient fused
                h = torch.sigmoid(h)
            else:
                h = mod(h)
        return nn.functional.normalize(h.float(), p=1, dim=-1), None

    def _body(self, inp, psf):
        if not self.use_graph:
            pred = self._forward(inp)
            self.cri(pred.float(), psf).backward()
            self.optim.step()
            return pred
        import ctypes as C_
        from aadff import _abi
        pred, loss = self._forward(inp, psf)
        if loss is None:
            loss = self.cri(pred.float(), psf)
        loss.backward()                                       # _LinearInto / _LinearReluInto write every dW / db into gbuf
        with torch.cuda.device(self.dev):
            _abi.call("aadff_adamw_step", _abi.ptr(self.flat), _abi.ptr(self.gbuf), int(self.bf16), _abi.ptr(self.m), _abi.ptr(self.v),
                      _abi.ptr(self.flat16), self.n, _abi.ptr(self.step_dev), _abi.ptr(self.scal), C_.c_float(self.lr0), self.T, C_.c_float(self.BETAS[0]),
                      C_.c_float(self.BETAS[1]), C_.c_float(self.EPS), C_.c_float(self.WD), _abi.stream_ptr(self.dev))
        return pred

    def __call__(self, inp, psf):
        if not self.use_graph:
            self.optim.zero_grad()
            pred = self._body(inp, psf)
            self.t += 1
            new_lr = 0.5 * self.lr0 * (1.0 + math.cos(math.pi * min(self.t, self.T) / self.T))
            for g in self.optim.param_groups:
                g["lr"] = new_lr
            return pred
        if self.fused is not None:
            self.t += 1
            return self.fused(inp, psf)
        self.inp.copy_(inp)
        self.psf.copy_(psf)
        if self.t < 3 or self.use_graph == "eager-static":    # eager warm-up on the capture stream (torch's capture recipe)
            self.side = getattr(self, "side", None) or torch.cuda.Stream(self.dev)
            self.side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(self.side):
                pred = self._body(self.inp, self.psf)
            torch.cuda.current_stream(self.dev).wait_stream(self.side)
        else:
            if self.graph is None:                            # capture records only: the replay below runs this step
                graph = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(graph, stream=self.side):
                        self.pred = self._body(self.inp, self.psf)
                    self.graph = graph
                except RuntimeError as e:                      # capture refused: run this and all later steps eagerly
                    import logging
                    logging.getLogger(__name__).warning("train step not graph-capturable (%s): running eagerly", e)
                    self.use_graph, self.graph = "eager-static", None
                    return self.__call__(inp, psf)
            self.graph.replay()
            pred = self.pred
        self.t += 1
        return pred


class PSFNet(Lensgroup):
    def __init__(self, filename, model_name="mlp", kernel_size=11, sensor_res=(512, 512), device="cuda"):
        super().__init__(filename=filename, sensor_res=sensor_res, device=device)
        self.in_features = 4
        self.kernel_size = kernel_size
        self.model_name = model_name
        self.init_net()
        self.spp = 4096
        self.patch_size = 64
        self.psf_grid = [sensor_res[0] // self.patch_size, sensor_res[1] // self.patch_size]
        self.d_max, self.d_min = -DMAX, -DMIN
        self.foc_d_arr = np.array([-500, -600, -700, -800, -900, -1000, -1250, -1500, -1750, -2000,
                                   -2500, -3000, -4000, -5000, -6000, -8000, -10000, -12000, -15000, -20000])
        self.foc_z_arr = (self.foc_d_arr - self.d_min) / (self.d_max - self.d_min)
        # "fp32" (default): inference runs the fused HIP kernel (csrc/psfnet.hip: the whole MLP + sigmoid +
        #   L1-normalise + per-pixel gather in one launch, fp32 operands carried as exact fp16 hi/lo pairs on
        #   MFMA with fp32 accumulation); rendered image within 1e-4 rel-L2 of the reference.
        # "torch": the same in stock torch fp32 ops + the HIP gather (what training always uses).
        # "fp16": the fused kernel in single-pass fp16 (one MFMA per product instead of three, half the LDS): PSFs within
        #   ~5e-4 relative of fp32, ~2.5x faster - opt-in, for producing training stacks on the fly.
        # "bf16": MLP GEMMs on bf16 MFMA under autocast (sigmoid / L1-normalise / gather stay fp32); the
        #   surrogate PSFs then differ by ~1e-3, far below the MLP's own fit error - opt-in.
        self.mlp_precision = "fp32"
        self._packed = None

    # ------------------------------------------------------------------ network
    def init_net(self):
        if self.model_name != "mlp":
            raise Exception("Unsupported PSF network architecture.")   # mlpconv/siren never construct in the reference either
        ks = self.kernel_size
        self.psfnet = MLP(in_features=4, out_features=ks ** 2, hidden_features=256, hidden_layers=8)
        self.psfnet.apply(initialize_weights)
        self.psfnet.to(self.device)

    def load_net(self, net_path):
        """Plain state_dict checkpoints, keys net.{0,2,..,20}.* (reference: psfnet.py:73-76;
        map_location added so CUDA-saved files load anywhere)."""
        self.psfnet.load_state_dict(torch.load(net_path, map_location=self.device))

    def _fused(self, dev):
        """PackedMLP for the fused kernel, or None when it does not apply (CPU, autograd, other modes/shapes)."""
        from aadff import psfnet_pack
        if self.mlp_precision not in ("fp32", "fp16") or torch.device(dev).type != "cuda" or not psfnet_pack.supported(self.psfnet):
            return None
        if self._packed is None or self._packed.key != psfnet_pack.PackedMLP.key_of(self.psfnet) or self._packed.wpack.device != torch.device(dev):
            self._packed = psfnet_pack.PackedMLP(self.psfnet, torch.device(dev))
        return self._packed

    def pred(self, inp):
        if not torch.is_grad_enabled() and inp.is_cuda:
            packed = self._fused(inp.device)
            if packed is not None:
                from aadff import psfnet_pack
                psf = psfnet_pack.forward(packed, inp.reshape(-1, inp.shape[-1]), 0, precision=int(self.mlp_precision == "fp16"))
                return psf.reshape(*inp.shape[:-1], self.kernel_size, self.kernel_size)
        psf = self.psfnet(inp)
        return psf.reshape(*psf.shape[:-1], self.kernel_size, self.kernel_size)

    def _pred_chunked(self, o):
        flat = o.reshape(-1, o.shape[-1])
        out = torch.empty((flat.shape[0], self.kernel_size ** 2), dtype=torch.float32, device=flat.device)
        bf16 = self.mlp_precision == "bf16" and flat.is_cuda
        for i in range(0, flat.shape[0], _MLP_CHUNK):
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
                h = self.psfnet.net[:-1](flat[i:i + _MLP_CHUNK]) if bf16 else None
            out[i:i + _MLP_CHUNK] = (torch.nn.functional.normalize(torch.sigmoid(h.float()), p=1, dim=-1)
                                     if bf16 else self.psfnet(flat[i:i + _MLP_CHUNK]))
        return out.reshape(*o.shape[:-1], self.kernel_size, self.kernel_size)

    @torch.no_grad()
    def render(self, img, depth, foc_dist):
        """Aberrated, defocused image from an all-in-focus image and a depth map (mm, < 0):
        per-pixel (x, y, z, foc_z) -> MLP -> per-pixel PSF -> local_psf_render."""
        dev = next(self.psfnet.parameters()).device       # coordinate grids are built where the MLP runs
        depth = depth.to(dev)
        if len(img.shape) == 3:
            H, W = depth.shape
            z = self.depth2z(depth)
            x, y = self._field_grid(H, W, dev)
            foc_z = self.depth2z(torch.full_like(depth, foc_dist))
            o = torch.stack((x, y, z, foc_z), -1)
        elif len(img.shape) == 4:
            N, C, H, W = img.shape
            z = self.depth2z(depth).squeeze(1)
            x, y = self._field_grid(H, W, dev)
            x, y = x.unsqueeze(0).expand(N, H, W), y.unsqueeze(0).expand(N, H, W)
            foc_z = self.depth2z(foc_dist.to(dev).reshape(N, 1, 1).expand(N, H, W))
            o = torch.stack((x, y, z, foc_z), -1).float()
        else:
            raise ValueError("img should be [C,H,W] or [N,C,H,W]")
        packed = self._fused(dev)
        if packed is not None:
            from aadff import psfnet_pack
            x = img if len(img.shape) == 4 else img.unsqueeze(0)
            out = psfnet_pack.forward(packed, o.reshape(-1, 4), 1, img=x.to(dev), ks=self.kernel_size, precision=int(self.mlp_precision == "fp16"))
            out = out if len(img.shape) == 4 else out.squeeze(0)
            return out.to(img.device)
        psf = self._pred_chunked(o)
        return local_psf_render(img, psf, self.kernel_size)

    @torch.no_grad()
    def render_stack(self, img, depth, foc_dists):
        """[N,C,S,H,W] focal stack of img [N,C,H,W] for depth [N,1,H,W] (mm, < 0) and focus distances [N,S] (mm, < 0):
        the loop `stack([render(img, depth, foc_dists[:, i]) for i], dim=2)` of 2_aber_aware_dff_aif.py:104-114 as
        ONE fused launch when the fused kernel applies (the (x, y, z, foc_z) rows of psfnet.py:424-437 are generated in
        the kernel from the depth map: no [N,S,H,W,4] tensor), else that loop."""
        dev = next(self.psfnet.parameters()).device
        N, C, H, W = img.shape
        foc_dists = foc_dists.to(dev).reshape(N, -1)
        S = foc_dists.shape[1]
        packed = self._fused(dev)
        if packed is None:
            return torch.stack([self.render(img, depth, foc_dists[:, i]) for i in range(S)], dim=2)
        from aadff import psfnet_pack
        xs, ys = self._field_axes(H, W, dev)
        out = psfnet_pack.render_rgbd(packed, img.to(dev), depth.to(dev).reshape(N, H, W), xs, ys, self.depth2z(foc_dists.float()),
                                      float(self.d_min), float(self.d_max - self.d_min), self.kernel_size,
                                      precision=int(self.mlp_precision == "fp16"))
        return out.to(img.device)

    def _field_axes(self, H, W, dev):
        """linspace(-1,1,W) and linspace(1,-1,H) (reference: psfnet.py:427-431) on `dev`, cached per (H, W, device)."""
        key = (H, W, str(dev))
        if getattr(self, "_axes_key", None) != key:
            self._axes, self._axes_key = (torch.linspace(-1, 1, W).to(dev), torch.linspace(1, -1, H).to(dev)), key
        return self._axes

    def _field_grid(self, H, W, dev):
        """x = linspace(-1,1,W) over columns, y = linspace(1,-1,H) over rows (reference: psfnet.py:427-431),
        cached per (H, W, device)."""
        key = (H, W, str(dev))
        if getattr(self, "_grid_key", None) != key:
            x, y = torch.meshgrid(torch.linspace(-1, 1, W), torch.linspace(1, -1, H), indexing="xy")
            self._grid_xy, self._grid_key = (x.to(dev), y.to(dev)), key
        return self._grid_xy

    def depth2z(self, depth):
        return torch.clamp((depth - self.d_min) / (self.d_max - self.d_min), min=0, max=1)

    def z2depth(self, z):
        return z * (self.d_max - self.d_min) + self.d_min

    # ------------------------------------------------------------------ fitting
    def get_training_data(self, bs=256, spp=4096):
        """One focus distance, `bs` random (x, y, z) points and their ray-traced PSFs
        (reference: psfnet.py:135-170; RNG order: np.random.choice, refocus draws,
        rand x, rand y, randn z, psf draws).  Returns device tensors (inp [bs,4], psf [bs,ks*ks]).
        With wavelength DEFAULT_WAVE and host sampling this runs through the pipelined producer
        (aadff/training.py: two launches, no copies); the reference's asserts are raised here, synchronously
        (set `self.defer_flag_check = True` to defer them to the producer's periodic poll; `check_flags()` is the lens's
        synchronous check, a METHOD - optics.py - and must not be shadowed by an attribute)."""
        plan = self._training_plan(bs, spp)
        if plan is None:
            return self._get_training_data_unpipelined(bs, spp)
        inp, psf = plan.next(prefetch=False)
        if not getattr(self, "defer_flag_check", False):
            plan.check_flags()
        return inp.clone(), psf.clone()

    def _training_plan(self, bs, spp):
        if self.sampler.on_device or self._gpu().type != "cuda":
            return None
        key = (int(bs), int(spp), self.kernel_size, tuple(self.sensor_res), str(self._gpu()), id(self.sampler))
        if getattr(self, "_tplan_key", None) != key:
            from aadff.training import TrainingDataPlan
            self._tplan, self._tplan_key = TrainingDataPlan(self, bs, spp), key
        return self._tplan

    def _get_training_data_unpipelined(self, bs=256, spp=4096):
        foc_z = np.random.choice(self.foc_z_arr)
        foc_dist = foc_z * (self.d_max - self.d_min) + self.d_min
        self.refocus(depth=foc_dist)
        x = (torch.rand(bs) - 0.5) * 2
        y = (torch.rand(bs) - 0.5) * 2
        z_gauss = torch.clamp(torch.randn(bs), min=-3, max=3)
        z = torch.zeros_like(z_gauss)
        z[z_gauss > 0] = (1 - foc_z) * z_gauss[z_gauss > 0] / 3 + foc_z
        z[z_gauss < 0] = foc_z * z_gauss[z_gauss < 0] / 3 + foc_z
        inp = torch.stack((x, y, z, torch.full_like(x, foc_z)), dim=-1)
        points = torch.stack((x, y, self.z2depth(z)), dim=-1)
        psf = self.psf(points=points, ks=self.kernel_size, spp=spp)
        return inp, psf.view(bs, -1)

    def train_psfnet(self, iters=10000, bs=128, lr=1e-4, spp=2048, evaluate_every=1000, result_dir="./results/temp",
                     autocast_bf16=False, graph=True):
        """Fit the MLP to ray-traced PSFs generated on the fly (reference: psfnet.py:79-132:
        MSE, AdamW, cosine schedule; checkpoints are plain state_dicts).  On a GPU the forward/backward/AdamW
        step (a few dozen small kernels on a 128-row batch) is captured once in a HIP graph and replayed;
        `graph=False` runs it eagerly.  Batches come from the pipelined producer: the host only draws the random
        numbers and enqueues two launches and one graph replay per iteration."""
        psfnet = self.psfnet
        dev = next(psfnet.parameters()).device
        step = _TrainStep(psfnet, lr, int(iters), bs, self.kernel_size ** 2, dev, autocast_bf16, graph and dev.type == "cuda")
        plan = self._training_plan(bs, spp)
        for i in tqdm(range(iters + 1)):
            if plan is not None:
                inp, psf = plan.next(prefetch=i < iters)    # device views; batch i+1 is traced while the step below runs
            else:
                inp, psf = self.get_training_data(bs=bs, spp=spp)
            pred = step(inp.to(dev), psf.to(dev))
            self._packed = None        # graph replays update the weights without bumping tensor versions: repack on next use
            if (i + 1) % evaluate_every == 0:
                ks = self.kernel_size
                if plan is not None:
                    plan.check_flags()
                psf = psf.to(dev)
                both = torch.stack((psf[:5].view(-1, ks, ks), pred[:5].detach().float().view(-1, ks, ks)), 1)
                save_image(make_grid(both.reshape(-1, 1, ks, ks) / both.max(), nrow=2), f"{result_dir}/iter{i + 1}.png")
                torch.save(psfnet.state_dict(), f"{result_dir}/iter{i + 1}_PSFNet_{self.model_name}.pkl")
        if plan is not None:
            plan.check_flags()
        torch.save(psfnet.state_dict(), f"{result_dir}/PSFNet_{self.model_name}.pkl")

    def _grid_points(self, psf_grid):
        """Centres of a psf_grid[0] x psf_grid[1] tiling of the normalised field, row-major from the top left
        (reference: psfnet.py:227-233, 320-325)."""
        x, y = torch.meshgrid(torch.linspace(-1 + 1 / (2 * psf_grid[1]), 1 - 1 / (2 * psf_grid[1]), psf_grid[1]),
                              torch.linspace(1 - 1 / (2 * psf_grid[0]), -1 + 1 / (2 * psf_grid[0]), psf_grid[0]), indexing="xy")
        return x.reshape(-1), y.reshape(-1)

    @torch.no_grad()
    def calc_psf_map(self, foc_dist, depth, psf_grid=(11, 11)):
        """Ray-traced PSF grid [3, psf_grid[0]*ks, psf_grid[1]*ks] (green PSF tiled, repeated over the channels by make_grid)
        of the lens focused to `foc_dist` for points at `depth` (reference: psfnet.py:215-243; its `self.psf(...,
        kernel_size=ks)` call names a keyword Lensgroup.psf does not have - `ks` is meant)."""
        self.refocus(depth=foc_dist)
        x, y = self._grid_points(psf_grid)
        o = torch.stack((x, y, torch.full_like(x, depth)), dim=-1)
        psf = self.psf(points=o, ks=self.kernel_size, spp=self.spp, center=True)            # [psf_grid^2, ks, ks]
        return make_grid(psf.unsqueeze(1), nrow=psf_grid[1], padding=0)

    @torch.no_grad()
    def get_training_psf_map(self, bs=8, psf_grid=(11, 11), psf_map_size=(128, 128)):
        """`bs` ray-traced PSF maps at one random focus distance and random depths around it, resized to `psf_map_size`:
        (inp [B,2] = (z, foc_z), psf_map_batch [B,3,*psf_map_size]) - the training data of the reference's PSF-map
        architectures (psfnet.py:172-211; RNG order: np.random.choice, torch.randn, then per map the refocus and PSF draws).
        torchvision's `F.resize` on a tensor is antialiased bilinear interpolation."""
        foc_z = np.random.choice(self.foc_z_arr)
        foc_dist = foc_z * (self.d_max - self.d_min) + self.d_min
        z_gauss = torch.clamp(torch.randn(bs), min=-3, max=3)
        z = torch.zeros_like(z_gauss)
        z[z_gauss > 0] = (1 - foc_z) * z_gauss[z_gauss > 0] / 3 + foc_z
        z[z_gauss < 0] = foc_z * z_gauss[z_gauss < 0] / 3 + foc_z
        depth = self.z2depth(z)
        inp = torch.stack((z, torch.full_like(z, foc_z)), dim=-1)
        maps = torch.stack([self.calc_psf_map(foc_dist, float(d), psf_grid=psf_grid) for d in depth], dim=0)
        maps = torch.nn.functional.interpolate(maps, size=tuple(psf_map_size), mode="bilinear", align_corners=False, antialias=True)
        return inp, maps

    def vis_psf_map(self, psf, filename=None):
        """Picture of a [N,N,k,k] / [N,N,k^2] / [N,k,k] PSF set (reference: psfnet.py:456-486)."""
        import matplotlib
        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
        if len(psf.shape) == 3 and psf.shape[0] == psf.shape[1] and int(round(psf.shape[2] ** 0.5)) ** 2 == psf.shape[2] and psf.shape[1] != psf.shape[2]:
            k = int(round(psf.shape[2] ** 0.5))
            psf = psf.reshape(psf.shape[0], psf.shape[1], k, k)
        if len(psf.shape) == 4:
            n = psf.shape[0]
            fig, axs = plt.subplots(n, n, squeeze=False)
            for i in range(n):
                for j in range(n):
                    axs[i, j].imshow(psf[i, j].detach().float().cpu().numpy(), vmin=0.0, vmax=0.1)
        elif len(psf.shape) == 3:
            n = psf.shape[0]
            fig, axs = plt.subplots(1, n, squeeze=False)
            for i in range(n):
                axs[0, i].imshow(psf[i].detach().float().cpu().numpy(), vmin=0.0, vmax=0.1)
        else:
            raise ValueError("vis_psf_map: PSF of shape [N,N,k,k], [N,N,k^2] or [N,k,k] expected")
        if filename is None:
            plt.show()
        else:
            fig.savefig(filename, bbox_inches="tight")
        plt.close(fig)

    @torch.no_grad()
    def evaluate_psf_score(self, vis=False, evaluate_model=None, result_dir="./"):
        """Mean L1 / L2 distance between ray-traced and predicted PSFs over every focus distance of `foc_z_arr`, 40 depths
        and the `psf_grid` field points (reference: psfnet.py:305-366, with its `self.psf(o=..., kernel_size=...)` call
        spelled as Lensgroup.psf takes it).  Prints the reference's line and also returns (avg_l1, avg_l2)."""
        psf_grid, ks, spp = self.psf_grid, self.kernel_size, self.spp
        psfnet = self.psfnet
        psfnet.eval()
        evaluate_model = getattr(self, "evaluate_model", "mlp") if evaluate_model is None else evaluate_model
        if evaluate_model != "mlp":
            raise Exception("Unimplemented")
        dev = next(psfnet.parameters()).device
        x, y = self._grid_points(psf_grid)
        l1_error, l2_error = [], []
        for foc_z in tqdm(self.foc_z_arr):
            foc_dist = foc_z * (self.d_max - self.d_min) + self.d_min
            self.refocus(depth=foc_dist)
            for z in np.linspace(0, 1, 40, endpoint=True):
                depth = z * (self.d_max - self.d_min) + self.d_min
                o = torch.stack((x, y, torch.full_like(x, depth)), dim=-1)
                psf_gt = self.psf(points=o, ks=ks, spp=spp, center=True).to(dev)             # [psf_grid^2, ks, ks]
                inp = torch.stack((x, y, torch.full_like(x, z), torch.full_like(x, foc_z)), dim=-1).to(dev)
                psf_pred = psfnet(inp).view(-1, ks, ks)
                l2_error.append(torch.sum((psf_gt - psf_pred) ** 2) / psf_gt.numel())
                l1_error.append(torch.sum((psf_gt - psf_pred).abs()) / psf_gt.numel())
                if vis:
                    gt_map = make_grid(psf_gt.unsqueeze(1), nrow=psf_grid[1], padding=0)
                    pred_map = make_grid(psf_pred.unsqueeze(1), nrow=psf_grid[1], padding=0)
                    scale = 1 / max(gt_map.max(), pred_map.max())
                    save_image(gt_map * scale, f"{result_dir}/psf_foc{-foc_dist}_depth{-depth}_gt.png")
                    save_image(pred_map * scale, f"{result_dir}/psf_foc{-foc_dist}_depth{-depth}_pred.png")
        avg_l2_error = sum(l2_error) / len(l2_error)
        avg_l1_error = sum(l1_error) / len(l1_error)
        print(f"avg l1 error: {avg_l1_error}, avg l2 error: {avg_l2_error}.")
        return float(avg_l1_error), float(avg_l2_error)

    @torch.no_grad()
    def evaluate_psf(self, result_dir="./"):
        """Ray-traced vs predicted PSFs at three field points, focus 1.5 m, depths
        1.2/1.5/2 m (reference: psfnet.py:248-302; images written as PNG tiles)."""
        ks = self.kernel_size
        self.psfnet.eval()
        dev = next(self.psfnet.parameters()).device
        x = torch.Tensor([0, 0.6, 0.98])
        foc_dist = -1500.0
        foc_z = float(self.depth2z(torch.tensor(foc_dist)))
        self.refocus(depth=foc_dist)
        for depth in (-1200.0, -1500.0, -2000.0):
            pts = torch.stack((x, x, torch.full_like(x, depth)), dim=-1)
            gt = self.psf(points=pts, ks=ks, center=True)
            z = float(self.depth2z(torch.tensor(depth)))
            inp = torch.stack((x, x, torch.full_like(x, z), torch.full_like(x, foc_z)), dim=-1).to(dev)
            pred = self.psfnet(inp).view(-1, ks, ks)
            both = torch.cat((gt.to(dev), pred), 0)
            save_image(make_grid((both / both.max()).unsqueeze(1), nrow=3), f"{result_dir}/foc{-foc_dist}_depth{-depth}.png")


class ThinLens(DeepObj):
    """Gaussian circle-of-confusion baseline (reference: psfnet.py:489-570)."""

    def __init__(self, foc_len, fnum, kernel_size, sensor_size, sensor_res, device="cpu"):
        self.d_max, self.d_min = DMAX, DMIN
        self.kernel_size = kernel_size
        self.foc_len, self.fnum = foc_len, fnum
        self.sensor_size, self.sensor_res = sensor_size, sensor_res
        self.ps = self.sensor_size[0] / self.sensor_res[0]
        self.device = device

    def coc(self, depth, foc_dist):
        if (depth < 0).any():
            depth, foc_dist = -depth, -foc_dist
        depth = torch.clamp(depth, self.d_min, self.d_max)
        coc = self.foc_len / self.fnum * torch.abs(depth - foc_dist) / depth * self.foc_len / (foc_dist - self.foc_len)
        return torch.clamp(coc / self.ps, min=0.1)

    @torch.no_grad()
    def render(self, img, depth, foc_dist):
        """img [N,C,H,W], depth [N,1,H,W], foc_dist [N] -> [N,C,H,W] (reference: psfnet.py:549-570).  One HIP kernel:
        the per-pixel Gaussian PSF is evaluated inside the gather (aadff_thinlens_render), the [N,H,W,ks,ks] PSF tensor
        of the reference is never built.  `render_psf_tensor` keeps the tensor form (tests, odd shapes)."""
        if len(img.shape) != 4:
            raise ValueError("ThinLens.render needs [N,C,H,W] (the reference's 3-D branch calls methods ThinLens lacks)")
        N, C, H, W = img.shape
        ks = self.kernel_size
        if C > 4 or ks not in (3, 5, 7, 9, 11, 13):
            return self.render_psf_tensor(img, depth, foc_dist)
        from aadff import _abi, ops  # noqa: F401
        _abi.require_gpu()
        dev = img.device if img.is_cuda else torch.device("cuda", torch.cuda.current_device())
        out = torch.ops.aadff.thinlens_render(_abi.f32c(img, dev), _abi.f32c(depth, dev), _abi.f32c(foc_dist, dev), ks, float(self.foc_len),
                                              float(self.fnum), float(self.ps), float(self.d_min), float(self.d_max))
        return out.to(img.device)

    @torch.no_grad()
    def render_psf_tensor(self, img, depth, foc_dist):
        """The reference's literal form: build the [N,H,W,ks,ks] Gaussian PSFs with torch ops, then local_psf_render."""
        ks, dev = self.kernel_size, img.device
        N, C, H, W = img.shape
        fd = foc_dist.unsqueeze(-1).unsqueeze(-1).unsqueeze(-1).repeat(1, 1, H, W)
        x, y = torch.meshgrid(torch.linspace(-ks / 2 + 1 / 2, ks / 2 - 1 / 2, ks),
                              torch.linspace(ks / 2 - 1 / 2, -ks / 2 + 1 / 2, ks), indexing="xy")
        x, y = x.to(dev), y.to(dev)
        rad = self.coc(depth, fd).squeeze(1).unsqueeze(-1).unsqueeze(-1) / 2
        psf = torch.exp(-(x ** 2 + y ** 2) / 2 / rad ** 2) / (2 * np.pi * rad ** 2)
        psf = psf * (x ** 2 + y ** 2 < rad ** 2)
        psf = psf / psf.sum((-1, -2)).unsqueeze(-1).unsqueeze(-1)
        return local_psf_render(img, psf, ks)
