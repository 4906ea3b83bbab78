"""Small helpers the reference scripts import from deeplens.utils (set_seed :95,
set_logger :107) plus torchvision-free `save_image` / `make_grid` with the semantics the
PSF path relies on (padding 0 => pure tiling; reference call site optics.py:1025)."""
import logging
import os
import random

import numpy as np
import torch


def set_seed(seed=0):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def set_logger(dir="./"):
    logger = logging.getLogger()
    logger.setLevel("DEBUG")
    fmt = logging.Formatter("%(asctime)s:%(levelname)s:%(message)s", "%Y-%m-%d %H:%M:%S")
    for h in (logging.StreamHandler(), logging.FileHandler(f"{dir}/output.log")):
        h.setFormatter(fmt)
        h.setLevel("INFO")
        logger.addHandler(h)


def make_grid(tensor, nrow=8, padding=0, pad_value=0.0, **kw):
    if padding != 0:
        raise NotImplementedError("only padding=0 (pure tiling) is used on the PSF path")
    if tensor.dim() == 3:
        tensor = tensor.unsqueeze(1)
    if tensor.shape[1] == 1:
        tensor = tensor.repeat(1, 3, 1, 1)
    n, c, h, w = tensor.shape
    xm = min(nrow, n)
    ym = (n + xm - 1) // xm
    grid = tensor.new_full((c, h * ym, w * xm), pad_value)
    for k in range(n):
        yy, xx = divmod(k, xm)
        grid[:, yy * h:(yy + 1) * h, xx * w:(xx + 1) * w] = tensor[k]
    return grid


def save_image(tensor, fp, **kw):
    """PNG writer via PIL (batches are tiled in one row)."""
    from PIL import Image
    t = tensor.detach().float().cpu()
    if t.dim() == 4:
        t = make_grid(t, nrow=t.shape[0])
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.shape[0] == 1:
        t = t.repeat(3, 1, 1)
    arr = t.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    Image.fromarray(arr).save(fp)
