"""Script-compat surface of the renderer's consumers (reference: dff/__init__.py:1-6).

`from dff import *` is how 2_aber_aware_dff_aif.py:25 / 2_aber_aware_dff_dfv.py:25 obtain `get_lens`, `get_dataset`,
`select_focus_dist`, the dataset classes, `AiFDepthNet` and the `mask_*` metrics.  This package provides what belongs to the
rendering path (`dataset`, `factory`, `utils`).  The depth-from-focus network (`dff/AiFNet.py`) and the evaluation
metrics (`dff/metrics.py`) are consumers of rendered stacks, outside this package's scope:

* with `AADFF_REFERENCE_ROOT=/path/to/reference/checkout` (set by `python -m aadff.run_script`) those two modules are loaded
  from that checkout, file by file, as `dff.AiFNet` / `dff.metrics` and their public names are re-exported, so the
  reference's scripts find every name they use;
* without it each of their names is bound to a stub that raises an ImportError saying exactly that when it is called.
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

from .dataset import *      # noqa: F401,F403
from .factory import *      # noqa: F401,F403
from .utils import *        # noqa: F401,F403
from . import dataset, factory, utils   # noqa: F401  (the reference's star import also binds the submodules)

# public names of the two out-of-scope modules (dff/AiFNet.py, dff/metrics.py:10-160 of the reference)
_CONSUMER_NAMES = {
    "AiFNet": ("conv3d_bn", "trans3d_bn", "Mixed", "AiFDepthNet"),
    "metrics": ("abs_rel", "sq_rel", "mae", "mse", "rmse", "rmse_log", "accuracy_k", "get_bumpiness", "get_bumpiness_non_mask",
                "AIF_DepthNEt_abs_rel", "AIF_DepthNEt_sq_rel", "mask_abs_rel", "mask_sq_rel", "mask_mse", "mask_mae", "mask_rmse",
                "mask_rmse_log", "mask_accuracy_k", "mask_mse_w_conf", "mask_mae_w_conf", "mask_mse_w_conf_wo_mask",
                "mask_mae_w_conf_wo_mask", "batch_PSNR", "batch_SSIM", "mask_psnr", "mask_ssim"),
}


class _OutOfScope:
    """Placeholder for a consumer-side name: importable (so `from dff import *` succeeds), loud when used."""

    def __init__(self, name, module, why=None):
        self.__name__, self._module, self._why = name, module, why

    def _fail(self, *a, **k):
        raise ImportError(
            f"dff.{self._module}.{self.__name__} is a consumer of rendered focal stacks (reference dff/{self._module}.py), not part of "
            f"the MI355X rendering package. " + (self._why or "Set AADFF_REFERENCE_ROOT to a checkout of the reference (or run the script "
            "through `python -m aadff.run_script`) and it is loaded from there."))

    __call__ = _fail

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        self._fail()


def _load_consumers():
    root = _os.environ.get("AADFF_REFERENCE_ROOT", "")
    for mod, names in _CONSUMER_NAMES.items():
        path, loaded, why = _os.path.join(root, "dff", mod + ".py"), None, None
        if root and _os.path.isfile(path):
            spec = _ilu.spec_from_file_location(f"dff.{mod}", path)
            loaded = _ilu.module_from_spec(spec)
            _sys.modules[f"dff.{mod}"] = loaded
            try:
                keep = _sys.dont_write_bytecode
                _sys.dont_write_bytecode = True            # nothing is ever written into the user's (possibly read-only) checkout
                try:
                    spec.loader.exec_module(loaded)
                finally:
                    _sys.dont_write_bytecode = keep
            except Exception as e:                                  # e.g. skimage missing for metrics.py
                del _sys.modules[f"dff.{mod}"]
                loaded, why = None, f"Loading {path} failed: {type(e).__name__}: {e}"
        if loaded is not None:
            globals()[mod] = loaded
            for k, v in vars(loaded).items():
                if not k.startswith("_") and k not in globals():
                    globals()[k] = v
        else:
            for k in names:
                globals().setdefault(k, _OutOfScope(k, mod, why))


_load_consumers()
