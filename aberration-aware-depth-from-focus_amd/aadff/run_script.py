"""Run one of the reference's scripts, unchanged, on this package:

    python -m aadff.run_script /path/to/reference/0_warm_up.py [script args ...]

Why a launcher: CPython puts the SCRIPT's directory at `sys.path[0]`, in front of `PYTHONPATH`, so `python 0_warm_up.py`
started inside the reference checkout always imports the checkout's own `deeplens/` and `dff/`, whatever `PYTHONPATH`
says.  This launcher runs the script with `runpy` after placing this package's directory in FRONT of the script's
directory: `deeplens`, `dff` and `pfmreader` resolve here, everything else the script imports from its own directory
(`DFV_models`, `configs/`, `ckpt/`, `lenses/` as relative paths) resolves there, and `AADFF_REFERENCE_ROOT` is set so that
`from dff import *` also finds the consumer-side modules of that checkout (dff/__init__.py).  The working directory becomes
the script's directory (the scripts open `configs/*.yml`, `./lenses/...`, `./ckpt/...` relative to it) unless
`AADFF_KEEP_CWD=1`.
"""
import os
import runpy
import sys

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def prepare(script):
    """sys.path / environment for `script`; returns the absolute script path."""
    script = os.path.abspath(script)
    here = os.path.dirname(script)
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in (PKG, here)]
    sys.path[0:0] = [PKG, here]
    os.environ.setdefault("AADFF_REFERENCE_ROOT", here)
    for name in ("deeplens", "dff", "pfmreader"):                 # anything imported before the path was fixed
        mod = sys.modules.get(name)
        if mod is not None and not os.path.abspath(getattr(mod, "__file__", "") or "").startswith(PKG + os.sep):
            for k in [k for k in sys.modules if k == name or k.startswith(name + ".")]:
                del sys.modules[k]
    if os.environ.get("AADFF_KEEP_CWD", "0") != "1":
        os.chdir(here)
    return script


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        print(__doc__)
        return 2
    script = prepare(argv[0])
    sys.argv = [script] + argv[1:]
    sys.dont_write_bytecode = True          # the script's own imports resolve inside the reference checkout: leave no __pycache__ there
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
