"""Portable Float Map reader for the Middlebury disparity files that `0_warm_up_with_pfm.py` feeds to `PSFNet.render`
(reference: pfmreader.py:5-50; same function names, importable as `from pfmreader import read_and_clean_pfm` with this
package directory on the path).  Header `Pf` (one channel) or `PF` (three), `width height`, a scale whose sign gives the
byte order (negative: little endian); rows are stored bottom-up.  numpy only - nothing here is on the hot path."""
import re

import numpy as np


def read_pfm(file):
    """-> float32 array [H, W] or [H, W, 3], top row first."""
    with open(file, "rb") as f:
        header = f.readline().rstrip()
        if header not in (b"PF", b"Pf"):
            raise Exception("Not a PFM file.")
        color = header == b"PF"
        dim_match = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("ascii"))
        if not dim_match:
            raise Exception("Malformed PFM header.")
        width, height = map(int, dim_match.groups())
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.fromfile(f, endian + "f")
    shape = (height, width, 3) if color else (height, width)
    return np.flipud(np.reshape(data, shape))


def read_and_clean_pfm(file_path):
    """read_pfm with NaN / +-inf (Middlebury marks unknown disparities with inf) replaced by 0."""
    return np.nan_to_num(read_pfm(file_path), nan=0.0, posinf=0.0, neginf=0.0)


def disparity_to_depth_mm(disp, fx, baseline, doffs):
    """Middlebury calibration rule Z = f * baseline / (d + doffs) [mm] (0_warm_up_with_pfm.py:21-29), as a positive depth."""
    return fx * baseline / (disp + doffs)
