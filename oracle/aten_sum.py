"""ORACLE (test infrastructure, not product code): the ORDER in which ATen's CPU kernel adds up `x.sum(0)` of a contiguous
float32 tensor [rows, ...] (aten/src/ATen/native/cpu/SumKernel.cpp, torch 2.10: cascade_sum -> vectorized_outer_sum), as a plain
numpy program.  The reference's chief-ray centre is such a sum over the 2048 rays of every field point
(`(ray.o * ray.ra.unsqueeze(-1)).sum(0)`, deeplens/optics.py:902-904); `aadff_strict_centroid` (csrc/strict.hip) follows this
program so that Lensgroup(parity="strict") gets the reference's centre bits without moving the rays to the host.
tests/test_oracle_golden.py checks it bit for bit against torch itself, for several shapes and thread counts.

What the kernel does, per output column (all accumulators float32):
  * columns below the last multiple of 32: rows are added in order into acc0; every 2^p rows (p = max(4, ceil_log2(rows) // 4))
    acc0 is added to acc1 and cleared, every 2^2p rows acc1 to acc2, ... (four levels); at the end ((acc0 + acc1) + acc2) + acc3;
  * the remaining columns: four interleaved partial sums (row % 4), each by the same cascade over rows // 4 rows, left-over
    rows added to partial 0, then ((p0 + p1) + p2) + p3.
ATen splits the columns over threads in multiples of 128 bytes, so which columns are "remaining" does not depend on the thread
count."""
import numpy as np


def ceil_log2(x):
    return 1 if x <= 2 else int(x - 1).bit_length()


def cascade(cols):
    """cols [rows, n] float32 -> [n]: the four-level cascade over the rows."""
    size = cols.shape[0]
    lp = max(4, ceil_log2(size) // 4)
    step, mask = 1 << lp, (1 << lp) - 1
    acc = np.zeros((4, cols.shape[1]), np.float32)
    i = 0
    while i + step <= size:
        for _ in range(step):
            acc[0] = acc[0] + cols[i]
            i += 1
        for j in range(1, 4):
            acc[j] = acc[j] + acc[j - 1]
            acc[j - 1] = 0
            if (i & (mask << (j * lp))) != 0:
                break
    while i < size:
        acc[0] = acc[0] + cols[i]
        i += 1
    for j in range(1, 4):
        acc[0] = acc[0] + acc[j]
    return acc[0]


def row_sum(cols):
    size = cols.shape[0]
    n4 = size // 4
    parts = [cascade(cols[k:4 * n4:4]) for k in range(4)]
    for i in range(4 * n4, size):
        parts[0] = parts[0] + cols[i]
    return ((parts[0] + parts[1]) + parts[2]) + parts[3]


def sum0(x):
    """x [rows, ...] float32 (C-contiguous) -> x.sum(0) with ATen's CPU summation order."""
    x = np.ascontiguousarray(x, np.float32)
    cols = x.reshape(x.shape[0], -1)
    nc = cols.shape[1]
    nv = (nc // 32) * 32
    out = np.empty(nc, np.float32)
    out[:nv] = cascade(cols[:, :nv])
    if nv < nc:
        out[nv:] = row_sum(cols[:, nv:])
    return out.reshape(x.shape[1:])
