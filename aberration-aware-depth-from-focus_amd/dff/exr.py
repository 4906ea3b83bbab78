"""Minimal OpenEXR scan-line reader (numpy + zlib) for the `disp.exr` files of the FlyingThings3D focal-stack set
(reference: dff/dataset.py:73, `cv.imread(path, IMREAD_ANYCOLOR | IMREAD_ANYDEPTH)`; OpenCV's EXR codec and the OpenEXR
library are not in this image, and nothing here is on the hot path).

Supported: single-part scan-line files, compression NONE / RLE / ZIPS / ZIP (Blender's default), channel types
HALF / FLOAT / UINT, no subsampling.  Everything else (tiles, multi-part, deep data, PIZ / PXR24 / B44 / DWA) raises
NotImplementedError naming the feature.  File layout per the OpenEXR file-format description: magic, version, attribute
list, offset table, then per chunk `y, size, data` with the rows of a chunk stored channel by channel in alphabetical
channel order; ZIP / RLE data is byte-delta predicted and split into even / odd bytes before compression.
"""
import struct
import zlib

import numpy as np

_MAGIC = 20000630
_PIX = {0: np.dtype("<u4"), 1: np.dtype("<f2"), 2: np.dtype("<f4")}
_LINES = {0: 1, 1: 1, 2: 1, 3: 16}            # NONE, RLE, ZIPS, ZIP
_NAMES = {4: "PIZ", 5: "PXR24", 6: "B44", 7: "B44A", 8: "DWAA", 9: "DWAB"}


def _cstr(buf, pos):
    end = buf.index(b"\0", pos)
    return buf[pos:end].decode("latin-1"), end + 1


def _unpredict(raw):
    """Inverse of OpenEXR's byte reordering + delta predictor (ImfZip.cpp / ImfRle.cpp): cumulative sum of (b - 128)
    modulo 256, then first half = even bytes, second half = odd bytes."""
    t = np.frombuffer(raw, dtype=np.uint8).astype(np.int64)
    if t.size > 1:
        t[1:] -= 128
        t = np.cumsum(t) & 0xFF
    t = t.astype(np.uint8)
    half = (t.size + 1) // 2
    out = np.empty(t.size, dtype=np.uint8)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def _unrle(data, want):
    out = bytearray()
    i = 0
    while i < len(data):
        n = struct.unpack_from("b", data, i)[0]
        i += 1
        if n < 0:
            out += data[i:i - n]
            i += -n
        else:
            out += bytes([data[i]]) * (n + 1)
            i += 1
    if len(out) != want:
        raise ValueError(f"EXR RLE chunk decodes to {len(out)} bytes, expected {want}")
    return bytes(out)


def read_exr(path):
    """-> float32 array [H, W] (one channel) or [H, W, C] with the channels in OpenCV's order: B, G, R(, A) for colour
    files, alphabetical otherwise."""
    with open(path, "rb") as f:
        buf = f.read()
    magic, version = struct.unpack_from("<ii", buf, 0)
    if magic != _MAGIC:
        raise ValueError(f"{path}: not an OpenEXR file")
    for bit, what in ((0x200, "tiled"), (0x800, "deep-data"), (0x1000, "multi-part")):
        if version & bit:
            raise NotImplementedError(f"{path}: {what} EXR files are not supported")
    pos, attrs = 8, {}
    while buf[pos] != 0:
        name, pos = _cstr(buf, pos)
        typ, pos = _cstr(buf, pos)
        size = struct.unpack_from("<i", buf, pos)[0]
        attrs[name] = (typ, buf[pos + 4:pos + 4 + size])
        pos += 4 + size
    pos += 1
    chans, cb, q = [], attrs["channels"][1], 0
    while cb[q] != 0:
        name, q = _cstr(cb, q)
        ptype, _plin, xs, ys = struct.unpack_from("<iB3xii", cb, q)
        q += 16
        if (xs, ys) != (1, 1):
            raise NotImplementedError(f"{path}: subsampled channel '{name}'")
        chans.append((name, _PIX[ptype]))
    comp = attrs["compression"][1][0]
    if comp not in _LINES:
        raise NotImplementedError(f"{path}: {_NAMES.get(comp, comp)} compression is not supported (NONE, RLE, ZIPS, ZIP are)")
    xmin, ymin, xmax, ymax = struct.unpack("<4i", attrs["dataWindow"][1])
    W, H = xmax - xmin + 1, ymax - ymin + 1
    lines = _LINES[comp]
    n_chunks = (H + lines - 1) // lines
    offsets = struct.unpack_from(f"<{n_chunks}Q", buf, pos)
    row_bytes = sum(dt.itemsize for _, dt in chans) * W
    planes = {name: np.empty((H, W), dtype=np.float32) for name, _ in chans}
    for off in offsets:
        y, size = struct.unpack_from("<ii", buf, off)
        data = buf[off + 8:off + 8 + size]
        rows = min(lines, ymax - y + 1)
        want = rows * row_bytes
        if size < want:                               # a chunk that did not shrink is stored raw
            if comp in (2, 3):
                data = _unpredict(zlib.decompress(data))
            elif comp == 1:
                data = _unpredict(_unrle(data, want))
        if len(data) != want:
            raise ValueError(f"{path}: chunk at y={y} has {len(data)} bytes, expected {want}")
        q = 0
        for r in range(rows):
            for name, dt in chans:
                planes[name][y - ymin + r] = np.frombuffer(data, dtype=dt, count=W, offset=q).astype(np.float32)
                q += dt.itemsize * W
    names = [n for n, _ in chans]
    if len(names) == 1:
        return planes[names[0]]
    order = [n for n in ("B", "G", "R", "A") if n in names] if {"R", "G", "B"} <= set(names) else names
    return np.stack([planes[n] for n in order], axis=-1)
