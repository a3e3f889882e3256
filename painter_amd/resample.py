"""Host-side index / coefficient tables for the device resize kernels (csrc/seggpt_io.hip).

The reference resizes with `PIL.Image.resize` (default BICUBIC for RGB; NEAREST for prompt targets) and with
`F.interpolate(mode='nearest')` (SegGPT/SegGPT_inference/seggpt_engine.py:62-71, :95-99).  The per-axis tables those libraries build
on the host are tiny (a few KB), depend only on (input size, output size) and carry all of the floating-point work; the per-pixel
work is integer and runs on the device.  So the tables are built here once per size pair, in numpy float64 with the same operation
order as the libraries (vectorised over the output axis, sequential over the taps), cached, and uploaded.
"""
import functools
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Pillow's 8-bit fixed point: 22 fractional bits


@functools.lru_cache(maxsize=64)
def bicubic_tables(in_size, out_size):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc, BICUBIC (a = -0.5, support 2), box = the whole axis.
    -> (bounds int32 [out][2] = (first tap, number of taps), coeffs int32 [out][ksize], ksize)."""
    scale = float(np.float32(in_size)) / out_size                   # the box is stored as C floats
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    inv = 1.0 / filterscale
    first = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    taps = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - first
    t = np.arange(ksize, dtype=np.int64)[None, :]
    x = np.abs(((t + first[:, None]).astype(np.float64) - center[:, None] + 0.5) * inv)
    inner = ((1.5 * x - 2.5) * x) * x + 1
    outer = (((x - 5) * x + 8) * x - 4) * -0.5
    w = np.where(x < 1.0, inner, np.where(x < 2.0, outer, 0.0))
    w = np.where(t < taps[:, None], w, 0.0)
    total = np.zeros(out_size, np.float64)
    for j in range(ksize):                                          # sequential over the taps, as the C loop sums them
        total = total + w[:, j]
    safe = np.where(total != 0.0, total, 1.0)
    k = np.where((total != 0.0)[:, None], w / safe[:, None], w)
    fixed = k * float(1 << PRECISION_BITS)
    coeffs = np.where(k < 0, np.trunc(-0.5 + fixed), np.trunc(0.5 + fixed)).astype(np.int32)
    bounds = np.stack([first, taps], axis=1).astype(np.int32)
    return np.ascontiguousarray(bounds), np.ascontiguousarray(coeffs), ksize


@functools.lru_cache(maxsize=64)
def pil_nearest_table(in_size, out_size):
    """Pillow Geometry.c ImagingScaleAffine: source index = (int) of a coordinate that starts at scale / 2 and is accumulated by
    repeated addition of scale; -1 marks a position outside the source (filled with 0)."""
    a = float(in_size) / out_size
    steps = np.full(out_size, a, np.float64)
    steps[0] = a * 0.5
    pos = np.add.accumulate(steps)                                  # sequential prefix sums = the C loop's xo += a
    idx = np.where(pos < 0.0, -1, np.trunc(pos)).astype(np.int64)
    idx = np.where((idx >= 0) & (idx < in_size), idx, -1)
    return np.ascontiguousarray(idx.astype(np.int32))


@functools.lru_cache(maxsize=64)
def torch_nearest_table(in_size, out_size):
    """Source indices of F.interpolate(mode='nearest') for the float64 NCHW-contiguous CPU tensor the reference passes
    (seggpt_engine.py:95-99): aten's generic nearest kernel computes scale = in / out and scale * dst in double, narrows the product
    to float32, floors and clamps."""
    scale = float(in_size) / float(out_size)
    src = np.floor((scale * np.arange(out_size, dtype=np.float64)).astype(np.float32))
    return np.ascontiguousarray(np.minimum(src.astype(np.int64), in_size - 1).astype(np.int32))
