"""Host-side constants of the hot path (pure numpy; built once per module, never on the per-step path)."""
import numpy as np


def _cubic_coeffs(t, A=-0.75):
    """PyTorch upsample_bicubic2d coefficients (cubic convolution, A = -0.75) for fractional offset t."""
    def c1(x):   # |x| <= 1
        return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0

    def c2(x):   # 1 < |x| < 2
        return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A
    return np.array([c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)])


def _axis_operator(n_in, n_out):
    """[n_out, n_in] 1-D bicubic resize operator, align_corners=False, border-clamped taps."""
    m = np.zeros((n_out, n_in), dtype=np.float64)
    scale = n_in / n_out
    for o in range(n_out):
        src = (o + 0.5) * scale - 0.5
        i0 = int(np.floor(src))
        t = src - i0
        w = _cubic_coeffs(t)
        for k in range(4):
            idx = min(max(i0 - 1 + k, 0), n_in - 1)
            m[o, idx] += w[k]
    return m


def abs_pos_operator(src, h, w):
    """M [h*w, src*src] with get_abs_pos(P)[l] = sum_s M[l, s] P[s]
    (Painter/util/vitdet_utils.py:140-157: F.interpolate(bicubic, align_corners=False) of the src x src grid).
    Identity when (h, w) == (src, src)."""
    if h == src and w == src:
        return np.eye(src * src, dtype=np.float32)
    my, mx = _axis_operator(src, h), _axis_operator(src, w)
    m = np.einsum("ai,bj->abij", my, mx).reshape(h * w, src * src)
    return m.astype(np.float32)


def sparse_rows(m):
    """Row-wise sparse form of a dense operator m [R, C]: (idx int32 [R, K], val float32 [R, K], K) with K = the largest number of
    non-zeros in a row; rows with fewer are padded with (0, 0.0).  The bicubic resize operator has 16 non-zeros per output token and
    (transposed) <= a few hundred per source cell, out of 196 / 1568 columns."""
    m = np.asarray(m)
    nz = m != 0
    K = max(1, int(nz.sum(1).max()))
    idx = np.zeros((m.shape[0], K), dtype=np.int32)
    val = np.zeros((m.shape[0], K), dtype=np.float32)
    for r in range(m.shape[0]):
        c = np.nonzero(nz[r])[0]
        idx[r, :len(c)] = c
        val[r, :len(c)] = m[r, c]
    return idx, val, K


def drop_path_rates(drop_path_rate, depth):
    """torch.linspace(0, drop_path_rate, depth) (models_painter.py:293) in float32 semantics."""
    if depth == 1:
        return [0.0]
    return [float(np.float32(drop_path_rate) * np.float32(i) / np.float32(depth - 1)) for i in range(depth)]
