"""TEST INFRASTRUCTURE ONLY -- CPU restatement of SegGPT's pre-/post-processing.

The reference does this work on the host with PIL, numpy and CPU torch (SegGPT/SegGPT_inference/seggpt_engine.py:56-103
`inference_image`, :106-181 `inference_video`, :26-53 `run_one_image`); painter_amd/seggpt_engine.py does it on the MI355X
(csrc/seggpt_io.hip).  This file is the checker for that device path: plain loops / numpy, one function per step, each citing
what it follows.  Only tests/ may import it; painter_amd/ never does.

Two steps of the reference live in third-party code that is not under /root/reference:
  * `PIL.Image.resize` -- Pillow (SegGPT_inference/requirements.txt does not pin it; this image has Pillow 12.2.0).  Restated from
    Pillow's published algorithm (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
    ImagingResampleHorizontal_8bpc / Vertical_8bpc; src/libImaging/Geometry.c: ImagingScaleAffine for NEAREST).
  * `F.interpolate(mode='nearest')` -- torch 2.x (aten/src/ATen/native/cpu/UpSampleKernel.cpp, generic nearest kernel).
Parity pinned: tests/test_seggpt_io_cpu.py checks every function here against Pillow itself, CPU torch itself and the unmodified
reference functions (imported with a stand-in model) in the build container, and against tests/golden/seggpt_io.npz anywhere.
"""
import math

import numpy as np

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406])       # seggpt_engine.py:9
IMAGENET_STD = np.array([0.229, 0.224, 0.225])        # seggpt_engine.py:10
PRECISION_BITS = 32 - 8 - 2                           # Pillow Resample.c: 8-bit pixels, 2 guard bits


# ------------------------------------------------------------------------------------------------ Pillow resize
def _bicubic(x):
    """Pillow Resample.c bicubic_filter (Keys kernel, a = -0.5), evaluated in double with one rounding per operation."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for BICUBIC (support 2.0), box = whole axis.
    -> (bounds int32 [out][2] = (first input index, tap count), coeffs int32 [out][ksize], ksize)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        ww = 0.0
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)          # C (int): truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [0.0] * ksize
        for x in range(xmax):
            w = _bicubic((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        for x in range(ksize):
            v = k[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _clip8(v):
    v = v >> PRECISION_BITS                        # arithmetic shift of the signed accumulator
    return 0 if v < 0 else (255 if v > 255 else v)


def pil_resample_pass(src, out_size, vertical):
    """One ImagingResampleHorizontal_8bpc / Vertical_8bpc pass over an [H][W][C] uint8 array (int32 accumulation from 2^21)."""
    src = np.asarray(src, np.uint8)
    h, w, c = src.shape
    bounds, kk, _ = pil_coeffs(h if vertical else w, out_size)
    s32 = src.astype(np.int64)
    if vertical:
        out = np.empty((out_size, w, c), np.uint8)
        for yy in range(out_size):
            first, cnt = bounds[yy]
            acc = np.full((w, c), 1 << (PRECISION_BITS - 1), np.int64)
            for t in range(cnt):
                acc += s32[first + t] * int(kk[yy, t])
            out[yy] = np.clip(acc >> PRECISION_BITS, 0, 255)
    else:
        out = np.empty((h, out_size, c), np.uint8)
        for xx in range(out_size):
            first, cnt = bounds[xx]
            acc = np.full((h, c), 1 << (PRECISION_BITS - 1), np.int64)
            for t in range(cnt):
                acc += s32[:, first + t] * int(kk[xx, t])
            out[:, xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out


def pil_resize_bicubic(src, size):
    """`Image.resize((w, h))` with Pillow's default filter for RGB images (BICUBIC): ImagingResample = horizontal pass (if the
    width changes) into an 8-bit intermediate, then vertical pass (if the height changes).  seggpt_engine.py:62, :66, :117, :136."""
    out_w, out_h = size
    img = np.asarray(src, np.uint8)
    if img.shape[1] != out_w:
        img = pil_resample_pass(img, out_w, vertical=False)
    if img.shape[0] != out_h:
        img = pil_resample_pass(img, out_h, vertical=True)
    return img.copy()


def pil_nearest_table(in_size, out_size):
    """Pillow ImagingScaleAffine source indices for `resize(..., Image.NEAREST)`: the source coordinate starts at scale / 2 and
    is ACCUMULATED in double (xo += scale), COORD() truncates; -1 = outside (filled with 0)."""
    a = float(in_size) / out_size                  # _imaging.c _resize: a[0] = (box[2] - box[0]) / xsize in double
    tab = np.full(out_size, -1, np.int32)
    xo = 0.0 + a * 0.5
    for x in range(out_size):
        xin = -1 if xo < 0.0 else int(xo)
        if 0 <= xin < in_size:
            tab[x] = xin
        xo += a
    return tab


def pil_resize_nearest(src, size):
    """`Image.resize((w, h), Image.NEAREST)` (seggpt_engine.py:70, :121)."""
    out_w, out_h = size
    img = np.asarray(src, np.uint8)
    if (img.shape[1], img.shape[0]) == (out_w, out_h):
        return img.copy()
    yt, xt = pil_nearest_table(img.shape[0], out_h), pil_nearest_table(img.shape[1], out_w)
    out = np.zeros((out_h, out_w, img.shape[2]), np.uint8)
    for y in range(out_h):
        if yt[y] < 0:
            continue
        for x in range(out_w):
            if xt[x] >= 0:
                out[y, x] = img[yt[y], xt[x]]
    return out


# ------------------------------------------------------------------------------------------------ stitch + normalise
def stitch(prompts, targets, query, target_div=None):
    """seggpt_engine.py:65-92 (image) / :139-157 (video): per prompt n, img = [prompt_n ; query] and tgt = [target_n ; target_n]
    stacked along H, `(v / 255. - mean) / std` in float64; run_one_image (:28-34, :47) then moves channels first and casts to
    float32.  `prompts`, `targets`: uint8 [N][R][W][3]; `query`: uint8 [R][W][3]; target_div[n] = 255 for an image-scale target,
    1 for the binary {0,1} masks the video path caches (:166-171).  -> (imgs, tgts) float32 [N][3][2R][W]."""
    prompts = np.asarray(prompts, np.uint8)
    targets = np.asarray(targets, np.uint8)
    n = prompts.shape[0]
    if target_div is None:
        target_div = np.full(n, 255.0)
    image = np.asarray(query, np.uint8) / 255.
    imgs, tgts = [], []
    for i in range(n):
        img2 = prompts[i] / 255.
        tgt2 = targets[i] / float(target_div[i])
        tgt = np.concatenate((tgt2, tgt2), axis=0)
        img = np.concatenate((img2, image), axis=0)
        img = img - IMAGENET_MEAN
        img = img / IMAGENET_STD
        tgt = tgt - IMAGENET_MEAN
        tgt = tgt / IMAGENET_STD
        imgs.append(img)
        tgts.append(tgt)
    imgs = np.stack(imgs, axis=0).transpose(0, 3, 1, 2).astype(np.float32)
    tgts = np.stack(tgts, axis=0).transpose(0, 3, 1, 2).astype(np.float32)
    return np.ascontiguousarray(imgs), np.ascontiguousarray(tgts)


# ------------------------------------------------------------------------------------------------ model output -> picture
def unpatchify_lower(pred0, res_h, res_w, patch):
    """models_seggpt.py:376-389 for sample 0, then seggpt_engine.py:51 `y[0, y.shape[1]//2:, :, :]`: float32 [L][p*p*3] tokens of a
    (2*res_h) x res_w canvas -> the lower res_h x res_w x 3 picture (still float32, still normalised)."""
    pred0 = np.asarray(pred0, np.float32)
    hp, wp = 2 * res_h // patch, res_w // patch
    assert pred0.shape == (hp * wp, patch * patch * 3)
    x = pred0.reshape(hp, wp, patch, patch, 3)
    full = x.transpose(0, 2, 1, 3, 4).reshape(hp * patch, wp * patch, 3)
    return full[res_h:]


def decode(pred0, res_h, res_w, patch):
    """seggpt_engine.py:51-53: `clip((output * std + mean) * 255, 0, 255)`.  float32 tensor * float64 array promotes to float64, every
    operation rounds once (no fused multiply-add).  -> float64 [res_h][res_w][3]."""
    o = unpatchify_lower(pred0, res_h, res_w, patch).astype(np.float64)
    o = o * IMAGENET_STD
    o = o + IMAGENET_MEAN
    o = o * 255
    return np.clip(o, 0, 255)


def torch_nearest_table(in_size, out_size):
    """Source indices of `F.interpolate(x, size=..., mode='nearest')` for the tensor the reference hands it (seggpt_engine.py:95-99):
    float64, NCHW-contiguous (the permute(0, 3, 1, 2) undoes the earlier nhwc view), CPU.  That takes aten's generic kernel
    (UpSampleKernel.cpp HelperInterpNearest): scale = in / out and scale * dst are computed in the tensor's own precision (double),
    the product is narrowed to float32 and floorf'ed, then clamped to in - 1.  (A float32 tensor or a channels-last one would use a
    float32 scale instead -- UpSample.h nearest_idx -- which differs e.g. for 448 -> 1080.)"""
    scale = float(in_size) / float(out_size)
    tab = np.empty(out_size, np.int32)
    for d in range(out_size):
        tab[d] = min(int(math.floor(np.float32(scale * d))), in_size - 1)
    return tab


def blend(pred0, input_image, res_h, res_w, patch):
    """seggpt_engine.py:95-102 (and :173-179): nearest-resize the decoded picture to the input's size, then
    `(input_image * (0.6 * output / 255 + 0.4)).astype(uint8)` in float64 (truncation).  input_image: uint8 [H0][W0][3]."""
    input_image = np.asarray(input_image, np.uint8)
    h0, w0 = input_image.shape[:2]
    out = decode(pred0, res_h, res_w, patch)
    yt, xt = torch_nearest_table(res_h, h0), torch_nearest_table(res_w, w0)
    out = out[yt][:, xt]
    return (input_image * (0.6 * out / 255 + 0.4)).astype(np.uint8)


def mask(pred0, res_h, res_w, patch):
    """seggpt_engine.py:166-171: the video path's next prompt target, `output.mean(-1).gt(128)` expanded to 3 channels.
    torch's CPU mean = sequential sum over the 3 channels, then one division by 3.  -> uint8 {0,1} [res_h][res_w][3]."""
    o = decode(pred0, res_h, res_w, patch)
    m = ((o[..., 0] + o[..., 1]) + o[..., 2]) / 3
    return np.repeat((m > 128).astype(np.uint8)[..., None], 3, axis=-1)
