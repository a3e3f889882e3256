"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the pixel work of Painter's training input pipeline.

The reference builds every training sample on the host (Painter/data/pairdataset.py:106-190 `__getitem__`, with the transform stack of
Painter/main_train.py:232-251 from Painter/data/pair_transforms.py): RandomResizedCrop (crop + PIL resize, BICUBIC or NEAREST per
side), ColorJitter on the image with probability 0.8 (PIL ImageEnhance + an HSV round trip), RandomHorizontalFlip, ToTensor,
Normalize, the second pair stitched underneath, an optional second RandomResizedCrop on the float canvases, the `valid` rules, the
mask.  painter_amd/pair_pipeline.py does the pixel work on the MI355X (csrc/pair_io.hip) from EXPLICIT random parameters (crop
boxes, jitter order and factors, flip flags) -- drawing those few scalars stays on the host with whatever the caller uses
(torchvision's get_params in a reference deployment).  This file is the checker for that device path; only tests/ import it.

Third-party code on this path that is not under /root/reference:
  * torchvision (un-pinned in Painter/requirements.txt, absent from this image).  What it does on this path for PIL inputs is glue
    around Pillow: F.resized_crop = img.crop((j, i, j + w, i + h)).resize(size[::-1], filter); F.hflip = transpose(FLIP_LEFT_RIGHT);
    F.adjust_brightness / contrast / saturation = ImageEnhance.Brightness / Contrast / Color(img).enhance(f); F.adjust_hue = HSV
    split, uint8 add of the hue shift, merge, convert back; F.to_tensor = uint8 -> float32 / 255; F.normalize = (x - mean) / std in
    float32; for tensors F.resized_crop = slice + torch interpolate.  The GLUE is restated from torchvision's published source and is
    NOT pinned (no torchvision here); everything under the glue is pinned against Pillow 12.2.0 and CPU torch themselves.
  * Pillow: Blend.c (float alpha, truncating / clipping), Convert.c (L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16, rgb2hsv,
    hsv2rgb with their float / double mix), ImageStat mean -- restated here, pinned by tests/test_pair_pipeline_cpu.py (HSV both ways
    over all 2^24 colours).
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import seggpt_io_oracle as IO

MEAN = [0.485, 0.456, 0.406]          # main_train.py:241
STD = [0.229, 0.224, 0.225]


# ------------------------------------------------------------------------------------------------ RandomResizedCrop on PIL images
def resized_crop(img, box, size, nearest):
    """pair_transforms.py:152-163 -> torchvision F.resized_crop on a PIL image: crop rows [i, i+h), columns [j, j+w), then
    Image.resize((size_w, size_h), BICUBIC | NEAREST).  img: uint8 [H][W][3]; box = (i, j, h, w); size = (height, width)."""
    i, j, h, w = box
    crop = np.ascontiguousarray(np.asarray(img, np.uint8)[i:i + h, j:j + w])
    fn = IO.pil_resize_nearest if nearest else IO.pil_resize_bicubic
    return fn(crop, (size[1], size[0]))


# ------------------------------------------------------------------------------------------------ ColorJitter (PIL ImageEnhance)
def _blend(in1, in2, alpha):
    """Pillow Blend.c ImagingBlend: alpha is a C float; out = (UINT8)(in1 + alpha * (in2 - in1)) in float32, truncated for
    0 <= alpha <= 1, clipped to [0, 255] first otherwise; alpha == 0 / 1 copy an operand."""
    a = np.float32(alpha)
    if a == 0.0:
        return in1.copy()
    if a == 1.0:
        return in2.copy()
    i1 = in1.astype(np.int32)
    d = (in2.astype(np.int32) - i1).astype(np.float32)
    t = (i1.astype(np.float32) + (a * d).astype(np.float32)).astype(np.float32)
    if 0 <= a <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32))).astype(np.uint8)


def gray(rgb):
    """Pillow Convert.c rgb2l."""
    r, g, b = [rgb[..., k].astype(np.int64) for k in range(3)]
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def adjust_brightness(rgb, f):
    return _blend(np.zeros_like(rgb), rgb, f)


def adjust_contrast(rgb, f):
    g = gray(rgb)
    mean = int(float(int(g.astype(np.int64).sum())) / g.size + 0.5)          # ImageStat.Stat(L).mean[0], int(mean + 0.5)
    return _blend(np.full_like(rgb, mean), rgb, f)


def adjust_saturation(rgb, f):
    return _blend(np.repeat(gray(rgb)[..., None], 3, -1), rgb, f)


def rgb2hsv(rgb):
    """Pillow Convert.c rgb2hsv_row: float32 ratios, hue assembled in double, narrowed to float32, fmod in double, (int) truncation."""
    r, g, b = [rgb[..., k].astype(np.int32) for k in range(3)]
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    eq = maxc == minc
    cr = np.where(eq, 1, maxc - minc).astype(np.float32)
    s = (cr / np.where(maxc == 0, 1, maxc).astype(np.float32)).astype(np.float32)
    rc = ((maxc - r).astype(np.float32) / cr).astype(np.float32)
    gc = ((maxc - g).astype(np.float32) / cr).astype(np.float32)
    bc = ((maxc - b).astype(np.float32) / cr).astype(np.float32)
    h = np.where(r == maxc, (bc - gc).astype(np.float32),
                 np.where(g == maxc, (2.0 + rc.astype(np.float64) - bc.astype(np.float64)).astype(np.float32),
                          (4.0 + gc.astype(np.float64) - rc.astype(np.float64)).astype(np.float32)))
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    return np.stack([np.where(eq, 0, uh), np.where(eq, 0, us), maxc], -1).astype(np.uint8)


def _c_round(x):
    return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5))


def hsv2rgb(hsv):
    """Pillow Convert.c hsv2rgb: sector and fraction from h * 6 / 255 in double, fs = s / 255 narrowed to float32, p / q / t rounded
    half away from zero."""
    h, s, v = [hsv[..., k] for k in range(3)]
    hh = h.astype(np.float64) * 6.0 / 255.0
    i = np.floor(hh).astype(np.int64)
    f = (hh - i.astype(np.float64)).astype(np.float32).astype(np.float64)
    fs = (s.astype(np.float64) / 255.0).astype(np.float32).astype(np.float64)
    vd = v.astype(np.float64)
    p = np.clip(_c_round(vd * (1.0 - fs)), 0, 255).astype(np.uint8)
    q = np.clip(_c_round(vd * (1.0 - fs * f)), 0, 255).astype(np.uint8)
    t = np.clip(_c_round(vd * (1.0 - fs * (1.0 - f))), 0, 255).astype(np.uint8)
    m = i % 6
    rr = np.choose(m, [v, q, p, p, t, v])
    gg = np.choose(m, [t, v, v, q, p, p])
    bb = np.choose(m, [p, p, t, v, v, q])
    z = s == 0
    return np.stack([np.where(z, v, rr), np.where(z, v, gg), np.where(z, v, bb)], -1).astype(np.uint8)


def hue_shift_byte(hue_factor):
    """torchvision _functional_pil.adjust_hue adds uint8(hue_factor * 255) with wrap-around: truncation toward zero, modulo 256."""
    return int(hue_factor * 255) & 0xff


def adjust_hue(rgb, shift):
    """shift: the uint8 added to the H plane (see hue_shift_byte)."""
    hsv = rgb2hsv(rgb)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + int(shift)).astype(np.uint8)
    return hsv2rgb(hsv)


def color_jitter(rgb, ops, factors):
    """pair_transforms.py:236-247: apply in the drawn order; ops[k] in {0 brightness, 1 contrast, 2 saturation, 3 hue, -1 none},
    factors[k] the factor (for hue: the uint8 shift)."""
    out = np.asarray(rgb, np.uint8).copy()
    for op, f in zip(ops, factors):
        if op == 0:
            out = adjust_brightness(out, f)
        elif op == 1:
            out = adjust_contrast(out, f)
        elif op == 2:
            out = adjust_saturation(out, f)
        elif op == 3:
            out = adjust_hue(out, int(f))
    return out


# ------------------------------------------------------------------------------------------------ flip, ToTensor, Normalize, stitch
def to_tensor_normalize(rgb, flip):
    """pair_transforms.py:199-203, :72, :101 with main_train.py:240-241: hflip, uint8 -> float32 / 255, (x - mean) / std in float32.
    -> torch float32 [3][H][W]."""
    a = np.asarray(rgb, np.uint8)
    if flip:
        a = a[:, ::-1]
    x = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    mean = torch.as_tensor(MEAN, dtype=torch.float32)[:, None, None]
    std = torch.as_tensor(STD, dtype=torch.float32)[:, None, None]
    return x.sub_(mean).div_(std)


def combine(first, second):
    """pairdataset.py:100-104: the second pair goes under the first."""
    return torch.cat([first, second], dim=1)


# ------------------------------------------------------------------------------------------------ second crop on float canvases
def resized_crop_tensor(x, box, size, nearest):
    """main_train.py:248-250 transform_train_seccrop on float32 [3][H][W] tensors: slice + torch interpolate (bicubic,
    align_corners=False, no antialias -- the crop is never larger than the output, so antialiasing would be the identity anyway).
    CPU torch IS the reference implementation of this step; the device kernel is compared to it with a float tolerance."""
    i, j, h, w = box
    c = x[:, i:i + h, j:j + w][None]
    if nearest:
        return F.interpolate(c, size=list(size), mode="nearest")[0]
    return F.interpolate(c, size=list(size), mode="bicubic", align_corners=False)[0]


# ------------------------------------------------------------------------------------------------ valid rules
VALID_NONE, VALID_LESS_ZERO, VALID_POSE, VALID_FG_ONLY = 0, 1, 2, 3


def valid_rule(pair_type):
    """pairdataset.py:155-180 -> (mode, black level before normalisation)."""
    if "nyuv2_image2depth" in pair_type:
        return VALID_LESS_ZERO, 1e-3 * 0.1
    if "ade20k_image2semantic" in pair_type or "coco_image2panoptic_sem_seg" in pair_type:
        return VALID_LESS_ZERO, 1e-5
    if "image2pose" in pair_type:
        return VALID_POSE, 1e-5
    if "image2panoptic_inst" in pair_type:
        return VALID_FG_ONLY, 1e-5
    return VALID_NONE, 0.0


def threshold(level):
    """pairdataset.py:157-158: `thres = torch.ones(3) * level; thres = (thres - mean) / std` in float32."""
    thres = torch.ones(3) * level
    return (thres - torch.tensor(MEAN)) / torch.tensor(STD)


def valid_map(target, pair_type):
    """pairdataset.py:152-180 on the final float32 target canvas [3][H][W]."""
    valid = torch.ones_like(target)
    mode, level = valid_rule(pair_type)
    if mode == VALID_NONE:
        return valid
    thres = threshold(level)
    if mode == VALID_LESS_ZERO:
        valid[target < thres[:, None, None]] = 0
    elif mode == VALID_POSE:
        valid[target > thres[:, None, None]] = 10.0
        fg = target > thres[:, None, None]
        if fg.sum() < 100 * 3:
            valid = valid * 0.
    elif mode == VALID_FG_ONLY:
        fg = target > thres[:, None, None]
        if fg.sum() < 100 * 3:
            valid = valid * 0.
    return valid


# ------------------------------------------------------------------------------------------------ one sample end to end
def build_sample(spec, input_size=(896, 448)):
    """pairdataset.py:106-190 with explicit random parameters.  spec: dict with
         pairs: list of 1 or 2 dicts {image, target (uint8 HWC), crop (i, j, h, w), jitter (ops, factors) or None, flip},
         interpolation1 / interpolation2: 'bicubic' | 'nearest', pair_type, seccrop: (i, j, h, w) or None.
    -> (image, target, valid) float32 [3][H][W]."""
    side = input_size[1]
    n1, n2 = spec["interpolation1"] == "nearest", spec["interpolation2"] == "nearest"
    imgs, tgts = [], []
    for p in spec["pairs"]:
        a = resized_crop(p["image"], p["crop"], (side, side), n1)
        b = resized_crop(p["target"], p["crop"], (side, side), n2)
        if p.get("jitter") is not None:
            a = color_jitter(a, *p["jitter"])
        imgs.append(to_tensor_normalize(a, p["flip"]))
        tgts.append(to_tensor_normalize(b, p["flip"]))
    image = imgs[0] if len(imgs) == 1 else combine(imgs[0], imgs[1])
    target = tgts[0] if len(tgts) == 1 else combine(tgts[0], tgts[1])
    if spec.get("seccrop") is not None:
        image = resized_crop_tensor(image, spec["seccrop"], tuple(image.shape[1:]), n1)
        target = resized_crop_tensor(target, spec["seccrop"], tuple(target.shape[1:]), n2)
    return image, target, valid_map(target, spec["pair_type"])
