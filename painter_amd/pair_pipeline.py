"""Painter's training input pipeline with the pixel work on the MI355X.

The reference builds each sample in a DataLoader worker (Painter/data/pairdataset.py:106-190 `PairDataset.__getitem__` under the
transform stack of Painter/main_train.py:232-251, classes in Painter/data/pair_transforms.py): decode two (image, target) pairs,
RandomResizedCrop each to 448 x 448 (PIL BICUBIC / NEAREST per side), ColorJitter the image with probability 0.8, random horizontal
flip, ToTensor, Normalize, stack the second pair under the first, optionally RandomResizedCrop the float canvases once more, derive
`valid`, draw the patch mask.  Here file decode and the drawing of the random parameters stay on the host -- a handful of scalars per
sample, from the same generators the reference uses -- and everything that touches pixels is a kernel (csrc/seggpt_io.hip resize
passes, csrc/pair_io.hip), batched over the samples of a step and bit-identical to PIL / torch on the same parameters
(tests/test_pair_pipeline_gpu.py).  The outputs are the model's inputs, already resident: `imgs`, `tgts`, `valid` float32
[B][3][896][448].

`SampleSpec` mirrors what `__getitem__` decides per sample; `sample_*` restate torchvision's parameter draws (torchvision is absent
from this image, so those few lines are NOT pinned -- in a deployment with torchvision call its own get_params and fill the spec).
There is no CPU fallback.
"""
import collections
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import resample as RS
from ._lib import check, lib

BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3
VALID_NONE, VALID_LESS_ZERO, VALID_POSE, VALID_FG_ONLY = 0, 1, 2, 3
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


@dataclass
class PairSpec:
    """One (image, target) pair of a sample and the random decisions of its transform stack."""
    image: np.ndarray                                   # decoded RGB uint8 [H][W][3]
    target: np.ndarray                                  # decoded RGB uint8 [H][W][3], same size
    crop: Tuple[int, int, int, int]                     # RandomResizedCrop.get_params -> (top, left, height, width)
    jitter_ops: Sequence[int] = ()                      # ColorJitter: ops in applied order (BRIGHTNESS ...), empty = not applied
    jitter_factors: Sequence[float] = ()                # factor per op; for HUE the hue_factor in [-0.5, 0.5]
    flip: bool = False


@dataclass
class SampleSpec:
    """What PairDataset.__getitem__ decides for one sample (pairdataset.py:106-190)."""
    pairs: List[PairSpec]                               # 1, or 2 with use_two_pairs
    pair_type: str = ""                                 # selects the interpolation modes (:113-124) and the valid rule (:155-180)
    seccrop: Optional[Tuple[int, int, int, int]] = None  # second RandomResizedCrop on the stitched canvases, or None
    interpolation: Tuple[str, str] = field(default=None)

    def __post_init__(self):
        if self.interpolation is None:
            self.interpolation = interpolation_modes(self.pair_type)


def interpolation_modes(pair_type):
    """pairdataset.py:113-124 -> (interpolation1 for images, interpolation2 for targets)."""
    if "depth" in pair_type or "pose" in pair_type:
        return "bicubic", "bicubic"
    if "image2" in pair_type:
        return "bicubic", "nearest"
    if "2image" in pair_type:
        return "nearest", "bicubic"
    return "bicubic", "bicubic"


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


def hue_shift_byte(hue_factor):
    """torchvision's PIL adjust_hue adds uint8(hue_factor * 255) to the H plane with wrap-around."""
    return int(hue_factor * 255) & 0xff


# ------------------------------------------------------------------------------------------------ parameter draws (host, unpinned)
def sample_resized_crop(height, width, scale, ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """torchvision RandomResizedCrop.get_params restated (torch global RNG): ten tries of a uniform area fraction and a log-uniform
    aspect ratio, then the central fallback crop.  -> (top, left, h, w)."""
    area = height * width
    log_ratio = torch.log(torch.tensor(ratio))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if 0 < w <= width and 0 < h <= height:
            i = torch.randint(0, height - h + 1, size=(1,)).item()
            j = torch.randint(0, width - w + 1, size=(1,)).item()
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def sample_color_jitter(brightness=0.4, contrast=0.4, saturation=0.2, hue=0.1, p=0.8):
    """RandomApply (pair_transforms.py:225-231) around torchvision ColorJitter.get_params restated (main_train.py:236-238 values):
    -> (ops, factors), empty when the jitter is skipped."""
    if p < torch.rand(1):
        return (), ()
    order = torch.randperm(4).tolist()
    b = float(torch.empty(1).uniform_(max(0.0, 1 - brightness), 1 + brightness))
    c = float(torch.empty(1).uniform_(max(0.0, 1 - contrast), 1 + contrast))
    s = float(torch.empty(1).uniform_(max(0.0, 1 - saturation), 1 + saturation))
    h = float(torch.empty(1).uniform_(-hue, hue))
    f = {BRIGHTNESS: b, CONTRAST: c, SATURATION: s, HUE: h}
    return tuple(order), tuple(f[o] for o in order)


def sample_flip(p=0.5):
    """pair_transforms.py:199-203."""
    return bool(torch.rand(1) < p)


# ------------------------------------------------------------------------------------------------ device pipeline
def _stream():
    return torch.cuda.current_stream().cuda_stream


class DevicePairPipeline:
    """Builds the model inputs of a step on the device from `SampleSpec`s."""

    def __init__(self, device, input_size=(896, 448)):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("painter_amd.pair_pipeline runs its pixel kernels on an MI355X only (no CPU fallback); got %s" % device)
        self.H, self.W = input_size
        self.side = input_size[1]                      # RandomResizedCrop(args.input_size[1]) -> side x side (main_train.py:234)
        self._tables = collections.OrderedDict()       # (kind, in, out) -> device table(s), least recently used first
        self._pin = None                               # pinned staging buffer for the packed crop boxes
        self._pin_event = None

    def _table(self, kind, in_size, out_size):
        """Pillow's per-axis tap / coefficient table for one (in, out) size pair, cached on the device.  Random crops make most lookups of
        a training step misses; a miss costs the host-side table build plus one or two small pageable uploads (a few KB each)."""
        key = (kind, in_size, out_size)
        t = self._tables.get(key)
        if t is not None:
            self._tables.move_to_end(key)
        else:
            if kind == "bicubic":
                bounds, coeffs, ksize = RS.bicubic_tables(in_size, out_size)
                t = (torch.from_numpy(bounds).to(self.device), torch.from_numpy(coeffs).to(self.device), ksize)
            else:
                t = torch.from_numpy(RS.pil_nearest_table(in_size, out_size)).to(self.device)
            while len(self._tables) >= 4096:           # crop sizes vary per sample: bound the cache, oldest entry first
                self._tables.popitem(last=False)
            self._tables[key] = t
        return t

    # ---- RandomResizedCrop on a decoded picture: PIL crop + resize
    def resized_crop(self, picture, box, out, nearest):
        """picture: uint8 [H][W][3] CUDA tensor; box = (top, left, h, w); out: uint8 [side][side][3] CUDA view written in place."""
        assert picture.is_cuda and picture.dtype == torch.uint8 and picture.is_contiguous() and picture.shape[2] == 3
        H, W, _ = picture.shape
        i, j, h, w = box
        assert 0 <= i and 0 <= j and h >= 1 and w >= 1 and i + h <= H and j + w <= W, (box, (H, W))
        oh, ow = out.shape[0], out.shape[1]
        row_bytes = W * 3
        src = picture.data_ptr() + (i * W + j) * 3
        if nearest:
            if (h, w) == (oh, ow):                      # PIL returns a copy when the size does not change
                out.copy_(picture[i:i + h, j:j + w])
                return out
            check(lib.pa_gather_u8_box(src, row_bytes, h, w, out.data_ptr(), oh, ow, 3, self._table("pil_nearest", h, oh).data_ptr(),
                                       self._table("pil_nearest", w, ow).data_ptr(), _stream()), "pa_gather_u8_box")
            return out
        if w != ow and h != oh:
            bounds, coeffs, ksize = self._table("bicubic", w, ow)
            mid = torch.empty((h, ow, 3), dtype=torch.uint8, device=self.device)
            check(lib.pa_resample_u8_box(src, row_bytes, h, w, mid.data_ptr(), h, ow, 3, bounds.data_ptr(), coeffs.data_ptr(), ksize, 0,
                                         _stream()), "pa_resample_u8_box")
            bounds, coeffs, ksize = self._table("bicubic", h, oh)
            check(lib.pa_resample_u8_box(mid.data_ptr(), ow * 3, h, ow, out.data_ptr(), oh, ow, 3, bounds.data_ptr(), coeffs.data_ptr(), ksize, 1,
                                         _stream()), "pa_resample_u8_box")
        elif w != ow:
            bounds, coeffs, ksize = self._table("bicubic", w, ow)
            check(lib.pa_resample_u8_box(src, row_bytes, h, w, out.data_ptr(), oh, ow, 3, bounds.data_ptr(), coeffs.data_ptr(), ksize, 0,
                                         _stream()), "pa_resample_u8_box")
        elif h != oh:
            bounds, coeffs, ksize = self._table("bicubic", h, oh)
            check(lib.pa_resample_u8_box(src, row_bytes, h, w, out.data_ptr(), oh, ow, 3, bounds.data_ptr(), coeffs.data_ptr(), ksize, 1,
                                         _stream()), "pa_resample_u8_box")
        else:
            out.copy_(picture[i:i + h, j:j + w])
        return out

    # ---- ColorJitter, batched, in place
    def color_jitter(self, images, ops, factors):
        """images: uint8 [B][h][w][3] CUDA; ops: int [B][4] (negative = none); factors: float [B][4] (HUE slot: hue_factor)."""
        B, h, w, _ = images.shape
        ops_h = np.full((B, 4), -1, np.int32)
        fac_h = np.zeros((B, 4), np.float32)
        for b in range(B):
            for k, (o, f) in enumerate(zip(ops[b], factors[b])):
                ops_h[b, k] = o
                fac_h[b, k] = float(hue_shift_byte(f)) if o == HUE else np.float32(f)
        if not (ops_h >= 0).any():
            return images
        ops_d = torch.from_numpy(ops_h).to(self.device)
        fac_d = torch.from_numpy(fac_h).to(self.device)
        ws = torch.empty(int(lib.pa_color_jitter_workspace_bytes(B)), dtype=torch.uint8, device=self.device)
        check(lib.pa_color_jitter(images.data_ptr(), ops_d.data_ptr(), fac_d.data_ptr(), ops_h.ctypes.data, ws.data_ptr(), B, h, w, _stream()),
              "pa_color_jitter")
        return images

    def to_tensor_normalize(self, images, flips, canvas, row0):
        B, h, w, _ = images.shape
        flip_d = torch.as_tensor([1 if f else 0 for f in flips], dtype=torch.int32).to(self.device)
        check(lib.pa_to_tensor_normalize(images.data_ptr(), flip_d.data_ptr(), canvas.data_ptr(), B, h, w, canvas.shape[2], row0, _stream()),
              "pa_to_tensor_normalize")
        return canvas

    def resized_crop_tensor(self, canvas, boxes, nearest):
        """canvas: float32 [B][C][H][W]; boxes: [B][4] = (top, left, h, w) -> new canvas of the same shape."""
        B, C, H, W = canvas.shape
        out = torch.empty_like(canvas)
        box_d = torch.as_tensor(np.asarray(boxes, np.int32).reshape(B, 4)).to(self.device)
        check(lib.pa_resized_crop_f32(canvas.data_ptr(), out.data_ptr(), box_d.data_ptr(), B, C, H, W, 1 if nearest else 0, _stream()),
              "pa_resized_crop_f32")
        return out

    def valid_map(self, tgts, pair_types):
        B, C, H, W = tgts.shape
        assert C == 3 and tgts.is_contiguous()
        modes = np.zeros(B, np.int32)
        thres = np.zeros((B, 3), np.float32)
        mean, std = torch.tensor(MEAN), torch.tensor(STD)
        for b, pt in enumerate(pair_types):
            modes[b], level = valid_rule(pt)
            thres[b] = ((torch.ones(3) * level - mean) / std).numpy()          # pairdataset.py:157-158, float32 on the host
        valid = torch.empty_like(tgts)
        ws = torch.empty(int(lib.pa_pair_valid_workspace_bytes(B)), dtype=torch.uint8, device=self.device)
        modes_d = torch.from_numpy(modes).to(self.device)          # named: a temporary would be recycled before the launch reads it
        thres_d = torch.from_numpy(thres).to(self.device)
        check(lib.pa_pair_valid(tgts.data_ptr(), valid.data_ptr(), modes_d.data_ptr(), thres_d.data_ptr(), ws.data_ptr(), B, H * W, _stream()),
              "pa_pair_valid")
        return valid

    # ---- RandomResizedCrop of every picture of a step: one upload, two launches
    _JOB = np.dtype([("src", "<u8"), ("dst", "<u8"), ("xbounds", "<u8"), ("xcoeffs", "<u8"), ("ybounds", "<u8"), ("ycoeffs", "<u8"),
                     ("src_row_bytes", "<i8"), ("h", "<i4"), ("w", "<i4"), ("xksize", "<i4"), ("yksize", "<i4"), ("nearest", "<i4"),
                     ("mid_row0", "<i4")])          # include/painter_hip.h: pa_crop_job

    def resized_crop_batch(self, pictures, boxes, nearest, out):
        """pictures: decoded uint8 [H][W][3] numpy arrays; boxes: (top, left, h, w) each; nearest: bool each; out: uint8
        [n][side][side][3] CUDA tensor written in place.  The crop boxes travel to the device packed in one pinned buffer, the job table in
        one more copy; resize tables for (in, out) size pairs not seen recently are built and uploaded by _table() (2-4 small copies each)."""
        n = len(pictures)
        oh, ow = out.shape[1], out.shape[2]
        assert out.is_cuda and out.dtype == torch.uint8 and out.is_contiguous() and out.shape[0] == n and out.shape[3] == 3
        sizes = []
        for pic, (i, j, h, w) in zip(pictures, boxes):
            H, W = pic.shape[:2]
            assert pic.dtype == np.uint8 and pic.ndim == 3 and pic.shape[2] == 3
            assert 0 <= i and 0 <= j and h >= 1 and w >= 1 and i + h <= H and j + w <= W, ((i, j, h, w), (H, W))
            sizes.append((h * w * 3 + 15) & ~15)
        total = sum(sizes)
        if self._pin_event is not None:
            self._pin_event.synchronize()               # the previous step's upload has left the staging buffer
        if self._pin is None or self._pin.numel() < total:
            self._pin = torch.empty(max(total, 1 << 22), dtype=torch.uint8).pin_memory()
        pin = self._pin.numpy()
        off = 0
        offs = []
        for pic, (i, j, h, w), sz in zip(pictures, boxes, sizes):
            pin[off:off + h * w * 3].reshape(h, w, 3)[...] = pic[i:i + h, j:j + w]
            offs.append(off)
            off += sz
        dev = torch.empty(total, dtype=torch.uint8, device=self.device)
        dev.copy_(self._pin[:total], non_blocking=True)
        self._pin_event = torch.cuda.Event()
        self._pin_event.record()
        jobs = np.zeros(n, self._JOB)
        mid_rows = 0
        keep = []
        for k, ((i, j, h, w), near) in enumerate(zip(boxes, nearest)):
            jobs["src"][k], jobs["dst"][k] = dev.data_ptr() + offs[k], out[k].data_ptr()
            jobs["src_row_bytes"][k], jobs["h"][k], jobs["w"][k], jobs["nearest"][k] = w * 3, h, w, 1 if near else 0
            if near:
                if (h, w) != (oh, ow):
                    yt, xt = self._table("pil_nearest", h, oh), self._table("pil_nearest", w, ow)
                    jobs["ybounds"][k], jobs["xbounds"][k] = yt.data_ptr(), xt.data_ptr()
                    keep += [yt, xt]
                continue
            jobs["mid_row0"][k] = mid_rows
            mid_rows += h
            if w != ow:
                tb, tc, ks = self._table("bicubic", w, ow)
                jobs["xbounds"][k], jobs["xcoeffs"][k], jobs["xksize"][k] = tb.data_ptr(), tc.data_ptr(), ks
                keep += [tb, tc]
            if h != oh:
                tb, tc, ks = self._table("bicubic", h, oh)
                jobs["ybounds"][k], jobs["ycoeffs"][k], jobs["yksize"][k] = tb.data_ptr(), tc.data_ptr(), ks
                keep += [tb, tc]
        jobs_d = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(self.device)
        mid = torch.empty(max(mid_rows, 1) * ow * 3, dtype=torch.uint8, device=self.device)
        check(lib.pa_resized_crop_u8_batch(jobs_d.data_ptr(), mid.data_ptr(), n, max(b[2] for b in boxes), max(b[3] for b in boxes), oh, ow,
                                           _stream()), "pa_resized_crop_u8_batch")
        del keep                                        # `keep` held the tables (the cache may evict them) until the launches were enqueued;
                                                        # after that the stream-ordered allocator cannot hand their memory out early
        return out

    def resized_crop_tensor_modes(self, canvas, boxes, modes):
        """canvas: float32 [B][C][H][W]; boxes [B][4]; modes [B] (0 bicubic, 1 nearest, 2 keep) -> new canvas, one launch."""
        B, C, H, W = canvas.shape
        out = torch.empty_like(canvas)
        box_d = torch.as_tensor(np.asarray(boxes, np.int32).reshape(B, 4)).to(self.device)
        mode_d = torch.as_tensor(np.asarray(modes, np.int32).reshape(B)).to(self.device)
        check(lib.pa_resized_crop_f32_modes(canvas.data_ptr(), out.data_ptr(), box_d.data_ptr(), mode_d.data_ptr(), B, C, H, W, _stream()),
              "pa_resized_crop_f32_modes")
        return out

    # ---- a whole step
    def build_batch(self, samples: Sequence[SampleSpec]):
        """-> (imgs, tgts, valid) float32 [B][3][H][W] on the device, what `collate(__getitem__ ...)` hands to the model.
        Launches per step (two-pair samples): 2 for every crop of the step (images and targets of both pairs are jobs of one table),
        <= 8 for the jitter (one sum + one apply per used slot, both pairs in one call), 4 ToTensor/Normalize/stitch, 2 second crops,
        2 valid -- independent of the batch size; uploads: the packed crop boxes, the job table, and five small parameter arrays."""
        B = len(samples)
        npairs = len(samples[0].pairs)
        assert npairs in (1, 2) and all(len(s.pairs) == npairs for s in samples) and npairs * self.side == self.H, \
            "input_size %s needs %d pair(s) of %d rows" % ((self.H, self.W), self.H // self.side, self.side)
        imgs = torch.empty((B, 3, self.H, self.W), dtype=torch.float32, device=self.device)
        tgts = torch.empty_like(imgs)
        # crops[0] = images, crops[1] = targets; row k * B + b = pair k of sample b
        crops = torch.empty((2, npairs * B, self.side, self.side, 3), dtype=torch.uint8, device=self.device)
        pictures, boxes, nearest = [], [], []
        for which in range(2):
            for k in range(npairs):
                for s in samples:
                    p = s.pairs[k]
                    assert p.image.shape == p.target.shape, "image and target of a pair share one crop box (pair_transforms.py:152-163)"
                    pictures.append(p.target if which else p.image)
                    boxes.append(p.crop)
                    nearest.append(s.interpolation[which] == "nearest")
        self.resized_crop_batch(pictures, boxes, nearest, crops.view(2 * npairs * B, self.side, self.side, 3))
        order = [(k, s) for k in range(npairs) for s in samples]
        self.color_jitter(crops[0], [s.pairs[k].jitter_ops for k, s in order], [s.pairs[k].jitter_factors for k, s in order])
        for k in range(npairs):
            flips = [s.pairs[k].flip for s in samples]
            self.to_tensor_normalize(crops[0, k * B:(k + 1) * B], flips, imgs, k * self.side)
            self.to_tensor_normalize(crops[1, k * B:(k + 1) * B], flips, tgts, k * self.side)
        # second crop: per sample a box and an interpolation mode, or "keep"
        if any(s.seccrop is not None for s in samples):
            boxes2 = [s.seccrop if s.seccrop is not None else (0, 0, self.H, self.W) for s in samples]
            for side, name in ((0, "imgs"), (1, "tgts")):
                modes = [2 if s.seccrop is None else (1 if s.interpolation[side] == "nearest" else 0) for s in samples]
                if side == 0:
                    imgs = self.resized_crop_tensor_modes(imgs, boxes2, modes)
                else:
                    tgts = self.resized_crop_tensor_modes(tgts, boxes2, modes)
        valid = self.valid_map(tgts, [s.pair_type for s in samples])
        return imgs, tgts, valid
