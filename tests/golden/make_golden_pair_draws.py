"""Records the random decisions of the reference's training transform stack.

    python tests/golden/make_golden_pair_draws.py        (build container only: needs /root/reference)  -> tests/golden/pair_draws.json

What runs: the UNMODIFIED Painter/data/pair_transforms.py, stacked exactly as Painter/main_train.py:233-251 stacks it (train stack, the
plain stack of transform_train2/3/val and the second-crop stack), on dummy pictures.  Its classes subclass torchvision.transforms, which
is absent from this image (and un-pinned in the reference's requirements), so the base classes come from the stand-ins below: they carry
ONLY what torchvision's own classes contribute to the draws -- `RandomResizedCrop.get_params` and `ColorJitter.get_params`, restated from
torchvision's published source (torchvision/transforms/transforms.py, 0.15: ten area / log-ratio tries + central fallback;
randperm(4) then the four uniform factors) -- and a recording `functional` module.  Everything about WHEN a draw happens is the
reference's own code: one get_params per pair shared by image and target (pair_transforms.py:139-150), RandomApply's `p < rand(1)` test
before the jitter's draws (:227-231), the jitter applied to the image only (:251-261), the flip coin after it (:199-203), the order
of the stack (main_train.py:233-241).  The fixture stores, per seed, the sequence of recorded events; tests/test_pair_draws_cpu.py
replays painter_amd.pair_pipeline.sample_* on the same seeds against it (and, where /root/reference is mounted, against a live run)."""
import enum
import importlib.util
import json
import math
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Painter/data/pair_transforms.py"

LOG = []


class Picture:
    """Stands for a decoded PIL picture: only its size matters to the draws."""

    def __init__(self, height, width, name):
        self.height, self.width, self.name = height, width, name


def _install_torchvision_stand_ins():
    tv = types.ModuleType("torchvision")
    T = types.ModuleType("torchvision.transforms")
    F = types.ModuleType("torchvision.transforms.functional")

    class InterpolationMode(enum.Enum):
        NEAREST = "nearest"
        BILINEAR = "bilinear"
        BICUBIC = "bicubic"

    class Compose:
        def __init__(self, transforms):
            self.transforms = transforms

    class ToTensor:
        pass

    class Normalize(torch.nn.Module):
        def __init__(self, mean, std, inplace=False):
            super().__init__()
            self.mean, self.std, self.inplace = mean, std, inplace

    class RandomHorizontalFlip(torch.nn.Module):
        def __init__(self, p=0.5):
            super().__init__()
            self.p = p

    class RandomApply(torch.nn.Module):
        def __init__(self, transforms, p=0.5):
            super().__init__()
            self.transforms, self.p = transforms, p

    class RandomResizedCrop(torch.nn.Module):
        def __init__(self, size, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), interpolation=InterpolationMode.BILINEAR):
            super().__init__()
            self.size = (size, size) if isinstance(size, int) else tuple(size)
            self.scale, self.ratio, self.interpolation = scale, ratio, interpolation

        @staticmethod
        def get_params(img, scale, ratio):      # torchvision/transforms/transforms.py, RandomResizedCrop.get_params
            height, width = img.height, img.width
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

    class ColorJitter(torch.nn.Module):
        def __init__(self, brightness=0, contrast=0, saturation=0, hue=0):
            super().__init__()
            rng = lambda v, center=1.0, lo=0.0: None if v == 0 else (max(center - v, lo), center + v)      # _check_input
            self.brightness, self.contrast, self.saturation = rng(brightness), rng(contrast), rng(saturation)
            self.hue = None if hue == 0 else (-hue, hue)

        @staticmethod
        def get_params(brightness, contrast, saturation, hue):   # torchvision/transforms/transforms.py, ColorJitter.get_params
            fn_idx = torch.randperm(4)
            b = None if brightness is None else float(torch.empty(1).uniform_(brightness[0], brightness[1]))
            c = None if contrast is None else float(torch.empty(1).uniform_(contrast[0], contrast[1]))
            s = None if saturation is None else float(torch.empty(1).uniform_(saturation[0], saturation[1]))
            h = None if hue is None else float(torch.empty(1).uniform_(hue[0], hue[1]))
            return fn_idx, b, c, s, h

    class RandomErasing(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    def rec(kind):
        def f(img, *args, **kw):
            LOG.append([kind, img.name] + [a.value if isinstance(a, enum.Enum) else (list(a) if isinstance(a, (tuple, list)) else a) for a in args])
            return img
        return f

    for name in ("resized_crop", "hflip", "adjust_brightness", "adjust_contrast", "adjust_saturation", "adjust_hue", "to_tensor", "normalize"):
        setattr(F, name, rec(name))
    F._interpolation_modes_from_int = lambda i: {0: InterpolationMode.NEAREST, 2: InterpolationMode.BILINEAR, 3: InterpolationMode.BICUBIC}[i]
    F.InterpolationMode = InterpolationMode
    for c in (Compose, ToTensor, Normalize, RandomHorizontalFlip, RandomApply, RandomResizedCrop, ColorJitter, RandomErasing):
        setattr(T, c.__name__, c)
    T.functional = F
    T.InterpolationMode = InterpolationMode
    tv.transforms = T
    sys.modules.update({"torchvision": tv, "torchvision.transforms": T, "torchvision.transforms.functional": F})


def load_reference_transforms():
    _install_torchvision_stand_ins()
    spec = importlib.util.spec_from_file_location("ref_pair_transforms", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def stacks(pt, input_size=(896, 448), min_random_scale=0.3):
    """main_train.py:233-251 with the reference's classes (ToTensor / Normalize make no draws and are left out: the stand-in pictures
    are not tensors)."""
    train = pt.Compose([pt.RandomResizedCrop(input_size[1], scale=(min_random_scale, 1.0), interpolation=3),
                        pt.RandomApply([pt.ColorJitter(0.4, 0.4, 0.2, 0.1)], p=0.8),
                        pt.RandomHorizontalFlip()])
    plain = pt.Compose([pt.RandomResizedCrop(input_size[1], scale=(0.9999, 1.0), interpolation=3)])
    seccrop = pt.Compose([pt.RandomResizedCrop(input_size, scale=(min_random_scale, 1.0), ratio=(0.3, 0.7), interpolation=3)])
    return {"train": train, "plain": plain, "seccrop": seccrop}


CASES = [("train", 480, 640, "bicubic", "nearest"), ("train", 375, 500, "bicubic", "bicubic"), ("plain", 480, 640, "nearest", "bicubic"),
         ("seccrop", 896, 448, "bicubic", "bicubic"), ("train", 120, 97, "bicubic", "nearest")]


def record(pt, seed):
    """-> list of (stack, events) for the CASES run back to back from one seed (as a sample's two pairs + second crop are)."""
    st = stacks(pt)
    torch.manual_seed(seed)
    out = []
    for kind, h, w, i1, i2 in CASES:
        del LOG[:]
        st[kind](Picture(h, w, "img"), Picture(h, w, "tgt"), interpolation1=i1, interpolation2=i2)
        out.append([kind, h, w, i1, i2, [list(e) for e in LOG]])
    return out


def main():
    pt = load_reference_transforms()
    fx = {str(seed): record(pt, seed) for seed in range(12)}
    with open(os.path.join(HERE, "pair_draws.json"), "w") as f:
        json.dump(fx, f)
    n = sum(len(ev) for cases in fx.values() for *_, ev in cases)
    print("pair_draws.json: %d seeds, %d recorded calls" % (len(fx), n))


if __name__ == "__main__":
    main()
