"""SURVEY.md 8f N2, host half: painter_amd.pair_dataset.PairSpecDataset against the UNMODIFIED reference `PairDataset`
(Painter/data/pairdataset.py) run on the same files and the same seeds.

The reference's transform classes need torchvision (absent here), so `PairDataset` gets stand-in transform stacks that make the
same parameter draws as PairSpecDataset (painter_amd.pair_pipeline.sample_*) and do the pixel work with the oracle's functions.
What this pins against the reference's own code: which stack a pair type gets, the interpolation modes, when the partner pair and
the half-mask coin are drawn, when the second crop applies, the stitch order, every `valid` rule, and the mask.  Then the samples
the reference returned must equal oracle.build_sample(spec) for the spec PairSpecDataset returned."""
import json
import random

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import pair_pipeline_oracle as O
from oracle import ref_import
from painter_amd import pair_dataset as PD
from painter_amd import pair_pipeline as PP
from tests import pair_pipeline_cases as C
from tests.seggpt_io_cases import picture

pytestmark = pytest.mark.skipif(not ref_import.reference_available(), reason="needs /root/reference (build container only)")

TYPES = ["coco_image2panoptic_sem_seg", "nyuv2_image2depth", "coco_image2pose", "coco_image2panoptic_inst", "ade20k_image2semantic",
         "ssid_2image_denoise", "lol_2image_enhance", "derain_2image"]
MIN_SCALE = 0.3


class StandInStack:
    """What main_train.py:233-254 builds with torchvision, as a callable PairStandardTransform can drive (pairdataset.py:196-203)."""

    def __init__(self, kind):
        self.kind = kind

    def __call__(self, img, tgt, interpolation1=None, interpolation2=None):
        n1, n2 = interpolation1 == "nearest", interpolation2 == "nearest"
        if self.kind == "seccrop":
            box = PP.sample_resized_crop(img.shape[1], img.shape[2], (MIN_SCALE, 1.0), ratio=(0.3, 0.7))
            return O.resized_crop_tensor(img, box, tuple(img.shape[1:]), n1), O.resized_crop_tensor(tgt, box, tuple(tgt.shape[1:]), n2)
        a, t = np.array(img), np.array(tgt)
        if self.kind == "full":
            crop = PP.sample_resized_crop(a.shape[0], a.shape[1], (MIN_SCALE, 1.0))
            ops, factors = PP.sample_color_jitter()
            flip = PP.sample_flip()
        else:
            crop = PP.sample_resized_crop(a.shape[0], a.shape[1], (0.9999, 1.0))
            ops, factors, flip = (), (), False
        a = O.resized_crop(a, crop, (448, 448), n1)
        t = O.resized_crop(t, crop, (448, 448), n2)
        if len(ops):
            a = O.color_jitter(a, ops, [O.hue_shift_byte(f) if o == PP.HUE else f for o, f in zip(ops, factors)])
        return O.to_tensor_normalize(a, flip), O.to_tensor_normalize(t, flip)


@pytest.fixture(scope="module")
def dataset_files(tmp_path_factory):
    root = tmp_path_factory.mktemp("pairs")
    lists = []
    for k, t in enumerate(TYPES):
        entries = []
        for i in range(3):
            h, w = 120 + 16 * ((k + i) % 4), 160 + 8 * ((2 * k + i) % 5)
            img = picture(1000 + 10 * k + i, h, w)
            if "pose" in t or "inst" in t:                      # sparse foreground on black, either side of the 300-element rule
                tg = np.zeros((h, w, 3), np.uint8)
                n = [0, 0, 40][i]
                tg[5:5 + n, 7:7 + n] = (200, 30, 90)
            elif "depth" in t:
                tg = picture(2000 + 10 * k + i, h, w)
                tg[: h // 3] = 0
            else:
                tg = picture(2000 + 10 * k + i, h, w, flat=True)
            Image.fromarray(img).save(root / ("%s_%d_img.png" % (t, i)))
            Image.fromarray(tg).save(root / ("%s_%d_tgt.png" % (t, i)))
            entries.append({"image_path": "%s_%d_img.png" % (t, i), "target_path": "%s_%d_tgt.png" % (t, i), "type": t})
        path = root / ("%s.json" % t)
        path.write_text(json.dumps(entries))
        lists.append(str(path))
    return str(root), lists


def _seed(s):
    torch.manual_seed(s)
    random.seed(s)
    np.random.seed(s)


@pytest.mark.parametrize("train", [True, False])
def test_spec_dataset_makes_the_references_decisions(dataset_files, train):
    root, lists = dataset_files
    ref_mod = ref_import.load_reference_pairdataset()
    MaskingGenerator = ref_import.load_reference_masking_generator().MaskingGenerator
    gen = MaskingGenerator((56, 28), num_masking_patches=784, max_num_patches=392, min_num_patches=16)
    if train:
        ref = ref_mod.PairDataset(root, lists, transform=StandInStack("full"), transform2=StandInStack("plain"), transform3=StandInStack("plain"),
                                  transform_seccrop=StandInStack("seccrop"), masked_position_generator=gen, use_two_pairs=True, half_mask_ratio=0.4)
        ours = PD.PairSpecDataset(root, lists, masked_position_generator=gen, use_two_pairs=True, half_mask_ratio=0.4, min_random_scale=MIN_SCALE)
    else:
        ref = ref_mod.PairDataset(root, lists, transform=StandInStack("plain"), transform2=None, transform3=None, masked_position_generator=gen,
                                  use_two_pairs=True, half_mask_ratio=1.0)
        ours = PD.PairSpecDataset(root, lists, masked_position_generator=gen, use_two_pairs=True, half_mask_ratio=1.0, train=False)
    assert len(ref) == len(ours) == 24 and ref.weights == ours.weights and ref.pair_type_dict == ours.pair_type_dict
    seen_sec = seen_half = seen_zero_valid = seen_ten = 0
    for index in range(len(ours)):
        for seed in (index, 100 + index):
            _seed(seed)
            image, target, mask, valid = ref[index]
            _seed(seed)
            spec, my_mask = ours[index]
            oi, ot, ov = O.build_sample(C.oracle_spec(spec))
            assert torch.equal(image, oi) and torch.equal(target, ot), (index, seed, spec.pair_type)
            assert torch.equal(valid, ov), (index, seed, spec.pair_type)
            assert np.array_equal(mask, my_mask)
            seen_sec += spec.seccrop is not None
            seen_half += int(np.array_equal(my_mask[:28], np.zeros((28, 28))) and my_mask[28:].all())
            seen_zero_valid += float(valid.sum()) == 0.0
            seen_ten += float(valid.max()) == 10.0
            assert spec.interpolation == PP.interpolation_modes(spec.pair_type)
    if train:
        assert seen_sec > 5 and seen_half > 5 and seen_zero_valid >= 2 and seen_ten >= 1
    else:
        assert seen_sec == 0 and seen_half == 48


def test_collate_specs(dataset_files):
    root, lists = dataset_files
    MaskingGenerator = ref_import.load_reference_masking_generator().MaskingGenerator
    gen = MaskingGenerator((56, 28), num_masking_patches=784, max_num_patches=392, min_num_patches=16)
    ds = PD.PairSpecDataset(root, lists, masked_position_generator=gen, half_mask_ratio=0.5)
    loader = torch.utils.data.DataLoader(ds, batch_size=4, collate_fn=PD.collate_specs, num_workers=0)
    specs, masks = next(iter(loader))
    assert len(specs) == 4 and isinstance(specs[0], PP.SampleSpec) and tuple(masks.shape) == (4, 56, 28) and int(masks[0].sum()) == 784


def test_load_image_depth_rescale_and_plain_files(dataset_files, tmp_path):
    """`_load_image` beside the reference's (pairdataset.py:81-98): a 16-bit nyuv2 `sync_depth` map (1e-4 m units) and a plain RGB file."""
    root, lists = dataset_files
    ref_mod = ref_import.load_reference_pairdataset()
    depth = (np.random.RandomState(3).rand(37, 53) * 65535).astype(np.uint16)
    depth[:4] = 0
    depth[4:8] = 10000                                           # 1 m -> 255 after the rescale, everything above saturates in convert("RGB")
    Image.fromarray(depth).save(tmp_path / "sync_depth_00001.png")
    Image.fromarray(picture(77, 37, 53)).save(tmp_path / "rgb_00001.png")
    sub = tmp_path / "list.json"
    sub.write_text(json.dumps([{"image_path": "rgb_00001.png", "target_path": "sync_depth_00001.png", "type": "nyuv2_image2depth"}]))
    ref = ref_mod.PairDataset(str(tmp_path), [str(sub)], transform=StandInStack("plain"), use_two_pairs=False)
    ours = PD.PairSpecDataset(str(tmp_path), [str(sub)], use_two_pairs=False)
    for name in ("sync_depth_00001.png", "rgb_00001.png"):
        a, b = ref._load_image(name), ours._load_image(name)
        assert a.mode == b.mode == "RGB" and np.array_equal(np.array(a), np.array(b))
    assert len(np.unique(np.array(ours._load_image("sync_depth_00001.png")))) > 20
