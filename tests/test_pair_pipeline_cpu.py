"""CPU tests of the training-input-pipeline row: pin oracle/pair_pipeline_oracle.py against Pillow and CPU torch
(the libraries the reference's transform stack bottoms out in; torchvision itself is absent, its glue is restated and documented as
unpinned), and check the host logic of painter_amd/pair_pipeline.py.  No GPU."""
import numpy as np
import pytest
import torch
from PIL import Image, ImageEnhance

from oracle import pair_pipeline_oracle as O
from painter_amd import pair_pipeline as PP
from tests import pair_pipeline_cases as C
from tests.seggpt_io_cases import picture

FACTORS = [0.0, 1.0, 0.6, 0.61234, 1.4, 1.39999, 0.8, 1.2, 0.95, 1.05, 2.5, 0.003]


def test_oracle_enhance_ops_are_pillow_bit_for_bit():
    img = picture(77, 97, 131)
    pil = Image.fromarray(img)
    assert np.array_equal(O.gray(img), np.array(pil.convert("L")))
    for f in FACTORS:
        assert np.array_equal(O.adjust_brightness(img, f), np.array(ImageEnhance.Brightness(pil).enhance(f))), f
        assert np.array_equal(O.adjust_contrast(img, f), np.array(ImageEnhance.Contrast(pil).enhance(f))), f
        assert np.array_equal(O.adjust_saturation(img, f), np.array(ImageEnhance.Color(pil).enhance(f))), f


def test_oracle_hsv_round_trip_is_pillow_over_all_colours():
    for r0 in range(0, 256, 16):
        grid = np.stack(np.meshgrid(np.arange(r0, r0 + 16), np.arange(256), np.arange(256), indexing="ij"), -1)
        grid = np.ascontiguousarray(grid.reshape(4096, 256, 3).astype(np.uint8))
        assert np.array_equal(O.rgb2hsv(grid), np.array(Image.fromarray(grid, "RGB").convert("HSV"))), r0
        assert np.array_equal(O.hsv2rgb(grid), np.array(Image.frombytes("HSV", (256, 4096), grid.tobytes()).convert("RGB"))), r0


@pytest.mark.parametrize("hue_factor", [-0.5, -0.1, -0.0312, 0.0, 0.05, 0.1, 0.5])
def test_oracle_adjust_hue_is_the_pil_recipe_torchvision_uses(hue_factor):
    img = picture(78, 64, 80)
    h, s, v = Image.fromarray(img).convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    np_h = (np_h.astype(np.int32) + O.hue_shift_byte(hue_factor)).astype(np.uint8)          # uint8 add with wrap-around
    ref = np.array(Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB"))
    assert np.array_equal(O.adjust_hue(img, O.hue_shift_byte(hue_factor)), ref)
    assert O.hue_shift_byte(hue_factor) == PP.hue_shift_byte(hue_factor)


@pytest.mark.parametrize("h,w,box,size", [(480, 640, (30, 50, 400, 500), (448, 448)), (375, 500, (0, 0, 375, 500), (448, 448)),
                                           (300, 700, (17, 123, 280, 333), (448, 448)), (448, 448, (0, 0, 448, 448), (448, 448)),
                                           (600, 450, (100, 0, 448, 300), (448, 448)), (64, 64, (5, 7, 9, 11), (32, 48))])
def test_oracle_resized_crop_is_pil_crop_then_resize(h, w, box, size):
    img = picture(h + w, h, w)
    i, j, bh, bw = box
    crop = Image.fromarray(img).crop((j, i, j + bw, i + bh))
    assert np.array_equal(O.resized_crop(img, box, size, False), np.array(crop.resize((size[1], size[0]), Image.BICUBIC)))
    assert np.array_equal(O.resized_crop(img, box, size, True), np.array(crop.resize((size[1], size[0]), Image.NEAREST)))


def test_oracle_to_tensor_normalize_values_and_flip():
    img = picture(5, 20, 30)
    for flip in (False, True):
        t = O.to_tensor_normalize(img, flip).numpy()
        a = img[:, ::-1] if flip else img
        ref = ((a.astype(np.float32) / np.float32(255)) - np.array(O.MEAN, np.float32)) / np.array(O.STD, np.float32)
        assert t.dtype == np.float32 and np.array_equal(t, ref.transpose(2, 0, 1))


def test_valid_rules_and_interpolation_modes():
    assert PP.interpolation_modes("nyuv2_image2depth") == ("bicubic", "bicubic")
    assert PP.interpolation_modes("coco_image2pose") == ("bicubic", "bicubic")
    assert PP.interpolation_modes("ade20k_image2semantic") == ("bicubic", "nearest")
    assert PP.interpolation_modes("ssid_2image_denoise") == ("nearest", "bicubic")
    assert PP.interpolation_modes("other") == ("bicubic", "bicubic")
    for t in ["nyuv2_image2depth", "ade20k_image2semantic", "coco_image2panoptic_sem_seg", "coco_image2pose", "coco_image2panoptic_inst", "x"]:
        assert PP.valid_rule(t) == O.valid_rule(t)
    black = float(((torch.zeros(3) - torch.tensor(O.MEAN)) / torch.tensor(O.STD))[0])
    tgt = torch.full((3, 8, 50), black)
    assert float(O.valid_map(tgt, "ade20k_image2semantic").sum()) == 0.0                      # all black -> ignored
    tgt[:, :2, :49] = 1.0                                                                      # 294 foreground elements
    assert float(O.valid_map(tgt, "coco_image2panoptic_inst").sum()) == 0.0
    assert float(O.valid_map(tgt, "coco_image2pose").sum()) == 0.0
    tgt[:, :2, :50] = 1.0                                                                      # 300
    assert float(O.valid_map(tgt, "coco_image2panoptic_inst").sum()) == tgt.numel()
    v = O.valid_map(tgt, "coco_image2pose")
    assert float(v.max()) == 10.0 and float(v.min()) == 1.0 and int((v == 10.0).sum()) == 300


def test_parameter_draws_have_the_documented_ranges():
    torch.manual_seed(0)
    for _ in range(50):
        i, j, h, w = PP.sample_resized_crop(375, 500, (0.3, 1.0))
        assert 0 <= i and 0 <= j and i + h <= 375 and j + w <= 500 and h * w >= 0.25 * 375 * 500
        ops, fac = PP.sample_color_jitter()
        assert len(ops) in (0, 4) and sorted(ops) == ([0, 1, 2, 3] if ops else [])
        for o, f in zip(ops, fac):
            lo, hi = {0: (0.6, 1.4), 1: (0.6, 1.4), 2: (0.8, 1.2), 3: (-0.1, 0.1)}[o]
            assert lo <= f <= hi
    i, j, h, w = PP.sample_resized_crop(100, 1000, (0.9999, 1.0))                              # aspect far outside [3/4, 4/3]: central fallback
    assert (h, w) == (100, 133) and i == 0 and j == (1000 - 133) // 2


def test_oracle_build_sample_shapes_and_specs():
    specs = C.batch_specs()
    img, tgt, valid = O.build_sample(C.oracle_spec(specs[2]))
    assert img.shape == tgt.shape == valid.shape == (3, 896, 448) and img.dtype == torch.float32
    with pytest.raises(RuntimeError, match="MI355X"):
        PP.DevicePairPipeline("cpu")
