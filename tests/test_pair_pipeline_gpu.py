"""GPU parity of the training-input-pipeline kernels (csrc/pair_io.hip and the box variants of csrc/seggpt_io.hip, through the C ABI
and painter_amd/pair_pipeline.py) against oracle/pair_pipeline_oracle.py -- itself pinned to Pillow and CPU torch
(tests/test_pair_pipeline_cpu.py) -- and against Pillow directly where one call does the step.  Byte and index work and the float32
elementwise steps: bit-exact.  The float32 bicubic crop: 1e-4 of the value range against torch's CPU kernel (measured 1e-5: different
summation order / fused multiply-adds on the host side; the cubic weights cancel)."""
import numpy as np
import pytest
import torch
from PIL import Image, ImageEnhance

from oracle import pair_pipeline_oracle as O
from tests import pair_pipeline_cases as C
from tests.seggpt_io_cases import picture

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from painter_amd import pair_pipeline as PP
    from painter_amd._lib import lib


@pytest.fixture(scope="module")
def pipe():
    return PP.DevicePairPipeline("cuda")


@pytest.mark.parametrize("h,w,box,size", [(480, 640, (30, 50, 400, 500), (448, 448)), (375, 500, (0, 0, 375, 500), (448, 448)),
                                           (300, 701, (17, 123, 280, 333), (448, 448)), (448, 448, (0, 0, 448, 448), (448, 448)),
                                           (600, 450, (100, 1, 448, 300), (448, 448)), (600, 450, (100, 1, 300, 448), (448, 448)),
                                           (64, 67, (5, 7, 9, 11), (32, 48)), (1200, 1999, (3, 5, 1100, 1901), (448, 448))])
def test_device_resized_crop_is_pil_crop_then_resize(pipe, h, w, box, size):
    img = picture(h + w, h, w)
    i, j, bh, bw = box
    crop = Image.fromarray(img).crop((j, i, j + bw, i + bh))
    dev = torch.from_numpy(img).cuda()
    for nearest, flt in ((False, Image.BICUBIC), (True, Image.NEAREST)):
        out = torch.zeros((size[0], size[1], 3), dtype=torch.uint8, device="cuda")
        pipe.resized_crop(dev, box, out, nearest)
        assert np.array_equal(out.cpu().numpy(), np.array(crop.resize((size[1], size[0]), flt))), nearest


def test_batched_resized_crop_is_pil_crop_then_resize_for_every_job(pipe):
    """pa_resized_crop_u8_batch: one job table for pictures of different sizes, boxes and interpolations (both passes, one pass skipped,
    both skipped, nearest with and without a size change), two launches -- every output equals PIL crop + resize."""
    cases = [(480, 640, (30, 50, 400, 500), False), (375, 500, (0, 0, 375, 500), True), (300, 701, (17, 123, 280, 333), False),
             (448, 448, (0, 0, 448, 448), False), (448, 448, (0, 0, 448, 448), True), (600, 450, (100, 1, 448, 300), False),
             (600, 450, (100, 1, 300, 448), False), (64, 67, (5, 7, 9, 11), True), (64, 67, (5, 7, 9, 11), False),
             (1200, 1999, (3, 5, 1100, 1901), False), (120, 97, (22, 31, 68, 55), True)]
    pictures = [picture(7 * k + h + w, h, w) for k, (h, w, _, _) in enumerate(cases)]
    out = torch.zeros((len(cases), 448, 448, 3), dtype=torch.uint8, device="cuda")
    for _ in range(2):                                          # twice: the second call reuses the pinned staging buffer and the tables
        pipe.resized_crop_batch(pictures, [c[2] for c in cases], [c[3] for c in cases], out)
        got = out.cpu().numpy()
        for k, (h, w, (i, j, bh, bw), near) in enumerate(cases):
            ref = np.array(Image.fromarray(pictures[k]).crop((j, i, j + bw, i + bh)).resize((448, 448), Image.NEAREST if near else Image.BICUBIC))
            assert np.array_equal(got[k], ref), (k, cases[k])
        out.zero_()


def test_second_crop_with_per_sample_modes_matches_the_single_mode_entry(pipe):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(5, 3, 896, 448, generator=g).cuda()
    boxes = [(10, 20, 700, 300), (0, 0, 896, 448), (100, 3, 512, 256), (5, 5, 448, 224), (0, 0, 896, 448)]
    modes = [0, 2, 1, 0, 1]
    got = pipe.resized_crop_tensor_modes(x, boxes, modes)
    bic = pipe.resized_crop_tensor(x, boxes, False)
    near = pipe.resized_crop_tensor(x, boxes, True)
    for b, m in enumerate(modes):
        ref = x[b] if m == 2 else (near[b] if m == 1 else bic[b])
        assert torch.equal(got[b], ref), b


def _jitter_batch():
    ops = [(PP.CONTRAST, PP.HUE, PP.BRIGHTNESS, PP.SATURATION), (PP.BRIGHTNESS, PP.SATURATION, PP.CONTRAST, PP.HUE), (),
           (PP.HUE, PP.SATURATION, PP.BRIGHTNESS, PP.CONTRAST), (PP.SATURATION,), (PP.CONTRAST, PP.CONTRAST, PP.HUE, PP.HUE)]
    fac = [(1.31, -0.07, 0.66, 1.18), (1.0, 0.0, 0.75, 0.1), (), (0.031, 0.93, 1.4, 1.0), (2.5,), (0.6, 1.39999, 0.5, -0.5)]
    return ops, fac


def test_device_color_jitter_matches_oracle_and_pillow(pipe):
    ops, fac = _jitter_batch()
    imgs = np.stack([picture(90 + b, 120, 136) for b in range(len(ops))])
    dev = torch.from_numpy(imgs).cuda()
    pipe.color_jitter(dev, ops, fac)
    got = dev.cpu().numpy()
    for b in range(len(ops)):
        ref = O.color_jitter(imgs[b], ops[b], [O.hue_shift_byte(f) if o == PP.HUE else f for o, f in zip(ops[b], fac[b])])
        assert np.array_equal(got[b], ref), b
    assert np.array_equal(got[4], np.array(ImageEnhance.Color(Image.fromarray(imgs[4])).enhance(2.5)))          # straight against Pillow
    # single ops over the factor list, batched: one sample per factor
    factors = [0.0, 1.0, 0.6, 0.61234, 1.4, 1.39999, 0.8, 1.2, 0.95, 1.05, 2.5, 0.003]
    base = picture(77, 97, 131)
    pil = Image.fromarray(base)
    for op, enh in ((PP.BRIGHTNESS, ImageEnhance.Brightness), (PP.CONTRAST, ImageEnhance.Contrast), (PP.SATURATION, ImageEnhance.Color)):
        dev = torch.from_numpy(np.stack([base] * len(factors))).cuda()
        pipe.color_jitter(dev, [(op,)] * len(factors), [(f,) for f in factors])
        out = dev.cpu().numpy()
        for k, f in enumerate(factors):
            assert np.array_equal(out[k], np.array(enh(pil).enhance(f))), (op, f)


def test_device_hsv_round_trip_over_all_colours(pipe):
    """Every RGB colour through rgb2hsv -> H + shift -> hsv2rgb on the device, against the Pillow-pinned oracle."""
    grid = np.stack(np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing="ij"), -1).astype(np.uint8).reshape(1, 4096, 4096, 3)
    for hue in (0.1451,):
        dev = torch.from_numpy(grid).cuda()
        pipe.color_jitter(dev, [(PP.HUE,)], [(hue,)])
        assert np.array_equal(dev.cpu().numpy()[0], O.adjust_hue(grid[0], O.hue_shift_byte(hue))), hue


def test_device_to_tensor_normalize_matches_oracle(pipe):
    imgs = np.stack([picture(30 + b, 448, 448) for b in range(3)])
    imgs[0, :2, :, 0] = np.arange(448) % 256                                    # every byte value
    flips = [False, True, True]
    canvas = torch.full((3, 3, 896, 448), 7.0, device="cuda")
    pipe.to_tensor_normalize(torch.from_numpy(imgs).cuda(), flips, canvas, 448)
    got = canvas.cpu()
    assert float((got[:, :, :448] - 7.0).abs().max()) == 0.0                    # only rows [448, 896) are written
    for b in range(3):
        assert torch.equal(got[b, :, 448:], O.to_tensor_normalize(imgs[b], flips[b])), b


@pytest.mark.parametrize("nearest", [False, True])
def test_device_float_crop_matches_torch_interpolate(pipe, nearest):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 896, 448, generator=g) * 1.2
    boxes = [(100, 40, 600, 300), (0, 0, 896, 448), (0, 10, 896, 400), (448, 224, 448, 224)]
    got = pipe.resized_crop_tensor(x.cuda(), boxes, nearest).cpu()
    for b, box in enumerate(boxes):
        ref = O.resized_crop_tensor(x[b], box, (896, 448), nearest)
        if nearest:
            assert torch.equal(got[b], ref), b
        else:
            assert float((got[b] - ref).abs().max()) <= 1e-4 * float(ref.abs().max()), (b, float((got[b] - ref).abs().max()))


def test_device_valid_rules_match_oracle(pipe):
    black = (torch.zeros(3) - torch.tensor(O.MEAN)) / torch.tensor(O.STD)
    g = torch.Generator().manual_seed(6)
    types = ["ade20k_image2semantic", "nyuv2_image2depth", "coco_image2pose", "coco_image2pose", "coco_image2panoptic_inst",
             "coco_image2panoptic_inst", "ssid_2image_denoise", "coco_image2panoptic_sem_seg"]
    tg = torch.randn(len(types), 3, 896, 448, generator=g)
    tg[0, :, :400] = black[:, None, None]
    tg[1, :, 100:300, 50:60] = black[:, None, None] - 1e-3
    for b, n in ((2, 299), (3, 300), (4, 299), (5, 300)):                      # the 300-foreground-element rule, either side
        tg[b] = black[:, None, None]
        flat = tg[b].reshape(-1)
        flat[torch.randperm(flat.numel(), generator=g)[:n]] = 1.5
    got = pipe.valid_map(tg.cuda(), types).cpu()
    for b, t in enumerate(types):
        assert torch.equal(got[b], O.valid_map(tg[b], t)), (b, t)
    assert float(got[2].sum()) == 0.0 and float(got[3].max()) == 10.0 and float(got[4].sum()) == 0.0 and float(got[5].min()) == 1.0


def test_build_batch_matches_the_oracle_sample_by_sample(pipe):
    specs = C.batch_specs()
    imgs, tgts, valid = pipe.build_batch(specs)
    assert tuple(imgs.shape) == (4, 3, 896, 448) and imgs.dtype == torch.float32
    imgs, tgts, valid = imgs.cpu(), tgts.cpu(), valid.cpu()
    for b, s in enumerate(specs):
        ri, rt, rv = O.build_sample(C.oracle_spec(s))
        for name, got, ref, near in (("imgs", imgs[b], ri, s.interpolation[0] == "nearest"), ("tgts", tgts[b], rt, s.interpolation[1] == "nearest")):
            if s.seccrop is None or near:
                assert torch.equal(got, ref), (b, name)
            else:
                assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max()), (b, name)
        if s.seccrop is None or s.interpolation[1] == "nearest" or PP.valid_rule(s.pair_type)[0] == PP.VALID_NONE:
            assert torch.equal(valid[b], rv), b


def test_c_abi_rejects_bad_arguments():
    buf = torch.zeros(256, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    p = buf.data_ptr()
    assert lib.pa_resample_u8_box(p, 5, 4, 4, p, 4, 2, 3, p, p, 1, 0, s) != 0                     # row pitch smaller than the row
    assert lib.pa_to_tensor_normalize(p, p, p, 1, 4, 4, 6, 3, s) != 0                              # rows [3, 7) do not fit a 6-row canvas
    assert lib.pa_resized_crop_f32(p, p, p, 1, 3, 4, 4, 0, s) != 0                                 # in place
    assert lib.pa_color_jitter(p, p, p, None, p, 0, 4, 4, s) != 0
    assert lib.pa_pair_valid(p, p, p, p, p, 1, 0, s) != 0
    assert lib.pa_resized_crop_f32_modes(p, p + 64, p, None, 1, 3, 4, 4, s) != 0                   # no modes array
    assert lib.pa_resized_crop_u8_batch(p, p, 0, 4, 4, 4, 4, s) != 0                               # no jobs
    assert lib.pa_resized_crop_u8_batch(p, p, 1, 4, 30000, 4, 4, s) != 0                           # a source row must fit the LDS stage
