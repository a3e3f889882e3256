"""GPU parity of the SegGPT pre-/post-processing kernels (csrc/seggpt_io.hip through the C ABI and painter_amd/seggpt_engine.py)
against oracle/seggpt_io_oracle.py -- itself pinned to Pillow, CPU torch and the unmodified reference (tests/test_seggpt_io_cpu.py) --
and against the digests the unmodified reference produced (tests/golden/seggpt_io.npz).  Everything here is byte / index / float64
work with a fixed operation order: the bar is bit-exact (np.array_equal), no tolerance."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import seggpt_io_oracle as O
from tests import seggpt_io_cases as C

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from painter_amd import seggpt_engine as E
    from painter_amd._lib import lib


@pytest.fixture(scope="module")
def io():
    return E.DeviceIO("cuda", res=C.RES, hres=C.HRES, patch=C.PATCH)


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "seggpt_io.npz"))


SIZES = [(37, 53, 56, 56), (100, 80, 56, 56), (56, 56, 56, 56), (60, 56, 56, 56), (56, 60, 56, 56), (200, 300, 64, 48), (17, 19, 40, 33),
         (333, 211, 47, 101), (50, 50, 125, 70), (1, 9, 4, 4), (301, 500, 448, 448), (1080, 1920, 448, 448), (3, 1200, 448, 448)]


@pytest.mark.parametrize("h,w,oh,ow", SIZES)
def test_device_resize_is_pillow_bit_for_bit(io, h, w, oh, ow):
    a = C.picture(h * 1000 + w, h, w)
    dev = io.upload(a)
    got = io.resize(dev, (ow, oh)).cpu().numpy()
    assert np.array_equal(got, np.array(Image.fromarray(a).resize((ow, oh))))
    if h * w <= 400 * 600:
        assert np.array_equal(got, O.pil_resize_bicubic(a, (ow, oh)))
    got = io.resize(dev, (ow, oh), nearest=True).cpu().numpy()
    assert np.array_equal(got, np.array(Image.fromarray(a).resize((ow, oh), Image.NEAREST)))
    if oh * ow <= 128 * 128:
        assert np.array_equal(got, O.pil_resize_nearest(a, (ow, oh)))


def test_device_stitch_matches_oracle(io):
    rng = np.random.default_rng(31)
    prompts = rng.integers(0, 256, (3, C.HRES, C.RES, 3), dtype=np.uint8)
    targets = rng.integers(0, 256, (3, C.HRES, C.RES, 3), dtype=np.uint8)
    targets[1:] = targets[1:] & 1                                   # cached video masks: {0,1}, used undivided
    query = rng.integers(0, 256, (C.HRES, C.RES, 3), dtype=np.uint8)
    div = [255.0, 1.0, 1.0]
    imgs, tgts = io.stitch(torch.from_numpy(prompts).cuda(), torch.from_numpy(targets).cuda(), torch.from_numpy(query).cuda(), div)
    ri, rt = O.stitch(prompts, targets, query, div)
    assert imgs.dtype == torch.float32 and tuple(imgs.shape) == (3, 3, 2 * C.HRES, C.RES)
    assert np.array_equal(imgs.cpu().numpy(), ri) and np.array_equal(tgts.cpu().numpy(), rt)
    # all 256 byte values through the normalisation, default divisor
    ramp = np.broadcast_to(np.arange(256, dtype=np.uint8).repeat(2)[:C.RES, None], (C.HRES, C.RES, 3)).copy()
    ramp = np.ascontiguousarray(np.roll(ramp, 7, axis=0))
    imgs, tgts = io.stitch(torch.from_numpy(ramp[None]).cuda(), torch.from_numpy(ramp[None]).cuda(), torch.from_numpy(ramp).cuda())
    ri, rt = O.stitch(ramp[None], ramp[None], ramp)
    assert np.array_equal(imgs.cpu().numpy(), ri) and np.array_equal(tgts.cpu().numpy(), rt)


def _tokens(seed):
    g = torch.Generator().manual_seed(seed)
    y = torch.randn((2 * C.HRES // C.PATCH) * (C.RES // C.PATCH), C.PATCH * C.PATCH * 3, generator=g) * 1.5
    near = (128.0 / 255 - torch.tensor(O.IMAGENET_MEAN).repeat(256)) / torch.tensor(O.IMAGENET_STD).repeat(256)
    y[-60:-30] = near.float()                                        # de-normalises to ~128 per channel: the mask threshold
    y[-30:] = 40.0                                                   # saturates at 255: blend factor 0.6 * 255 / 255 + 0.4
    y[-90:-60] = -40.0                                               # saturates at 0: blend factor exactly 0.4
    return y


@pytest.mark.parametrize("seed", [1, 2])
def test_device_decode_and_mask_match_oracle(io, seed):
    y = _tokens(seed)
    yd = y.cuda()
    assert np.array_equal(io.decode(yd).cpu().numpy(), O.decode(y.numpy(), C.HRES, C.RES, C.PATCH))
    assert np.array_equal(io.mask(yd).cpu().numpy(), O.mask(y.numpy(), C.HRES, C.RES, C.PATCH))
    assert np.array_equal(io.decode(yd[None].expand(2, -1, -1)).cpu().numpy(), O.decode(y.numpy(), C.HRES, C.RES, C.PATCH))


@pytest.mark.parametrize("h0,w0", [(301, 500), (1080, 1920), (448, 448), (896, 896), (100, 37), (449, 447), (2160, 3840)])
def test_device_blend_matches_oracle(io, h0, w0):
    y = _tokens(3)
    frame = C.picture(h0 + w0, h0, w0)
    frame[: h0 // 8] = 255                                           # 255 * 0.9999999999999999 must truncate as numpy does
    frame[h0 // 8: h0 // 4] = 200
    got = io.blend(y.cuda(), io.upload(frame)).cpu().numpy()
    assert np.array_equal(got, O.blend(y.numpy(), frame, C.HRES, C.RES, C.PATCH))


def _write_case(tmp_path):
    q, prompts, targets = C.image_case_inputs()
    Image.fromarray(q).save(tmp_path / "q.png")
    pp, tp = [], []
    for i, (a, b) in enumerate(zip(prompts, targets)):
        pp.append(str(tmp_path / ("p%d.png" % i)))
        tp.append(str(tmp_path / ("t%d.png" % i)))
        Image.fromarray(a).save(pp[-1])
        Image.fromarray(b).save(tp[-1])
    return str(tmp_path / "q.png"), pp, tp


def test_inference_image_reproduces_the_references_output_file(tmp_path, golden):
    """painter_amd.seggpt_engine.inference_image with the stand-in network == what the unmodified reference wrote (golden digest)."""
    qp, pp, tp = _write_case(tmp_path)
    model = C.StandInModel()
    E.inference_image(model, "cuda", qp, pp, tp, str(tmp_path / "out.png"))
    out = np.array(Image.open(tmp_path / "out.png"))
    assert model.calls == [dict(n=2, masked=784, seg=2.0, merge=0, valid_ok=True)], model.calls
    assert np.array_equal(out[::25, ::25], golden["image_out_sample"])
    assert C.digest(out) == str(golden["image_out_digest"])
    model1 = C.StandInModel()
    E.inference_image(model1, "cuda", qp, pp[:1], tp[:1], str(tmp_path / "out1.png"))
    assert model1.calls[0]["merge"] == -1
    assert C.digest(np.array(Image.open(tmp_path / "out1.png"))) == str(golden["image1_out_digest"])


def test_inference_frames_reproduces_the_references_video_frames(golden):
    vc = C.VIDEO_CASE
    frames = [C.picture(*s) for s in vc["frames"]]
    prompt = C.picture(*vc["prompt"])
    prompt_t = C.picture(vc["prompt"][0] + 100, vc["prompt"][1], vc["prompt"][2], flat=True)
    model = C.StandInModel()
    outs = list(E.inference_frames(model, "cuda", iter(frames), vc["num_frames"], prompt, prompt_t))
    assert [c["n"] for c in model.calls] == [1, 2, 3]
    assert np.array_equal(outs[-1][::20, ::20], golden["video_out_sample"])
    for i, o in enumerate(outs):
        assert C.digest(o) == str(golden["video_out_digest_%d" % i]), i
    assert len(list(E.inference_frames(C.StandInModel(), "cuda", iter(frames[:2]), 0, prompt, prompt_t))) == 2     # Cache(0): no prompts kept


def test_run_one_image_keeps_the_reference_signature_and_result():
    q, prompts, targets = C.image_case_inputs()
    image = O.pil_resize_bicubic(q, (C.RES, C.HRES))
    p = np.stack([O.pil_resize_bicubic(a, (C.RES, C.HRES)) for a in prompts])
    t = np.stack([O.pil_resize_nearest(a, (C.RES, C.HRES)) for a in targets])
    img = np.stack([(np.concatenate((a / 255., image / 255.), axis=0) - E.imagenet_mean) / E.imagenet_std for a in p])
    tgt = np.stack([(np.concatenate((a / 255., a / 255.), axis=0) - E.imagenet_mean) / E.imagenet_std for a in t])
    out = E.run_one_image(img, tgt, C.StandInModel(), "cuda")
    imgs, tgts = O.stitch(p, t, image)
    y = C.standin_tokens(torch.from_numpy(imgs), torch.from_numpy(tgts)).numpy()
    assert out.dtype == torch.float64 and out.device.type == "cpu"
    assert np.array_equal(out.numpy(), O.decode(y[0], C.HRES, C.RES, C.PATCH))


def test_inference_image_with_the_vit_large_network(tmp_path):
    """The real seggpt_vit_large_patch16_input896x448 (random weights, bf16 build) between the device pre- and post-processing:
    the written picture must be the oracle's blend of the tokens the network itself produced."""
    from painter_amd import models_seggpt
    torch.manual_seed(0)
    net = models_seggpt.seggpt_vit_large_patch16_input896x448(compute_dtype="bf16").to("cuda").eval()
    net.seg_type = "instance"
    seen = {}
    inner = net.forward

    def spy(*a, **k):
        r = inner(*a, **k)
        seen["imgs"], seen["y"] = a[0], r[1]
        return r
    net.forward = spy
    qp, pp, tp = _write_case(tmp_path)
    E.inference_image(net, "cuda", qp, pp, tp, str(tmp_path / "out.png"))
    out = np.array(Image.open(tmp_path / "out.png"))
    y0 = seen["y"][0].float().cpu().numpy()
    assert seen["imgs"].shape == (2, 3, 896, 448) and np.isfinite(y0).all()
    q, prompts, targets = C.image_case_inputs()
    assert np.array_equal(out, O.blend(y0, q, C.HRES, C.RES, C.PATCH))
    # the video loop with the real network (feature ensemble from the second frame on); the caller's grad mode is left alone
    assert torch.is_grad_enabled()
    frames = [C.picture(41, 270, 480), C.picture(42, 270, 480)]
    for i, blended in enumerate(E.inference_frames(net, "cuda", iter(frames), 1, prompts[0], targets[0])):
        assert torch.is_grad_enabled()
        assert seen["imgs"].shape[0] == i + 1
        assert np.array_equal(blended, O.blend(seen["y"][0].float().cpu().numpy(), frames[i], C.HRES, C.RES, C.PATCH))


def test_c_abi_rejects_bad_arguments():
    buf = torch.zeros(64, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    assert lib.pa_resample_u8(buf.data_ptr(), 4, 4, buf.data_ptr(), 4, 2, 5, buf.data_ptr(), buf.data_ptr(), 1, 0, s) != 0      # 5 channels
    assert lib.pa_resample_u8(buf.data_ptr(), 4, 4, buf.data_ptr(), 3, 2, 3, buf.data_ptr(), buf.data_ptr(), 1, 0, s) != 0      # rows change in a horizontal pass
    assert lib.pa_seggpt_decode(buf.data_ptr(), buf.data_ptr(), 30, 32, 16, s) != 0                                             # 30 % 16
    assert lib.pa_seggpt_stitch(buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), 0, 4, 4, s) != 0
