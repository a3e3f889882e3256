"""CPU tests of the SegGPT pre-/post-processing row: pin oracle/seggpt_io_oracle.py against Pillow, CPU torch, the
reference-generated golden digests (tests/golden/seggpt_io.npz, made by tests/golden/make_golden_seggpt_io.py from the unmodified
seggpt_engine.py) and -- when /root/reference is mounted -- the reference functions themselves; and check the host tables the product
uploads (painter_amd/resample.py) against the oracle's.  No GPU, no compute calls into libpainter_hip.so."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from PIL import Image

from oracle import ref_import
from oracle import seggpt_io_oracle as O
from painter_amd import resample as RS
from tests import seggpt_io_cases as C

SIZES = [(37, 53, 56, 56), (100, 80, 56, 56), (23, 31, 56, 56), (56, 56, 56, 56), (60, 56, 56, 56), (200, 300, 64, 48),
         (17, 19, 40, 33), (333, 211, 47, 101), (50, 50, 100, 100), (50, 50, 125, 70), (1, 9, 4, 4), (301, 500, 448, 448)]


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "seggpt_io.npz"))


@pytest.mark.parametrize("h,w,oh,ow", SIZES)
def test_oracle_resize_is_pillow_bit_for_bit(h, w, oh, ow):
    a = C.picture(h * 1000 + w, h, w)
    assert np.array_equal(O.pil_resize_bicubic(a, (ow, oh)), np.array(Image.fromarray(a).resize((ow, oh))))
    assert np.array_equal(O.pil_resize_nearest(a, (ow, oh)), np.array(Image.fromarray(a).resize((ow, oh), Image.NEAREST)))


def test_oracle_nearest_table_is_torch_interpolate_for_the_references_tensor():
    rng = np.random.default_rng(3)
    cases = [(448, 1080), (448, 1920), (448, 448), (448, 896), (448, 301), (448, 500), (448, 3), (448, 4000), (7, 100)]
    cases += [(int(a), int(b)) for a, b in zip(rng.integers(2, 600, 60), rng.integers(1, 3000, 60))]
    for i, o in cases:
        t = torch.arange(i, dtype=torch.float64).reshape(i, 1, 1).expand(i, 1, 3).contiguous()          # [H][W][3] like `output`
        r = F.interpolate(t[None, ...].permute(0, 3, 1, 2), size=[o, 1], mode='nearest').permute(0, 2, 3, 1)[0]
        assert np.array_equal(r[:, 0, 0].numpy().astype(np.int64), O.torch_nearest_table(i, o)), (i, o)


def test_product_tables_match_the_oracles():
    rng = np.random.default_rng(5)
    cases = [(1080, 448), (1920, 448), (448, 448), (300, 448), (448, 1080), (449, 448), (4000, 448), (2, 448), (448, 2), (1, 5), (5, 1)]
    cases += [(int(a), int(b)) for a, b in zip(rng.integers(1, 2500, 60), rng.integers(1, 900, 60))]
    for i, o in cases:
        b, k, ks = O.pil_coeffs(i, o)
        b2, k2, ks2 = RS.bicubic_tables(i, o)
        assert ks == ks2 and np.array_equal(b, b2) and np.array_equal(k, k2), (i, o)
        assert np.array_equal(O.pil_nearest_table(i, o), RS.pil_nearest_table(i, o)), (i, o)
        assert np.array_equal(O.torch_nearest_table(i, o), RS.torch_nearest_table(i, o)), (i, o)


def test_oracle_mask_is_the_references_torch_expression():
    """seggpt_engine.py:166-171 on a tensor with the layout run_one_image returns, including means that sit exactly on 128."""
    g = torch.Generator().manual_seed(7)
    y = torch.randn(1, (2 * C.HRES // C.PATCH) * (C.RES // C.PATCH), C.PATCH * C.PATCH * 3, generator=g)
    y[0, -40:] = (128.0 / 255 - torch.tensor(O.IMAGENET_MEAN, dtype=torch.float64).repeat(256)
                  / 1).float() / torch.tensor(O.IMAGENET_STD).repeat(256).float()       # de-normalises to ~128 in every channel
    m = C.StandInModel()
    img = m.unpatchify(y)
    img = torch.einsum('nchw->nhwc', img)
    output = img[0, img.shape[1] // 2:, :, :]
    output = torch.clip((output * O.IMAGENET_STD + O.IMAGENET_MEAN) * 255, 0, 255)
    ref = output.mean(-1).gt(128).float().unsqueeze(-1).expand(-1, -1, 3).numpy()
    assert np.array_equal(output.numpy(), O.decode(y[0].numpy(), C.HRES, C.RES, C.PATCH))
    assert np.array_equal(ref.astype(np.uint8), O.mask(y[0].numpy(), C.HRES, C.RES, C.PATCH))


def test_oracle_pipeline_reproduces_the_reference_image_outputs(golden):
    q, prompts, targets = C.image_case_inputs()
    imgs, tgts, _, out = C.oracle_inference_image(q, prompts, targets)
    assert C.digest(imgs) == str(golden["image_imgs_digest"]) and C.digest(tgts) == str(golden["image_tgts_digest"])
    assert np.array_equal(out[::25, ::25], golden["image_out_sample"])
    assert C.digest(out) == str(golden["image_out_digest"])
    _, _, _, out1 = C.oracle_inference_image(q, prompts[:1], targets[:1])
    assert C.digest(out1) == str(golden["image1_out_digest"])


def test_oracle_pipeline_reproduces_the_reference_video_outputs(golden):
    vc = C.VIDEO_CASE
    frames = [C.picture(*s) for s in vc["frames"]]
    prompt = C.picture(*vc["prompt"])
    prompt_t = C.picture(vc["prompt"][0] + 100, vc["prompt"][1], vc["prompt"][2], flat=True)
    outs, _ = C.oracle_inference_frames(frames, prompt, prompt_t, vc["num_frames"])
    assert np.array_equal(outs[-1][::20, ::20], golden["video_out_sample"])
    for i, o in enumerate(outs):
        assert C.digest(o) == str(golden["video_out_digest_%d" % i]), i


@pytest.mark.skipif(not ref_import.reference_available(), reason="needs /root/reference (build container only)")
def test_reference_run_one_image_matches_oracle_decode():
    """The unmodified run_one_image (seggpt_engine.py:26-53) with the stand-in network, against stitch + decode of the oracle."""
    eng = ref_import.load_reference_seggpt_engine()
    q, prompts, targets = C.image_case_inputs()
    image = O.pil_resize_bicubic(q, (C.RES, C.HRES))
    p = np.stack([O.pil_resize_bicubic(a, (C.RES, C.HRES)) for a in prompts])
    t = np.stack([O.pil_resize_nearest(a, (C.RES, C.HRES)) for a in targets])
    # the arrays the reference builds at :73-92, restated with its own expressions
    img = np.stack([(np.concatenate((a / 255., image / 255.), axis=0) - eng.imagenet_mean) / eng.imagenet_std for a in p])
    tgt = np.stack([(np.concatenate((a / 255., a / 255.), axis=0) - eng.imagenet_mean) / eng.imagenet_std for a in t])
    out = eng.run_one_image(img, tgt, C.StandInModel(), "cpu")
    imgs, tgts = O.stitch(p, t, image)
    y = C.standin_tokens(torch.from_numpy(imgs), torch.from_numpy(tgts)).numpy()
    assert out.dtype == torch.float64 and np.array_equal(out.numpy(), O.decode(y[0], C.HRES, C.RES, C.PATCH))


def test_engine_module_mirrors_the_reference_interface():
    import inspect

    from painter_amd import seggpt_engine as E
    assert list(inspect.signature(E.run_one_image).parameters) == ["img", "tgt", "model", "device"]
    assert list(inspect.signature(E.inference_image).parameters) == ["model", "device", "img_path", "img2_paths", "tgt2_paths", "out_path"]
    assert list(inspect.signature(E.inference_video).parameters) == ["model", "device", "vid_path", "num_frames", "img2_paths",
                                                                     "tgt2_paths", "out_path"]
    c = E.Cache(2)
    for i in range(4):
        c.append(i)
    assert list(c) == [2, 3]
    z = E.Cache(0)
    z.append(1)
    assert list(z) == []
    with pytest.raises(RuntimeError, match="MI355X"):
        E.DeviceIO("cpu")
    if ref_import.reference_available():
        eng = ref_import.load_reference_seggpt_engine()
        for name in ("run_one_image", "inference_image", "inference_video"):
            assert list(inspect.signature(getattr(eng, name)).parameters) == list(inspect.signature(getattr(E, name)).parameters)


@pytest.mark.skipif(not ref_import.reference_available(), reason="needs /root/reference (build container only)")
def test_oracle_pipeline_on_the_references_example_pictures(tmp_path):
    """the unmodified inference_image on the repository's own examples (JPEG pictures of several sizes, PNG targets), stand-in
    network, against the oracle's composition -- real decoded pictures instead of the seeded synthetic ones."""
    eng = ref_import.load_reference_seggpt_engine()
    ex = os.path.join(ref_import.SEGGPT_DIR, "examples")
    rgb = lambda name: np.array(Image.open(os.path.join(ex, name)).convert("RGB"))          # noqa: E731
    for query, prompts in (("hmbb_3.jpg", ["hmbb_1", "hmbb_2"]), ("video_3.jpg", ["video_1"])):
        out_path = str(tmp_path / (query + ".png"))
        eng.inference_image(C.StandInModel(), "cpu", os.path.join(ex, query), [os.path.join(ex, p + ".jpg") for p in prompts],
                            [os.path.join(ex, p + "_target.png") for p in prompts], out_path)
        _, _, _, out = C.oracle_inference_image(rgb(query), [rgb(p + ".jpg") for p in prompts], [rgb(p + "_target.png") for p in prompts])
        assert np.array_equal(np.array(Image.open(out_path)), out), query
