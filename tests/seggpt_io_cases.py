"""Shared inputs for the SegGPT pre-/post-processing tests: seeded synthetic pictures, a stand-in for the network
and the oracle's end-to-end composition.  TEST INFRASTRUCTURE -- imports oracle/."""
import hashlib
import types

import numpy as np
import torch

from oracle import seggpt_io_oracle as O

RES, HRES, PATCH = 448, 448, 16            # seggpt_engine.py:57 hard-codes 448 x 448; the model's patch size is 16


def picture(seed, h, w, flat=False):
    """A seeded RGB uint8 picture: smooth gradients + blocks + noise (so bicubic taps see edges); `flat` = a palette-like mask."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    if flat:
        cells = rng.integers(0, 4, size=(h // 23 + 2, w // 31 + 2))
        palette = np.array([[0, 0, 0], [255, 0, 0], [0, 255, 64], [250, 250, 250]], np.uint8)
        return np.ascontiguousarray(palette[cells[yy // 23, xx // 31]])
    base = np.stack([(xx * 255) // max(w - 1, 1), (yy * 255) // max(h - 1, 1), ((xx + yy) * 7) % 256], axis=-1)
    blocks = rng.integers(0, 256, size=(h // 9 + 2, w // 11 + 2, 3))[yy // 9, xx // 11]
    noise = rng.integers(-40, 41, size=(h, w, 3))
    return np.clip((base + blocks) // 2 + noise, 0, 255).astype(np.uint8)


def patchify(canvas, p=PATCH):
    """Inverse of models_seggpt.py:376-389 unpatchify: [N][3][H][W] -> [N][L][p*p*3]."""
    n, c, h, w = canvas.shape
    x = canvas.reshape(n, c, h // p, p, w // p, p).permute(0, 2, 4, 3, 5, 1)
    return x.reshape(n, (h // p) * (w // p), p * p * c).contiguous()


def standin_tokens(imgs, tgts):
    """What the stand-in network returns for (imgs, tgts) float32 [N][3][2R][W] CPU tensors: a fixed float32 function of BOTH canvases
    and of every sample, so that any difference in resize / normalise / stitch reaches the output, with enough gain that the
    de-normalised picture saturates on both sides.  CPU torch only (the GPU tests move tensors to the host for this call)."""
    assert imgs.device.type == "cpu" and imgs.dtype == torch.float32
    c = imgs * 0.6
    c = c + tgts * 0.4
    c = c + imgs.flip(2) * 0.5
    c = c + tgts.flip(3) * 0.25
    c[0] = c[0] + c.mean(0)
    g = torch.Generator().manual_seed(1234)
    c = c + 0.05 * torch.randn(c.shape[1:], generator=g)
    return patchify(c)


class StandInModel:
    """Duck-types what run_one_image touches on the model (seggpt_engine.py:36-48): patch_embed.num_patches, seg_type, __call__,
    unpatchify."""
    seg_type = "instance"
    patch_size = PATCH

    def __init__(self):
        self.patch_embed = types.SimpleNamespace(num_patches=(2 * HRES // PATCH) * (RES // PATCH))
        self.calls = []

    def __call__(self, x, tgt, bool_masked_pos, valid, seg_type, feat_ensemble):
        self.calls.append(dict(n=x.shape[0], masked=int(bool_masked_pos.sum()), seg=float(seg_type.sum()), merge=feat_ensemble,
                               valid_ok=bool((valid == 1).all())))
        y = standin_tokens(x.detach().float().cpu(), tgt.detach().float().cpu())
        return None, y.to(x.device), bool_masked_pos

    def unpatchify(self, x):
        p = PATCH
        w = int((x.shape[1] * 0.5) ** .5)
        h = w * 2
        x = x.reshape(x.shape[0], h, w, p, p, 3).permute(0, 5, 1, 3, 2, 4)
        return x.reshape(x.shape[0], 3, h * p, w * p)


IMAGE_CASE = dict(query=(11, 301, 500), prompts=[(12, 480, 640), (13, 333, 200)])       # (seed, height, width)
VIDEO_CASE = dict(frames=[(21, 360, 640), (22, 360, 640), (23, 360, 640)], prompt=(24, 240, 320), num_frames=2)


def image_case_inputs():
    q = picture(*IMAGE_CASE["query"])
    prompts = [picture(*s) for s in IMAGE_CASE["prompts"]]
    targets = [picture(s[0] + 100, s[1], s[2], flat=True) for s in IMAGE_CASE["prompts"]]
    return q, prompts, targets


def oracle_inference_image(q, prompts, targets):
    """The oracle's composition of seggpt_engine.py:56-103 -> (stitched imgs, tgts, tokens of all samples, blended uint8 picture)."""
    image = O.pil_resize_bicubic(q, (RES, HRES))
    p = np.stack([O.pil_resize_bicubic(a, (RES, HRES)) for a in prompts])
    t = np.stack([O.pil_resize_nearest(a, (RES, HRES)) for a in targets])
    imgs, tgts = O.stitch(p, t, image)
    y = standin_tokens(torch.from_numpy(imgs), torch.from_numpy(tgts)).numpy()
    return imgs, tgts, y, O.blend(y[0], q, HRES, RES, PATCH)


def oracle_inference_frames(frames, img2, tgt2, num_frames):
    """The oracle's composition of the loop of seggpt_engine.py:130-179 -> list of blended frames, list of cached masks."""
    img2 = O.pil_resize_bicubic(img2, (RES, HRES))
    tgt2 = O.pil_resize_nearest(tgt2, (RES, HRES))
    fcache, tcache, outs, masks = [], [], [], []
    for frame in frames:
        image = O.pil_resize_bicubic(frame, (RES, HRES))
        p = np.stack([img2] + fcache)
        t = np.stack([tgt2] + tcache)
        div = [255.0] + [1.0] * len(tcache)
        imgs, tgts = O.stitch(p, t, image, div)
        y = standin_tokens(torch.from_numpy(imgs), torch.from_numpy(tgts)).numpy()
        if num_frames > 0:
            fcache.append(image)
            tcache.append(O.mask(y[0], HRES, RES, PATCH))
            if len(fcache) > num_frames:
                fcache.pop(0)
                tcache.pop(0)
        masks.append(O.mask(y[0], HRES, RES, PATCH))
        outs.append(O.blend(y[0], frame, HRES, RES, PATCH))
    return outs, masks


def digest(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(str(a.dtype).encode() + str(a.shape).encode() + a.tobytes()).hexdigest()
