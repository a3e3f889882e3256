"""Drop-in for the reference's `seggpt_engine` module (SegGPT/SegGPT_inference/seggpt_engine.py) with the pre- and
post-processing on the MI355X.

Same entry points, argument meaning and outputs as the reference: `run_one_image` (:26-53), `inference_image` (:56-103),
`inference_video` (:106-181), `Cache` (:13-23).  What moved: the reference decodes a file, then resizes (PIL), normalises and
stitches (numpy float64), runs the model, and de-normalises, clips, nearest-resizes (CPU torch float64), blends (numpy float64) and
thresholds the video prompt mask on the host; here only file decode / encode stay on the host and every array operation is one
kernel of csrc/seggpt_io.hip on the decoded uint8 pixels, bit-exact with the host path (tests/test_seggpt_io_gpu.py).  A 1080p
frame crosses PCIe once in each direction as uint8 (6 MB) instead of never leaving the host as float64 (50 MB per temporary).

`torch.manual_seed(2)` (:93, :160 -- "make random mask reproducible") is not replayed: nothing on this path draws random numbers.
There is no CPU fallback: a missing libpainter_hip.so raises on first use.
"""
import numpy as np
import torch
from PIL import Image

from . import resample as RS
from ._lib import check, lib

imagenet_mean = np.array([0.485, 0.456, 0.406])
imagenet_std = np.array([0.229, 0.224, 0.225])


class Cache(list):
    """seggpt_engine.py:13-23: a FIFO of at most `max_size` entries; max_size <= 0 keeps nothing."""

    def __init__(self, max_size=0):
        super().__init__()
        self.max_size = max_size

    def append(self, x):
        if self.max_size <= 0:
            return
        super().append(x)
        if len(self) > self.max_size:
            self.pop(0)


def _stream():
    return torch.cuda.current_stream().cuda_stream


class DeviceIO:
    """The device-side image operations of this path.  Images are uint8 [H][W][3] CUDA tensors (RGB)."""

    def __init__(self, device, res=448, hres=448, patch=16):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("painter_amd.seggpt_engine runs its image kernels on an MI355X only (no CPU fallback); got %s" % device)
        self.res, self.hres, self.patch = res, hres, patch          # reference naming: res = width, hres = height (:57)
        self._tables = {}

    # ---- plumbing
    def upload(self, array):
        a = np.ascontiguousarray(array)
        assert a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3, (a.dtype, a.shape)
        return torch.from_numpy(a).to(self.device)

    def _table(self, kind, in_size, out_size):
        key = (kind, in_size, out_size)
        t = self._tables.get(key)
        if t is None:
            if kind == "bicubic":
                bounds, coeffs, ksize = RS.bicubic_tables(in_size, out_size)
                t = (torch.from_numpy(bounds).to(self.device), torch.from_numpy(coeffs).to(self.device), ksize)
            else:
                fn = RS.pil_nearest_table if kind == "pil_nearest" else RS.torch_nearest_table
                t = torch.from_numpy(fn(in_size, out_size)).to(self.device)
            self._tables[key] = t
        return t

    @staticmethod
    def _img(t):
        assert t.is_cuda and t.dtype == torch.uint8 and t.dim() == 3 and t.is_contiguous(), (t.dtype, tuple(t.shape))
        return t

    # ---- PIL.Image.resize
    def resize(self, img, size, nearest=False):
        """`Image.resize(size)` (BICUBIC) or `Image.resize(size, Image.NEAREST)`; size = (width, height) as in PIL."""
        img = self._img(img)
        h, w, c = img.shape
        ow, oh = size
        if (w, h) == (ow, oh):
            return img.clone()
        if nearest:
            out = torch.empty((oh, ow, c), dtype=torch.uint8, device=self.device)
            check(lib.pa_gather_u8(img.data_ptr(), h, w, out.data_ptr(), oh, ow, c, self._table("pil_nearest", h, oh).data_ptr(),
                                   self._table("pil_nearest", w, ow).data_ptr(), _stream()), "pa_gather_u8")
            return out
        if w != ow:
            bounds, coeffs, ksize = self._table("bicubic", w, ow)
            mid = torch.empty((h, ow, c), dtype=torch.uint8, device=self.device)
            check(lib.pa_resample_u8(img.data_ptr(), h, w, mid.data_ptr(), h, ow, c, bounds.data_ptr(), coeffs.data_ptr(), ksize, 0,
                                     _stream()), "pa_resample_u8")
            img = mid
        if h != oh:
            bounds, coeffs, ksize = self._table("bicubic", h, oh)
            out = torch.empty((oh, ow, c), dtype=torch.uint8, device=self.device)
            check(lib.pa_resample_u8(img.data_ptr(), h, ow, out.data_ptr(), oh, ow, c, bounds.data_ptr(), coeffs.data_ptr(), ksize, 1,
                                     _stream()), "pa_resample_u8")
            img = out
        return img

    # ---- seggpt_engine.py:65-92 / :139-157
    def stitch(self, prompts, targets, query, target_div=None):
        """prompts, targets: uint8 [N][hres][res][3]; query: uint8 [hres][res][3] -> (imgs, tgts) float32 [N][3][2*hres][res]."""
        n = prompts.shape[0]
        want = (n, self.hres, self.res, 3)
        assert tuple(prompts.shape) == want and tuple(targets.shape) == want and tuple(query.shape) == want[1:], \
            (tuple(prompts.shape), tuple(targets.shape), tuple(query.shape))
        for t in (prompts, targets, query):
            assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()
        div = torch.full((n,), 255.0, dtype=torch.float64) if target_div is None else torch.as_tensor(target_div, dtype=torch.float64)
        assert div.numel() == n
        div = div.to(self.device)
        imgs = torch.empty((n, 3, 2 * self.hres, self.res), dtype=torch.float32, device=self.device)
        tgts = torch.empty_like(imgs)
        check(lib.pa_seggpt_stitch(prompts.data_ptr(), targets.data_ptr(), div.data_ptr(), query.data_ptr(), imgs.data_ptr(),
                                   tgts.data_ptr(), n, self.hres, self.res, _stream()), "pa_seggpt_stitch")
        return imgs, tgts

    # ---- model output -> pictures
    def _pred0(self, pred):
        p0 = pred[0] if pred.dim() == 3 else pred
        p0 = p0.detach().to(torch.float32).contiguous()
        assert p0.is_cuda and tuple(p0.shape) == ((2 * self.hres // self.patch) * (self.res // self.patch), self.patch * self.patch * 3), \
            tuple(p0.shape)
        return p0

    def decode(self, pred):
        """:49-53 -> float64 [hres][res][3] in [0, 255]."""
        p0 = self._pred0(pred)
        out = torch.empty((self.hres, self.res, 3), dtype=torch.float64, device=self.device)
        check(lib.pa_seggpt_decode(p0.data_ptr(), out.data_ptr(), self.hres, self.res, self.patch, _stream()), "pa_seggpt_decode")
        return out

    def mask(self, pred):
        """:166-171 -> uint8 {0,1} [hres][res][3]."""
        p0 = self._pred0(pred)
        out = torch.empty((self.hres, self.res, 3), dtype=torch.uint8, device=self.device)
        check(lib.pa_seggpt_mask(p0.data_ptr(), out.data_ptr(), self.hres, self.res, self.patch, _stream()), "pa_seggpt_mask")
        return out

    def blend(self, pred, image):
        """:95-102 / :173-179 -> uint8 [H0][W0][3], the input picture dimmed to 40 % outside the predicted mask colours."""
        p0 = self._pred0(pred)
        image = self._img(image)
        h0, w0, _ = image.shape
        out = torch.empty_like(image)
        check(lib.pa_seggpt_blend(p0.data_ptr(), image.data_ptr(), out.data_ptr(), h0, w0, self._table("torch_nearest", self.hres, h0).data_ptr(),
                                  self._table("torch_nearest", self.res, w0).data_ptr(), self.hres, self.res, self.patch, _stream()),
              "pa_seggpt_blend")
        return out


def _forward(model, imgs, tgts):
    """The model call of run_one_image (:36-47): second half of the canvas masked, everything valid, the module's seg_type,
    cross-prompt feature ensemble when there is more than one prompt.  -> float32 tokens [N][L][p*p*3]."""
    n = imgs.shape[0]
    num_patches = model.patch_embed.num_patches
    bool_masked_pos = torch.zeros(num_patches)
    bool_masked_pos[num_patches // 2:] = 1
    bool_masked_pos = bool_masked_pos.unsqueeze(dim=0)
    valid = torch.ones_like(tgts)
    if model.seg_type == 'instance':
        seg_type = torch.ones([n, 1])
    else:
        seg_type = torch.zeros([n, 1])
    feat_ensemble = 0 if n > 1 else -1
    _, y, _ = model(imgs, tgts, bool_masked_pos.to(imgs.device), valid, seg_type.to(imgs.device), feat_ensemble)
    return y


def _io_for(model, device, res, hres):
    return DeviceIO(device, res=res, hres=hres, patch=int(model.patch_size))


@torch.no_grad()
def run_one_image(img, tgt, model, device):
    """Reference signature (:26-53): img, tgt = normalised float arrays [N][2*hres][res][3] (host).  Returns the de-normalised,
    clipped lower half of sample 0 as a float64 CPU tensor [hres][res][3]; the decode runs on the device."""
    x = torch.as_tensor(np.asarray(img)).permute(0, 3, 1, 2).float().to(device)
    t = torch.as_tensor(np.asarray(tgt)).permute(0, 3, 1, 2).float().to(device)
    io = _io_for(model, device, res=x.shape[3], hres=x.shape[2] // 2)
    y = _forward(model, x.contiguous(), t.contiguous())
    return io.decode(y).cpu()


def _open_rgb(path):
    return np.array(Image.open(path).convert("RGB"))


@torch.no_grad()
def inference_image(model, device, img_path, img2_paths, tgt2_paths, out_path):
    """:56-103.  File decode / encode on the host; resize, normalise, stitch, model, decode, nearest-resize and blend on the device."""
    res, hres = 448, 448
    io = _io_for(model, device, res, hres)
    input_image = io.upload(_open_rgb(img_path))
    image = io.resize(input_image, (res, hres))
    prompts, targets = [], []
    for img2_path, tgt2_path in zip(img2_paths, tgt2_paths):
        prompts.append(io.resize(io.upload(_open_rgb(img2_path)), (res, hres)))
        targets.append(io.resize(io.upload(_open_rgb(tgt2_path)), (res, hres), nearest=True))
    imgs, tgts = io.stitch(torch.stack(prompts), torch.stack(targets), image)
    y = _forward(model, imgs, tgts)
    output = io.blend(y, input_image)
    Image.fromarray(output.cpu().numpy()).save(out_path)


class _FrameLoop:
    """State of the video loop (:125-132): the resized prompt pair and the FIFO caches of earlier (frame, predicted mask) pairs."""

    def __init__(self, model, device, num_frames, img2, tgt2):
        self.res, self.hres = 448, 448
        self.model = model
        self.io = _io_for(model, device, self.res, self.hres)
        self.img2 = self.io.resize(self.io.upload(img2), (self.res, self.hres))
        self.tgt2 = self.io.resize(self.io.upload(tgt2), (self.res, self.hres), nearest=True)
        self.frames_cache, self.target_cache = Cache(num_frames), Cache(num_frames)

    @torch.no_grad()
    def step(self, frame):
        io = self.io
        input_image = io.upload(frame)
        image = io.resize(input_image, (self.res, self.hres))
        prompts = torch.stack([self.img2] + list(self.frames_cache))
        targets = torch.stack([self.tgt2] + list(self.target_cache))
        div = [255.0] + [1.0] * len(self.target_cache)       # cached targets are {0,1} masks, used as they are (:166-171)
        imgs, tgts = io.stitch(prompts, targets, image, div)
        y = _forward(self.model, imgs, tgts)
        self.frames_cache.append(image)
        self.target_cache.append(io.mask(y))
        return io.blend(y, input_image).cpu().numpy()


def inference_frames(model, device, frames, num_frames, img2, tgt2):
    """The loop body of inference_video (:130-179) without the cv2 container I/O: `frames` yields RGB uint8 [H][W][3] arrays, `img2`
    and `tgt2` are the prompt image / prompt target as RGB uint8 arrays of any size, `num_frames` is the size of the prompt cache
    of earlier (frame, predicted mask) pairs.  Yields one blended RGB uint8 [H][W][3] array per frame.  (Grad mode is switched off
    per frame inside `_FrameLoop.step`, not around the generator, so the consumer's code between frames keeps its own grad mode.)"""
    loop = _FrameLoop(model, device, num_frames, img2, tgt2)
    for frame in frames:
        yield loop.step(frame)


def inference_video(model, device, vid_path, num_frames, img2_paths, tgt2_paths, out_path):
    """:106-181.  Container decode / encode through OpenCV as in the reference (imported here: it is not needed for anything else)."""
    import cv2
    cap = cv2.VideoCapture(vid_path)
    fps = cap.get(cv2.CAP_PROP_FPS)
    width = int(cap.get(cv2.CAP_PROP_FRAME_WIDTH))
    height = int(cap.get(cv2.CAP_PROP_FRAME_HEIGHT))
    writer = cv2.VideoWriter(out_path, cv2.VideoWriter_fourcc(*'mp4v'), fps, (width, height), True)
    if img2_paths is None:
        _, frame = cap.read()                                  # the first frame becomes the prompt and is not segmented (:116-118)
        img2 = np.ascontiguousarray(frame[:, :, ::-1])
    else:
        img2 = _open_rgb(img2_paths[0])
    tgt2 = _open_rgb(tgt2_paths[0])

    def frames():
        while True:
            ret, frame = cap.read()
            if not ret:
                return
            yield np.ascontiguousarray(frame[:, :, ::-1])

    for out in inference_frames(model, device, frames(), num_frames, img2, tgt2):
        writer.write(np.ascontiguousarray(out[:, :, ::-1]))
    writer.release()
