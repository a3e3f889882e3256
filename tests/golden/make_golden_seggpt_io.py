"""Generate tests/golden/seggpt_io.npz by running the UNMODIFIED reference functions
(SegGPT/SegGPT_inference/seggpt_engine.py: run_one_image, inference_image, inference_video) on CPU.

Build container only (needs /root/reference):   python tests/golden/make_golden_seggpt_io.py

The network is replaced by tests/seggpt_io_cases.StandInModel (a fixed float32 function of both canvases); inputs are the seeded
pictures of tests/seggpt_io_cases.py written as PNG files, so the fixture stores only SHA-256 digests of what the reference handed
to the model and of what it wrote, plus strided samples for debugging.  OpenCV is absent from this image: inference_video runs
against an in-memory stand-in for the five cv2 calls it makes (VideoCapture.get/read, VideoWriter_fourcc, VideoWriter.write/release).
"""
import os
import sys
import tempfile

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_import                     # noqa: E402
from tests import seggpt_io_cases as C            # noqa: E402


class RecordingModel(C.StandInModel):
    def __call__(self, x, tgt, bool_masked_pos, valid, seg_type, feat_ensemble):
        self.inputs = getattr(self, "inputs", [])
        self.inputs.append((C.digest(x.numpy()), C.digest(tgt.numpy())))
        return super().__call__(x, tgt, bool_masked_pos, valid, seg_type, feat_ensemble)


def install_memory_cv2(videos, written):
    """videos: {path: (fps, [BGR frames])}; written: {path: [BGR frames]} filled by VideoWriter.write."""
    cv2 = sys.modules["cv2"]
    cv2.CAP_PROP_FPS, cv2.CAP_PROP_FRAME_WIDTH, cv2.CAP_PROP_FRAME_HEIGHT = 5, 3, 4

    class VideoCapture:
        def __init__(self, path):
            self.fps, self.frames = videos[path]
            self.i = 0

        def get(self, prop):
            h, w = self.frames[0].shape[:2]
            return {5: self.fps, 3: float(w), 4: float(h)}[prop]

        def read(self):
            if self.i >= len(self.frames):
                return False, None
            self.i += 1
            return True, self.frames[self.i - 1].copy()

    class VideoWriter:
        def __init__(self, path, fourcc, fps, size, color):
            self.out = written.setdefault(path, [])
            self.size = size

        def write(self, frame):
            assert frame.dtype == np.uint8 and frame.shape[:2] == (self.size[1], self.size[0])
            self.out.append(frame.copy())

        def release(self):
            pass

    cv2.VideoCapture, cv2.VideoWriter = VideoCapture, VideoWriter
    cv2.VideoWriter_fourcc = lambda *a: 0


def main():
    eng = ref_import.load_reference_seggpt_engine()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        # ---- inference_image (:56-103)
        q, prompts, targets = C.image_case_inputs()
        Image.fromarray(q).save(os.path.join(tmp, "q.png"))
        p_paths, t_paths = [], []
        for i, (a, b) in enumerate(zip(prompts, targets)):
            p_paths.append(os.path.join(tmp, "p%d.png" % i))
            t_paths.append(os.path.join(tmp, "t%d.png" % i))
            Image.fromarray(a).save(p_paths[-1])
            Image.fromarray(b).save(t_paths[-1])
        model = RecordingModel()
        eng.inference_image(model, "cpu", os.path.join(tmp, "q.png"), p_paths, t_paths, os.path.join(tmp, "out.png"))
        result = np.array(Image.open(os.path.join(tmp, "out.png")))
        assert model.calls == [dict(n=2, masked=784, seg=2.0, merge=0, valid_ok=True)], model.calls
        out["image_imgs_digest"], out["image_tgts_digest"] = model.inputs[0]
        out["image_out_digest"] = C.digest(result)
        out["image_out_sample"] = result[::25, ::25]

        # ---- run_one_image (:26-53) on the oracle-independent normalised arrays the reference itself built: replay its own lines
        # through the module (single prompt -> feat_ensemble -1)
        model1 = RecordingModel()
        eng.inference_image(model1, "cpu", os.path.join(tmp, "q.png"), p_paths[:1], t_paths[:1], os.path.join(tmp, "out1.png"))
        assert model1.calls[0]["merge"] == -1 and model1.calls[0]["n"] == 1
        out["image1_out_digest"] = C.digest(np.array(Image.open(os.path.join(tmp, "out1.png"))))

        # ---- inference_video (:106-181) over an in-memory container
        vc = C.VIDEO_CASE
        frames_rgb = [C.picture(*s) for s in vc["frames"]]
        prompt = C.picture(*vc["prompt"])
        prompt_t = C.picture(vc["prompt"][0] + 100, vc["prompt"][1], vc["prompt"][2], flat=True)
        Image.fromarray(prompt).save(os.path.join(tmp, "vp.png"))
        Image.fromarray(prompt_t).save(os.path.join(tmp, "vt.png"))
        written = {}
        install_memory_cv2({"mem.mp4": (25.0, [f[:, :, ::-1].copy() for f in frames_rgb])}, written)
        vmodel = RecordingModel()
        eng.inference_video(vmodel, "cpu", "mem.mp4", vc["num_frames"], [os.path.join(tmp, "vp.png")], [os.path.join(tmp, "vt.png")],
                            "memout.mp4")
        assert [c["n"] for c in vmodel.calls] == [1, 2, 3], vmodel.calls
        for i, f in enumerate(written["memout.mp4"]):
            out["video_out_digest_%d" % i] = C.digest(np.ascontiguousarray(f[:, :, ::-1]))
            out["video_imgs_digest_%d" % i], out["video_tgts_digest_%d" % i] = vmodel.inputs[i]
        out["video_out_sample"] = np.ascontiguousarray(written["memout.mp4"][-1][::20, ::20, ::-1])
    path = os.path.join(HERE, "seggpt_io.npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
    print("wrote", path, os.path.getsize(path), "bytes")
    for k, v in out.items():
        if "digest" in k:
            print(" ", k, v[:16])


if __name__ == "__main__":
    main()
