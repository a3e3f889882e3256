"""Measure the training-input-pipeline row: one step's batch (B two-pair samples from 480 x 640 sources) built on the
device by painter_amd.pair_pipeline (host->device copies of the decoded uint8 pictures included) against the same pixel work done the
reference's way on the host (PIL crop / resize / ImageEnhance / HSV, torch ToTensor / Normalize / cat / valid), single process.
Decode and the parameter draws are outside both.    python tools/pair_pipeline_bench.py [--batch 8 --iters 20] -> one JSON line"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
from PIL import Image, ImageEnhance

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from painter_amd import pair_pipeline as PP          # noqa: E402

MEAN = torch.tensor(PP.MEAN)[:, None, None]
STD = torch.tensor(PP.STD)[:, None, None]


def host_pair(p, near1, near2):
    i, j, h, w = p.crop
    a = Image.fromarray(p.image).crop((j, i, j + w, i + h)).resize((448, 448), Image.NEAREST if near1 else Image.BICUBIC)
    t = Image.fromarray(p.target).crop((j, i, j + w, i + h)).resize((448, 448), Image.NEAREST if near2 else Image.BICUBIC)
    for o, f in zip(p.jitter_ops, p.jitter_factors):
        if o == PP.BRIGHTNESS:
            a = ImageEnhance.Brightness(a).enhance(f)
        elif o == PP.CONTRAST:
            a = ImageEnhance.Contrast(a).enhance(f)
        elif o == PP.SATURATION:
            a = ImageEnhance.Color(a).enhance(f)
        else:
            hh, s, v = a.convert("HSV").split()
            nh = (np.array(hh, dtype=np.uint8).astype(np.int32) + PP.hue_shift_byte(f)).astype(np.uint8)
            a = Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB")
    if p.flip:
        a, t = a.transpose(Image.FLIP_LEFT_RIGHT), t.transpose(Image.FLIP_LEFT_RIGHT)
    out = []
    for pic in (a, t):
        x = torch.from_numpy(np.array(pic)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
        out.append(x.sub_(MEAN).div_(STD))
    return out


def host_sample(s):
    near1, near2 = s.interpolation[0] == "nearest", s.interpolation[1] == "nearest"
    parts = [host_pair(p, near1, near2) for p in s.pairs]
    img = torch.cat([x[0] for x in parts], dim=1)
    tgt = torch.cat([x[1] for x in parts], dim=1)
    mode, level = PP.valid_rule(s.pair_type)
    valid = torch.ones_like(tgt)
    if mode == PP.VALID_LESS_ZERO:
        thres = (torch.ones(3) * level - torch.tensor(PP.MEAN)) / torch.tensor(PP.STD)
        valid[tgt < thres[:, None, None]] = 0
    return img, tgt, valid


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cpu-iters", type=int, default=2)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    specs = []
    for b in range(a.batch):
        pairs = []
        for k in range(2):
            img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
            tgt = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
            ops, fac = PP.sample_color_jitter()
            pairs.append(PP.PairSpec(img, tgt, PP.sample_resized_crop(480, 640, (0.3, 1.0)), ops, fac, PP.sample_flip()))
        specs.append(PP.SampleSpec(pairs=pairs, pair_type="coco_image2panoptic_sem_seg"))
    pipe = PP.DevicePairPipeline("cuda")
    for _ in range(3):
        out = pipe.build_batch(specs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        out = pipe.build_batch(specs)
    torch.cuda.synchronize()
    dev_ms = (time.perf_counter() - t0) / a.iters * 1e3
    ref = [host_sample(s) for s in specs]
    same = all(torch.equal(out[0][b].cpu(), ref[b][0]) and torch.equal(out[1][b].cpu(), ref[b][1]) and torch.equal(out[2][b].cpu(), ref[b][2])
               for b in range(a.batch))
    t0 = time.perf_counter()
    for _ in range(a.cpu_iters):
        [host_sample(s) for s in specs]
    host_ms = (time.perf_counter() - t0) / a.cpu_iters * 1e3
    print(json.dumps({"what": "training input pipeline pixel work for one step (two-pair samples from 480x640 sources), decode and parameter draws excluded",
                      "batch": a.batch, "device_ms_per_batch_incl_uploads": round(dev_ms, 3), "host_reference_path_ms_per_batch_1_process": round(host_ms, 2),
                      "bit_identical_to_host_path": same}))


if __name__ == "__main__":
    main()
