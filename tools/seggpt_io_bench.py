"""Measure the SegGPT pre-/post-processing row on one frame size: device kernels (csrc/seggpt_io.hip) against the
reference's host path (the same PIL / numpy / CPU-torch calls seggpt_engine.py:130-179 makes per video frame), model call excluded.

    python tools/seggpt_io_bench.py [--h 1080 --w 1920 --prompts 2 --iters 50] -> one JSON line
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from painter_amd import seggpt_engine as E          # noqa: E402

RES = HRES = 448


def host_frame(frame, prompts, targets, y0):
    """One frame of the reference's host work around the model call (seggpt_engine.py:134-157, :49-53, :163-179)."""
    image = np.array(Image.fromarray(frame).resize((RES, HRES))) / 255.
    ib, tb = [], []
    for p, t in zip(prompts, targets):
        tgt = np.concatenate((t, t), axis=0)
        img = np.concatenate((p, image), axis=0)
        ib.append((img - E.imagenet_mean) / E.imagenet_std)
        tb.append((tgt - E.imagenet_mean) / E.imagenet_std)
    x = torch.einsum('nhwc->nchw', torch.tensor(np.stack(ib))).float()
    t = torch.einsum('nhwc->nchw', torch.tensor(np.stack(tb))).float()
    y = y0.reshape(1, 56, 28, 16, 16, 3)
    y = torch.einsum('nhwpqc->nchpwq', y).reshape(1, 3, 896, 448)
    y = torch.einsum('nchw->nhwc', y)
    output = y[0, y.shape[1] // 2:, :, :]
    output = torch.clip((output * E.imagenet_std + E.imagenet_mean) * 255, 0, 255)
    mask = output.mean(-1).gt(128).float().unsqueeze(-1).expand(-1, -1, 3).numpy()
    output = F.interpolate(output[None, ...].permute(0, 3, 1, 2), size=[frame.shape[0], frame.shape[1]], mode='nearest').permute(0, 2, 3, 1)[0].numpy()
    out = (frame * (0.6 * output / 255 + 0.4)).astype(np.uint8)
    return x, t, mask, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--prompts", type=int, default=2)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--cpu-iters", type=int, default=5)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    frame = rng.integers(0, 256, (a.h, a.w, 3), dtype=np.uint8)
    prompts_u8 = rng.integers(0, 256, (a.prompts, HRES, RES, 3), dtype=np.uint8)
    targets_u8 = rng.integers(0, 256, (a.prompts, HRES, RES, 3), dtype=np.uint8)
    y0 = torch.randn(1568, 768)

    io = E.DeviceIO("cuda")
    dp, dt, dy = torch.from_numpy(prompts_u8).cuda(), torch.from_numpy(targets_u8).cuda(), y0.cuda()
    pinned_in = torch.from_numpy(frame).pin_memory()
    pinned_out = torch.empty_like(pinned_in)

    def device_frame(with_pcie):
        dev = pinned_in.to("cuda", non_blocking=True) if with_pcie else dframe
        image = io.resize(dev, (RES, HRES))
        imgs, tgts = io.stitch(dp, dt, image)
        m = io.mask(dy)
        out = io.blend(dy, dev)
        if with_pcie:
            pinned_out.copy_(out, non_blocking=True)
        return imgs, tgts, m, out

    dframe = torch.from_numpy(frame).cuda()
    res = {}
    for with_pcie in (False, True):
        for _ in range(5):
            device_frame(with_pcie)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            device_frame(with_pcie)
        torch.cuda.synchronize()
        res["device_ms_pcie" if with_pcie else "device_ms_resident"] = (time.perf_counter() - t0) / a.iters * 1e3

    # per-kernel device time (events on torch's current stream, which the kernels are launched on)
    def timed(fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            fn()
        s.record()
        for _ in range(a.iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / a.iters * 1e3
    image = io.resize(dframe, (RES, HRES))
    kern = {"resize_frame_us": timed(lambda: io.resize(dframe, (RES, HRES))), "stitch_us": timed(lambda: io.stitch(dp, dt, image)),
            "mask_us": timed(lambda: io.mask(dy)), "blend_us": timed(lambda: io.blend(dy, dframe))}
    frame_bytes = a.h * a.w * 3
    kern["blend_GBps"] = 2 * frame_bytes / (kern["blend_us"] * 1e-6) / 1e9            # frame read once + written once

    # host path, same libraries as the reference; check it agrees with the device before timing it
    pf, tf = [p / 255. for p in prompts_u8], [t / 255. for t in targets_u8]
    hx, ht, hm, hout = host_frame(frame, pf, tf, y0)
    dx, dtg, dm, dout = device_frame(False)
    same = bool(np.array_equal(hout, dout.cpu().numpy()) and np.array_equal(hx.numpy(), dx.cpu().numpy())
                and np.array_equal(ht.numpy(), dtg.cpu().numpy()) and np.array_equal(hm.astype(np.uint8), dm.cpu().numpy()))
    t0 = time.perf_counter()
    for _ in range(a.cpu_iters):
        host_frame(frame, pf, tf, y0)
    host_ms = (time.perf_counter() - t0) / a.cpu_iters * 1e3
    print(json.dumps({"what": "SegGPT per-frame pre+post processing, model call excluded", "frame": [a.h, a.w], "prompts": a.prompts,
                      "device_ms_resident": round(res["device_ms_resident"], 4), "device_ms_with_pcie": round(res["device_ms_pcie"], 4),
                      "host_reference_path_ms": round(host_ms, 2), "host_threads": torch.get_num_threads(),
                      "bit_identical_to_host_path": same, "kernels": {k: round(v, 2) for k, v in kern.items()}}))


if __name__ == "__main__":
    main()
