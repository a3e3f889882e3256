"""End-to-end SegGPT video loop (painter_amd.seggpt_engine.inference_frames) with the ViT-L network, random weights, synthetic 1080p
frames: frames/s including host<->device copies of the uint8 frames, the device pre-/post-processing, and the forward with
1 + num_frames prompts (cross-prompt feature ensemble).  Prints one JSON line.

    python tools/seggpt_video_bench.py [--h 1080 --w 1920 --num-frames 2 --frames 40]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                            # noqa: E402
from painter_amd import models_seggpt                   # noqa: E402
from painter_amd import seggpt_engine as E              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--num-frames", type=int, default=2)
    ap.add_argument("--frames", type=int, default=40)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    net = models_seggpt.seggpt_vit_large_patch16_input896x448(compute_dtype="bf16")
    bench.randomize_parameters(net, seed=1)
    net = net.to("cuda").eval()
    net.seg_type = "instance"
    frames = [rng.integers(0, 256, (a.h, a.w, 3), dtype=np.uint8) for _ in range(4)]
    prompt = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    target = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)

    def run(n):
        it = E.inference_frames(net, "cuda", (frames[i % 4] for i in range(n)), a.num_frames, prompt, target)
        outs = 0
        for out in it:
            outs += 1
        return outs

    run(a.num_frames + 3)                                # warm-up: weight casts, tables, workspaces for every prompt count
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = run(a.frames)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # steady-state forward alone at the steady prompt count, for the split
    N = 1 + a.num_frames
    imgs = torch.randn(N, 3, 896, 448, device="cuda")
    tg = torch.randn(N, 3, 896, 448, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            E._forward(net, imgs, tg)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    with torch.no_grad():
        for _ in range(20):
            E._forward(net, imgs, tg)
    torch.cuda.synchronize()
    fwd_ms = (time.perf_counter() - t1) / 20 * 1e3
    print(json.dumps({"what": "SegGPT video loop end to end (inference_frames), ViT-L bf16, random weights, synthetic frames",
                      "frame": [a.h, a.w], "prompt_cache": a.num_frames, "frames": n, "ms_per_frame": round(dt / n * 1e3, 3),
                      "frames_per_sec": round(n / dt, 2), "forward_ms_at_steady_prompt_count": round(fwd_ms, 3),
                      "note": "the first num_frames frames run with fewer prompts; pageable host frames, synchronous copies"}))


if __name__ == "__main__":
    main()
