"""Times the decoder-head 3x3 conv kernels at the BASELINE configs[1] size (B=8, 896x448, 64 channels, bf16)."""
import sys

import torch

sys.path.insert(0, ".")
from painter_amd import ops  # noqa: E402

DEV, T = "cuda", torch.bfloat16


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    B, Hi, Wi, P = 8, 896, 448, 16
    g = torch.Generator().manual_seed(0)
    x = torch.randn((B, Hi, Wi, 64), generator=g).to(T).to(DEV)
    dy = torch.randn((B, Hi, Wi, 64), generator=g).to(T).to(DEV)
    w3 = (torch.randn((64, 64, 3, 3), generator=g) * 0.05).to(DEV)
    v64 = torch.randn(64, generator=g).to(DEV) * 0.1
    w1, b1 = torch.randn((3, 64), generator=g).to(DEV) * 0.1, torch.zeros(3, device=DEV)
    w3r, wf = ops.conv3x3_pack(w3, T)
    fl = 2.0 * B * Hi * Wi * 576 * 64
    for name, fn in [("tail fwd", lambda: ops.decoder_tail_fwd(x, w3r, v64, 1 + v64, v64, w1, b1, 1e-6, save_y3=True)),
                     ("dgrad+unshuffle", lambda: ops.conv3x3_dgrad_unshuffle(dy, wf, B, Hi // P, Wi // P, P)),
                     ("wgrad", lambda: ops.conv3x3_wgrad(dy, x))]:
        us = timeit(fn)
        print("%-18s %9.1f us  %7.1f TFLOP/s" % (name, us, fl / us / 1e6))


if __name__ == "__main__":
    main()
