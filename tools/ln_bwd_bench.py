"""LayerNorm backward, the two variants of pa_debug_set(10, .) interleaved in one process: 0 = rows split over the workgroup's waves (round 5,
D >= 1024), 1 = one wave per row (rounds 1 - 4).  Times the launch with its partial-row reduction (as the engine's non-deferred calls run
it) at the ViT-L (R = 12544 and 25088, D = 1024) and ViT-H/14 (R = 8192, D = 1280) shapes, with the bf16 copy and the column sums the
engine asks for, and prints the rate over the algorithmic bytes (dy 2 + x 4 + dres 4 + dx 4 + dxT 2 bytes per element)."""
import statistics
import sys

import torch

sys.path.insert(0, ".")
from painter_amd import ops                      # noqa: E402
from painter_amd._lib import lib                 # noqa: E402

DEV = torch.device("cuda")
T = torch.bfloat16


def timeit(f, n=40):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    saved = lib.pa_debug_get(10)
    try:
        for R, D in ((12544, 1024), (25088, 1024), (8192, 1280)):
            x = torch.randn(R, D, generator=g).to(DEV)
            gam, bet = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
            _, mean, rstd = ops.layernorm_fwd(x, gam, bet, 1e-6, T)
            dy = torch.randn(R, D, generator=g).to(T).to(DEV)
            dres = torch.randn(R, D, generator=g).to(DEV)
            dxT = torch.empty(R, D, dtype=T, device=DEV)
            cs = torch.empty(D, device=DEV)
            res = {0: [], 1: []}
            for _ in range(4):
                for v in (0, 1):
                    lib.pa_debug_set(10, v)
                    res[v].append(timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, gam, dres=dres, dx=dres, dxT=dxT, dxT_colsum=cs)))
            for v in (0, 1):
                us = statistics.median(res[v])
                print("R %5d D %4d variant %d: %6.1f us (kernel + partial-row reduction)  %.2f TB/s algorithmic  (%s)"
                      % (R, D, v, us, R * D * 16 / us / 1e6, " ".join("%.1f" % t for t in res[v])), flush=True)
    finally:
        lib.pa_debug_set(10, saved)


if __name__ == "__main__":
    main()
