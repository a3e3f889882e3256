"""Round 6 diagnostics: what does a contraction tile of gemm256 cost by operand layout?  The same problem size -- 64 output tiles of 256 x 256,
196 contraction tiles each, one workgroup per tile (no K split, epilogue negligible) -- as
  KK  y  = x . W^T          (pa_linear_fwd:   both operands contraction-major, ds_read_b128 fragments)
  KM  dX = dY . W           (pa_linear_dgrad: B operand M-major, transposing ds_read_b64_tr_b16 fragments)
  MM  dW = dY^T . X         (pa_linear_wgrad: both operands M-major)
and the same three on the whole chip (256 tiles).  Prints microseconds per launch and per contraction tile.

    python tools/gemm_layout_probe.py
"""
import sys

import torch

sys.path.insert(0, ".")
from painter_amd import ops                      # noqa: E402
from painter_amd._lib import EPI_BIAS, lib       # noqa: E402


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    T = torch.bfloat16
    K = 12544
    saved = lib.pa_debug_get(3)
    lib.pa_debug_set(12, 1)                        # uniform tiles
    lib.pa_debug_set(4, 1)                         # 256-row tiles
    for rows, cols in ((4096, 1024), (4096, 4096)):
        tiles = (rows // 256) * (cols // 256)
        lib.pa_debug_set(3, tiles)                 # weight gradient: one workgroup per tile, no K split
        x = torch.randn(rows, K, device="cuda").to(T)
        w = (torch.randn(cols, K, device="cuda") * 0.05).to(T)
        b = torch.zeros(cols, device="cuda")
        wd = (torch.randn(K, cols, device="cuda") * 0.05).to(T)            # dgrad: dX [rows, cols] = dY [rows, K] . W [K, cols]
        dyw, xw = torch.randn(K, rows, device="cuda").to(T), torch.randn(K, cols, device="cuda").to(T)     # wgrad: dW [rows, cols] = dY^T . X
        res = (("KK forward", timeit(lambda: ops.linear_fwd(x, w, b, EPI_BIAS))), ("KM data gradient", timeit(lambda: ops.linear_dgrad(x, wd))),
               ("MM weight gradient", timeit(lambda: ops.linear_wgrad(dyw, xw))))
        # shader clock / board power while the forward launch loops for ~3 s (amdgpu hwmon, 100 ms: bench.py's sampler)
        import bench
        import time
        smp = bench.ClockPowerSampler()
        smp.start()
        t0 = time.time()
        while time.time() - t0 < 3.0:
            for _ in range(50):
                ops.linear_fwd(x, w, b, EPI_BIAS)
            torch.cuda.synchronize()
        cp = smp.stop()
        print("%d x %d output = %d tiles, K = %d (%d contraction tiles)" % (rows, cols, tiles, K, K // 64))
        print("  looping the forward launch: %s" % (cp,))
        for name, us in res:
            print("  %-20s %8.1f us   %6.3f us per contraction tile   %6.0f TFLOP/s" % (name, us, us / (K // 64), 2.0 * rows * cols * K / us / 1e6), flush=True)
        del x, w, wd, dyw, xw
    lib.pa_debug_set(3, saved)
    lib.pa_debug_set(12, 0)
    lib.pa_debug_set(4, 0)


if __name__ == "__main__":
    main()
