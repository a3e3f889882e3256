"""What does the fc1 epilogue cost?  The fc1 shape (12544 x 4096 x 1024) with: bias only (one bf16 store), bias + GELU (one store),
bias + GELU + pre-activation (two stores, the training forward), for both row-tile heights of gemm256 (pa_debug_set(4, 1 | 2): 256 rows = 784
tiles in 3.06 rounds, 224 rows = 896 tiles in 3.5 rounds).  Diagnostics."""
import sys

import torch

sys.path.insert(0, ".")
from painter_amd import ops                                                # noqa: E402
from painter_amd._lib import EPI_BIAS, EPI_BIAS_GELU, lib                  # noqa: E402
from tools.gemm_bench import timeit                                        # noqa: E402

DEV, T = "cuda", torch.bfloat16


def main():
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: (torch.rand(s, generator=g) * 2 - 1).to(T).to(DEV)
    M = 12544
    x1, w = rnd(M, 1024), rnd(4096, 1024) * 0.05
    b = rnd(4096).float()
    a, pre = torch.empty(M, 4096, dtype=T, device=DEV), torch.empty(M, 4096, dtype=torch.uint8, device=DEV)      # (second output: the 8-bit gelu' code, C ABI 6)
    cases = [("bias, one store", lambda: ops.linear_fwd(x1, w, b, EPI_BIAS, out=a)),
             ("bias + GELU, one store", lambda: ops.linear_fwd(x1, w, b, EPI_BIAS_GELU, out=a, out2=None)),
             ("bias + GELU, two stores", lambda: ops.linear_fwd(x1, w, b, EPI_BIAS_GELU, out=a, out2=pre))]
    for mode, kname in ((0, "256-row tile"), (1, "224-row tile")):
        for name, fn in cases:
            res = []
            for rep in range(3):
                lib.pa_debug_set(4, 1 + mode)
                res.append(timeit(fn, iters=30) * 1e3)
            print("%s  %-26s %7.1f us" % (kname, name, min(res)), flush=True)
    lib.pa_debug_set(4, 0)


if __name__ == "__main__":
    main()
