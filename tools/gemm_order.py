"""Tile order A/B for gemm256: blocked 4 x 8 patches per XCD (default) against plain row-major (pa_debug_set(2, 1)), interleaved in
one process, on the ViT-L B = 8 shapes.  Diagnostics."""
import sys

import torch

sys.path.insert(0, ".")
from painter_amd import ops                                                # noqa: E402
from painter_amd._lib import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, lib  # noqa: E402
from tools.gemm_bench import timeit                                        # noqa: E402

DEV, T = "cuda", torch.bfloat16


def main():
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: (torch.rand(s, generator=g) * 2 - 1).to(T).to(DEV)
    M = 12544
    x1, x4 = rnd(M, 1024), rnd(M, 4096)
    w_fc1, w_qkv, w_proj, w_fc2 = rnd(4096, 1024) * 0.05, rnd(3072, 1024) * 0.05, rnd(1024, 1024) * 0.05, rnd(1024, 4096) * 0.05
    b4, b3, b1 = torch.zeros(4096, device=DEV), torch.zeros(3072, device=DEV), torch.zeros(1024, device=DEV)
    o4a, o4b = torch.empty(M, 4096, dtype=T, device=DEV), torch.empty(M, 4096, dtype=T, device=DEV)
    o3 = torch.empty(M, 3072, dtype=T, device=DEV)
    o32, resid = torch.empty(M, 1024, device=DEV), torch.zeros(M, 1024, device=DEV)
    dy1, dy3, hpre = rnd(M, 1024), rnd(M, 3072), rnd(M, 4096)
    aux8 = ops.gelu_aux_encode(hpre.float().sigmoid())          # any 8-bit codes serve for timing (C ABI 6)
    dx1, dx4 = torch.empty(M, 1024, dtype=T, device=DEV), torch.empty(M, 4096, dtype=T, device=DEV)
    dw41, dw14, dw31 = torch.empty(4096, 1024, device=DEV), torch.empty(1024, 4096, device=DEV), torch.empty(3072, 1024, device=DEV)
    cases = [("fc1+gelu fwd", lambda: ops.linear_fwd(x1, w_fc1, b4, EPI_BIAS_GELU, out=o4a, out2=o4b)),
             ("qkv fwd", lambda: ops.linear_fwd(x1, w_qkv, b3, EPI_BIAS, out=o3)),
             ("proj fwd+resid", lambda: ops.linear_fwd(x1, w_proj, b1, EPI_BIAS_RESID, out=o32, resid=resid)),
             ("fc2 fwd+resid", lambda: ops.linear_fwd(x4, w_fc2, b1, EPI_BIAS_RESID, out=o32, resid=resid)),
             ("fc2 dgrad+gelu'", lambda: ops.linear_dgrad(dy1, w_fc2, gelu_aux=aux8, out=dx4)),
             ("fc1 dgrad", lambda: ops.linear_dgrad(o4a, w_fc1, out=dx1)),
             ("qkv dgrad", lambda: ops.linear_dgrad(dy3, w_qkv, out=dx1)),
             ("fc1 wgrad", lambda: ops.linear_wgrad(o4a, x1, out=dw41)),
             ("fc2 wgrad", lambda: ops.linear_wgrad(dy1, x4, out=dw14)),
             ("qkv wgrad", lambda: ops.linear_wgrad(dy3, x1, out=dw31))]
    for name, fn in cases:
        res = {0: [], 1: []}
        for rep in range(3):
            for order in (0, 1):
                lib.pa_debug_set(2, order)
                res[order].append(timeit(fn, iters=30) * 1e3)
        lib.pa_debug_set(2, 0)
        print("%-18s blocked %7.1f us   row-major %7.1f us   (all: %s | %s)" % (name, min(res[0]), min(res[1]), ["%.1f" % v for v in res[0]], ["%.1f" % v for v in res[1]]),
              flush=True)


if __name__ == "__main__":
    main()
