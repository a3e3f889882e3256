"""Does de-phasing the first round of workgroups pay?  Times the multi-round GEMMs (fc1+GELU forward, qkv forward, fc2 dgrad with the
GELU' epilogue, the decoder GEMM) for several values of the start-up stagger (pa_debug_set(0, shader cycles)).  Diagnostics."""
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
    x1, w_fc1, w_qkv, w_proj = rnd(M, 1024), rnd(4096, 1024) * 0.05, rnd(3072, 1024) * 0.05, rnd(1024, 1024) * 0.05
    x4, w_fc2 = rnd(M, 4096), rnd(1024, 4096) * 0.05
    b4, b3, b1 = torch.zeros(4096, device=DEV), torch.zeros(3072, device=DEV), torch.zeros(1024, device=DEV)
    o4a, o4b = torch.empty(M, 4096, dtype=T, device=DEV), torch.empty(M, 4096, dtype=T, device=DEV)
    o3 = torch.empty(M, 3072, dtype=T, device=DEV)
    o32, resid = torch.empty(M, 1024, device=DEV), torch.zeros(M, 1024, device=DEV)
    dy1, hpre = rnd(M, 1024), rnd(M, 4096)
    aux8 = ops.gelu_aux_encode(hpre.float().sigmoid())          # any 8-bit codes serve for timing (C ABI 6: the bf16 GELU side input is a uint8 code)
    dx1 = torch.empty(M, 1024, dtype=T, device=DEV)
    cases = [("fc1+gelu fwd (784 tiles)", lambda: ops.linear_fwd(x1, w_fc1, b4, EPI_BIAS_GELU, out=o4a, out2=o4b)),
             ("qkv fwd (588 tiles)", lambda: ops.linear_fwd(x1, w_qkv, b3, EPI_BIAS, out=o3)),
             ("fc2 dgrad+gelu' (784 tiles)", lambda: ops.linear_dgrad(dy1, w_fc2, gelu_aux=aux8)),
             ("proj fwd+resid (196 tiles)", lambda: ops.linear_fwd(x1, w_proj, b1, EPI_BIAS_RESID, out=o32, resid=resid)),
             ("fc2 fwd+resid (196 tiles)", lambda: ops.linear_fwd(x4, w_fc2, b1, EPI_BIAS_RESID, out=o32, resid=resid)),
             ("fc1 dgrad (196 tiles)", lambda: ops.linear_dgrad(o4a, w_fc1, out=dx1))]
    for name, fn in cases:
        res = []
        for st in (0, 8000, 16000, 24000, 32000, 48000, 64000, 0):
            lib.pa_debug_set(0, st)
            res.append("%d: %.1f" % (st, timeit(fn, iters=30) * 1e3))
        lib.pa_debug_set(0, 0)
        print("%-30s us by stagger cycles  %s" % (name, "  ".join(res)), flush=True)


if __name__ == "__main__":
    main()
