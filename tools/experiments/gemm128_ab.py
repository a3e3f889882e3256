"""gemm128 (128 x 256 tiles, 4 waves, two workgroups per CU) against gemm256 (256 x 256, 8 waves, one per CU) on the forward GEMMs of a
ViT-L block at B = 8: time, interleaved in one process (pa_debug_set(4, 1 + mode)), and bit-equality of the outputs (both accumulate
the 16-deep MFMA steps in ascending k).  Diagnostics."""
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
    b4, b3, b1 = rnd(4096).float(), rnd(3072).float(), rnd(1024).float()
    resid = torch.randn(M, 1024, generator=g).to(DEV)

    def fc1():
        a, b = torch.empty(M, 4096, dtype=T, device=DEV), torch.empty(M, 4096, dtype=T, device=DEV)
        ops.linear_fwd(x1, w_fc1, b4, EPI_BIAS_GELU, out=a, out2=b)
        return a, b

    def qkv():
        return (ops.linear_fwd(x1, w_qkv, b3, EPI_BIAS),)

    def proj():
        o = torch.empty(M, 1024, device=DEV)
        ops.linear_fwd(x1, w_proj, b1, EPI_BIAS_RESID, out=o, resid=resid)
        return (o,)

    def fc2():
        o = torch.empty(M, 1024, device=DEV)
        ops.linear_fwd(x4, w_fc2, b1, EPI_BIAS_RESID, out=o, resid=resid)
        return (o,)

    for name, fn, fl in (("fc1+gelu", fc1, 2.0 * M * 4096 * 1024), ("qkv", qkv, 2.0 * M * 3072 * 1024), ("proj+resid", proj, 2.0 * M * 1024 * 1024),
                         ("fc2+resid", fc2, 2.0 * M * 1024 * 4096)):
        outs = {}
        for mode in (0, 1):
            lib.pa_debug_set(4, 1 + mode)
            outs[mode] = fn()
        torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
        worst = max(float((a.float() - b.float()).abs().max()) for a, b in zip(outs[0], outs[1]))
        res = {0: [], 1: []}
        for rep in range(3):
            for mode in (0, 1):
                lib.pa_debug_set(4, 1 + mode)
                res[mode].append(timeit(fn, iters=30) * 1e3)
        lib.pa_debug_set(4, 0)
        t0, t1 = min(res[0]), min(res[1])
        print("%-11s gemm256 %7.1f us (%5.0f TFLOP/s)   gemm128 %7.1f us (%5.0f TFLOP/s)   bit-identical %s (max abs diff %.3g)"
              % (name, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6, same, worst), flush=True)
        lib.pa_debug_set(4, 2)
        st = []
        for cyc in (4000, 8000, 12000, 16000, 24000, 32000, 0):
            lib.pa_debug_set(5, cyc)
            st.append("%d: %.1f" % (cyc, timeit(fn, iters=30) * 1e3))
        lib.pa_debug_set(5, 0)
        lib.pa_debug_set(4, 0)
        print("            gemm128 with the second resident workgroup started late by (cycles: us)  " + "  ".join(st), flush=True)


if __name__ == "__main__":
    main()
