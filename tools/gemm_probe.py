"""Diagnostics for the 256x256 GEMM on the K=1024 shapes: what the epilogue stores and the lock-step start cost.
Uses pa_debug_set (stagger / no-store), never part of the product path."""
import sys

import torch

sys.path.insert(0, ".")
from painter_amd import ops                                                # noqa: E402
from painter_amd._lib import EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, lib  # noqa: E402

DEV, T = "cuda", torch.bfloat16


def timeit(fn, iters=20, warm=3):
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
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: (torch.rand(s, generator=g) * 2 - 1).to(T).to(DEV)
    M = 12544
    x1k, w4k, w3k, w1k = rnd(M, 1024), rnd(4096, 1024) * 0.05, rnd(3072, 1024) * 0.05, rnd(1024, 1024) * 0.05
    x4k, w14 = rnd(M, 4096), rnd(1024, 4096) * 0.05
    aux8 = ops.gelu_aux_encode(x4k.float().sigmoid())          # any 8-bit codes serve for timing (C ABI 6)
    b4k, b3k, b1k = torch.zeros(4096, device=DEV), torch.zeros(3072, device=DEV), torch.zeros(1024, device=DEV)
    act, pre = torch.empty(M, 4096, dtype=T, device=DEV), torch.empty(M, 4096, dtype=T, device=DEV)
    qkv = torch.empty(M, 3072, dtype=T, device=DEV)
    o32, resid = torch.empty(M, 1024, device=DEV), torch.zeros(M, 1024, device=DEV)
    dy1k = rnd(M, 1024)
    dx4k = torch.empty(M, 4096, dtype=T, device=DEV)
    dx1k = torch.empty(M, 1024, dtype=T, device=DEV)
    cases = [
        ("fc1 fwd gelu  (N=4096,K=1024)", lambda: ops.linear_fwd(x1k, w4k, b4k, EPI_BIAS_GELU, out=act, out2=pre)),
        ("fc1 fwd bias  (N=4096,K=1024)", lambda: ops.linear_fwd(x1k, w4k, b4k, EPI_BIAS, out=act)),
        ("qkv fwd bias  (N=3072,K=1024)", lambda: ops.linear_fwd(x1k, w3k, b3k, EPI_BIAS, out=qkv)),
        ("proj fwd resid(N=1024,K=1024)", lambda: ops.linear_fwd(x1k, w1k, b1k, EPI_BIAS_RESID, out=o32, resid=resid)),
        ("fc2 fwd resid (N=1024,K=4096)", lambda: ops.linear_fwd(x4k, w14, b1k, EPI_BIAS_RESID, out=o32, resid=resid)),
        ("fc2 dgrad gelu(N=4096,K=1024)", lambda: ops.linear_dgrad(dy1k, w14, gelu_aux=aux8, out=dx4k)),
        ("fc1 dgrad     (N=1024,K=4096)", lambda: ops.linear_dgrad(x4k, w4k, out=dx1k)),
    ]
    variants = [("base", 0, 0), ("nostore", 0, 1), ("stag8k", 8000, 0), ("stag16k", 16000, 0), ("stag32k", 32000, 0)]
    print("%-32s" % "case" + "".join("%10s" % v[0] for v in variants) + "   (us)")
    for name, fn in cases:
        row = []
        for _, stag, nost in variants:
            lib.pa_debug_set(0, stag)
            lib.pa_debug_set(1, nost)
            row.append(timeit(fn))
        lib.pa_debug_set(0, 0)
        lib.pa_debug_set(1, 0)
        print("%-32s" % name + "".join("%10.1f" % t for t in row))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
