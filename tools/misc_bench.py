"""Times a few secondary kernels at the ViT-L B=8 shapes (relpos gradient contraction, LayerNorm, column sums)."""
import sys

import torch

sys.path.insert(0, ".")
from painter_amd import ops  # noqa: E402

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
    from painter_amd._lib import lib, check
    B, H, Hp, Wp = 8, 16, 56, 28
    L = Hp * Wp
    R = B * L
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(R, 3 * H * 64, generator=g).to(T).to(DEV)
    nrp = lib.pa_relpos_rows_padded(Hp, Wp)
    dG = torch.randn(R, H * nrp, generator=g).to(T).to(DEV)
    drcat = torch.empty((nrp, 64), dtype=torch.float32, device=DEV)
    ws = ops.workspace(lib.pa_attn_bwd_relpos_workspace_bytes(1, B, L, H, Hp, Wp, 64), qkv.device)
    f = lambda: check(lib.pa_attn_bwd_relpos(1, dG.data_ptr(), qkv.data_ptr(), qkv.stride(0), drcat.data_ptr(), ws.data_ptr(), B, L, H, Hp, Wp, 64, ops.stream()), "x")
    print("relpos_grad %.1f us" % timeit(f))
    x = torch.randn(R, 1024, generator=g).to(DEV)
    gam, bet = torch.ones(1024, device=DEV), torch.zeros(1024, device=DEV)
    y, mean, rstd = ops.layernorm_fwd(x, gam, bet, 1e-6, T)
    print("ln_fwd %.1f us" % timeit(lambda: ops.layernorm_fwd(x, gam, bet, 1e-6, T)))
    dy = torch.randn(R, 1024, generator=g).to(T).to(DEV)
    dres = torch.randn(R, 1024, generator=g).to(DEV)
    dxT = torch.empty(R, 1024, dtype=T, device=DEV)
    print("ln_bwd %.1f us" % timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, gam, dres=dres, dx=dres, dxT=dxT)))
    d4 = torch.randn(R, 4096, generator=g).to(T).to(DEV)
    print("colsum4096 %.1f us  colsum1024 %.1f us" % (timeit(lambda: ops.colsum(d4)), timeit(lambda: ops.colsum(dy))))


if __name__ == "__main__":
    main()
