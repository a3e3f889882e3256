"""Isolation test: fused attention forward/backward (explicit packed-fp32 VALU code) and the fc1+GELU GEMM epilogue on the main
stream, repeated with fixed inputs while weight-gradient GEMMs (MFMA) run on a side stream.  Any bitwise difference is reported."""
import sys

import torch

sys.path.insert(0, ".")
from painter_amd import ops  # noqa: E402
from painter_amd._lib import EPI_BIAS_GELU  # noqa: E402

DEV, T = "cuda", torch.bfloat16


def main():
    g = torch.Generator().manual_seed(0)
    B, H, Hp, Wp = 4, 16, 56, 28
    L = Hp * Wp
    qkv = torch.randn(B * L, 3 * H * 64, generator=g).to(T).to(DEV)
    dout = torch.randn(B * L, H * 64, generator=g).to(T).to(DEV)
    rel_h = (torch.randn(2 * Hp - 1, 64, generator=g) * 0.05).to(DEV)
    rel_w = (torch.randn(2 * Wp - 1, 64, generator=g) * 0.05).to(DEV)
    rcat = ops.relpos_pack(rel_h, rel_w, Hp, Wp, T)
    rcatT = ops.relpos_pack_t(rel_h, rel_w, Hp, Wp, T)
    x = torch.randn(B * L, 1024, generator=g).to(T).to(DEV)
    w = (torch.randn(4096, 1024, generator=g) * 0.05).to(T).to(DEV)
    bias = torch.randn(4096, generator=g).to(DEV) * 0.1
    sdy = torch.randn(12544, 4096, generator=g).to(T).to(DEV)
    sx = torch.randn(12544, 1024, generator=g).to(T).to(DEV)
    side = torch.cuda.Stream()

    def one(with_side):
        if with_side:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    ops.linear_wgrad(sdy, sx)
        out, lse = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125)
        dqkv, drcat = ops.attn_bwd(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125)
        act, pre = ops.linear_gelu(x, w, bias)
        if with_side:
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        return out, lse, dqkv, drcat, act, pre

    ref = one(False)
    names = ["attn out", "lse", "dqkv", "drel_pos", "gelu act", "gelu pre"]
    bad = {n: 0 for n in names}
    N = 400
    for it in range(N):
        for n, a, b in zip(names, one(True), ref):
            if not torch.equal(a, b):
                bad[n] += 1
    print("differing launches of", N, ":", bad)


if __name__ == "__main__":
    main()
