"""Times the fused attention forward / backward at the ViT-L shape (B' = 8 and 16, 16 heads, 56x28 tokens, hd 64)."""
import sys

import torch

sys.path.insert(0, ".")
from painter_amd import ops   # noqa: E402

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
    return e0.elapsed_time(e1) / iters


def main():
    H, Hp, Wp = 16, 56, 28
    L = Hp * Wp
    g = torch.Generator().manual_seed(0)
    for B in (8, 16):
        qkv = (torch.randn(B * L, 3 * H * 64, generator=g)).to(T).to(DEV)
        dout = torch.randn(B * L, H * 64, generator=g).to(T).to(DEV)
        rel_h = (torch.randn(2 * Hp - 1, 64, generator=g) * 0.05).to(DEV)
        rel_w = (torch.randn(2 * Wp - 1, 64, generator=g) * 0.05).to(DEV)
        rcat = ops.relpos_pack(rel_h, rel_w, Hp, Wp, T)
        rcatT = ops.relpos_pack_t(rel_h, rel_w, Hp, Wp, T)
        out, lse = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125)
        fl = 4.0 * B * H * L * L * 64
        tf = timeit(lambda: ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125))
        tb = timeit(lambda: ops.attn_bwd(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125))
        print("B'=%d  fwd %.3f ms (%.0f TFLOP/s)   bwd %.3f ms (%.0f TFLOP/s algorithmic, 2.5x fwd)" % (B, tf, fl / tf / 1e9, tb, 2.5 * fl / tb / 1e9))


if __name__ == "__main__":
    main()
