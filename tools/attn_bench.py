"""Times the fused attention forward / backward at the ViT-L shape (B' = 8 and 16, 16 heads, 56x28 tokens, hd 64) for the
generation-3 kernels (default) and generation 2 (pa_attn_set_generation(2)), interleaved in one process; prints the accuracy of
both against an fp64 reference on a small slice as well.  Env knobs of the kernels apply (PA_ATTN3_FWD_STAGES, PA_ATTN3_DQ_WAVES, ...)."""
import sys

import torch

sys.path.insert(0, ".")
from painter_amd import ops   # noqa: E402
from painter_amd._lib import lib   # noqa: E402

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
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    for B in (8, 16):
        qkv = (torch.randn(B * L, 3 * H * 64, generator=g)).to(T).to(DEV)
        dout = torch.randn(B * L, H * 64, generator=g).to(T).to(DEV)
        rel_h = (torch.randn(2 * Hp - 1, 64, generator=g) * 0.05).to(DEV)
        rel_w = (torch.randn(2 * Wp - 1, 64, generator=g) * 0.05).to(DEV)
        rcat = ops.relpos_pack(rel_h, rel_w, Hp, Wp, T)
        rcatT = ops.relpos_pack_t(rel_h, rel_w, Hp, Wp, T)
        fl = 4.0 * B * H * L * L * 64
        res = {2: [], 5: [], 0: [], 4: []}
        for _ in range(rounds):
            for gen_ in (2, 0, 5, 4):
                lib.pa_attn_set_generation(gen_)
                out, lse, tables = ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True)
                tf = timeit(lambda: ops.attn_fwd(qkv, rcat, B, L, H, Hp, Wp, 0.125, need_tables=True))
                tb = timeit(lambda: ops.attn_bwd_core(qkv, rcat, rcatT, out, dout, lse, B, L, H, Hp, Wp, 0.125, tables=tables))
                res[gen_].append((tf, tb))
        lib.pa_attn_set_generation(0)
        for gen_, name in ((2, "gen2"), (0, "gen3 4-wave"), (5, "gen3 pipelined dq"), (4, "gen3 paired 8-wave")):
            tf = min(r[0] for r in res[gen_])
            tb = min(r[1] for r in res[gen_])
            print("B'=%d %s  fwd %.3f ms (%.0f TFLOP/s)   bwd core %.3f ms (%.0f TFLOP/s algorithmic, 2.5x fwd)   all rounds fwd %s bwd %s"
                  % (B, name, tf, fl / tf / 1e9, tb, 2.5 * fl / tb / 1e9, ["%.3f" % r[0] for r in res[gen_]], ["%.3f" % r[1] for r in res[gen_]]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
